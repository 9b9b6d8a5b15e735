import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import bench
print(json.dumps(bench.secondary_battery(0)))
