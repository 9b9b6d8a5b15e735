import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, json
print(json.dumps(bench.secondary_battery(0))[:300])
