import sys, json; sys.path.insert(0,'.')
import bench
print(json.dumps(bench.secondary_battery_rollout(0), indent=0))
