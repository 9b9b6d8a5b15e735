# environments the projection screen leaves undecided per step on bench.py's own workload (needs a -DEVC_COUNT_UNDECIDED build):
# SUSTAINGYM_AMD_LIB=sustaingym_amd/variants/lib_undec.so python tools/scratch/undecided_bench.py [sync|stagger]
import os, sys, warnings
sys.path.insert(0, os.getcwd()); warnings.simplefilter('ignore')
import numpy as np
from bench import EvWorkload
phase = sys.argv[1] if len(sys.argv) > 1 else 'sync'
w = EvWorkload('caltech', 65536, 0, 0, phase=phase)
w.run(288)
prev = w.eng.read_metrics()['tie_snap_near_boundary']
counts = []
for i in range(288):
    w.run(1)
    cur = w.eng.read_metrics()['tie_snap_near_boundary']; counts.append(cur - prev); prev = cur
c = np.array(counts)
print(phase, 'undecided environments per step: mean %.0f  max %.0f  by 4 h %s' % (c.mean(), c.max(), [int(c[i*48:(i+1)*48].mean()) for i in range(6)]))
