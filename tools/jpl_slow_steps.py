"""Per step of a JPL GMM day (one launch per step): queued environments and the slow kernel's duration (GPU box)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import bench
w = bench.EvWorkload(sys.argv[1] if len(sys.argv) > 1 else 'jpl', 65536, 0, 0, project=True, episodes='gmm', phase='sync')
w.run(288)
w.eng.set_pipeline(1)
eng = w.eng
eng.enable_timing(True)
rows = []
for t in range(288):
    w.run(1)
    a, b = eng.last_step_ms()
    rows.append((t, eng.last_slow_count(), a * 1e3, b * 1e3))
eng.enable_timing(False)
rows = np.array(rows)
for t in range(90, 150, 4):
    print(f't={int(rows[t,0])}: queued {int(rows[t,1])}, streaming {rows[t,2]:.1f} us, slow kernel {rows[t,3]:.1f} us')
q = rows[:, 1]; s = rows[:, 3]
for lo, hi in ((1, 50), (50, 200), (200, 500), (500, 1000), (1000, 2000), (2000, 10**6)):
    m = (q >= lo) & (q < hi)
    if m.any(): print(f'queued in [{lo},{hi}): {int(m.sum())} steps, slow kernel mean {s[m].mean():.1f} us, per queued env {1e3 * s[m].sum() / q[m].sum():.1f} ns')
