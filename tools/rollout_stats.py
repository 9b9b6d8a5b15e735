#!/usr/bin/env python
"""tools/rollout_stats.py [site] — where the fused rollout's time goes on GMM greedy days, per wavefront (GPU box): 100 MHz
time stamps of the period loop, of the rare projection branch (exact rows + in-row water-filling + the solve call) and of the
solve call itself, with the number of visits and calls.  Needs a library built with -DEVC_ROLLOUT_STATS=1 (evc_rollout.hip)
and -DEVC_TIMELINE=1 (evc_engine.hip: evc_debug_read_slow_list), SUSTAINGYM_AMD_LIB pointing at it."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

site = sys.argv[1] if len(sys.argv) > 1 else 'caltech'
w = bench.EvWorkload(site, 65536, 0, 0, project=True, episodes='gmm', phase='sync')
torch = w.torch
for _ in range(8):                       # the engine's register-budget tuner settles on its build
    w.eng.rollout(policy='greedy', steps=288)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    w.eng.rollout(policy='greedy', steps=288)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f'{site}: {dt / 288 * 1e6:.2f} us per period, build at {w.eng.last_rollout_waves()} wavefronts per SIMD')
raw = np.zeros(65536, dtype=np.int32)
w.eng.lib.evc_debug_read_slow_list(w.eng.handle, raw.ctypes.data_as(C.c_void_p), 65536)
st_all = raw.view(np.uint32).reshape(16384, 4).astype(float)
st = st_all[:8192]                   # wavefronts 0 .. 8191: loop / rare branch / solve call; 8192 ..: the split of the visit
tot, rare, call = st[:, 0] * 0.01, st[:, 1] * 0.01, st[:, 2] * 0.01
calls, visits = np.floor(st[:, 3] % 65536), np.floor(st[:, 3] / 65536)
pc = lambda a: ' '.join(f'{x:9.1f}' for x in np.percentile(a, [1, 10, 50, 90, 99, 100]))
print('per wavefront (us over the 288 periods)      1        10        50        90        99       100')
print('whole loop                          ', pc(tot))
print('inside the rare branch              ', pc(rare))
print('inside the solve call               ', pc(call))
print('solve calls                         ', pc(calls))
print('visits of the rare branch           ', pc(visits))
print(f'per visit {rare.sum() / max(visits.sum(), 1):.2f} us (calls included), per call {call.sum() / max(calls.sum(), 1):.1f} us; loop without the branch {(tot - rare).mean() / 288:.2f} us per period')
print(f'means: loop {tot.mean():.1f} us, rare {rare.mean():.1f}, call {call.mean():.1f} ({call.sum() / max(calls.sum(), 1):.1f} us per call); '
      f'launch / mean wavefront loop = {dt * 1e6 / tot.mean():.2f} (8 wavefront rounds of 2 per SIMD would be 8.0)')
s2 = st_all[8192:]
ex, fl = s2[:, 1] * 0.01, s2[:, 2] * 0.01
short, fills = np.floor(s2[:, 3] % 65536), np.floor(s2[:, 3] / 65536)
passes, fcalls = np.floor(s2[:, 0] % 65536), np.floor(s2[:, 0] / 65536)
v2 = max(visits.mean(), 1e-9)
print(f'split of a visit (second half of the wavefronts): exact rows / shortcut test {ex.mean() / v2:.2f} us, fillings + second evaluation {fl.mean() / v2:.2f} us per visit; '
      f'{short.mean() / v2:.2f} of the visits take the caps-only shortcut, {fills.mean() / v2:.2f} have a filling; '
      f'{fcalls.mean() / v2:.2f} class fillings per visit, {passes.mean() / max(fcalls.mean(), 1e-9):.2f} Newton passes per filling')
w.close()
