#!/usr/bin/env python
"""tools/wg_timeline.py [pipeline] — anatomy of one launch of the streaming kernel from per-wavefront time stamps (GPU box).

Needs a library built with -DEVC_TIMELINE=1 (or 2):  tools/build_variant.sh timeline "-DEVC_TIMELINE=1"
and SUSTAINGYM_AMD_LIB pointing at it.  Every wavefront stores 16 stamps of the 100 MHz counter into Params::slow_list:
0 entry, 1 after the prologue's barrier, 2+2i top of its i-th quad, 3+2i before the next quad's rows are requested
(EVC_TIMELINE=2: after a forced s_waitcnt vmcnt(0) at the top of the quad instead), 14 after its last quad, 15 end."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    pipeline = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    w = bench.EvWorkload('caltech', 65536, 0, 0, project=True, pipeline=pipeline)
    w.run(64)
    w.eng.join()
    w.torch.cuda.synchronize()
    w.run(40)                                   # steady state: the stamps of the LAST launch survive
    w.eng.join()
    w.torch.cuda.synchronize()
    raw = np.zeros(65536, dtype=np.int32)
    rc = w.eng.lib.evc_debug_read_slow_list(w.eng.handle, raw.ctypes.data_as(C.c_void_p), 65536)
    assert rc == 0
    st = raw.view(np.uint32)[:3072 * 16].reshape(3072, 16).astype(np.int64)
    t0 = st[:, 0].min()
    us = (st - t0) * 0.01                       # 100 MHz -> microseconds
    ok = (st[:, 15] >= st[:, 0]) & (st[:, 15] - st[:, 0] < 100000)
    us = us[ok]
    print(f'waves with stamps: {ok.sum()} of {len(ok)}; pipeline={pipeline}')
    q = lambda a: ' '.join(f'{x:6.2f}' for x in np.percentile(a, [1, 10, 50, 90, 99, 100]))
    print('percentiles              1     10     50     90     99    100')
    print('entry (us after first) ', q(us[:, 0]))
    print('prologue (0 -> 1)      ', q(us[:, 1] - us[:, 0]))
    if os.environ.get('EVC_TIMELINE_MODE') == '3':       # -DEVC_TIMELINE=3: slots 2 .. 13 are twelve points inside the wavefront's second quad
        names = ['top -> entries', 'entries (decode, action image, class sums)', 'screen', 'pilots, charge', 'excess', 'event pass',
                 'take quad + issue next rows', 'obs image', 'reward', 'autoreset', 'obs + write-back']
        have = ok[ok] & (us[:, 13] > us[:, 2]) & (us[:, 2] > us[:, 1])
        print(f'second quad: n={have.sum()}; whole iteration', q((us[:, 13] - us[:, 2])[have]))
        for k, nm in enumerate(names):
            print(f'  {nm:<44}', q((us[:, 3 + k] - us[:, 2 + k])[have]), f' mean {np.mean((us[:, 3 + k] - us[:, 2 + k])[have]):.3f}')
        w.close()
        return
    for i in range(6):
        a, b = us[:, 2 + 2 * i], us[:, 3 + 2 * i]
        have = (b >= a) & (a >= us[:, 1] - 1e-9) & (b <= us[:, 15] + 1e-9)
        if have.sum() < 10:
            continue
        nxt = us[:, 4 + 2 * i] if i < 5 else us[:, 14]
        h2 = have & (nxt >= b)
        print(f'quad {i}: n={have.sum():5d} top->mid', q((b - a)[have]), '| mid->next top', q((nxt - b)[h2]) if h2.sum() else '')
    print('loop exit (abs)        ', q(us[:, 14]))
    print('end (abs)              ', q(us[:, 15]))
    print('exit -> end (drain)    ', q(us[:, 15] - us[:, 14]))
    w.close()


if __name__ == '__main__':
    main()
