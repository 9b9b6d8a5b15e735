"""Per-wavefront time stamps of the lean streaming kernel (lib built with -DEVC_WG_TIMING): start, after the prologue,
end of streaming, end — spread over the launch."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('SUSTAINGYM_AMD_LIB', os.path.join(ROOT, 'sustaingym_amd/variants/lib_wgt.so'))
import numpy as np, torch
import bench
from sustaingym_amd import _lib
EP = os.environ.get('WG_EPISODES', 'synthetic')
w = bench.EvWorkload(os.environ.get('WG_SITE', 'caltech'), 65536, 0, 0, project=True, episodes=EP, phase='stagger' if EP == 'synthetic' else 'sync')
w.run(int(os.environ.get('WG_SKIP', '300')))
torch.cuda.synchronize()
lib = _lib.load()
grid = 768
for rep in range(3):
    w.run(1 if rep == 0 else 40); torch.cuda.synchronize()      # rep > 0: the LAST of 40 back-to-back launches (steady state)
    buf = np.zeros(grid * 4 * 8 * 2, np.int32)
    lib.evc_debug_read_slow_list(w.eng.handle, buf.ctypes.data_as(C.c_void_p), len(buf))
    raw = buf.view(np.uint64).reshape(grid, 4, 8).astype(np.float64) * 0.01      # us
    st = raw[:, :, :4].copy()
    cnt = raw[:, :, 5].sum() * 100
    print(f'   rare visits this launch: {cnt:.0f}, mean {raw[:,:,4].sum()/max(cnt,1):.2f} us, max {raw[:,:,6].max():.2f} us; water-filling passes per visit {raw[:,:,7].sum()*100/max(cnt,1):.1f}')
    t0 = st[:, :, 0].min()
    st -= t0
    print(f'launch {rep}: start spread {st[:,:,0].max():.2f} us | prologue {np.mean(st[:,:,1]-st[:,:,0]):.2f} (max {np.max(st[:,:,1]-st[:,:,0]):.2f}) | '
          f'streaming end: mean {st[:,:,2].mean():.2f} p50 {np.percentile(st[:,:,2],50):.2f} p90 {np.percentile(st[:,:,2],90):.2f} p99 {np.percentile(st[:,:,2],99):.2f} max {st[:,:,2].max():.2f} | '
          f'kernel end {st[:,:,3].max():.2f} | per-WG end: p50 {np.percentile(st[:,:,3].max(axis=1),50):.2f} p90 {np.percentile(st[:,:,3].max(axis=1),90):.2f}')
    # by XCD (block % 8)
    starts = st[:, :, 0].min(axis=1)
    late = np.flatnonzero(starts > 1.5)
    print('   late-starting WGs:', len(late), 'block ids', late[:12], '...', late[-6:], 'mod 8 hist', np.bincount(late % 8, minlength=8) if len(late) else None)
    nq = np.array([((st[b,:,2]-st[b,:,1]).sum()) for b in range(grid)])
    ends = st[:, :, 3].max(axis=1)
    print('   WG end vs block index thirds (bx<32 have 24 quads):', [round(float(ends[(np.arange(grid)//8 >= a) & (np.arange(grid)//8 < b)].mean()),2) for a,b in ((0,32),(32,64),(64,96))])
    print('   end by block%8:', [round(float(ends[x::8].max()), 2) for x in range(8)], ' wave streaming time mean', round(float(np.mean(st[:,:,2]-st[:,:,1])),2))
w.close()
