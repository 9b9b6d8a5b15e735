import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, ctypes
t=torch.zeros(1<<20, device='cuda'); print('torch dev', hex(t.data_ptr()))
t2=torch.zeros(1<<28, device='cuda'); print('torch dev big', hex(t2.data_ptr()))
p=torch.zeros(1<<20).pin_memory(); print('torch pinned', hex(p.data_ptr()))
c=torch.zeros(1<<26); print('torch cpu 256MB', hex(c.data_ptr()))
a=np.zeros(1<<20); print('numpy 8MB', hex(a.ctypes.data))
b=np.zeros(1<<26); print('numpy 512MB', hex(b.ctypes.data))
hip=ctypes.CDLL('libamdhip64.so')
ptr=ctypes.c_void_p(); hip.hipHostMalloc(ctypes.byref(ptr), 256, 0); print('hipHostMalloc 256B', hex(ptr.value))
ptr2=ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(ptr2), 1<<20); print('hipMalloc 1MB', hex(ptr2.value))
print(open('/proc/self/maps').read().count('\n'),'maps')
import re
for l in open('/proc/self/maps'):
    if '[heap]' in l: print(l.strip())
