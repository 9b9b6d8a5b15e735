#!/usr/bin/env python
"""tools/probes/hbm_rw_mix.py — what one MI355X moves per second for pure writes, pure reads and a copy (GPU box), with library
kernels (torch fill_ / sum / copy_) on 1 GiB buffers: the ceilings a write-heavy stream such as bat_rollout's trajectory
(3 bytes written per byte read) can be priced against, next to the 8 TB/s of the data sheet."""
import json
import torch

dev = torch.device('cuda', 0)
n = 1 << 28                      # float32 elements: 1 GiB
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


gib = n * 4
out = {}
t = timed(lambda: a.fill_(1.0)); out['fill_write_only_TBps'] = round(gib / t / 1e12, 3)
t = timed(lambda: a.sum()); out['sum_read_only_TBps'] = round(gib / t / 1e12, 3)
t = timed(lambda: b.copy_(a)); out['copy_read_plus_write_TBps'] = round(2 * gib / t / 1e12, 3)
# 1 read : 3 writes, the trajectory kernel's mix: one source quarter broadcast into three destination quarters + itself read
q = n // 4
t = timed(lambda: (b[:q].copy_(a[:q]), b[q:2 * q].copy_(a[:q]), b[2 * q:3 * q].copy_(a[:q])))
out['three_copies_of_one_quarter_TBps_counting_3r_3w'] = round(6 * q * 4 / t / 1e12, 3)
print(json.dumps(out))
