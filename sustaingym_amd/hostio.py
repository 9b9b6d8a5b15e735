"""Tensor <-> numpy transfers through page-locked staging memory.

``tensor.cpu()`` / ``torch.from_numpy(a).cuda()`` hand PAGEABLE host memory to the HIP runtime, which pins the
pages on the fly, lets the GPU access them and unpins them again.  On this stack (ROCm 7.2, MI355X) that path
intermittently ends in "Memory access fault by GPU ... on address <inside the host heap>" — seen in 2-3 of ~60
runs of the GPU test-suite, in the library's own pageable copies (rounds 1-2) and in torch's alike (DESIGN.md
§11).  Everything in this package and its tests therefore stages through pinned memory: the C library through
its own bounce buffers (csrc/evc_hostcopy.h), Python through these two helpers (torch's caching pinned
allocator makes the staging blocks cheap after first use).
"""
from __future__ import annotations

import numpy as np


def to_host(tensor) -> np.ndarray:
    """CUDA/HIP tensor -> fresh numpy array (device -> pinned staging -> pageable copy)."""
    import torch
    if not tensor.is_cuda:
        return tensor.detach().numpy().copy()
    src = tensor.detach().contiguous()
    stage = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    stage.copy_(src, non_blocking=True)
    torch.cuda.current_stream(src.device).synchronize()
    return stage.numpy().copy()


def to_device(array, device='cuda'):
    """numpy array -> CUDA/HIP tensor (pageable -> pinned staging -> device)."""
    import torch
    src = torch.from_numpy(np.ascontiguousarray(array))
    stage = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    stage.copy_(src)
    dev = torch.device(device)
    out = stage.to(dev, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    return out
