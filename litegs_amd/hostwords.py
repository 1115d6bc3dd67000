"""Pinned host words the DEVICE stores into asynchronously (sizing feedback of the GPU-driven protocol, speculation mirrors, exchange
headers), handed out by the library's arena (csrc/fused.hip ``lg_host_words_alloc``).

Why not ``torch.zeros(n).pin_memory()``: a kernel that stores into a pinned tensor is invisible to torch's host allocator, which recycles
the block as soon as the Python tensor dies and may unmap it on ``empty_cache()`` -- a launch still in flight then writes into memory that
belongs to somebody else, or to nobody (a GPU memory access fault).  The arena is never unmapped, and a freed range is re-issued only
after a device synchronisation.  The reference keeps such words alive for the life of the process as well (litegs/data.py:236-241)."""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import lib


class HostWords:
    """n int32 words of pinned, device-visible host memory; ``.a`` is a numpy view, ``.ptr`` the address (the same for host and device)."""

    def __init__(self, n: int):
        self.n = int(n)
        self.ptr = lib().lg_host_words_alloc(self.n)
        if not self.ptr:
            raise MemoryError(f"lg_host_words_alloc({n}) failed")
        self.a = np.ctypeslib.as_array(ctypes.cast(self.ptr, ctypes.POINTER(ctypes.c_int32)), shape=(self.n,))

    def addr(self, index: int = 0) -> int:
        return self.ptr + 4 * int(index)

    def tensor(self):
        """a torch int32 CPU tensor over the same words (``is_pinned()`` is true: the operator path of the reference takes such tensors)"""
        import torch
        return torch.from_numpy(self.a)

    def close(self):
        if self.ptr:
            ptr, self.ptr, self.a = self.ptr, 0, None
            try:
                lib().lg_host_words_free(ptr, self.n)
            except Exception:           # interpreter shutdown: the arena dies with the process
                pass

    def __del__(self):
        self.close()


def pinned_int32(shape):
    """(tensor, owner): a zeroed int32 CPU tensor over arena words for device stores (keep `owner` alive as long as the tensor is used)"""
    import torch
    n = 1
    for d in shape:
        n *= int(d)
    owner = HostWords(max(n, 1))
    return torch.from_numpy(owner.a[:n]).view(*shape), owner
