"""Shared helpers for the -m gpu parity tests (all compute goes through the C ABI of libletkf_amd.so)."""
import numpy as np
import torch

from __graft_entry__ import load_package

pkg = load_package()
_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        assert torch.cuda.is_available()
        pkg.build()
        _ctx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
    return _ctx


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()
