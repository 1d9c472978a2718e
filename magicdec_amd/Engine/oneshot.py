"""C1 of SURVEY.md section 8e: one-shot sum-all-reduce of the per-layer bf16 partials over xGMI peer-mapped buffers
(`md_allreduce_oneshot`, csrc/allreduce.hip), one communicator per process group.

Replaces `dist.all_reduce` at Engine/SnapKV/model.py:336,455 (and the StreamingLLM twins) for the latency-bound
decode messages; anything larger than the registered buffer (prefill chunks) stays on RCCL.  RCCL / gloo is still
the bootstrap transport: the IPC handles are exchanged with `dist.all_gather_object`.

OPT-IN (`MAGICDEC_ONESHOT_AR=1`): the kernel and the IPC set-up are validated with two processes sharing one
GPU (tests/test_gpu_allreduce.py) -- the only multi-process configuration available to this round -- not yet on
a real multi-GPU xGMI node, so RCCL remains the default collective."""
from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist

from .. import _lib
from .._lib import check

HANDLE_BYTES = 64           # MD_AR_HANDLE_BYTES
DEFAULT_MAX_BYTES = 4 << 20


def enabled() -> bool:
    return os.environ.get("MAGICDEC_ONESHOT_AR", "0") == "1" and torch.cuda.is_available()


class OneShotAllReduce:
    def __init__(self, group, max_bytes: int = DEFAULT_MAX_BYTES):
        self.lib = _lib.load()
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.max_bytes = int(max_bytes)
        comm = ctypes.c_void_p()
        check(self.lib.md_ar_create(self.rank, self.world, self.max_bytes, ctypes.byref(comm)), "md_ar_create")
        self.comm = comm
        mine = ctypes.create_string_buffer(2 * HANDLE_BYTES)
        check(self.lib.md_ar_get_handles(self.comm, mine), "md_ar_get_handles")
        gathered = [None] * self.world
        dist.all_gather_object(gathered, mine.raw, group=group)
        blob = b"".join(gathered)
        assert len(blob) == 2 * HANDLE_BYTES * self.world
        check(self.lib.md_ar_open_peers(self.comm, ctypes.create_string_buffer(blob, len(blob))), "md_ar_open_peers")
        dist.barrier(group=group)       # nobody starts reducing before every rank has mapped its peers

    def fits(self, t: torch.Tensor) -> bool:
        return (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() % 8 == 0
                and t.numel() * 2 <= self.max_bytes and t.data_ptr() % 16 == 0)

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the group (same result bits on every rank)."""
        if not self.fits(t):
            raise ValueError("OneShotAllReduce: tensor must be contiguous bf16 on the GPU, numel % 8 == 0, and fit "
                             f"the registered buffer ({self.max_bytes} bytes)")
        p = ctypes.c_void_p(t.data_ptr())
        check(self.lib.md_allreduce_oneshot(self.comm, p, p, t.numel(),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "md_allreduce_oneshot")
        return t

    def status(self) -> int:
        """0 = ok, 1 = some call gave up waiting for a peer (synchronises the device)."""
        s = ctypes.c_int(0)
        check(self.lib.md_ar_status(self.comm, ctypes.byref(s)), "md_ar_status")
        return s.value

    def close(self):
        if self.comm:
            self.lib.md_ar_destroy(self.comm)
            self.comm = None
