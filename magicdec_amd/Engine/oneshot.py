"""C1 of SURVEY.md section 8e: one-shot sum-all-reduce of the per-layer bf16 partials over xGMI peer-mapped buffers
(`md_allreduce_oneshot`, csrc/allreduce.hip), one communicator per process group.

Replaces `dist.all_reduce` at Engine/SnapKV/model.py:336,455 (and the StreamingLLM twins) for the latency-bound
decode messages; anything larger than the registered buffer (prefill chunks) stays on RCCL.  RCCL / gloo is still
the bootstrap transport: the IPC handles are exchanged with `dist.all_gather_object`.

Selection: `MAGICDEC_ONESHOT_AR=1` asks for it (bench.py does for N > 1; the Engine default is RCCL).  The kernel and
the IPC set-up are validated with 2 and 3 processes sharing one GPU (tests/test_gpu_allreduce.py) -- the only
multi-process configuration available to the development box -- so on a real xGMI node `try_create` treats the
first use as a probe: every stage (allocation, handle export, peer mapping) is agreed on collectively, then a
self-test compares a few all-reduces with the bootstrap backend's (RCCL) results; any rank failing any stage makes
ALL ranks fall back to RCCL, loudly."""
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


def _all_ok(ok: bool, group) -> bool:
    """True iff every rank of the group reports ok (over the bootstrap backend)."""
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(t, group=group)
    return int(t.item()) == 0


def try_create(group, max_bytes: int = DEFAULT_MAX_BYTES):
    """Collective: returns a validated OneShotAllReduce on every rank of `group`, or None on every rank."""
    import warnings
    ar, err = None, None
    try:
        ar = OneShotAllReduce.__new__(OneShotAllReduce)
        ar._alloc(group, max_bytes)
    except Exception as e:  # noqa: BLE001
        err = f"set-up: {e}"
    if _all_ok(err is None, group):
        try:
            ar._exchange()
        except Exception as e:  # noqa: BLE001
            err = f"peer mapping: {e}"
    # _exchange contains a collective, so it is only entered when every rank allocated successfully
    ok = _all_ok(err is None, group)
    if ok:
        try:
            good = ar.self_test()
            if not good:
                err = "self-test mismatch against the bootstrap backend's all-reduce (or a peer time-out)"
        except Exception as e:  # noqa: BLE001
            err = f"self-test: {e}"
        ok = _all_ok(err is None, group)
    if not ok:
        if dist.get_rank(group) == 0 or err is not None:
            warnings.warn(f"[magicdec_amd] one-shot all-reduce disabled, using the bootstrap backend's collective "
                          f"(rank {dist.get_rank(group)}: {err or 'another rank failed'})", RuntimeWarning, stacklevel=2)
        if ar is not None and getattr(ar, "comm", None):
            ar.close()
        return None
    return ar


class OneShotAllReduce:
    def __init__(self, group, max_bytes: int = DEFAULT_MAX_BYTES):
        self._alloc(group, max_bytes)
        self._exchange()
        dist.barrier(group=group)       # nobody starts reducing before every rank has mapped its peers

    def _alloc(self, group, max_bytes):
        self.comm = None
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
        self._mine = mine.raw

    def _exchange(self):
        """One collective (the handle all-gather), entered by every rank, followed by local work that may fail;
        the caller synchronises afterwards (barrier, or try_create's agreement all-reduce)."""
        gathered = [None] * self.world
        dist.all_gather_object(gathered, self._mine, group=self.group)
        blob = b"".join(gathered)
        assert len(blob) == 2 * HANDLE_BYTES * self.world
        check(self.lib.md_ar_open_peers(self.comm, ctypes.create_string_buffer(blob, len(blob))), "md_ar_open_peers")

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

    def self_test(self) -> bool:
        """Local verdict: a few all-reduces agree with the bootstrap backend's (different summation order, so a bf16
        tolerance) and no spin timed out.  Collective (all ranks must call it)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        ok = True
        for k, n in enumerate((2048, 64 * 2048, min(256 * 4096, self.max_bytes // 2))):
            for rep in range(2):                       # both halves of the double buffer
                g = torch.Generator(device=dev).manual_seed(1000 * k + 10 * rep + self.rank)
                x = torch.randn(n, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
                ref = x.clone()
                dist.all_reduce(ref, group=self.group)
                y = x.clone()
                self.all_reduce_(y)
                torch.cuda.synchronize()
                tol = 2.0 ** -6 * float(ref.float().abs().max()) + 1e-3
                ok = ok and bool((y.float() - ref.float()).abs().max() <= tol) and bool(torch.isfinite(y.float()).all())
        return ok and self.status() == 0

    def status(self) -> int:
        """0 = ok, 1 = some call gave up waiting for a peer (synchronises the device)."""
        s = ctypes.c_int(0)
        check(self.lib.md_ar_status(self.comm, ctypes.byref(s)), "md_ar_status")
        return s.value

    def close(self):
        if self.comm:
            self.lib.md_ar_destroy(self.comm)
            self.comm = None
