"""C1 of SURVEY.md section 8e: the per-layer sum-all-reduce of the bf16 partials over xGMI peer-mapped buffers
(`md_allreduce` / `md_allreduce_add_rmsnorm`, csrc/allreduce.hip), one communicator per process group.

Replaces `dist.all_reduce` at Engine/SnapKV/model.py:334-335,453-454 (and the StreamingLLM twins) for the
latency-bound decode messages -- one-shot for the small ones (a 1B draft step: 256 KiB), two-shot (reduce-scatter +
all-gather through the registered buffers) for the 2 MiB verify message on >= 4 ranks -- and, fused into the same
launch, the residual add + RMSNorm that consumes the result.  Anything larger than the registered buffer (prefill
chunks) stays on RCCL.  RCCL / gloo is still the bootstrap transport: the IPC handles are exchanged with
`dist.all_gather_object`.

Selection: `MAGICDEC_ONESHOT_AR=1` asks for it; the Engine default is RCCL.  The kernels and the IPC set-up are
validated with 2 and 3 processes sharing one GPU (tests/test_gpu_allreduce.py) -- the only multi-process configuration
available to the development box -- so on a real xGMI node `try_create` treats the first use as a probe: every stage
(allocation, handle export, peer mapping) is agreed on collectively, then a self-test compares all-reduces of every
algorithm with the bootstrap backend's (RCCL) results; any rank failing any stage makes ALL ranks fall back to RCCL,
loudly.  A peer that goes missing later is never papered over: the kernel poisons its output with NaN and sets the
status word; the decode loops read the status word with every iteration's flag read (`status_async`) and raise at
the iteration it happened in, and `check(collective=True)` at the end of a batch makes every rank raise together."""
from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist

from .. import _lib
from .._lib import check

HANDLE_BYTES = 64           # MD_AR_HANDLE_BYTES
DEFAULT_MAX_BYTES = 4 << 20
ALGO_AUTO, ALGO_ONESHOT, ALGO_TWOSHOT = 0, 1, 2      # MD_AR_ALGO_*
PUBLISH_WRITE_THROUGH, PUBLISH_FENCE = 0, 1          # MD_AR_PUBLISH_*


def publish_mode_from_env() -> int:
    """MAGICDEC_AR_PUBLISH=fence: the release-fence publish (include/magicdec_hip.h, md_ar_set_publish) -- the arm a
    multi-GPU run falls back to when the write-through hand-off fails its bit-exact stress on real links (bench.py);
    unset / "wt": the write-through publish."""
    v = os.environ.get("MAGICDEC_AR_PUBLISH", "wt").lower()
    if v not in ("wt", "write_through", "fence"):
        raise ValueError(f"MAGICDEC_AR_PUBLISH must be 'wt' or 'fence', got '{v}'")
    return PUBLISH_FENCE if v == "fence" else PUBLISH_WRITE_THROUGH


class AllReduceTimeout(RuntimeError):
    pass


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
        self.publish = PUBLISH_WRITE_THROUGH
        self.set_publish(publish_mode_from_env())
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

    def set_publish(self, mode: int):
        """PUBLISH_WRITE_THROUGH | PUBLISH_FENCE; every rank of the group must choose the same (it is part of the
        protocol's timing, not of its data layout, so a mixed group would still be correct -- only slower on one side)."""
        check(self.lib.md_ar_set_publish(self.comm, int(mode)), "md_ar_set_publish")
        self.publish = int(mode)

    def fits(self, t: torch.Tensor) -> bool:
        return (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() % 8 == 0
                and t.numel() * 2 <= self.max_bytes and t.data_ptr() % 16 == 0)

    def all_reduce_(self, t: torch.Tensor, algo: int = ALGO_AUTO) -> torch.Tensor:
        """In-place sum over the group (same result bits on every rank)."""
        if not self.fits(t):
            raise ValueError("OneShotAllReduce: tensor must be contiguous bf16 on the GPU, numel % 8 == 0, and fit "
                             f"the registered buffer ({self.max_bytes} bytes)")
        p = ctypes.c_void_p(t.data_ptr())
        check(self.lib.md_allreduce(self.comm, p, p, t.numel(), int(algo),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "md_allreduce")
        return t

    def fits_fused(self, partial: torch.Tensor, weight: torch.Tensor) -> bool:
        return (partial.dim() == 2 and self.fits(partial) and partial.shape[1] % 8 == 0 and partial.shape[1] <= 8192
                and weight.is_contiguous() and weight.dtype == torch.bfloat16)

    def all_reduce_add_rmsnorm(self, partial, resid, weight, eps, algo: int = ALGO_AUTO):
        """(h, y) with h = resid + all_reduce(partial) (bf16 add) and y = rmsnorm(h) * weight, one launch."""
        rows, dim = partial.shape
        if not self.fits_fused(partial, weight) or not resid.is_contiguous() or resid.shape != partial.shape:
            raise ValueError("all_reduce_add_rmsnorm: [rows, dim] contiguous bf16 tensors, dim % 8 == 0, dim <= 8192")
        h = torch.empty_like(partial)
        y = torch.empty_like(partial)
        pv = lambda t: ctypes.c_void_p(t.data_ptr())
        check(self.lib.md_allreduce_add_rmsnorm(self.comm, pv(partial), pv(resid), pv(weight), pv(h), pv(y), rows, dim,
                                                float(eps), int(algo),
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "md_allreduce_add_rmsnorm")
        return h, y

    def check(self, collective: bool = False):
        """Raise if any call since the last check gave up waiting for a peer (its output rows are NaN).
        Synchronises the device.  collective=True (the decode loops' end-of-batch check; every rank of the group must
        call it): the status words are max-reduced over the bootstrap backend first, so that all ranks raise together
        instead of one rank raising while the others wait for it in their next collective."""
        bad = self.status()
        if collective and self.world > 1:
            dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
            t = torch.tensor([bad], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            bad = int(t.item())
        if bad != 0:
            raise AllReduceTimeout(f"rank {self.rank}: an xGMI all-reduce timed out waiting for a peer (on this rank or "
                                   "another rank of its group); the affected hidden states were poisoned with NaN -- "
                                   "the results are invalid")

    def status_async(self):
        """Queue a 4-byte copy of the status word into pinned host memory on the current stream and return the pinned
        tensor: after the stream's next synchronisation `int(t[0]) != 0` means a time-out.  The decode loops piggyback
        this on the one host read of an iteration (harness._read_flags), so a time-out stops the loop at the iteration
        it happened in instead of iterating on NaN hidden states until the end of the batch."""
        if getattr(self, "_status_host", None) is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        check(self.lib.md_ar_status_async(self.comm, ctypes.c_void_p(self._status_host.data_ptr()),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "md_ar_status_async")
        return self._status_host

    def self_test(self) -> bool:
        """Local verdict: all-reduces of every algorithm -- plain AND fused with the residual add + RMSNorm, the form the
        decode path actually runs -- agree with the bootstrap backend's (different summation order, so a bf16
        tolerance) and no spin timed out.  Collective (all ranks must call it)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        ok = True
        for k, n in enumerate((2048, 64 * 2048, min(256 * 4096, self.max_bytes // 2))):
            for rep, algo in enumerate((ALGO_ONESHOT, ALGO_TWOSHOT, ALGO_ONESHOT, ALGO_TWOSHOT)):   # both buffer halves
                g = torch.Generator(device=dev).manual_seed(1000 * k + 10 * rep + self.rank)
                x = torch.randn(n, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
                ref = x.clone()
                dist.all_reduce(ref, group=self.group)
                y = x.clone()
                self.all_reduce_(y, algo)
                torch.cuda.synchronize()
                tol = 2.0 ** -6 * float(ref.float().abs().max()) + 1e-3
                ok = ok and bool((y.float() - ref.float()).abs().max() <= tol) and bool(torch.isfinite(y.float()).all())
        # the decode path runs the FUSED instantiations (Transformer._reduce_add_norm): validate them too, against the
        # bootstrap backend's all-reduce followed by the add + RMSNorm kernel (h: bf16 tolerance of the summation order;
        # y: the same tolerance relative to |y|)
        from .. import ops
        for k, (rows, dim) in enumerate(((64, 2048), (256, 4096))):
            if rows * dim * 2 > self.max_bytes:
                continue
            for rep, algo in enumerate((ALGO_ONESHOT, ALGO_TWOSHOT, ALGO_ONESHOT, ALGO_TWOSHOT)):
                g = torch.Generator(device=dev).manual_seed(5000 + 100 * k + 10 * rep + self.rank)
                part = torch.randn(rows, dim, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
                g2 = torch.Generator(device=dev).manual_seed(7000 + 100 * k + rep)        # same on every rank
                resid = torch.randn(rows, dim, device=dev, generator=g2, dtype=torch.float32).to(torch.bfloat16)
                w = (1.0 + 0.1 * torch.randn(dim, device=dev, generator=g2, dtype=torch.float32)).to(torch.bfloat16)
                ref = part.clone()
                dist.all_reduce(ref, group=self.group)
                h_ref, y_ref = ops.add_rmsnorm(resid, ref, w, 1e-5)
                h, y = self.all_reduce_add_rmsnorm(part, resid, w, 1e-5, algo)
                torch.cuda.synchronize()
                for got, want in ((h, h_ref), (y, y_ref)):
                    tol = 2.0 ** -5 * float(want.float().abs().max()) + 1e-3
                    ok = (ok and bool(torch.isfinite(got.float()).all())
                          and bool((got.float() - want.float()).abs().max() <= tol))
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
