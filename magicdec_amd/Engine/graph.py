"""hipGraph capture of the decode steps (the MI355X counterpart of the reference's `compile()`:
torch.compile(mode="max-autotune") + CUDA graphs, Engine/SnapKV/backend.py:116-125).

A draft step of a 1B model is ~150 kernel launches for ~0.6 ms of HBM time: eager launching is host-bound
(MI355X_MICROARCH.md "graph-replay-floor").  Every kernel of a step reads its lengths / page table from device
memory at execution time and nothing in a step touches the host, so a step is captured once per
(kind, n_tokens) and replayed.

The harness may REBIND the length tensors between calls (tests/SnapKV/longspec_benchmark.py:228-256).  Graphs
bake addresses, so each captured step owns static copies of the state tensors it reads; before a replay any
attribute that is no longer the static tensor is copied into it and re-bound to it (same values, so the
reference's semantics are unchanged); in the steady state of our own loop (in-place updates) nothing is copied.
"""
from __future__ import annotations

import warnings

import torch

_STATE_BY_KIND = {
    "fwd": ("cachelens", "qo_indptr", "paged_kv_indices", "paged_kv_indptr", "paged_kv_last_page_len"),
    "draft": ("cachelens", "qo_indptr", "draft_paged_kv_indices", "draft_paged_kv_indptr",
              "draft_paged_kv_last_page_len"),
    "spec": ("draft_cachelens", "qo_indptr", "draft_paged_kv_indices", "draft_paged_kv_indptr",
             "draft_paged_kv_last_page_len"),
    "verify": ("cachelens", "qo_indptr", "paged_kv_indices", "paged_kv_indptr", "paged_kv_last_page_len",
               "draft_paged_kv_indices", "draft_paged_kv_indptr", "draft_paged_kv_last_page_len"),
}


class _Captured:
    def __init__(self):
        self.graph = None
        self.static_in = None
        self.static_out = None
        self.names = ()       # state attributes this step reads


def _statics(backend):
    """attr name -> static tensor, shared by all captured steps of one back-end."""
    if not hasattr(backend, "_gstate"):
        backend._gstate = {}
    return backend._gstate


def _bind_state(backend, ent: _Captured):
    """Make every state attribute the static tensor (copying the current values if it was rebound)."""
    src, dst = [], []
    gs = _statics(backend)
    for name in ent.names:
        st = gs[name]
        cur = getattr(backend, name)
        if cur is not st:
            if cur.shape != st.shape:
                raise RuntimeError(f"captured step: '{name}' changed shape {tuple(st.shape)} -> {tuple(cur.shape)}; "
                                   "call clear_graphs() after setup_caches/encode with a new batch geometry")
            src.append(cur)
            dst.append(st)
            setattr(backend, name, st)
    if src:
        torch._foreach_copy_(dst, src)


def _all_ranks_ok(backend, ok: bool) -> bool:
    """True iff the capture succeeded on every rank of the back-end's tensor-parallel group (trivially `ok` without
    one).  Collective over that group: every rank reaches it once per newly captured step."""
    group = getattr(getattr(backend, "model", None), "process_group", None)
    if group is None:
        return ok
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(t, group=group)
    return int(t.item()) == 0


_FAILED_GRAPHS = []      # see _recover_from_failed_capture: graph objects of failed captures are never destroyed
_CAPTURE_POISONED = [None]   # set to the reason once a capture was INVALIDATED in this process: this PyTorch build then
#                              refuses every later capture ("Cannot register the state during capturing stage": the
#                              default generator still believes it is being captured), so later steps do not try


def _recover_from_failed_capture(g, home_stream):
    """Leave the process able to launch eagerly after an invalidated capture.  `torch.cuda.graph.__exit__` ends the capture
    first and restores the stream second: when the end itself raises (hipErrorStreamCaptureInvalidated) the capture stream
    stays current, and the runtime's sticky error fails the next launch check ("operation failed due to a previous error
    during capture" -- seen in the 2-rank rehearsal on one GPU, where gloo's collectives cannot be captured).  So: end the
    capture again if it is still open, go back to the caller's stream, read the sticky error away, drain the device.
    The graph object itself is LEAKED on purpose: the destructor of a `torch.cuda.CUDAGraph` whose capture was invalidated
    throws ("The graph should be registered to the state", HIPGeneratorImpl::unregister_graph) -- from a destructor, i.e.
    `terminate` -- whenever the garbage collector gets to it (reproduced: tests/test_gpu_engine.py::
    test_a_failed_graph_capture_falls_back_to_eager_and_keeps_working)."""
    import ctypes
    _FAILED_GRAPHS.append(g)
    ctypes.pythonapi.Py_IncRef(ctypes.py_object(g))      # never deallocated, not even at interpreter shutdown
    try:
        if torch.cuda.is_current_stream_capturing():
            g.capture_end()
    except Exception:  # noqa: BLE001 -- the end of an invalidated capture reports the invalidation once more
        pass
    torch.cuda.set_stream(home_stream)
    from .. import _lib
    for _ in range(4):                      # one read per queued error is enough; a few in case both calls left one
        try:
            if _lib.load().md_clear_last_hip_error() == 0:
                break
        except Exception:  # noqa: BLE001
            break
    try:
        torch.cuda.synchronize()
    except Exception:  # noqa: BLE001 -- torch's own check may still see (and thereby clear) the error
        try:
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001
            pass


def run_captured(backend, key, fn, input_ids):
    kind = key[0]
    full_key = (key, tuple(input_ids.shape), tuple(getattr(backend, "paged_kv_indices").shape),
                tuple(getattr(backend, "draft_paged_kv_indices").shape) if hasattr(backend, "draft_paged_kv_indices")
                else None)
    ent = backend._graphs.get(full_key)
    if ent is None:
        ent = _Captured()
        gs = _statics(backend)
        names = []
        for name in _STATE_BY_KIND[kind]:
            cur = getattr(backend, name, None)
            if cur is None:
                continue
            if name not in gs or gs[name].shape != cur.shape:
                gs[name] = cur.clone()
            names.append(name)
        ent.names = tuple(names)
        _bind_state(backend, ent)
        ent.static_in = input_ids.clone()
        # warm up on a side stream (hipBLASLt workspaces, lazy module loads, LDS attributes); a step never
        # changes the lengths it reads, and re-running it rewrites the same KV rows with the same values
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                fn(ent.static_in)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        err = None
        home = torch.cuda.current_stream()
        if _CAPTURE_POISONED[0] is not None:
            # (the warm-up above and the agreement below still run: the ranks of a TP group must issue the same collectives)
            err = RuntimeError(f"an earlier capture in this process was invalidated ({_CAPTURE_POISONED[0]}); this PyTorch "
                               "build cannot capture again afterwards")
        else:
            try:
                # thread_local: the RCCL watchdog thread of torch.distributed polls hipEventQuery concurrently; in the
                # default "global" capture mode such a call from ANY thread invalidates the capture and kills the process
                # ("operation not permitted when stream is capturing", seen intermittently with a TP group)
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    ent.static_out = fn(ent.static_in)
            except Exception as e:  # noqa: BLE001 -- e.g. a collective this RCCL build (or gloo) cannot capture
                err = e
                invalidated = "capture" in str(e).lower() and "hip error" in str(e).lower()
                _recover_from_failed_capture(g, home)
                if invalidated:
                    _CAPTURE_POISONED[0] = f"step {key}: {type(e).__name__}"
        # Under tensor parallelism every rank captures the same step at the same point of the same program, but a
        # capture can fail on ONE rank only (an allocation, a watchdog race): a rank replaying a graph while another
        # launches eagerly still issues the same collectives in the same order, yet the choice must not depend on
        # luck -- the ranks agree on the outcome (one tiny all-reduce, outside any capture) and fall back together.
        if not _all_ranks_ok(backend, err is None):
            # Capture executes nothing and the warm-up runs above are idempotent, so the step can still be run
            # eagerly.  Loud, once: silently running eager would misreport what was measured.
            why = f"{type(err).__name__}: {err}" if err is not None else "it failed on another rank of the TP group"
            warnings.warn(f"[magicdec_amd] hipGraph capture of step {key} failed ({why}); this back-end continues "
                          "WITHOUT graphs on every rank", RuntimeWarning, stacklevel=2)
            backend._use_graphs = False
            backend._graphs.clear()
            del g
            return fn(input_ids)
        ent.graph = g
        backend._graphs[full_key] = ent
    _bind_state(backend, ent)
    if input_ids.data_ptr() != ent.static_in.data_ptr():
        ent.static_in.copy_(input_ids)
    ent.graph.replay()
    return ent.static_out.clone()


def clear_graphs(backend):
    backend._graphs.clear()
    if hasattr(backend, "_gstate"):
        backend._gstate.clear()
