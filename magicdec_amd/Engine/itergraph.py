"""One hipGraph for a WHOLE speculative iteration (gamma draft steps + the verify pass + the fused accept/rollback).

`Engine/graph.py` captures each decode step; an iteration is then 4 graph launches plus ~25 tiny eager launches of the
harness (token-buffer slices, length updates, the accept kernel) with host gaps between them: 0.4-0.6 ms of a 31 ms
iteration on one GPU (profiles/r01_bench_cfg3_iter_breakdown.csv: "idle"), and the part of an iteration that does not
shrink under tensor parallelism.  Nothing in an iteration reads device memory on the host -- the one host read (the
terminal / next_double flags) happens AFTER it -- so the whole body is capturable: one launch per iteration.

MEASURED (profiles/r02_ab_iteration_graph.txt, same box, back to back): on one GPU the
whole-iteration graph is NOT faster -- 31.53-31.67 ms against 31.51-31.64 ms per iteration with per-step graphs (the host already
runs ahead of the GPU; the per-node cost inside one large graph is the same) and each body costs a ~25 ms capture.
It is therefore OFF by default (MAGICDEC_ITER_GRAPH=1 turns it on; graphs == eager is tested with it on) and kept for
the tensor-parallel case, where a rank's kernels are 3-8x shorter and the host-side launch rate is the limit.

There are two bodies per loop (the first draft step consumes one token, or two after an all-accept iteration:
tests/SnapKV/longspec_benchmark.py:165-188), selected by the flag the host read at the end of the previous iteration.

Graphs bake addresses.  Every tensor an iteration reads or updates in place is therefore a STATIC tensor shared by all
captured bodies (for the back-ends: the same statics `Engine/graph.py` uses for its step graphs); before a replay any
attribute that was rebound by the caller (encode() / clear_kv() create fresh length tensors, the reference's harness
rebinds them too) is copied into its static tensor and bound to it.  A body may itself rebind an attribute to a tensor
it creates (`paged_kv_last_page_len = paged_kv_last_page_len - n + cachelen_update`, backend_draft.py:166-168): python
does not run on a replay, so the bindings found after the capture are re-applied after every replay.
"""
from __future__ import annotations

import os
import warnings

import torch

from . import graph as stepgraph

LOOP_FIELDS = ("tokens_buffer", "output", "num_nodes", "accept_nums", "bonus", "double_buffer", "cachelens_update",
               "flags")
CAPTURE_AFTER = 2          # iterations of a kind run with step graphs before its body is captured (everything warm)


def enabled(*backends) -> bool:
    if os.environ.get("MAGICDEC_ITER_GRAPH", "0") != "1":
        return False
    bs = [b for b in backends if b is not None]
    return bool(bs) and all(getattr(b, "_use_graphs", False) and torch.device(b.device).type == "cuda" for b in bs)


def _backend_names(b):
    names = []
    for kind_names in stepgraph._STATE_BY_KIND.values():
        for n in kind_names:
            if n not in names and getattr(b, n, None) is not None:
                names.append(n)
    return names


class _Body:
    def __init__(self):
        self.graph = None
        self.forced = None
        self.post = []          # (owner, name, tensor): attribute bindings the body itself makes
        self.seen = 0


class IterationGraphs:
    """Per (engine, draft) pair; lives on the engine object."""

    def __init__(self, engine, draft):
        self.engine, self.draft = engine, draft
        self.bodies = {}
        self.loop_static = {}
        self.failed = False

    # ---- static binding
    def _owners(self, st):
        out = [(b, _backend_names(b), stepgraph._statics(b)) for b in (self.engine, self.draft) if b is not None]
        out.append((st, [n for n in LOOP_FIELDS if getattr(st, n, None) is not None], self.loop_static))
        return out

    def _bind(self, st):
        src, dst = [], []
        for owner, names, statics in self._owners(st):
            for n in names:
                cur = getattr(owner, n)
                s = statics.get(n)
                if s is None or s.shape != cur.shape or s.dtype != cur.dtype:
                    if s is not None:
                        self.bodies.clear()          # geometry changed: every baked address is void
                    statics[n] = s = cur.clone()
                    setattr(owner, n, s)
                elif cur is not s:
                    src.append(cur)
                    dst.append(s)
                    setattr(owner, n, s)
        if src:
            torch._foreach_copy_(dst, src)

    def _bindings(self, st):
        return {(id(o), n): getattr(o, n) for o, names, _ in self._owners(st) for n in names}

    # ---- run
    def run(self, key, st, body, forced):
        """Runs `body(forced)` for this iteration -- eagerly (step graphs) the first CAPTURE_AFTER times a key is seen,
        as one graph replay afterwards.  Returns nothing; the caller reads the flags."""
        if self.failed:
            return body(forced)
        ent = self.bodies.get(key)
        if ent is None:
            ent = self.bodies[key] = _Body()
        if ent.graph is None:
            ent.seen += 1
            if ent.seen <= CAPTURE_AFTER:
                return body(forced)
            self._bind(st)
            if self.bodies.get(key) is not ent:      # _bind found a new geometry and dropped the bodies
                ent = self.bodies[key] = _Body()
                ent.seen = CAPTURE_AFTER + 1
            ent.forced = forced.clone() if forced is not None else None
            before = self._bindings(st)
            flags = [(b, b._use_graphs) for b in (self.engine, self.draft) if b is not None]
            for b, _ in flags:
                b._use_graphs = False                # the steps run inline inside the capture
            g = torch.cuda.CUDAGraph()
            try:
                torch.cuda.synchronize()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    body(ent.forced)
            except Exception as e:  # noqa: BLE001 -- e.g. a collective this RCCL build cannot capture
                torch.cuda.synchronize()
                for b, f in flags:
                    b._use_graphs = f
                warnings.warn(f"[magicdec_amd] hipGraph capture of a whole iteration failed ({type(e).__name__}: {e}); "
                              "continuing with per-step graphs", RuntimeWarning, stacklevel=2)
                self.failed = True
                self.bodies.clear()
                # the capture executed nothing but python-side rebindings happened: restore them
                for owner, names, _ in self._owners(st):
                    for n in names:
                        setattr(owner, n, before[(id(owner), n)])
                return body(forced)
            for b, f in flags:
                b._use_graphs = f
            after = self._bindings(st)
            owners = {id(o): o for o, _, _ in self._owners(st)}
            ent.post = [(owners[oid], n, t) for (oid, n), t in after.items() if t is not before[(oid, n)]]
            ent.graph = g
        else:
            self._bind(st)
            if self.bodies.get(key) is not ent:      # geometry changed under us: start over for this key
                return self.run(key, st, body, forced)
        if forced is not None:
            ent.forced.copy_(forced)
        ent.graph.replay()
        for owner, n, t in ent.post:
            setattr(owner, n, t)


def get(engine, draft) -> IterationGraphs:
    ig = getattr(engine, "_iter_graphs", None)
    if ig is None or ig.draft is not draft:
        ig = engine._iter_graphs = IterationGraphs(engine, draft)
    return ig
