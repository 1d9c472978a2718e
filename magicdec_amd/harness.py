"""The speculative draft/verify decode loops (L5 of SURVEY.md section 1) on the GPU.

These are the loop bodies of the reference's benchmark scripts
(tests/SnapKV/longspec_benchmark.py:131-295 and its StreamingLLM twin,
tests/*/selfspec_benchmark.py, tests/baseline_benchmark.py:72-90) with the ~15
per-iteration ATen launches and 4 host syncs of the verify loop replaced by ONE
integer kernel (ops.accept_rollback) and ONE 8-byte flag read per iteration.
The Engine attributes are updated in place exactly as the reference harness
updates them, including its quirks (e.g. the SnapKV longspec harness rolls back
`draft.paged_kv_last_page_len`, which the compressed draft does not use).

Used by the CLI scripts under tests/, by bench.py and by the parity tests.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field

import torch

from . import ops


@dataclass
class LoopState:
    """Per-batch device state of the spec-decode loop."""
    tokens_buffer: torch.Tensor
    output: torch.Tensor
    num_nodes: torch.Tensor
    accept_nums: torch.Tensor
    bonus: torch.Tensor
    double_buffer: torch.Tensor
    cachelens_update: torch.Tensor
    flags: torch.Tensor
    flags_host: torch.Tensor
    iters: int = 0
    accept_trace: list = field(default_factory=list)
    # called (with the state) after an iteration's accept kernel was queued and BEFORE the host read: device-side
    # statistics queued here cost the host nothing on the critical path -- the GPU is still busy with the verify step
    before_host_read: object = None
    _flags_event: object = None


def new_state(B, gamma, out_cols, device, input_ids=None):
    st = LoopState(tokens_buffer=torch.zeros((B, gamma + 1), dtype=torch.long, device=device),
                   output=torch.zeros((B, out_cols), dtype=torch.long, device=device),
                   num_nodes=torch.zeros(B, dtype=torch.long, device=device),
                   accept_nums=torch.zeros(B, dtype=torch.long, device=device),
                   bonus=torch.zeros(B, dtype=torch.long, device=device),
                   double_buffer=torch.zeros((B, 2), dtype=torch.long, device=device),
                   cachelens_update=torch.ones(B, dtype=torch.long, device=device),
                   flags=torch.zeros(2, dtype=torch.int32, device=device),
                   flags_host=(torch.zeros(2, dtype=torch.int32).pin_memory() if torch.device(device).type == "cuda"
                               else torch.zeros(2, dtype=torch.int32)))
    if input_ids is not None:
        st.output[:, :input_ids.shape[1]] = input_ids
        st.num_nodes += input_ids.shape[1]
    return st


# MAGICDEC_HOST_READ_SPIN=0: block in hipStreamSynchronize for the iteration's host read instead of polling an event
_SPIN_ON_HOST_READ = os.environ.get("MAGICDEC_HOST_READ_SPIN", "1") != "0"


def _sync(t):
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def _read_flags(st: LoopState, collectives=()):
    """The one host read of an iteration: (terminal, next_double).  With xGMI all-reduce communicators attached
    (`collectives`, Engine/oneshot.py) their time-out status words ride on the same stream synchronisation, and a
    time-out raises HERE -- at the iteration whose hidden states were poisoned, not at the end of the batch."""
    if st.flags.is_cuda:
        st.flags_host.copy_(st.flags, non_blocking=True)
        status = [ar.status_async() for ar in collectives]
        # the GPU idles from here until the next iteration's first launch: poll an event instead of blocking in
        # hipStreamSynchronize (whose wake-up alone was most of a 93 us gap per iteration,
        # profiles/r04_iteration_gaps.txt)
        if not _SPIN_ON_HOST_READ:
            torch.cuda.current_stream().synchronize()
        else:
            if st._flags_event is None:
                st._flags_event = torch.cuda.Event()
            ev = st._flags_event
            ev.record()
            while not ev.query():
                time.sleep(0)           # give up the GIL and the time slice between polls (with 8 ranks and
                #                         their RCCL proxy threads on one host a hard spin would pin a core per rank)
        for ar, s in zip(collectives, status):
            if int(s[0]) != 0:
                from .Engine.oneshot import AllReduceTimeout
                raise AllReduceTimeout(f"rank {ar.rank}: an xGMI all-reduce timed out waiting for a peer during "
                                       f"iteration {st.iters}; its hidden states were poisoned with NaN")
    else:
        st.flags_host.copy_(st.flags)
    return bool(st.flags_host[0]), bool(st.flags_host[1])


def _collectives_of(*engines):
    """The xGMI all-reduce communicators attached to the engines' models (tp.apply_tp, MAGICDEC_ONESHOT_AR=1)."""
    out = []
    for e in engines:
        ar = getattr(getattr(e, "model", None), "_oneshot", None) if e is not None else None
        if ar is not None and ar not in out:
            out.append(ar)
    return tuple(out)


class PhaseTimers:
    """--benchmark of the reference scripts (tests/SnapKV/longspec_benchmark.py:159-203,305-307): synchronise around
    the draft phase, the target verification and the verify loop and accumulate their wall times."""

    def __init__(self, device):
        self.device = device
        self.draft = self.target = self.verify_loop = 0.0
        self._t = 0.0

    def reset(self):
        self.draft = self.target = self.verify_loop = 0.0

    def start(self):
        _sync_dev(self.device)
        self._t = time.time()

    def lap(self, what):
        _sync_dev(self.device)
        now = time.time()
        setattr(self, what, getattr(self, what) + now - self._t)
        self._t = now


def _sync_dev(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def warn_on_page_overflow(where):
    """Rows dropped by the append kernels because a request's last page was full (the reference's page tables do not
    grow during decode; there flashinfer would have written into the next request's page).  Read once per batch."""
    n = ops.page_overflow_count(reset=True)
    if n:
        import warnings
        warnings.warn(f"{where}: {n} KV rows fell beyond their request's mapped pages and were dropped "
                      f"(last_page_len grew past the page size -- the decode loop outran the page table)",
                      RuntimeWarning, stacklevel=2)
    return n


def check_collectives(*engines):
    """End-of-batch check, collective over each communicator's group: raise on EVERY rank if an xGMI all-reduce of any
    rank timed out (Engine/oneshot.py; the per-iteration read in _read_flags catches a rank's own time-outs earlier)."""
    for ar in _collectives_of(*engines):
        ar.check(collective=True)


def _draft_round(step_fn, st: LoopState, gamma, next_double):
    """gamma draft steps; after an all-accept iteration the first step consumes two tokens
    (tests/SnapKV/longspec_benchmark.py:165-188)."""
    for i in range(gamma):
        if i == 0 and next_double:
            nt = step_fn(st.double_buffer, st.cachelens_update)
            st.tokens_buffer[:, 1:2] = nt.gather(1, st.cachelens_update.view(-1, 1) - 1)
        else:
            st.tokens_buffer[:, i + 1:i + 2] = step_fn(st.tokens_buffer[:, i].view(-1, 1), None)


def _iterate(engine, draft, st: LoopState, body, forced):
    """Runs one iteration body (each decode step inside it is one hipGraph replay when the back-ends were compile()d)
    and makes the iteration's one host read.  (A whole-iteration graph was built and measured in rounds 2 and 3: 31.6
    vs 31.6 ms at TP1, 8.77 vs 8.85 ms for a TP8 rank's compute -- the host already runs ahead of the GPU -- and was
    removed: profiles/r02_ab_iteration_graph.txt, profiles/r03_ab_iteration_graph_tp8.txt.)"""
    body(forced)
    st.iters += 1
    if st.before_host_read is not None:
        st.before_host_read(st)
    return _read_flags(st, _collectives_of(engine, draft))


def longspec_iteration(engine, draft, st: LoopState, gamma, eot_1, eot_2, max_nodes, next_double,
                       forced_accept=None, bcast=None, timers: PhaseTimers = None):
    """One iteration of the longspec loop: gamma draft steps, one verify, the fused accept/rollback.
    Returns (terminal, next_double).

    Tensor parallel (tests/SnapKV/longspec_benchmark.py:163-189): `draft` is None on ranks outside the draft
    sub-group; when the draft group is smaller than the target group the gamma draft tokens are broadcast from
    `bcast = (src_rank, group)` before the verify.  Every rank runs the (replicated, all-integer) accept kernel."""
    def body(forced):
        if timers is not None:
            timers.start()
        if draft is not None:
            _draft_round(lambda ids, cu: draft.inference(ids, cachelen_update=cu), st, gamma, next_double)
        if bcast is not None:
            import torch.distributed as dist
            dist.broadcast(st.tokens_buffer, src=bcast[0], group=bcast[1])
        if timers is not None:
            timers.lap("draft")
        target_tokens = engine.inference(st.tokens_buffer)
        if timers is not None:
            timers.lap("target")
        if forced is not None:
            target_tokens = _force_accept(st.tokens_buffer, target_tokens, forced, gamma)
        ops.accept_rollback(st.tokens_buffer, target_tokens, st.output, st.num_nodes, engine.cachelens,
                            engine.paged_kv_last_page_len, draft.cachelens if draft is not None else None,
                            draft.paged_kv_last_page_len if draft is not None else None, gamma,
                            gamma, gamma, eot_1, eot_2, max_nodes, st.accept_nums, st.bonus, st.double_buffer,
                            st.cachelens_update, st.flags)
    res = _iterate(engine, draft, st, body, forced_accept)
    if timers is not None:
        timers.lap("verify_loop")
    return res


def selfspec_iteration(engine, st: LoopState, gamma, eot_1, eot_2, max_nodes, next_double, streaming,
                       forced_accept=None, timers: PhaseTimers = None):
    """One iteration of tests/SnapKV/selfspec_benchmark.py:121-211 (streaming=False: draft rolled back by
    gamma+1 and advanced by accept_nums, no two-token step) or tests/StreamingLLM/selfspec_benchmark.py:121-238."""
    def body(forced):
        if timers is not None:
            timers.start()
        if streaming:
            _draft_round(lambda ids, cu: engine.speculate(ids, cachelen_update=cu), st, gamma, next_double)
        else:
            _draft_round(lambda ids, cu: engine.speculate(ids), st, gamma, False)
        if timers is not None:
            timers.lap("draft")
        target_tokens = engine.verify(st.tokens_buffer)
        if timers is not None:
            timers.lap("target")
        if forced is not None:
            target_tokens = _force_accept(st.tokens_buffer, target_tokens, forced, gamma)
        if streaming:
            ops.accept_rollback(st.tokens_buffer, target_tokens, st.output, st.num_nodes, engine.cachelens,
                                engine.paged_kv_last_page_len, engine.draft_cachelens,
                                engine.draft_paged_kv_last_page_len, gamma, gamma, gamma, eot_1, eot_2, max_nodes,
                                st.accept_nums, st.bonus, st.double_buffer, st.cachelens_update, st.flags)
        else:
            ops.accept_rollback(st.tokens_buffer, target_tokens, st.output, st.num_nodes, engine.cachelens,
                                engine.paged_kv_last_page_len, engine.draft_cachelens,
                                engine.draft_paged_kv_last_page_len, gamma, gamma + 1, gamma + 1, eot_1, eot_2,
                                max_nodes, st.accept_nums, st.bonus, None, None, st.flags)
    res = _iterate(engine, None, st, body, forced_accept)
    if timers is not None:
        timers.lap("verify_loop")
    return res


def _force_accept(tokens_buffer, target_tokens, forced_accept, gamma):
    """Fixed-acceptance replay: with random-init weights the measured acceptance is
    meaningless, so benchmarks may replace the target's tokens by ones that accept exactly
    forced_accept[b]-1 draft tokens of row b (all the draft/verify/accept work still runs)."""
    tt = target_tokens.clone()
    j = torch.arange(gamma, device=tt.device).view(1, -1)
    keep = j < (forced_accept.view(-1, 1) - 1)
    draft = tokens_buffer[:, 1:gamma + 1]
    tt[:, :gamma] = torch.where(keep, draft, draft + 1)
    return tt


def run_longspec_batch(engine, draft, input_ids, gamma, max_len, eot_1, eot_2, forced_accept_fn=None,
                       trace_fn=None, bcast=None, barrier=None, timers: PhaseTimers = None):
    """A whole batch: prefill both models, loop until termination.  Returns (state, seconds in the loop)."""
    B, S = input_ids.shape
    st = new_state(B, gamma, max_len + 1, input_ids.device, input_ids)
    st.tokens_buffer[:, :1] = engine.encode(input_ids=input_ids)[:, -1:]
    if draft is not None:
        draft.encode(input_ids=input_ids)
    if barrier is not None:
        barrier()
    _sync(input_ids)
    t0 = time.perf_counter()
    terminal, nd = False, False
    while not terminal:
        fa = forced_accept_fn(st) if forced_accept_fn is not None else None
        terminal, nd = longspec_iteration(engine, draft, st, gamma, eot_1, eot_2, S + 80, nd, fa, bcast, timers)
        if trace_fn is not None:
            trace_fn(st)
    _sync(input_ids)
    dt = time.perf_counter() - t0
    if input_ids.is_cuda:
        warn_on_page_overflow("longspec batch")
        check_collectives(engine, draft)
    return st, dt


def run_selfspec_batch(engine, input_ids, gamma, max_len, eot_1, eot_2, streaming, forced_accept_fn=None,
                       trace_fn=None, timers: PhaseTimers = None):
    B, S = input_ids.shape
    st = new_state(B, gamma, max_len + 1, input_ids.device, input_ids)
    st.tokens_buffer[:, :1] = engine.encode(input_ids=input_ids)[:, -1:]
    if streaming:
        engine.draft_encode(input_ids=input_ids)
    _sync(input_ids)
    t0 = time.perf_counter()
    terminal, nd = False, False
    while not terminal:
        fa = forced_accept_fn(st) if forced_accept_fn is not None else None
        terminal, nd = selfspec_iteration(engine, st, gamma, eot_1, eot_2, S + 80, nd, streaming, fa, timers)
        if trace_fn is not None:
            trace_fn(st)
    _sync(input_ids)
    dt = time.perf_counter() - t0
    if input_ids.is_cuda:
        warn_on_page_overflow("selfspec batch")
        check_collectives(engine)
    return st, dt


def run_baseline_batch(engine, input_ids, max_len, eot_1, eot_2, check_eot_every=1):
    """tests/baseline_benchmark.py:72-90: greedy autoregressive decode until max_len or EOT."""
    output = input_ids.clone()
    next_tokens = engine.encode(input_ids=input_ids)[:, -1:]
    output = torch.cat((output, next_tokens), dim=-1)
    _sync(input_ids)
    t0 = time.perf_counter()
    steps = 0
    terminate = False
    while output.size(1) < max_len and not terminate:
        next_tokens = engine.inference(input_ids=next_tokens.clone())
        output = torch.cat((output, next_tokens), dim=-1)
        steps += 1
        if steps % check_eot_every == 0:
            last = next_tokens[:, -1]
            terminate = bool(((last == eot_1) | (last == eot_2)).any())
    _sync(input_ids)
    dt = time.perf_counter() - t0
    if input_ids.is_cuda:
        warn_on_page_overflow("baseline batch")
        check_collectives(engine)
    return output, steps, dt
