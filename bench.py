"""Benchmark of the MagicDec draft/verify decode loop on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "decode tokens/s/node + speedup vs autoregressive, Llama-3.1-8B B=64
prefix=16K" == configs[2], the tests/SnapKV/longspec_benchmark.py command of the reference's README.md:69):
Llama-3.1-8B target (TP = N) + Llama-3.2-1B SnapKV draft (budget 257, TP = min(N,4)), gamma=3, B=64,
prefix_len 16032, max_len 16128, bf16, seeded random-init weights of those architectures (no checkpoints on
the box), synthetic PG-19-shaped token batch.  It fits one MI355X (126 GiB target KV + 32 GiB draft KV +
17 GiB weights), so the N=1 line is this exact configuration at TP=1.

A "step" is one speculative iteration: gamma draft steps, one (gamma+1)-token verify over the full KV, the
fused accept/rollback.  Prefill (outside the reference's timed region too) runs once before the clock starts;
when a batch reaches prefix+80 generated tokens the length counters are restored to their post-prefill values
(the KV rows beyond the prefix are overwritten), exactly what the next batch would see.

Acceptance: random-init weights make measured acceptance meaningless (~0), so `value` uses the fixed-acceptance
replay of SURVEY.md section 8d -- per-row accept counts drawn from a seeded truncated geometric with alpha=0.8
(the reference's measured rate, index.html:595-601), E[tokens/iter]=2.952 at gamma=3; every kernel of the
iteration still runs.  The same line also reports the run with the models' own (random-weight) acceptance and
the autoregressive baseline (tests/baseline_benchmark.py loop) that defines "speedup".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (target cfg dir, draft cfg dir, B, prefix, max_len, budget, gamma)
    "cfg3": ("Meta-Llama-3.1-8B", "Llama-3.2-1B", 64, 16032, 16128, 257, 3),
    "cfg3-small": ("Meta-Llama-3.1-8B", "Llama-3.2-1B", 8, 2080, 2176, 257, 3),
    "tiny": ("llama-68m-gqa", "llama-68m-gqa", 4, 416, 512, 129, 3),
    # eight kv heads: shards down to ONE kv head per rank at TP8 with a TP4 draft sub-group -- the reference README's
    # 8-GPU topology in miniature (tests/test_bench_cpu.py runs it over gloo with 4 and 8 ranks)
    "tiny-kh8": ("llama-68m-kh8", "llama-68m-kh8", 4, 416, 512, 129, 3),
    # configs[1] of BASELINE.json: self-speculation, StreamingLLM draft cache (one model, two caches)
    "cfg2": ("Meta-Llama-3.1-8B", None, 32, 8065, 8192, 257, 3),
    # configs[3] / configs[4]: TP8 configurations -- on one GPU only as `--emulate-tp 8` (one rank's compute)
    "cfg4": ("Meta-Llama-3.1-70B", "Llama-3.2-1B", 32, 32641, 32768, 513, 3, "longspec-stream"),
    "cfg5": ("Qwen2.5-32B", None, 128, 65440, 65536, 257, 3, "selfspec-snapkv"),
    "tiny-selfspec-snapkv": ("llama-68m-gqa", None, 4, 416, 512, 129, 3, "selfspec-snapkv"),
    "tiny-longspec-stream": ("llama-68m-gqa", "llama-68m-gqa", 4, 416, 512, 129, 3, "longspec-stream"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    ap.add_argument("--alpha", type=float, default=0.8, help="acceptance rate of the fixed-acceptance replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graphs", dest="graphs", action="store_true", default=None,
                    help="capture decode steps into hipGraphs (engine.compile()); default: on (also under TP: the RCCL "
                         "all-reduces are captured with the step)")
    ap.add_argument("--no-graphs", dest="graphs", action="store_false")
    ap.add_argument("--checkpoints", type=Path, default=Path("checkpoints"))
    ap.add_argument("--force-tp", action="store_true",
                    help="development: take the tensor-parallel code path (RCCL init, sharding, all-reduces, TP argmax "
                         "merge) even with one rank, to exercise it on a 1-GPU box")
    ap.add_argument("--emulate-tp", type=int, default=0, metavar="N",
                    help="development: shard the models as rank 0 of an N-way tensor-parallel group but run alone "
                         "(collectives over a 1-rank group): measures one rank's compute of a TP-N run on a 1-GPU box; "
                         "tokens are meaningless (partial sums are not reduced)")
    ap.add_argument("--kv-dtype", default="bf16", choices=["bf16", "fp8"],
                    help="full-context KV cache storage (fp8 = OCP e4m3fn, BASELINE configs[4]; default bf16 = the "
                         "reference's)")
    ap.add_argument("--kv-layout", default="HND", choices=["NHD", "HND"],
                    help="page layout of the full-context KV cache: HND (the Engine API's default) keeps the rows of a "
                         "kv head contiguous -- same tokens, verify attention 85 %% instead of 82 %% of the HBM peak "
                         "(bf16), 81 %% instead of 71 %% (fp8), profiles/r02_layout_ab.txt; NHD = the reference's "
                         "flashinfer layout (DESIGN.md section 3.1)")
    ap.add_argument("--draft-tp", type=int, default=4,
                    help="size of the draft sub-group (reference README: 4 of 8); 0 = REPLICATED draft: every rank runs "
                         "the whole draft model (no draft collectives, no token broadcast; greedy drafting is "
                         "deterministic, so all ranks draft the same tokens)")
    ap.add_argument("--pmc", dest="pmc", action="store_true", default=None,
                    help="measure roofline.traffic live: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the "
                         "verify-attention launch at this run's shard shape in a subprocess (tools/attn_bench.py); "
                         "default: on for the cfg* workloads on rank 0")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false")
    ap.add_argument("--weights", default="peaked:40:12:miss=0.2", metavar="random|peaked[:emb_rms[:peak]][:miss=f]",
                    help="synthetic weights when no checkpoint exists on the box: 'random' = seeded normal(0, 0.02) "
                         "(acceptance ~0: `value` comes from the fixed-acceptance replay); 'peaked' = the same layers "
                         "with a dominant embedding and a head tied to it through a permutation (Engine/utils._peak_) "
                         "-- peaked next-token distributions, so measured_acceptance_run reports what the draft / "
                         "verify kernels really accept; ':miss=f' makes a separate DRAFT model mispredict a seeded "
                         "fraction f of the vocabulary (per-step acceptance ~1 - f), and measured_acceptance_sweep "
                         "runs the loop at miss = 0.4 / 0.3 / 0.2 beside the replay at alpha = 0.6 / 0.7 / 0.8.  "
                         "Default (round 6): peaked with miss = 0.2, so that the driver's line carries a MEASURED "
                         "acceptance next to the replay that defines `value`; timings are weight-value independent "
                         "(ms_per_step of the two agree to 0.1 %%: profiles/r06_bench_cfg3_peaked_miss_sweep_c7.json)")
    return ap.parse_args(argv)


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()


class AttnTimer:
    """Duration of every verify-attention launch (md_paged_attn with gamma+1 query rows per request) of an eager pass,
    taken from the kernel's OWN begin / end timestamps (md_debug_attn_timing: hipExtLaunchKernel start / stop events on
    the launching stream) -- the quantity a rocprofv3 kernel trace reports as the kernel's duration, so that
    roofline.avg_launch_ms can be checked against profiles/*_kernel_stats.csv.  (Stream events recorded around a launch
    also see the dispatch overhead: +1.2 % at this kernel's 0.62 ms, r03_call1.)"""

    def __init__(self, dev="cuda"):
        self.ms = []
        self.cuda = torch.device(dev).type == "cuda"
        self.n_verify = 0      # only launches with this many query rows per request (gamma+1) are timed
        self._on = False

    @property
    def enabled(self):
        return self._on

    @enabled.setter
    def enabled(self, on):
        self._on = bool(on)
        if self.cuda:
            from magicdec_amd import _lib
            lib = _lib.load()
            if not on:
                self.collect()
            lib.md_debug_attn_timing(1 if on else 0, self.n_verify)

    def collect(self):
        if self.cuda:
            import ctypes
            from magicdec_amd import _lib
            buf = (ctypes.c_float * 4096)()
            n = _lib.load().md_debug_attn_timing_read(buf, 4096)
            self.ms.extend(buf[:n])

    def clear(self):
        self.collect()
        self.ms = []

    def mean_ms(self):
        self.collect()
        return sum(self.ms) / max(len(self.ms), 1)


def truncated_geometric(alpha, gamma, shape, gen, device):
    """accept_nums in [1, gamma+1]: P(j accepted drafts) = alpha^j (1-alpha) for j<gamma, alpha^gamma for j=gamma."""
    u = torch.rand(shape, generator=gen, device=device)
    a = torch.ones(shape, dtype=torch.long, device=device)
    for j in range(1, gamma + 1):
        a += (u < alpha ** j).long()
    return a


def self_launch(n, argv=None):
    """`python bench.py --gpus N` (N > 1) started plainly, i.e. not under torch.distributed.run: re-execute the SAME
    command line as N ranks of one node -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free port> <this script> <same argv>`, the form the driver uses and the reference's README
    prescribes for its own scripts (README.md:59-69: torchrun --nproc_per_node=8; Engine/tp.py:54-64 reads the ranks from
    the environment) -- and hand its exit code on.  Rank 0 of the children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    argv = list(sys.argv if argv is None else argv)
    with socket.socket() as s:                      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL / peer-mapped buffers need it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(argv[0])] + argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd[:10])} ...", file=sys.stderr,
          flush=True)
    return subprocess.call(cmd, env=env)


def main(device=None):
    """`device` is None in production (one cuda device per rank); tests/test_bench_cpu.py passes "cpu" from a wrapper
    that first installs its device-op stand-ins, to drive the launcher + rendezvous + JSON contract without a GPU."""
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (a launcher exported another world size)"
    if device is None:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
        # MAGICDEC_TP_SINGLE_GPU=1 (development / rehearsal, Engine/tp.init_dist): every rank on GPU 0, gloo as the bootstrap
        # transport -- the multi-rank control flow of this file end to end on a 1-GPU box (RCCL refuses two ranks per device)
        dev_index = 0 if os.environ.get("MAGICDEC_TP_SINGLE_GPU", "0") == "1" else local_rank
        torch.cuda.set_device(dev_index)
        device = f"cuda:{dev_index}"
    line = run(args, device)
    # RCCL prints its version banner through C stdio, which is fully buffered on a pipe and would otherwise be
    # flushed at process exit, i.e. AFTER the JSON line: flush it first so that the JSON line is the last line
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stderr.flush()
    if line is not None:
        print(json.dumps(line), flush=True)


def run(args, dev):
    """The benchmark proper.  `dev` is a cuda device in production; tests/test_bench_cpu.py drives the same code on
    "cpu" (gloo, device ops replaced by oracle stand-ins) to check the multi-rank control flow without a GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    on_gpu = torch.device(dev).type == "cuda"
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))

    from magicdec_amd import harness, _lib
    from magicdec_amd.Engine import gemm_policy
    gemm_mode = {"auto": "md_linear (weight-streaming) for the long weight streams at <= 128 rows, md_linear_block (block-tile) "
                         "for the wide 129..256-row products of the verify pass" +
                         (" [off: MAGICDEC_BLOCK=0]" if gemm_policy.block_mode() == "0" else "") +
                         ", hipBLASLt otherwise (Engine/gemm_policy.py)", "hip": "md_linear everywhere it supports the shape",
                 "lib": "hipBLASLt only"}[gemm_policy.mode()]
    from magicdec_amd.Engine import model_core
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
    from magicdec_amd.Engine.utils import setup_seed
    if on_gpu:
        _lib.load()
    model_core.transformer_configs.setdefault(
        "llama-68m-gqa", dict(block_size=2048, n_layer=2, n_head=16, n_local_heads=4, dim=1024, intermediate_size=2048,
                              vocab_size=32000))

    model_core.transformer_configs.setdefault(
        "llama-68m-kh8", dict(block_size=2048, n_layer=2, n_head=32, n_local_heads=8, dim=2048, intermediate_size=1024,
                              vocab_size=4096))      # g = 4: the reference's SnapKV select needs 8g >= 32 rows per chunk

    wl = WORKLOADS[args.workload]
    tgt_name, drf_name, B, S, ML, BUDGET, G = wl[:7]
    kind = wl[7] if len(wl) > 7 else ("selfspec-stream" if drf_name is None else "longspec-snapkv")
    if args.workload == "cfg5" and "--kv-dtype" not in sys.argv:
        args.kv_dtype = "fp8"                       # BASELINE configs[4] names the fp8 KV cache
    emu = getattr(args, "emulate_tp", 0)
    use_tp = world > 1 or getattr(args, "force_tp", False) or emu > 1
    group = draft_group = None
    rank_group = list(range(emu if emu > 1 else world))
    replicate_draft = args.draft_tp == 0
    draft_ranks = list(rank_group) if replicate_draft else list(range(min(len(rank_group), args.draft_tp)))
    if use_tp:
        from magicdec_amd.Engine.tp import init_dist
        _, group, draft_group = init_dist([0] if (emu > 1 or replicate_draft) else draft_ranks)
    in_draft = rank in draft_ranks
    # Collective of the per-layer partial sums: RCCL unless MAGICDEC_ONESHOT_AR=1 (Engine/oneshot.py; validated
    # against RCCL at start-up, falls back on any disagreement).  Not the default: one-shot pulls (N-1) x the message
    # into every rank -- right for the 256 KiB draft messages, no better than RCCL's ring for the 2 MiB verify
    # message at N=8 (14 MiB inbound per rank) -- and it has only been exercised with processes sharing one GPU.
    # FIRST on a multi-GPU run, before the models are loaded and before the long phases (prefill ~20 s, the timed loops,
    # the PMC passes): what one per-layer collective of this run costs, RCCL against the xGMI kernels -- these have never
    # crossed a real link, and a short lease on an 8-GPU node should yield this report even if it yields nothing else
    # (VERDICT r3 next #5c).  It goes to stderr and to gpurun_out/ at once, and into the JSON line at the end.
    # Round 5 (VERDICT r4 missing #2 / next #1b): the report also SELECTS the collective of the run.  The reference runs
    # with its low-latency intra-node all-reduce on (README.md:59, ENABLE_INTRA_NODE_COMM=1); ours was opt-in because it
    # had never crossed a link.  With MAGICDEC_ONESHOT_AR unset, the fused xGMI all-reduce + add + RMSNorm becomes the
    # run's collective iff the CHILD processes -- where a fault or hang costs nothing -- validated it against RCCL on these
    # very GPUs (no error, no time-out, a 640-call bit-exact stress: collective_microbench) and measured it faster than RCCL
    # + the add+norm launch on the verify message;
    # otherwise RCCL.  Rank 0 decides, every rank follows (one broadcast); Engine/oneshot.try_create still self-tests it
    # at load and falls back collectively.
    coll_report = None
    ar_selection = ("MAGICDEC_ONESHOT_AR=" + os.environ["MAGICDEC_ONESHOT_AR"] + " (set by the caller)"
                    if "MAGICDEC_ONESHOT_AR" in os.environ else "rccl (default; no multi-GPU measurement to choose by)")
    if use_tp and world > 1 and on_gpu and os.environ.get("MAGICDEC_BENCH_COLLECTIVES", "1") != "0":
        dim_t = model_core.ModelArgs.from_name(tgt_name).dim
        dist.barrier()
        _sync(dev)
        shapes = [("verify", B * (G + 1), dim_t), ("autoregressive", B, dim_t)]
        if drf_name is not None:          # the same list on every rank (ranks outside the draft group hold no draft model)
            shapes.append(("draft_step", B, model_core.ModelArgs.from_name(drf_name).dim))
        coll_report = collective_microbench_isolated(shapes)
        if rank == 0:
            print("[collectives_us] " + json.dumps(coll_report), file=sys.stderr, flush=True)
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", f"collectives_us_n{world}.json"), "w") as f:
                    json.dump(coll_report, f)
            except OSError:
                pass
        if "MAGICDEC_ONESHOT_AR" not in os.environ:
            arm, why = xgmi_verdict(coll_report) if rank == 0 else (None, "")
            flag = torch.tensor([{"wt": 1, "fence": 2}.get(arm, 0)], dtype=torch.int32, device=dev)
            dist.broadcast(flag, src=0)
            os.environ["MAGICDEC_ONESHOT_AR"] = "1" if int(flag.item()) else "0"
            if int(flag.item()) == 2:
                os.environ["MAGICDEC_AR_PUBLISH"] = "fence"      # read by every OneShotAllReduce this process creates
            ar_selection = (("xgmi fused all-reduce + add + RMSNorm" +
                             (" [release-fence publish]" if int(flag.item()) == 2 else " [write-through publish]")
                             if int(flag.item()) else "rccl") +
                            f" (chosen from this run's collectives_us: {why})") if rank == 0 else ""
            if rank == 0:
                print(f"[collectives_us] all-reduce of this run: {ar_selection}", file=sys.stderr, flush=True)
        dist.barrier()

    setup_seed(123)
    weights = getattr(args, "weights", "random")
    if weights != "random":
        os.environ["MAGICDEC_SYNTH_WEIGHTS"] = weights

    kv_layout = getattr(args, "kv_layout", "HND")
    selfspec = kind.startswith("selfspec")
    streaming = kind.endswith("stream")            # the draft-side cache: StreamingLLM ring, else SnapKV select
    t_load = time.time()
    if selfspec and streaming:
        from magicdec_amd.Engine.StreamingLLM.backend import LMBackend as SelfSpecBackend
        engine = SelfSpecBackend(dtype=torch.bfloat16, device=dev, dec_len=G + 1)
        engine.load_model(args.checkpoints / tgt_name / "model.pth", use_tp=use_tp, rank_group=rank_group, group=group)
        engine.setup_caches(max_batch_size=B, max_seq_length=ML, draft_budget=BUDGET, kv_dtype=args.kv_dtype,
                            kv_layout=kv_layout)
    elif selfspec:
        engine = LMBackend(dtype=torch.bfloat16, device=dev, dec_len=G + 1, draft_dec_len=1)
        engine.load_model(args.checkpoints / tgt_name / "model.pth", use_tp=use_tp, rank_group=rank_group, group=group)
        engine.setup_caches(max_batch_size=B, max_seq_length=ML, draft_budget=BUDGET, window_size=32,
                            kv_dtype=args.kv_dtype, kv_layout=kv_layout)
    else:
        engine = LMBackend(dtype=torch.bfloat16, device=dev, dec_len=G + 1)
        engine.load_model(args.checkpoints / tgt_name / "model.pth", use_tp=use_tp, rank_group=rank_group, group=group)
        engine.setup_caches(max_batch_size=B, max_seq_length=ML, kv_dtype=args.kv_dtype, kv_layout=kv_layout)
    draft = None
    if in_draft and not selfspec:
        draft_tp = (not replicate_draft) and (len(draft_ranks) > 1 or getattr(args, "force_tp", False))
        if streaming:
            from magicdec_amd.Engine.StreamingLLM.backend_draft import LMBackend_Draft as StreamDraft
            draft = StreamDraft(dtype=torch.bfloat16, device=dev)
            draft.load_model(args.checkpoints / drf_name / "model.pth", use_tp=draft_tp, rank_group=draft_ranks,
                             group=draft_group)
            if replicate_draft and use_tp:
                draft.model.replica_group = group
            draft.setup_caches(max_batch_size=B, draft_budget=BUDGET)
        else:
            draft = LMBackend_Draft(dtype=torch.bfloat16, device=dev, draft_budget=BUDGET)
            draft.load_model(args.checkpoints / drf_name / "model.pth", use_tp=draft_tp, rank_group=draft_ranks,
                             group=draft_group)
            if replicate_draft and use_tp:
                draft.model.replica_group = group
            draft.setup_caches(max_batch_size=B, max_seq_length=ML, draft_budget=BUDGET)
    if args.graphs is None:
        # Also under TP: the per-layer RCCL all-reduces are captured with the step (validated with a 1-rank RCCL
        # group on the development box, profiles/r01_tp1rank_rccl_graphs.log; a capture failure falls back to
        # eager launching with a warning, Engine/graph.py).  A TP8 shard's step is ~10 us kernels: eager
        # launching would be host-bound by >2x.
        # (the one-GPU rehearsal mode bootstraps over gloo, whose collectives cannot be captured: eager steps there)
        args.graphs = (on_gpu and os.environ.get("MAGICDEC_NO_GRAPHS", "0") != "1"
                       and not (use_tp and os.environ.get("MAGICDEC_TP_SINGLE_GPU", "0") == "1"))
    if args.graphs:
        engine.compile()
        if draft is not None:
            draft.compile()
    timer = AttnTimer(dev)
    timer.n_verify = G + 1
    t_load = time.time() - t_load

    # synthetic PG-19-shaped batch: uniform token ids, BOS in column 0 (Data/data_converter.py:54), seed 123
    vocab = engine.model.tok_embeddings.weight.shape[0]
    g = torch.Generator().manual_seed(123)
    input_ids = torch.randint(0, vocab, (B, S), generator=g)
    input_ids[:, 0] = 1
    input_ids = input_ids.to(dev)
    eot_1, eot_2 = -1, -2      # synthetic ids carry no EOT semantics

    # ---- prefill (untimed, as in the reference)
    _sync(dev)
    t_pf = time.time()
    st = harness.new_state(B, G, ML + 1, dev, input_ids)
    st.tokens_buffer[:, :1] = engine.encode(input_ids=input_ids)[:, -1:]
    if draft is not None:
        draft.encode(input_ids=input_ids)
    if selfspec and streaming:
        engine.draft_encode(input_ids=input_ids)
    _sync(dev)
    t_pf = time.time() - t_pf
    snap = {"e": (engine.cachelens.clone(), engine.paged_kv_last_page_len.clone())}
    if selfspec:
        snap["s"] = (engine.draft_cachelens.clone(), engine.draft_paged_kv_last_page_len.clone())
    if draft is not None:
        snap["d"] = (draft.cachelens.clone(), draft.paged_kv_last_page_len.clone(),
                     draft.draft_paged_kv_last_page_len.clone() if not streaming else None)
    first_tok = st.tokens_buffer[:, :1].clone()

    def restore():
        engine.cachelens.copy_(snap["e"][0])
        engine.paged_kv_last_page_len.copy_(snap["e"][1])
        if selfspec:
            engine.draft_cachelens.copy_(snap["s"][0])
            engine.draft_paged_kv_last_page_len.copy_(snap["s"][1])
        if draft is not None:
            draft.cachelens.copy_(snap["d"][0])
            draft.paged_kv_last_page_len.copy_(snap["d"][1])
            if not streaming:
                draft.draft_paged_kv_last_page_len.copy_(snap["d"][2])
        st.num_nodes.fill_(S)
        st.tokens_buffer.zero_()
        st.tokens_buffer[:, :1] = first_tok

    def barrier():
        if use_tp:
            dist.barrier()
        _sync(dev)

    bcast = (draft_ranks[0], group) if (use_tp and len(draft_ranks) != len(rank_group)) else None   # replicated: None

    def iteration(next_double, forced):
        """One speculative iteration across the TP group (draft sub-group drafts, tokens broadcast, all verify)."""
        if selfspec:
            return harness.selfspec_iteration(engine, st, G, eot_1, eot_2, S + 80, next_double, streaming, forced)
        return harness.longspec_iteration(engine, draft, st, G, eot_1, eot_2, S + 80, next_double, forced, bcast)

    debug_iters = os.environ.get("MAGICDEC_BENCH_DEBUG_ITERS", "0") == "1"

    # The reference's longspec loop rolls back draft.paged_kv_last_page_len while a SnapKV draft appends through
    # draft_paged_kv_last_page_len (tests/SnapKV/longspec_benchmark.py:247-256 vs Engine/SnapKV/backend_draft.py:113-173): the
    # compressed cache's page length grows by one per draft step and is never rolled back, so its last page is full after
    # page_size - budget % page_size draft steps (127 at budget 257) -- reproduced bug for bug (tests/test_gpu_engine_fuzz.py).
    # A batch at alpha >= ~0.55 reaches prefix + 80 tokens earlier; at lower acceptance the counters are restored just before
    # the page would overflow (rank-independent arithmetic: every rank restores at the same iteration), and the rows any run
    # still dropped are reported (config.kv_rows_dropped_beyond_mapped_pages).
    draft_step_cap = (128 - BUDGET % 128) if (drf_name is not None and not selfspec and not streaming) else None
    early_restores = [0]

    def run_spec(n_warm, n_steps, forced_table):
        restore()
        nd = False
        since = [0]                                   # draft steps since the counters were last restored

        def guard(nd_):
            if draft_step_cap is not None:
                if since[0] + G + 1 > draft_step_cap:
                    restore()
                    early_restores[0] += 1
                    since[0] = 0
                    nd_ = False
                since[0] += G + (1 if nd_ else 0)
            return nd_
        tokens = torch.zeros((), dtype=torch.long, device=dev)
        timer.enabled = False
        # the token counter is queued behind the accept kernel, before the iteration's host read (the host is ahead of
        # the GPU there; after the read it would sit on the critical path of the next iteration)
        def count(state):
            tokens.add_(state.accept_nums.sum())
        st.before_host_read = count
        for i in range(n_warm):
            # exactly the statements of a timed step: the first execution of any torch op in a process loads its code
            # object (the token counter's int sum + add cost 18.6 ms in the first timed iteration when the warm-up
            # loop did not run them: profiles/r02_bench_first_iteration.txt)
            nd = guard(nd)
            term, nd = iteration(nd, forced_table[i] if forced_table is not None else None)
            if term:
                restore()
                nd = False
                since[0] = 0
        tokens.zero_()
        timer.clear()
        timer.enabled = True
        barrier()
        t0 = time.perf_counter()
        stamps = []
        for i in range(n_steps):
            nd = guard(nd)
            term, nd = iteration(nd, forced_table[n_warm + i] if forced_table is not None else None)
            stamps.append((time.perf_counter(), bool(term), bool(nd)))   # the iteration ended with a host read
            if term:
                restore()
                nd = False
                since[0] = 0
        barrier()
        dt = time.perf_counter() - t0
        timer.enabled = False
        st.before_host_read = None
        if debug_iters and rank == 0:
            prev, out = t0, []
            for t, te, d2 in stamps:
                out.append(f"{(t - prev) * 1e3:.2f}{'T' if te else ''}{'d' if d2 else ''}")
                prev = t
            print(f"[bench debug] {n_steps} iterations in {dt * 1e3:.1f} ms (+{(time.perf_counter() - t0 - dt) * 1e3:.1f}): "
                  + " ".join(out), file=sys.stderr, flush=True)
        return dt, int(tokens.item())

    def prime():
        """Run every step variant once outside the timed regions: the first use of a shape captures its hipGraph
        (tens of ms) -- e.g. the two-token draft step only occurs after an all-accept iteration, which a short warm-up
        may never reach.  (The reference's scripts discard their first 6 batches for the same reason,
        tests/SnapKV/longspec_benchmark.py:308.)  The length counters are restored afterwards."""
        two = st.double_buffer.clone()
        cu = torch.ones(B, dtype=torch.long, device=dev)
        if selfspec:
            engine.speculate(first_tok.clone())
            if streaming:
                engine.speculate(two, cachelen_update=cu)
            engine.verify(st.tokens_buffer.clone())
            engine.verify(first_tok.clone())
        else:
            if draft is not None:
                draft.inference(first_tok.clone())
                draft.inference(two, cachelen_update=cu)
            engine.inference(st.tokens_buffer.clone())
            engine.inference(first_tok.clone())
        restore()
    prime()

    # Probation of the xGMI all-reduce inside the run's own processes and hipGraphs (round 5): the children validated the
    # kernels on these links, try_create self-tested them eagerly at load; prime() has now replayed every captured step
    # variant once with them inside.  If ANY rank saw a time-out there, every rank drops to RCCL here -- a point where all
    # ranks stand at the same statement and no other collective is in flight -- instead of failing in the timed region,
    # where a time-out is fatal by design (AllReduceTimeout at the iteration it happened in).
    ar_probation = None
    if use_tp and world > 1 and on_gpu and os.environ.get("MAGICDEC_ONESHOT_AR") == "1":
        models = [engine.model] + ([draft.model] if draft is not None else [])
        ar_probation = xgmi_probation(models, dev)
        if ar_probation["drop"]:
            from magicdec_amd.Engine.graph import clear_graphs
            _sync(dev)                                # nothing of the communicators is in flight on any rank (the
            for m in models:                          # all-reduce inside xgmi_probation ordered the ranks)
                if getattr(m, "_oneshot", None) is not None:
                    m._oneshot.close()                # md_ar_destroy: registered buffers, signal area, IPC mappings
                m._oneshot = None                     # Transformer._reduce_add_norm / _reduce fall back to RCCL
            clear_graphs(engine)
            if draft is not None:
                clear_graphs(draft)
            ar_selection += f" -- DROPPED in probation ({ar_probation['why']}): every rank runs rccl"
            if rank == 0:
                print(f"[collectives_us] xGMI all-reduce dropped in probation ({ar_probation['why']}): every rank drops "
                      "to RCCL", file=sys.stderr, flush=True)
            prime()

    gen = torch.Generator(device=dev if on_gpu else "cpu").manual_seed(2024)
    forced = truncated_geometric(args.alpha, G, (args.warmup + args.steps, B), gen, dev)
    dt_replay, tok_replay = run_spec(args.warmup, args.steps, forced)
    if debug_iters:                      # development: is the first timed region slower than a repeat of itself?
        for _ in range(2):
            run_spec(args.warmup, args.steps, forced)
    if args.graphs:
        # HIP events cannot be recorded inside a replayed graph: time the verify-attention launches in an eager
        # pass of the same iterations (same kernels, same shapes, same stream) right after the timed region
        was = engine._use_graphs            # False if a capture failed and the back-end fell back to eager
        engine._use_graphs = False
        run_spec(1, min(args.steps, 8), forced)
        engine._use_graphs = was
    attn_ms = timer.mean_ms()
    n_attn = len(timer.ms)
    dt_meas, tok_meas = run_spec(min(args.warmup, 2), max(args.steps // 4, 4), None)
    meas_steps = max(args.steps // 4, 4)
    # sensitivity of the headline to the assumed acceptance rate: the same loop replayed at other alphas (short runs)
    sens_steps = max(min(args.steps, 12), 4)
    sens_raw = {}
    for al in ((0.5, 0.6, 0.7, 0.8, 0.9) if on_gpu else (args.alpha,)):
        if abs(al - args.alpha) < 1e-9:
            sens_raw[al] = (dt_replay / args.steps, tok_replay / args.steps)
            continue
        f_al = truncated_geometric(al, G, (2 + sens_steps, B), gen, dev)
        dt_al, tok_al = run_spec(2, sens_steps, f_al)
        sens_raw[al] = (dt_al / sens_steps, tok_al / sens_steps)

    # ---- measured acceptance at KNOWN draft quality: with `--weights peaked[...]` the draft's head is
    # rewritten in place (Engine/utils.repeak_head_: same storage, so the captured graphs stay valid; the draft cache does not
    # depend on the head) to mispredict a seeded fraction `miss` of the vocabulary, i.e. a draft whose per-step acceptance
    # rate is ~1 - miss, and the SAME loop runs with the accept kernel's own decisions.  Each point sits beside the
    # fixed-acceptance replay at alpha = 1 - miss: equal tokens/s there shows the replay is a timing-neutral stand-in.
    acc_sweep = None
    # (rank-independent condition: ranks outside the draft sub-group hold no draft model but run the same loops;
    #  --emulate-tp: partial sums are not reduced, tokens are noise)
    if on_gpu and emu <= 1 and weights.startswith("peaked") and drf_name is not None and not selfspec:
        from magicdec_amd.Engine.utils import parse_peaked, repeak_head_
        miss_cfg = parse_peaked(weights)[2]
        acc_sweep = {}
        for miss in (0.4, 0.3, 0.2):
            if in_draft and draft is not None:
                repeak_head_(draft.model, miss)
            dt_m, tok_m = run_spec(2, sens_steps, None)
            acc_sweep[miss] = (dt_m / sens_steps, tok_m / sens_steps)
        if in_draft and draft is not None:
            repeak_head_(draft.model, miss_cfg)

    # ---- autoregressive baseline (tests/baseline_benchmark.py loop: one token per target step)
    target_step = engine.verify if selfspec else engine.inference
    restore()
    nt = first_tok.clone()
    for _ in range(min(args.warmup, 3)):
        nt = target_step(nt)
    restore()
    base_steps = max(args.steps // 2, 8)
    barrier()
    t0 = time.perf_counter()
    for _ in range(base_steps):
        nt = target_step(nt)
    barrier()
    dt_base = time.perf_counter() - t0

    def allmax(x):
        if not use_tp:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    dt_replay, dt_meas, dt_base = allmax(dt_replay), allmax(dt_meas), allmax(dt_base)
    sens_raw = {al: (allmax(dt), tok) for al, (dt, tok) in sens_raw.items()}
    if acc_sweep is not None:
        acc_sweep = {m: (allmax(dt), tok) for m, (dt, tok) in acc_sweep.items()}

    value = tok_replay / dt_replay
    base_tps = B * base_steps / dt_base
    cfg = engine.model.config
    H_loc, KH_loc, D = cfg.n_head, cfg.n_local_heads, cfg.head_dim
    L_kv = S + 40 + G + 1                                 # mean kv length over a batch's 80 generated tokens
    kv_elem = 1 if args.kv_dtype == "fp8" else 2
    attn_bytes = B * L_kv * KH_loc * D * 2 * kv_elem + 2 * B * (G + 1) * H_loc * D * 2
    achieved = attn_bytes / (attn_ms * 1e-3) / 1e9 if attn_ms > 0 else 0.0

    traffic, traffic_source = None, None
    want_pmc = args.pmc if getattr(args, "pmc", None) is not None else args.workload.startswith("cfg")
    if want_pmc and on_gpu and rank == 0:
        traffic, traffic_source = measure_traffic(B, S + 40 + G + 1, KH_loc, H_loc, D, G + 1, args.kv_dtype == "fp8",
                                                  kv_layout == "HND")
    # the same kernel on the OTHER page layout, stand-alone: the line says which layout the headline
    # ran and what the reference's own layout (NHD) would give
    other = "NHD" if kv_layout == "HND" else "HND"
    roofline_other = None
    if on_gpu and rank == 0 and args.workload.startswith("cfg") and os.environ.get("MAGICDEC_BENCH_LAYOUT_AB", "1") != "0":
        roofline_other = layout_roofline(engine, timer, B, G + 1, L_kv, H_loc, KH_loc, D, args.kv_dtype == "fp8", other,
                                         attn_bytes, dev)
    # rows the append kernels dropped because a request's last page was full (md_page_overflow_count; the reference would
    # have written them into another request's page): must be 0 -- over every run of this process, not only the headline
    kv_dropped = None
    if on_gpu:
        from magicdec_amd import ops as _ops
        kv_dropped = int(_ops.page_overflow_count(reset=True))
    ar_timeouts = None
    if use_tp:
        ars = [m._oneshot for m in ([engine.model] + ([draft.model] if draft is not None else []))
               if getattr(m, "_oneshot", None) is not None]
        ar_timeouts = sum(a.status() for a in ars) if ars else None      # must be 0: a time-out invalidates the run
    line = {
        "metric": "decode tokens/s/node + speedup vs autoregressive, Llama-3.1-8B B=64 prefix=16K",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world,
        # the group size the collective backend itself reports (RCCL under "nccl"): a multi-GPU record whose ranks did not
        # form ONE communicator would show here, not only in n_gpus (= WORLD_SIZE)
        "rccl_ranks": (dist.get_world_size(group) if (use_tp and group is not None) else 1),
        "collective_backend": (dist.get_backend(group) if (use_tp and group is not None) else None),
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt_replay / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "bf16", "kv_cache_dtype": args.kv_dtype, "kv_cache_layout": kv_layout,
        "data": "synthetic",
        "config": {"workload": (f"{args.workload}: {tgt_name} self-speculation TP{len(rank_group)}, "
                                f"{'StreamingLLM' if streaming else 'SnapKV'} draft cache "
                                f"budget {BUDGET} gamma {G}, B={B} prefix={S} max_len={ML}") if selfspec else
                               (f"{args.workload}: {tgt_name} target TP{len(rank_group)} + {drf_name} "
                                f"{'StreamingLLM' if streaming else 'SnapKV'} draft "
                                f"{'replicated on every rank' if replicate_draft and use_tp else 'TP' + str(len(draft_ranks))} "
                                f"budget {BUDGET} gamma {G}, B={B} prefix={S} max_len={ML}"),
                   "acceptance": f"fixed replay alpha={args.alpha} (E[tokens/iter]={tok_replay / args.steps / B:.3f})",
                   "weights": ("seeded random init (no checkpoints on the box)" if weights == "random" else
                               f"seeded synthetic '{weights}': random layers, dominant embedding, head tied to it through "
                               "a permutation (peaked next-token distributions; Engine/utils._peak_)"),
                   "hip_graphs": bool(engine._use_graphs),
                   "gemm": gemm_mode,
                   "fused_linear": {"auto": "md_linear_fused (linear + rope/append | residual add | SiLU*mul in one "
                                            "launch) for the launch-bound small products (Engine/gemm_policy.py)",
                                    "0": "off", "1": "forced on"}[gemm_policy.fused_mode()],
                   # weights held TWICE during decode (streaming layout + row-major): after prefill the row-major tensor of
                   # every weight that decode reads in the streaming layout only is released (Transformer.release_rowmajor);
                   # what remains is read in both layouts at different row counts (configs[2]: nothing)
                   "packed_weight_copies_bytes": int(getattr(engine.model, "packed_bytes", 0)
                                                     + (getattr(draft.model, "packed_bytes", 0) if draft is not None
                                                        else 0)),
                   "rowmajor_weight_bytes_released_after_prefill": int(
                       getattr(engine.model, "released_bytes", 0)
                       + (getattr(draft.model, "released_bytes", 0) if draft is not None else 0)),
                   "kv_rows_dropped_beyond_mapped_pages": kv_dropped,
                   "draft_cache_restores_before_its_page_end": early_restores[0] if draft_step_cap is not None else None,
                   **({"emulated_tp_rank0_of": emu} if emu > 1 else {}),
                   "allreduce": (None if not use_tp else
                                 "oneshot-ipc" if getattr(engine.model, "_oneshot", None) is not None else "rccl"),
                   "allreduce_selection": ar_selection,
                   "allreduce_probation": ar_probation,
                   "allreduce_timeouts": ar_timeouts,
                   "allreduce_plan": allreduce_plan(engine, draft, B, G, len(rank_group),
                                                    1 if replicate_draft else len(draft_ranks)) if use_tp
                   else None},
        "speedup_vs_autoregressive": round(value / base_tps, 4),
        "speedup_condition": speedup_condition({al: tok / dt / base_tps for al, (dt, tok) in sens_raw.items()},
                                               args.alpha),
        "alpha_sensitivity": {f"{al:.1f}": {"tokens_per_s": round(tok / dt, 1), "speedup": round(tok / dt / base_tps, 3),
                                            "tokens_per_iter_per_seq": round(tok / B, 3),
                                            "ms_per_step": round(dt * 1e3, 3)}
                              for al, (dt, tok) in sorted(sens_raw.items())},
        "autoregressive_tokens_per_s": round(base_tps, 2),
        "autoregressive_ms_per_step": round(dt_base / base_steps * 1e3, 4),
        "measured_acceptance_run": {"tokens_per_s": round(tok_meas / dt_meas, 2),
                                    "tokens_per_iter_per_seq": round(tok_meas / meas_steps / B, 3),
                                    "accepted_drafts_per_drafted": round((tok_meas / meas_steps / B - 1) / G, 4),
                                    "speedup_vs_autoregressive": round(tok_meas / dt_meas / base_tps, 4),
                                    "ms_per_step": round(dt_meas / meas_steps * 1e3, 4),
                                    "weights": weights},
        "measured_acceptance_sweep": (None if acc_sweep is None else {
            f"miss={m:.1f}": acceptance_point(dt, tok, B, G, base_tps, sens_raw.get(round(1.0 - m, 1)))
            for m, (dt, tok) in sorted(acc_sweep.items(), reverse=True)}),
        "prefill_s": round(t_pf, 2), "load_s": round(t_load, 2),
        # what "identical to the reference" means for this path (checked by `pytest -m gpu` and smoke(), not here)
        "parity": {"tokens": "identity with the CPU oracle; zero flips over 16 832 positions at the real layer widths and "
                             "B = 64 (peaked weights, rejections in the run) and over 1 670 on the tiny peaked pair; on the "
                             "near-flat logits of random-init tiny models argmax flips are MEASURED against three yardstick "
                             "oracles per lock-step run of ~450-1 700 positions: hip 2-18, float64-linear oracle 0-13, "
                             "bf16-P oracle (the tensor-core attention algorithm) 1-20, both 2-20 -- the bf16 P alone "
                             "explains the HIP count; gate hip <= max(bf16-P, both) + 3 + one standard deviation "
                             "(tests/test_gpu_engine.py, profiles/r05_parity_report.txt)",
                   "logits": "max |hip - oracle| <= 2 x the error of a correctly-rounded bf16 implementation + 2 bf16 "
                             "ulp (measured 0.012-0.047 on logits of magnitude 1-4; 1 ulp = 0.0625 on the peaked full-"
                             "width run), NOT 1e-3 absolute: every bf16 linear output carries 2^-8 relative rounding",
                   "integer_paths": "page tables, accept/rollback, gather, top-k order GIVEN the scores: bit-exact; SnapKV "
                                    "pooled scores bit-identical to the reference fixtures; ties at the selection "
                                    "threshold go to the lowest index (reference: torch.topk, unspecified among equal "
                                    "scores -- fixture index sets differ by <= 4 of 225 per (request, kv head), all at "
                                    "the threshold score)"},
        "roofline": {"kernel": f"paged_attn_kernel<{D},{1 if (G + 1) * (H_loc // KH_loc) <= 16 else 2},"
                               f"{'true' if args.kv_dtype == 'fp8' else 'false'}> (verify attention, md_paged_attn, "
                               f"{kv_layout} pages)",
                     "bound": "hbm",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "bytes_per_launch": attn_bytes, "avg_launch_ms": round(attn_ms, 4), "launches_timed": n_attn,
                     "timing": "kernel begin/end timestamps of every verify-attention launch of an eager pass "
                               "(hipExtLaunchKernel start/stop events on the launching stream) = the duration a "
                               "rocprofv3 kernel trace reports"},
    }
    line["roofline"]["kv_cache_layout"] = kv_layout
    if roofline_other is not None:
        line["roofline_" + other.lower()] = roofline_other
    if kv_layout == "NHD":
        line["roofline_nhd"] = dict(line["roofline"])       # the headline itself ran the reference's layout
    if rank == 0 and not args.no_cpu_baseline and world == 1 and not selfspec:
        line["cpu_baseline"] = cpu_baseline(tgt_name, drf_name, S, BUDGET, G, args.alpha)
    if coll_report is not None and rank == 0:
        line["collectives_us"] = coll_report
    if use_tp:
        dist.barrier()
        dist.destroy_process_group()
    return line if rank == 0 else None


def xgmi_probation(models, dev, rounds=4):
    """Probation of the xGMI all-reduce inside the run's OWN processes, communicators and hipGraphs (rounds 5-6).  The
    children validated the kernels on these links (collective_microbench), try_create self-tested them eagerly at load and
    prime() has replayed every captured step variant once with them inside.  Two checks, one collective decision:
      * time-outs: any rank whose status word is set (a bounded spin gave up, output rows poisoned);
      * numerics: a short BIT-EXACT stress on each model's own communicator -- `rounds` x 6 all-reduces of the
        run's message sizes, algorithms interleaved, queued back to back, against the rank-ordered fp32 sum of the
        RCCL-all-gathered inputs (the kernels' definition).  A stale read that rounds differently would otherwise corrupt
        tokens silently while the timings still looked valid.
    Returns {"drop": bool, "why": str, "status": [...], "mismatched_elements": n, "calls": n} -- identical on every rank
    (max-reduced), so that all ranks drop to RCCL at this common point or none does."""
    from magicdec_amd.Engine import oneshot
    ars = [m._oneshot for m in models if getattr(m, "_oneshot", None) is not None]
    status = [a.status() for a in ars]
    mism = calls = 0
    algos = (oneshot.ALGO_ONESHOT, oneshot.ALGO_TWOSHOT, oneshot.ALGO_AUTO)
    for ai, ar in enumerate(ars):
        dim = models[ai].tok_embeddings.weight.shape[1]
        sizes = [n for n in (64 * dim, 256 * dim, 2048) if n * 2 <= ar.max_bytes]
        gen = torch.Generator(device=dev).manual_seed(9100 + 17 * ai + ar.rank)
        for rnd in range(rounds):
            xs, wants = [], []
            for j in range(6):
                n = sizes[(rnd + j) % len(sizes)]
                x = (torch.randn(n, device=dev, generator=gen, dtype=torch.float32) * 3).to(torch.bfloat16)
                if dist.get_backend(ar.group) == "nccl":
                    parts = [torch.empty_like(x) for _ in range(ar.world)]
                    dist.all_gather(parts, x, group=ar.group)
                else:                             # gloo (ranks sharing one GPU): all_gather takes host tensors only
                    host = [torch.empty(n, dtype=torch.bfloat16) for _ in range(ar.world)]
                    dist.all_gather(host, x.cpu(), group=ar.group)
                    parts = [h.to(dev) for h in host]
                acc = parts[0].float()
                for r in range(1, ar.world):
                    acc = acc + parts[r].float()
                xs.append(x)
                wants.append(acc.to(torch.bfloat16))
            torch.cuda.synchronize()
            dist.barrier(group=ar.group)
            for j, x in enumerate(xs):
                ar.all_reduce_(x, algos[(rnd + j) % 3])
            torch.cuda.synchronize()
            for x, w_ in zip(xs, wants):
                mism += int((x.view(torch.int16) != w_.view(torch.int16)).sum())
                calls += 1
        status[ai] = max(status[ai], ar.status())
    t = torch.tensor([max(status) if status else 0, mism], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    timed_out, bad_bits = int(t[0]) != 0, int(t[1]) != 0
    why = "; ".join(w for w, on in (("a peer time-out in the first graph-replayed steps or the stress", timed_out),
                                    ("the bit-exact stress on the run's own communicator mismatched", bad_bits)) if on)
    return {"drop": timed_out or bad_bits, "why": why or "passed", "status_this_rank": status,
            "status_max_over_ranks": int(t[0]), "mismatched_elements_max_over_ranks": int(t[1]), "calls": calls}


def layout_roofline(engine, timer, B, n, L_kv, H, KH, D, fp8, layout, attn_bytes, dev, launches=24):
    """The verify-attention launch of this run's shard shape on pages of `layout`, timed like `roofline` (the kernel's own
    begin / end timestamps, md_debug_attn_timing) on TWO scratch layer caches of random values used alternately (2 x 4.2 GB at
    cfg3: Infinity-Cache cold, like consecutive layers of the run) with the page table of a request that holds L_kv rows.
    Reported beside `roofline` so that the driver record carries both the Engine default (HND) and the layout the
    reference's flashinfer plan uses (NHD, Engine/SnapKV/backend.py:30) -- same kernel, bit-identical outputs
    (tests/test_gpu_hnd.py).  Returns the roofline-shaped dict, or {"error": ...} when the scratch caches do not fit."""
    from magicdec_amd import ops
    mp = (L_kv + 127) // 128
    try:
        free = torch.cuda.mem_get_info(dev)[0]
        need = 2 * B * mp * 2 * 128 * KH * D * (1 if fp8 else 2)
        if need + (4 << 30) > free:
            return {"error": f"scratch caches need {need >> 20} MiB, {free >> 20} MiB free"}
        g = torch.Generator(device=dev).manual_seed(77)
        shape = (B * mp, 2, KH, 128, D) if layout == "HND" else (B * mp, 2, 128, KH, D)
        caches = []
        for _ in range(2):
            c = torch.empty(shape, device=dev, dtype=torch.bfloat16)
            c.normal_(generator=g)
            caches.append(c.to(ops.FP8_DTYPE) if fp8 else c)
        scales = (torch.full((KH,), 0.5, device=dev), torch.full((KH,), 0.25, device=dev)) if fp8 else None
        q = torch.randn(B * n, H, D, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
        indices = torch.arange(B * mp, dtype=torch.int32, device=dev)
        indptr = torch.arange(B + 1, dtype=torch.int32, device=dev) * mp
        last = torch.full((B,), L_kv - (mp - 1) * 128, dtype=torch.int32, device=dev)
        qo = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
        ws = engine.model.workspace
        for i in range(4):
            ops.paged_attention(q, caches[i % 2], qo, indices, indptr, last, n, mp, ws, kv_scales=scales, kv_layout=layout)
        _sync(dev)
        timer.clear()
        timer.enabled = True
        for i in range(launches):
            ops.paged_attention(q, caches[i % 2], qo, indices, indptr, last, n, mp, ws, kv_scales=scales, kv_layout=layout)
        _sync(dev)
        timer.enabled = False
        ms, cnt = timer.mean_ms(), len(timer.ms)
        timer.clear()
        del caches
        torch.cuda.empty_cache()
    except RuntimeError as e:                 # out of memory on a shape this helper was not sized for
        timer.enabled = False
        return {"error": f"{type(e).__name__}: {str(e)[:120]}"}
    achieved = attn_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"kernel": f"paged_attn_kernel<{D},...> (verify attention, md_paged_attn, {layout} pages; stand-alone launches at "
                      f"this run's shard shape, kv length {L_kv})",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": attn_bytes,
            "avg_launch_ms": round(ms, 4), "launches_timed": cnt}


def alpha_of(tokens_per_iter, gamma):
    """The per-step acceptance rate a of a truncated-geometric draft with E[tokens / iteration] = sum_{j<=gamma} a^j."""
    lo, hi = 0.0, 1.0
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if sum(mid ** j for j in range(gamma + 1)) < tokens_per_iter:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def acceptance_point(dt, tok, B, gamma, base_tps, replay):
    """One point of measured_acceptance_sweep: the loop run with the accept kernel's own decisions on a draft of known
    quality, and the fixed-acceptance replay (dt, tokens per step) at the matching alpha beside it."""
    tpi = tok / B
    out = {"tokens_per_s": round(tok / dt, 1), "ms_per_step": round(dt * 1e3, 3), "tokens_per_iter_per_seq": round(tpi, 3),
           "accepted_drafts_per_drafted": round((tpi - 1) / gamma, 4),
           "alpha_equivalent": round(alpha_of(tpi, gamma), 4),
           "speedup_vs_autoregressive": round(tok / dt / base_tps, 3)}
    if replay is not None:
        rdt, rtok = replay
        out["replay_at_matching_alpha"] = {"tokens_per_s": round(rtok / rdt, 1), "ms_per_step": round(rdt * 1e3, 3),
                                           "tokens_per_iter_per_seq": round(rtok / B, 3)}
        # the quantity the replay stands in for: time per iteration (tokens per iteration differ by sampling noise)
        out["ms_per_step_vs_replay"] = round(dt / rdt, 4)
    return out


def speedup_condition(speedup_by_alpha, alpha, targets=(1.0, 1.8)):
    """What the headline speed-up is conditional on: `value` replays a FIXED acceptance rate, so
    "x over autoregressive" holds iff the real draft reaches that rate.  Linear interpolation of the measured
    alpha -> speed-up sweep at break-even (1.0x) and at north_star's target (>= 1.8x)."""
    pts = sorted(speedup_by_alpha.items())
    head = f"speedup_vs_autoregressive is the fixed-acceptance replay at alpha={alpha}"
    if len(pts) < 2:
        return head + "; no alpha sweep in this run"
    out = []
    for target in targets:
        if pts[0][1] >= target:
            out.append(f">= {target}x already at alpha = {pts[0][0]} (lowest rate swept)")
            continue
        for (a0, s0), (a1, s1) in zip(pts, pts[1:]):
            if s0 < target <= s1:
                a = a0 + (target - s0) * (a1 - a0) / (s1 - s0)
                out.append(f">= {target}x iff alpha >= {a:.2f} (between the measured {a0} -> {s0:.2f}x and {a1} -> {s1:.2f}x)")
                break
        else:
            out.append(f"{target}x is not reached at any swept alpha (best {pts[-1][1]:.2f}x at {pts[-1][0]})")
    return head + "; " + "; ".join(out)


def allreduce_plan(engine, draft, B, G, tp, draft_tp):
    """Which implementation carries each per-layer all-reduce of this run, by message (what the JSON line documents so
    that a scaling number can be attributed): RCCL (captured inside the step's hipGraph), or -- MAGICDEC_ONESHOT_AR=1 and
    the start-up self-test against RCCL passed -- the xGMI kernels of csrc/allreduce.hip under MD_AR_ALGO_AUTO
    (two-shot for >= 4 ranks and messages > 512 KiB, else one-shot), fused with the residual add + RMSNorm."""
    def one(model, rows, world):
        if model is None or world <= 1:
            return None
        dim = model.tok_embeddings.weight.shape[1]
        nbytes = rows * dim * 2
        if getattr(model, "_oneshot", None) is None:
            algo = "rccl all_reduce (in-graph) + md_add_rmsnorm"
        else:
            algo = ("xgmi two-shot" if (world >= 4 and nbytes > 512 * 1024) else "xgmi one-shot") + " fused add+rmsnorm"
        return {"rows": rows, "bytes": nbytes, "ranks": world, "per_forward": 2 * len(model.layers), "impl": algo}
    return {"verify": one(engine.model, B * (G + 1), tp), "autoregressive": one(engine.model, B, tp),
            "draft_step": one(draft.model if draft is not None else None, B, draft_tp),
            "argmax_merge": "2 x rccl all_reduce of [rows, ranks] per forward",
            "draft_tokens": ("rccl broadcast per iteration" if draft_tp not in (1, tp) else
                             ("none: the draft model is replicated on every rank" if draft_tp == 1 and tp > 1 else None))}


def collective_microbench_isolated(shapes, iters=30, timeout_s=120, dry=False):
    """collective_microbench in one CHILD process per rank (tools/collective_bench.py): same ranks and GPUs, a
    rendezvous port of its own, the parents waiting on the host (a GPU-side barrier would spin on the CUs the children
    are timing).  The xGMI kernels map peer memory and had never run over real links before the first multi-GPU bench;
    whatever they do -- fault, hang, disagree with RCCL -- costs this report, not the benchmark's line.  Collective
    over the ranks of the job (every rank must call it); returns the report on rank 0 (None elsewhere); failures are
    reported as {"error": ...}."""
    import subprocess
    import tempfile
    rank = int(os.environ.get("RANK", "0"))
    base = int(os.environ.get("MASTER_PORT", "29500"))
    port = 20000 + (base * 7 + 1234) % 20000
    if port == base:
        port += 1
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}   # the child hosts its own store
    env["MASTER_PORT"] = str(port)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    out = os.path.join(tempfile.gettempdir(), f"md_collectives_{base}_{os.getpid()}.json")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "collective_bench.py"), "--shapes",
           ",".join(f"{n}:{r}:{d}" for n, r, d in shapes), "--iters", str(iters), "--out", out] + (["--dry"] if dry else [])
    err = None
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=ROOT)
        try:
            _, stderr = p.communicate(timeout=timeout_s)
            if p.returncode != 0:
                err = f"child exited with {p.returncode}: {stderr.decode(errors='replace')[-400:]}"
        except subprocess.TimeoutExpired:
            p.kill()                      # this exact child (never by pattern)
            p.communicate()
            err = f"child timed out after {timeout_s} s"
    except OSError as e:
        err = f"{type(e).__name__}: {e}"
    if rank != 0:
        return None
    try:
        with open(out) as f:
            res = json.load(f)
        os.remove(out)
        if err is not None and isinstance(res, dict):
            res["rank0_child"] = err
        return res
    except (OSError, ValueError):
        return {"error": err or "the rank-0 child wrote no report"}


def xgmi_verdict(report):
    """(collective of this run, why) from the children's collectives_us report, collective in {"wt", "fence", None}:
    "wt" = the xGMI fused all-reduce with the write-through publish, "fence" = the same kernels with the release-fence
    publish (MAGICDEC_AR_PUBLISH=fence), None = RCCL.  An arm qualifies only if every shape was measured with the xGMI
    kernels validated against RCCL (try_create's self-test inside the child), no child failed or timed out, no kernel spin
    timed out, ITS bit-exact stress passed, and its fused all-reduce + add + RMSNorm beat RCCL + the add+norm launch on the
    verify message.  The write-through arm is preferred (it is the faster one); when its stress fails -- the one hardware
    assumption it rests on does not hold on these links -- the fence arm is still eligible."""
    if not isinstance(report, dict) or not report:
        return None, "no report"
    if "error" in report or "rank0_child" in report:
        return None, "a child process failed: " + str(report.get("error") or report.get("rank0_child"))[:160]
    meta = ("xgmi_stress", "xgmi_fence_stress")
    for name, r in report.items():
        if name in meta:
            continue
        if not isinstance(r, dict) or "xgmi_fused_add_rmsnorm_auto" not in r:
            return None, f"xGMI kernels unavailable for '{name}' (set-up or self-test against RCCL failed)"
    v = report.get("verify")
    if v is None:
        return None, "no verify message in the report"

    def passed(st):
        return (isinstance(st, dict) and st.get("mismatched_elements_all_ranks", 1) == 0
                and st.get("timeouts_all_ranks", 1) == 0 and st.get("calls", 0) >= 100)
    b = v["rccl_allreduce_then_add_rmsnorm"]
    why = []
    wt_ok = passed(report.get("xgmi_stress")) and all(r.get("xgmi_timeouts", 1) == 0 for n, r in report.items()
                                                      if n not in meta)
    if wt_ok:
        a = v["xgmi_fused_add_rmsnorm_auto"]
        if a < b:
            return "wt", f"verify message: xgmi fused {a} us vs rccl + add+norm {b} us"
        why.append(f"write-through arm slower than rccl on the verify message ({a} vs {b} us)")
    else:
        why.append(f"the write-through arm did not pass (stress {report.get('xgmi_stress')}, or a spin timed out)")
    # the fence arm: its stress ran AFTER the write-through one on the same communicator, and the status word is sticky --
    # a time-out there is only attributable to the fence arm if the first stress ended clean
    st_f = report.get("xgmi_fence_stress")
    wt_clean_status = isinstance(report.get("xgmi_stress"), dict) and report["xgmi_stress"].get("timeouts_all_ranks", 1) == 0
    if passed(st_f) and wt_clean_status and "xgmi_fence_fused_add_rmsnorm_auto" in v:
        a = v["xgmi_fence_fused_add_rmsnorm_auto"]
        if a < b:
            return "fence", f"{why[0]}; fence-publish arm on the verify message: {a} us vs rccl + add+norm {b} us"
        why.append(f"fence arm slower than rccl on the verify message ({a} vs {b} us)")
    else:
        why.append(f"the fence arm did not pass either ({st_f})")
    return None, "; ".join(why)


def collective_microbench(group, shapes, dev, iters=30):
    """Mean time (us, max over ranks) of ONE per-layer all-reduce of this run's hidden-state messages [rows, dim] bf16:
    RCCL (`dist.all_reduce`, what the timed run used) followed by the add + RMSNorm kernel, against the xGMI kernels of
    csrc/allreduce.hip (one-shot, two-shot, and fused with the add + RMSNorm), each queued `iters` times back to back
    between two events.  The xGMI communicator is created here (collective, validated against RCCL by its self-test;
    `None` -> reported as unavailable).  Every rank runs the same sequence; the kernels' spins are bounded."""
    from magicdec_amd import ops
    from magicdec_amd.Engine import oneshot
    out = {}
    ar = oneshot.try_create(group)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier(group=group)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / iters * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return round(float(t.item()), 2)

    for name, rows, dim in shapes:
        x = torch.randn(rows, dim, device=dev, dtype=torch.float32).to(torch.bfloat16)
        res = torch.randn(rows, dim, device=dev, dtype=torch.float32).to(torch.bfloat16)
        w = torch.ones(dim, device=dev, dtype=torch.bfloat16)
        buf = x.clone()
        r = {"bytes": rows * dim * 2}
        r["rccl_allreduce"] = timed(lambda: dist.all_reduce(buf, group=group))
        r["rccl_allreduce_then_add_rmsnorm"] = timed(lambda: (dist.all_reduce(buf, group=group),
                                                              ops.add_rmsnorm(res, buf, w, 1e-5)))
        if ar is not None and ar.fits(buf):
            r["xgmi_oneshot"] = timed(lambda: ar.all_reduce_(buf, oneshot.ALGO_ONESHOT))
            r["xgmi_twoshot"] = timed(lambda: ar.all_reduce_(buf, oneshot.ALGO_TWOSHOT))
            r["xgmi_fused_add_rmsnorm_auto"] = timed(lambda: ar.all_reduce_add_rmsnorm(x, res, w, 1e-5))
            # third arm (round 6): the same fused kernel with the RELEASE-FENCE publish (md_ar_set_publish) -- what a
            # run degrades to if the write-through hand-off fails its stress on real links
            ar.set_publish(oneshot.PUBLISH_FENCE)
            r["xgmi_fence_fused_add_rmsnorm_auto"] = timed(lambda: ar.all_reduce_add_rmsnorm(x, res, w, 1e-5))
            ar.set_publish(oneshot.PUBLISH_WRITE_THROUGH)
            r["xgmi_timeouts"] = ar.status()
        else:
            r["xgmi"] = "unavailable (set-up or self-test against RCCL failed on some rank)"
        out[name] = r
    if ar is not None:
        # Bit-exact stress of the kernels over THESE links before a run may rely on them (round 5): the hand-off rests on
        # write-through stores being visible to a peer when the relaxed flag is (csrc/allreduce.hip), which the
        # shared-GPU tests cannot probe and try_create's tolerance self-test would miss if it failed rarely.  16 inputs
        # per round are all-gathered over RCCL, summed locally in rank order (fp32, one rounding = the kernel's
        # definition), then 16 xGMI calls are queued back to back with no host synchronisation, algorithms and sizes
        # interleaved, and compared bit for bit.
        sizes = sorted({rows * dim for _, rows, dim in shapes} | {2048})
        algos = (oneshot.ALGO_ONESHOT, oneshot.ALGO_TWOSHOT, oneshot.ALGO_AUTO)
        world, rk = dist.get_world_size(group), dist.get_rank(group)

        def stress(rounds, seed):
            mism = calls = 0
            gen = torch.Generator(device=dev).manual_seed(seed + rk)
            for rnd in range(rounds):
                xs, wants = [], []
                for j in range(16):
                    n = sizes[(rnd + j) % len(sizes)]
                    x = (torch.randn(n, device=dev, generator=gen, dtype=torch.float32) * 3).to(torch.bfloat16)
                    if dist.get_backend(group) == "nccl":
                        parts = [torch.empty_like(x) for _ in range(world)]
                        dist.all_gather(parts, x, group=group)
                    else:                         # gloo (the shared-GPU test): all_gather takes host tensors only
                        host = [torch.empty(n, dtype=torch.bfloat16) for _ in range(world)]
                        dist.all_gather(host, x.cpu(), group=group)
                        parts = [h.to(dev) for h in host]
                    acc = parts[0].float()
                    for r in range(1, world):
                        acc = acc + parts[r].float()
                    xs.append(x)
                    wants.append(acc.to(torch.bfloat16))
                torch.cuda.synchronize()
                dist.barrier(group=group)
                for j, x in enumerate(xs):
                    ar.all_reduce_(x, algos[(rnd + j) % 3])
                torch.cuda.synchronize()
                for x, w_ in zip(xs, wants):
                    mism += int((x.view(torch.int16) != w_.view(torch.int16)).sum())
                    calls += 1
            t = torch.tensor([mism, ar.status()], dtype=torch.int64, device=dev)
            dist.all_reduce(t, group=group)
            return {"calls": calls, "mismatched_elements_all_ranks": int(t[0]), "timeouts_all_ranks": int(t[1])}

        out["xgmi_stress"] = stress(40, 4321)
        # the fence arm's own stress (status is sticky: only meaningful while the first arm left it at 0)
        ar.set_publish(oneshot.PUBLISH_FENCE)
        out["xgmi_fence_stress"] = stress(20, 8765)
        ar.set_publish(oneshot.PUBLISH_WRITE_THROUGH)
        ar.close()
    return out


def measure_traffic(B, L_kv, KH, H, D, n, fp8, hnd=False):
    """roofline.traffic measured in THIS invocation: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not
    fit one pass on gfx950; --kernel-trace only, no other trace domain) over tools/attn_bench.py, which launches
    md_paged_attn at exactly this run's per-layer shard shape on > 256 MiB of KV per launch (Infinity-Cache cold).
    Corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies 64 B
    per 128-B request for wide coalesced streaming reads, so read bytes = FETCH_SIZE * 1024 * 2.  The split-KV merge
    kernel's bytes (when it runs) are added.  Returns (bytes per launch | None, provenance string)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="md_pmc_", dir="/tmp")
        cmd = ["timeout", "150", "rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d,
               "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "tools", "attn_bench.py"), "--iters", "4",
               "--B", str(B), "--S", str(L_kv), "--KH", str(KH), "--H", str(H), "--D", str(D), "--n", str(n),
               "--fp8", "1" if fp8 else "0", "--hnd", "1" if hnd else "0"]
        env = {k: v for k, v in os.environ.items()
               if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env["TMPDIR"] = "/tmp"
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=200)
        except (subprocess.TimeoutExpired, OSError) as e:
            return None, f"rocprofv3 --pmc {counter} failed: {type(e).__name__}"
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None, f"rocprofv3 --pmc {counter}: no counter_collection.csv"
        per_kernel = {}
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter and ("paged_attn_kernel" in r["Kernel_Name"]
                                                         or "attn_merge" in r["Kernel_Name"]):
                    key = "merge" if "attn_merge" in r["Kernel_Name"] else "attn"
                    per_kernel.setdefault(key, []).append(float(r["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if "attn" not in per_kernel:
            return None, f"rocprofv3 --pmc {counter}: no md_paged_attn dispatch in the trace"
        res[counter] = sum(sum(v) / len(v) for v in per_kernel.values())
        res[counter + "_n"] = len(per_kernel["attn"])
    traffic = int(res["FETCH_SIZE"] * 1024 * 2 + res["WRITE_SIZE"] * 1024)
    return traffic, (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) over "
                     f"tools/attn_bench.py --B {B} --S {L_kv} --KH {KH} --H {H} --D {D} --n {n} --fp8 {int(fp8)}; mean of "
                     f"{res['FETCH_SIZE_n']} launches; bytes = FETCH_SIZE KiB * 1024 * 2 (gfx950 wide-read correction) "
                     f"+ WRITE_SIZE KiB * 1024")


def cpu_baseline(tgt_name, drf_name, S, budget, gamma, alpha, Bc=4):
    """The oracle (oracle/magicdec_ref.py + oracle/flashinfer_ref.py: the torch-eager CPU restatement of the reference
    path) timed on this host's cores on a bounded sample of the same workload: ONE decoder layer of each model at the
    real shapes with a batch of `Bc` requests at the real prefix length (KV pre-filled with random values), plus ONE
    final-norm + lm-head pass and ONE embedding lookup per model, each timed separately.  An iteration is assembled as

        gamma * (n_layer_draft * t_layer_draft + t_head_draft + t_embed_draft)
              + (n_layer_target * t_layer_target(gamma+1 rows) + t_head_target + t_embed_target)

    (layer cost is depth-independent; embedding and head are counted once per forward, not once per layer), and
    converted to tokens/s with the same replayed acceptance.  The K1 / K2 / K6 micro-benchmarks of SURVEY.md section
    8d (verify attention, draft attention, SnapKV select at the per-layer shapes, batch Bc) are reported in GB/s of
    algorithmic bytes.  A reported baseline, not an optimisation target."""
    import torch
    import torch.nn.functional as F
    from magicdec_amd.Engine.model_core import ModelArgs
    from oracle import flashinfer_ref as fr
    from oracle import magicdec_ref as mr
    # threads actually used: the M = 4..16-row products and the per-request attention loops of the restatement do not
    # scale past a few cores -- on the 256-core host of the GPU box torch.set_num_threads(256) ran the same sample 80x
    # SLOWER than 8 threads (thread wake-up / NUMA traffic dominates), so the pool is capped and the cap is reported
    ncores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(ncores)
    t_all = time.perf_counter()

    def best(fn, reps=2):
        fn()                                   # warm (allocations, oneDNN primitive caches)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    def model_of(name):
        a = ModelArgs.from_name(name)
        cfg = mr.RefConfig(n_layer=1, n_head=a.n_head, n_local_heads=a.n_local_heads, dim=a.dim,
                           intermediate_size=a.intermediate_size, vocab_size=a.vocab_size, rope_base=a.rope_base,
                           scaling_factor=a.scaling_factor, low_freq_factor=a.low_freq_factor,
                           high_freq_factor=a.high_freq_factor,
                           original_max_position_embeddings=a.original_max_position_embeddings)
        return cfg, mr.init_state_dict(cfg, 1), a.n_layer

    def time_model(name, mode, n_rows, kv_len):
        """(t_layer, t_head, t_embed, t_attention_only, attention bytes) of one forward with n_rows query rows per
        request over kv_len cached positions (the appended rows included)."""
        cfg, sd, n_layer = model_of(name)
        eng = mr.RefEngine(mode, cfg, sd, Bc, S + 96, budget if mode == "snapkv_draft" else 0, max_pos=S + 256)
        m = eng.model
        if mode == "target":
            caches, prefix = eng.caches, ""
            npg = (kv_len + 127) // 128
            eng.paged_kv_indptr = torch.arange(Bc + 1, dtype=torch.int32) * npg
            eng.paged_kv_indices = torch.cat([torch.arange(b * eng.ppr, b * eng.ppr + npg, dtype=torch.int32)
                                              for b in range(Bc)])
            eng.paged_kv_last_page_len = torch.full((Bc,), kv_len - (npg - 1) * 128, dtype=torch.int32)
        else:
            caches, prefix = eng.draft_caches, "draft_"
            eng.draft_paged_kv_last_page_len = torch.full((Bc,), kv_len - (eng.dppr - 1) * 128, dtype=torch.int32)
        caches[0].normal_()
        eng.cachelens.fill_(S)
        tab = eng._tab(prefix)
        attn_fn = eng._attn_std(n_rows, eng.cachelens, caches, tab)
        x = torch.randn(Bc, n_rows, cfg.dim).to(torch.bfloat16)
        ids = torch.randint(0, cfg.vocab_size, (Bc, n_rows))
        t_layer = best(lambda: m.block(x, 0, attn_fn))
        t_head = best(lambda: m.head(x))
        t_embed = best(lambda: F.embedding(ids, sd["tok_embeddings.weight"]))
        q = torch.randn(Bc * n_rows, cfg.n_head, cfg.head_dim).to(torch.bfloat16)
        qo = torch.arange(Bc + 1, dtype=torch.int32) * n_rows
        t_attn = best(lambda: m.attn(q, caches[0], qo, tab))
        nbytes = (Bc * kv_len * cfg.n_local_heads * cfg.head_dim * 2 * 2
                  + 2 * Bc * n_rows * cfg.n_head * cfg.head_dim * 2)
        return t_layer, t_head, t_embed, t_attn, nbytes, n_layer, cfg, eng

    tl_t, th_t, te_t, ta_t, nb_t, n_t, _, _ = time_model(tgt_name, "target", gamma + 1, S + gamma + 1)
    # the SAME target-layer sample once on every host core (BASELINE.md section 4 asks for the physical cores): printed so
    # that capping the pool at 16 threads is evidence, not assertion
    all_cores = os.cpu_count() or 1
    tl_all = None
    if all_cores > ncores and os.environ.get("MAGICDEC_CPU_ALL_CORES", "1") != "0":
        torch.set_num_threads(all_cores)
        try:
            cfg_a, sd_a, _ = model_of(tgt_name)
            eng_a = mr.RefEngine("target", cfg_a, sd_a, Bc, S + 96, 0, max_pos=S + 256)
            npg = (S + gamma + 1 + 127) // 128
            eng_a.paged_kv_indptr = torch.arange(Bc + 1, dtype=torch.int32) * npg
            eng_a.paged_kv_indices = torch.cat([torch.arange(b * eng_a.ppr, b * eng_a.ppr + npg, dtype=torch.int32)
                                                for b in range(Bc)])
            eng_a.paged_kv_last_page_len = torch.full((Bc,), S + gamma + 1 - (npg - 1) * 128, dtype=torch.int32)
            eng_a.caches[0].normal_()
            eng_a.cachelens.fill_(S)
            fn_a = eng_a._attn_std(gamma + 1, eng_a.cachelens, eng_a.caches, eng_a._tab(""))
            x_a = torch.randn(Bc, gamma + 1, cfg_a.dim).to(torch.bfloat16)
            tl_all = best(lambda: eng_a.model.block(x_a, 0, fn_a), reps=1)
            del eng_a
        finally:
            torch.set_num_threads(ncores)
    tl_d, th_d, te_d, ta_d, nb_d, n_d, cfg_d, eng_d = time_model(drf_name, "snapkv_draft", 1, budget + 1)
    # K6: SnapKV select of ONE request of the draft model over the full prefix (the oracle loops over requests)
    g = cfg_d.n_head // cfg_d.n_local_heads
    k_ctx = torch.randn(S, cfg_d.n_local_heads, cfg_d.head_dim).to(torch.bfloat16)
    v_ctx = torch.randn(S, cfg_d.n_local_heads, cfg_d.head_dim).to(torch.bfloat16)
    q_win = torch.randn(32, cfg_d.n_head, cfg_d.head_dim).to(torch.bfloat16)
    t0 = time.perf_counter()
    mr.snapkv_select(q_win, k_ctx, v_ctx, g, 32, budget)
    t_k6 = time.perf_counter() - t0
    nb_k6 = S * cfg_d.n_local_heads * cfg_d.head_dim * 2 + 2 * budget * cfg_d.n_local_heads * cfg_d.head_dim * 2

    cfg1 = cpu_baseline_cfg1()
    full_b1 = cpu_baseline_full_iteration_b1(tgt_name, drf_name, S, budget, gamma)
    iter_s = gamma * (n_d * tl_d + th_d + te_d) + (n_t * tl_t + th_t + te_t)
    e_tok = sum(alpha ** j for j in range(gamma + 1))
    full_b1["tokens_per_s"] = round(e_tok / full_b1["iteration_s"], 3)      # B = 1, the same replayed acceptance
    return {"value": round(Bc * e_tok / iter_s, 4), "unit": "tokens/s", "cores": ncores,
            "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle (torch-eager CPU restatement), batch {Bc}, prefix {S}, random KV: 1 of {n_t} target layers "
                      f"at {gamma + 1} rows/request ({tl_t * 1e3:.1f} ms) + target head ({th_t * 1e3:.1f} ms) + embedding "
                      f"({te_t * 1e3:.2f} ms); 1 of {n_d} draft layers at 1 row/request over the {budget}-row SnapKV cache "
                      f"({tl_d * 1e3:.1f} ms) + draft head ({th_d * 1e3:.1f} ms) + embedding ({te_d * 1e3:.2f} ms); iteration = "
                      f"{gamma} x ({n_d} x layer + head + embedding) + ({n_t} x layer + head + embedding) = {iter_s:.2f} s; "
                      f"replay alpha={alpha}; best of 2 after a warm-up call",
            "micro_GBps": {"K1_verify_attention": round(nb_t / ta_t / 1e9, 2),
                           "K2_draft_attention": round(nb_d / ta_d / 1e9, 3),
                           "K6_snapkv_select_one_request": round(nb_k6 / t_k6 / 1e9, 3)},
            "thread_choice": ({"target_layer_ms_at_cores": round(tl_t * 1e3, 1), "cores": ncores,
                               "target_layer_ms_at_all_host_cores": round(tl_all * 1e3, 1), "all_host_cores": all_cores,
                               "note": "the same sample (one target layer, 4 rows / request) on torch's pool capped at "
                                       "`cores` and on every host core: the cap is the FASTER of the two"}
                              if tl_all is not None else None),
            "cfg1_end_to_end": cfg1,
            "full_iteration_b1": full_b1,
            "wall_s": round(time.perf_counter() - t_all, 1)}


def cpu_baseline_full_iteration_b1(tgt_name, drf_name, S, budget, gamma):
    """ONE whole speculative iteration of the oracle at B = 1 and the real prefix ("one full
    spec-decode iteration of Llama-3.1-8B shapes at B=1, S=16032"), not an assembly of per-layer timings: `gamma` draft
    forwards of the full-depth draft model over its `budget`-row SnapKV cache and one (gamma+1)-token verify forward of
    the full-depth target over `S` cached positions, head and embedding included, timed end to end.  Two shortcuts keep
    it a bounded sample: the KV caches are zero-filled and already hold the prefix (no CPU prefill of 16K tokens), and
    all layers of a model share ONE layer's weight tensors (allocating 16 GB of random weights would take longer than
    the measurement; the shared 0.4 GB layer is friendlier to the host caches than 32 distinct ones, so this number is,
    if anything, optimistic for the CPU)."""
    from magicdec_amd.Engine.model_core import ModelArgs
    from oracle import magicdec_ref as mr

    def engine(name, mode):
        a = ModelArgs.from_name(name)
        one = mr.RefConfig(n_layer=1, n_head=a.n_head, n_local_heads=a.n_local_heads, dim=a.dim,
                           intermediate_size=a.intermediate_size, vocab_size=a.vocab_size, rope_base=a.rope_base,
                           scaling_factor=a.scaling_factor, low_freq_factor=a.low_freq_factor,
                           high_freq_factor=a.high_freq_factor,
                           original_max_position_embeddings=a.original_max_position_embeddings)
        sd = mr.init_state_dict(one, 1)
        for i in range(1, a.n_layer):                  # every layer = layer 0's tensors (aliases, no copies)
            for k in [k for k in sd if k.startswith("layers.0.")]:
                sd[k.replace("layers.0.", f"layers.{i}.", 1)] = sd[k]
        full = mr.RefConfig(**{**one.__dict__, "n_layer": a.n_layer})
        eng = mr.RefEngine(mode, full, sd, 1, S + 96, budget if mode == "snapkv_draft" else 0, max_pos=S + 256)
        return eng

    t_all = time.perf_counter()
    tgt = engine(tgt_name, "target")
    kv_len = S + gamma + 1
    npg = (kv_len + 127) // 128
    tgt.paged_kv_indptr = torch.tensor([0, npg], dtype=torch.int32)
    tgt.paged_kv_indices = torch.arange(npg, dtype=torch.int32)
    tgt.paged_kv_last_page_len = torch.full((1,), kv_len - (npg - 1) * 128, dtype=torch.int32)
    tgt.cachelens.fill_(S)
    drf = engine(drf_name, "snapkv_draft")
    drf.cachelens.fill_(S)
    ids4 = torch.randint(4, 1000, (1, gamma + 1))
    tok = ids4[:, :1]

    def iteration():
        nonlocal tok
        t = tok
        for _ in range(gamma):
            t = drf.inference(t)
        tgt.paged_kv_last_page_len.fill_(kv_len - gamma - 1 - (npg - 1) * 128)    # verify re-appends the same rows
        tgt.cachelens.fill_(S)
        drf.cachelens.fill_(S)
        drf.draft_paged_kv_last_page_len.fill_(budget - (drf.dppr - 1) * 128)
        return tgt.inference(ids4)
    iteration()                                          # warm (allocations, oneDNN primitive caches)
    t0 = time.perf_counter()
    iteration()
    dt = time.perf_counter() - t0
    return {"workload": f"one iteration, B=1, prefix {S}: {gamma} x {drf_name} draft forward ({budget}-row SnapKV cache) + "
                        f"1 x {tgt_name} {gamma + 1}-token verify forward, full depth, zero-filled KV, layers share one "
                        f"layer's weights", "iteration_s": round(dt, 3), "kind": "port (the oracle run end to end)",
            "setup_s": round(time.perf_counter() - t_all - 2 * dt, 1)}


def cpu_baseline_cfg1(prefix=129, max_len=256):
    """BASELINE.json configs[0] END TO END on the host cores (the reference's own
    CPU-runnable case): the reference's "68m" model entry (Engine/SnapKV/model.py:67: MHA, 12 heads, dim 768, 2 layers,
    vocab 32000; seeded random weights), B = 1, prefix 129, greedy autoregressive decode to 256 positions with the
    loop of tests/baseline_benchmark.py:72-90 -- the whole run of the oracle (oracle/harness_ref.baseline_batch:
    prefill + 126 decode steps), not an extrapolation.  tokens/s is generated tokens over the decode loop's wall time,
    as the script reports it."""
    from magicdec_amd.Engine.model_core import ModelArgs
    from oracle import harness_ref as hr
    from oracle import magicdec_ref as mr
    a = ModelArgs.from_name("68m")
    cfg = mr.RefConfig(n_layer=a.n_layer, n_head=a.n_head, n_local_heads=a.n_local_heads, dim=a.dim,
                       intermediate_size=a.intermediate_size, vocab_size=a.vocab_size, rope_base=a.rope_base)
    sd = mr.init_state_dict(cfg, 68)
    eng = mr.RefEngine("target", cfg, sd, 1, max_len)
    g = torch.Generator().manual_seed(123)
    ids = torch.randint(4, cfg.vocab_size, (1, prefix), generator=g)
    ids[:, 0] = 1
    t0 = time.perf_counter()
    first = eng.encode(ids)[:, -1:]
    t_prefill = time.perf_counter() - t0
    steps, nt = 0, first
    t0 = time.perf_counter()
    while prefix + 1 + steps < max_len:
        nt = eng.inference(nt.clone())
        steps += 1
    t_decode = time.perf_counter() - t0
    return {"workload": f"cfg1: llama-68m (reference '68m' entry, MHA) autoregressive, B=1 prefix={prefix} -> {max_len}",
            "tokens_per_s": round(steps / t_decode, 2), "decode_steps": steps, "decode_s": round(t_decode, 3),
            "prefill_s": round(t_prefill, 3), "kind": "port (the oracle run end to end)"}


if __name__ == "__main__":
    main()
