"""Command lines of the reference's benchmark scripts (same flags, asserts and printed lines):
tests/SnapKV/longspec_benchmark.py:16-46, tests/StreamingLLM/longspec_benchmark.py, tests/*/selfspec_benchmark.py,
tests/baseline_benchmark.py.  The loops themselves live in magicdec_amd/harness.py."""
from __future__ import annotations

import argparse
import builtins
import time
from pathlib import Path

import torch
import torch.distributed as dist
from torch.utils.data.dataloader import DataLoader

from . import harness
from .data import convert_pg19_dataset, load_tokenizer
from .Engine.utils import setup_seed


# Per-script defaults of the reference's five entry points (their argparse blocks differ: e.g. the self-speculation scripts
# default to B = 45, prefix 100000, gamma 7, budget 4097; tests/SnapKV/selfspec_benchmark.py:17-36).  Checkpoint paths are
# the one deliberate difference: "checkpoints/..." relative to the working directory instead of "/scratch/models/...".
_CKPT_8B = Path("checkpoints/meta-llama/Meta-Llama-3.1-8B/model.pth")
_CKPT_1B = Path("checkpoints/meta-llama/Llama-3.2-1B/model.pth")
_DEFAULTS = {
    "longspec": dict(model=_CKPT_1B, B=1, prefix_len=4000, max_len=64, gamma=5),
    "selfspec": dict(model=_CKPT_8B, B=45, prefix_len=100000, max_len=100096, gamma=7, draft_budget=4097),
    "baseline": dict(model=_CKPT_8B, B=16, prefix_len=8065, max_len=8192),
}


def _common(parser, script, kind=None):
    d = _DEFAULTS[script]
    parser.add_argument('--model', type=Path, default=d["model"], help='model')
    parser.add_argument('--model_name', type=str, default="meta-llama/Meta-Llama-3.1-8B", help='model name')
    parser.add_argument('--dataset', type=str, default="pg19", help='Dataset name.')
    parser.add_argument('--rank_group', nargs='+', type=int, help='Target group of ranks')
    parser.add_argument('--compile', action='store_true', help='Capture the decode steps into hipGraphs.')
    parser.add_argument('--B', type=int, default=d["B"], help='Batch size.')
    parser.add_argument('--prefix_len', type=int, default=d["prefix_len"], help='Prefix length')
    parser.add_argument('--max_len', type=int, default=d["max_len"], help='Generate length')
    parser.add_argument('--seed', type=int, default=123, help='Random seed.')
    parser.add_argument('--printoutput', action='store_true', help='Whether to print the generated text.')
    parser.add_argument('--benchmark', action='store_true', help='Per-phase timing (adds synchronisations).')
    parser.add_argument('--kv_dtype', type=str, default="bf16", choices=["bf16", "fp8"],
                        help='Storage of the full-context KV cache (fp8 = OCP e4m3fn; not in the reference).')
    parser.add_argument('--kv_layout', type=str, default=None, choices=["NHD", "HND"],
                        help='Page layout of the full-context KV cache.  Default: the Engine default '
                             '(backend_core.default_kv_layout(): HND -- the rows of one kv head contiguous inside a '
                             'page, the layout bench.py measures -- unless MAGICDEC_KV_LAYOUT says otherwise); '
                             'NHD is the layout the reference plans flashinfer with.')
    if script != "baseline":
        budget = d.get("draft_budget", -1 if kind == "SnapKV" else 1025)     # longspec: SnapKV -1, StreamingLLM 1025
        parser.add_argument('--draft_budget', type=int, default=budget, help='Draft KV budget.')
        parser.add_argument('--gamma', type=int, default=d["gamma"], help='speculation length')
        if kind == "SnapKV":
            parser.add_argument('--window_size', type=int, default=32, help='SnapKV observation window')


def longspec_parser(kind="SnapKV"):
    """The flags and defaults of tests/<kind>/longspec_benchmark.py (+ --kv_dtype / --kv_layout)."""
    parser = argparse.ArgumentParser(description='Process model configuration and partitions.')
    _common(parser, "longspec", kind)
    parser.add_argument('--target', type=Path, default=_CKPT_8B, help='target model')
    parser.add_argument('--draft_rank_group', nargs='+', type=int, help='Draft group of ranks')
    # not in the reference: every rank of --rank_group runs the WHOLE draft model (no draft-side collectives, no
    # per-iteration token broadcast; greedy drafting is deterministic, so all ranks draft the same tokens).  Needs
    # --draft_rank_group to name at least two ranks only because the reference keys `use_tp` on it (kept verbatim).
    parser.add_argument('--replicate_draft', action='store_true',
                        help='run the whole draft model on every target rank instead of sharding it over --draft_rank_group')
    return parser


def selfspec_parser(kind="SnapKV"):
    """The flags and defaults of tests/<kind>/selfspec_benchmark.py (+ --kv_dtype / --kv_layout)."""
    parser = argparse.ArgumentParser(description='Process model configuration and partitions.')
    _common(parser, "selfspec", kind)
    return parser


def baseline_parser():
    """The flags and defaults of tests/baseline_benchmark.py (+ --dataset / --benchmark / --kv_dtype / --kv_layout)."""
    parser = argparse.ArgumentParser(description='Process model configuration and partitions.')
    _common(parser, "baseline")
    return parser


def _eot(tokenizer):
    eot_1 = tokenizer.eos_token_id
    eot_2 = tokenizer.unk_token_id if tokenizer.unk_token_id is not None else tokenizer.encode("<|eot_id|>")[-1]
    return eot_1, eot_2


def _avg_len(num_gen_tokens, target_steps, batch_size):
    """"avg generate len per sentence" exactly as the reference prints it: its token counter is a 0-d int64 tensor, so
    the two divisions happen in float32 (tests/SnapKV/longspec_benchmark.py:307 prints 2.6111111640930176, not
    2.611111111111111)."""
    return (torch.tensor(num_gen_tokens) / target_steps / batch_size).item()


def _device():
    return 'cuda' if torch.cuda.is_available() else 'cpu'


def _dataset(args, tokenizer, vocab):
    if args.dataset != "pg19":
        raise ValueError(f"Unknown dataset {args.dataset}")
    return convert_pg19_dataset(tokenizer=tokenizer, seq_len=args.prefix_len, vocab_size=vocab,
                                num_sequences=10 * args.B)


def longspec_main(kind: str, argv=None):
    """kind: 'SnapKV' | 'StreamingLLM' -- tests/<kind>/longspec_benchmark.py."""
    args = longspec_parser(kind).parse_args(argv)
    assert args.prefix_len < args.max_len
    assert (args.max_len + 127) // 128 == args.prefix_len // 128 + 1
    if kind == "SnapKV":
        if args.draft_budget != -1:
            assert (args.prefix_len - args.window_size) % 128 == 0
            assert (args.draft_budget - 1) % 128 == 0
    else:
        assert (args.draft_budget - 1) % 128 == 0

    draft_tp = len(args.draft_rank_group) > 1
    DEVICE = _device()
    from .Engine.tp import init_dist
    use_tp = len(args.draft_rank_group) > 1
    global_group = draft_group = None
    rank = 0
    print_ = builtins.print
    if use_tp:
        rank, global_group, draft_group = init_dist(args.draft_rank_group)
        if rank != args.rank_group[0]:
            print_ = lambda *a, **k: None
        DEVICE = f"cuda:{rank}" if torch.cuda.is_available() else "cpu"
    setup_seed(args.seed)
    print_(f"Using device={DEVICE}")
    replicate = bool(getattr(args, "replicate_draft", False)) and use_tp
    draft_target_equal = len(args.draft_rank_group) == len(args.rank_group) or replicate
    MAX_LEN_TARGET, BATCH_SIZE, DTYPE = args.max_len, args.B, torch.bfloat16

    from .Engine.SnapKV.backend import LMBackend
    if kind == "SnapKV":
        from .Engine.SnapKV.backend_draft import LMBackend_Draft
    else:
        from .Engine.StreamingLLM.backend_draft import LMBackend_Draft
    engine = LMBackend(dtype=DTYPE, device=DEVICE, dec_len=args.gamma + 1)
    engine.load_model(args.target, use_tp=use_tp, rank_group=args.rank_group, group=global_group)
    if args.compile:
        engine.compile()
    engine.setup_caches(max_batch_size=BATCH_SIZE, max_seq_length=MAX_LEN_TARGET, kv_dtype=args.kv_dtype,
                        kv_layout=args.kv_layout)

    draft = None
    if (not use_tp) or replicate or rank in args.draft_rank_group:
        if kind == "SnapKV":
            draft = LMBackend_Draft(dtype=DTYPE, device=DEVICE, draft_budget=args.draft_budget)
        else:
            draft = LMBackend_Draft(dtype=DTYPE, device=DEVICE)
        draft.load_model(args.model, use_tp=use_tp and draft_tp and not replicate, rank_group=args.draft_rank_group,
                         group=draft_group)
        if replicate:       # every rank runs the whole draft and must choose the same kernels (Transformer._pack_weights)
            draft.model.replica_group = global_group
        if args.compile:
            draft.compile()
        if kind == "SnapKV":
            draft.setup_caches(max_batch_size=BATCH_SIZE, max_seq_length=MAX_LEN_TARGET, draft_budget=args.draft_budget)
        else:
            draft.setup_caches(max_batch_size=BATCH_SIZE, draft_budget=args.draft_budget)
    if use_tp:
        dist.barrier()

    tokenizer = load_tokenizer(args.model_name)
    eot_1, eot_2 = _eot(tokenizer)
    print_(f"eot_1: {eot_1}, eot_2: {eot_2}")
    dataset = _dataset(args, tokenizer, engine.model.tok_embeddings.weight.shape[0])
    dataloader = DataLoader(dataset, batch_size=BATCH_SIZE, shuffle=False, drop_last=True)
    num_eval_steps = min(10, len(dataloader))
    bcast = (args.draft_rank_group[0], global_group) if (use_tp and not draft_target_equal) else None
    barrier = dist.barrier if use_tp else None

    total_time, num_gen_tokens, target_steps = 0.0, 0, 0
    timers = harness.PhaseTimers(DEVICE) if args.benchmark else None
    for step, batch in enumerate(dataloader):
        if step >= num_eval_steps:
            break
        input_ids = batch[0].to(DEVICE)
        st, dt = harness.run_longspec_batch(engine, draft, input_ids, args.gamma, MAX_LEN_TARGET, eot_1, eot_2,
                                            bcast=bcast, barrier=barrier, timers=timers)
        total_time += dt
        target_steps += st.iters
        num_gen_tokens += int(st.num_nodes.sum() - (input_ids.shape[1] + 1) * BATCH_SIZE)
        if args.printoutput:
            for i in range(BATCH_SIZE):
                print_("Sequence ", i)
                print_(tokenizer.decode(st.output[i, args.prefix_len:st.num_nodes[i]]))
        if kind == "SnapKV":   # only tests/SnapKV/longspec_benchmark.py:305 prints the latency; the StreamingLLM twin (:302) not
            print_("total time :{:.5f}s, time per iter :{:.5f}s, decoding step: {}, large model step: {}, avg latency: {}".format(
                total_time, total_time / target_steps, num_gen_tokens, target_steps, total_time / num_gen_tokens * BATCH_SIZE))
        else:
            print_("total time :{:.5f}s, time per iter :{:.5f}s, decoding step: {}, large model step: {}".format(
                total_time, total_time / target_steps, num_gen_tokens, target_steps))
        if args.benchmark:     # tests/SnapKV/longspec_benchmark.py:305-307
            print_("target time :{:.5f}s, draft time :{:.5f}s, verify loop : {}, avg generate len per sentence: {}".format(
                timers.target / target_steps, timers.draft / target_steps, timers.verify_loop / target_steps,
                _avg_len(num_gen_tokens, target_steps, BATCH_SIZE)))
        if step < 5:
            total_time, num_gen_tokens, target_steps = 0.0, 0, 0
            if timers is not None:
                timers.reset()
        if use_tp:
            dist.barrier()
    print_(f"Final tokens per second :{num_gen_tokens / total_time}")
    return num_gen_tokens / total_time


def selfspec_main(kind: str, argv=None):
    """kind: 'SnapKV' | 'StreamingLLM' -- tests/<kind>/selfspec_benchmark.py."""
    args = selfspec_parser(kind).parse_args(argv)
    assert args.prefix_len < args.max_len
    assert (args.max_len + 127) // 128 == args.prefix_len // 128 + 1
    assert (args.draft_budget - 1) % 128 == 0
    if kind == "SnapKV":
        assert (args.prefix_len - args.window_size) % 128 == 0
    DEVICE = _device()
    use_tp = len(args.rank_group) > 1
    global_group = None
    rank = 0
    print_ = builtins.print
    if use_tp:
        from .Engine.tp import init_dist
        rank, global_group = init_dist()
        if rank != args.rank_group[0]:
            print_ = lambda *a, **k: None
        DEVICE = f"cuda:{rank}" if torch.cuda.is_available() else "cpu"
    setup_seed(args.seed)
    print_(f"Using device={DEVICE}")
    MAX_LEN_TARGET, BATCH_SIZE, DTYPE = args.max_len, args.B, torch.bfloat16
    streaming = kind == "StreamingLLM"
    if streaming:
        from .Engine.StreamingLLM.backend import LMBackend
        engine = LMBackend(dtype=DTYPE, device=DEVICE, dec_len=args.gamma + 1)
    else:
        from .Engine.SnapKV.backend import LMBackend
        engine = LMBackend(dtype=DTYPE, device=DEVICE, dec_len=args.gamma + 1, draft_dec_len=1)
    engine.load_model(args.model, use_tp=use_tp, rank_group=args.rank_group, group=global_group)
    if args.compile:
        engine.compile()
    if streaming:
        engine.setup_caches(max_batch_size=BATCH_SIZE, max_seq_length=MAX_LEN_TARGET, draft_budget=args.draft_budget,
                            kv_dtype=args.kv_dtype, kv_layout=args.kv_layout)
    else:
        engine.setup_caches(max_batch_size=BATCH_SIZE, max_seq_length=MAX_LEN_TARGET, draft_budget=args.draft_budget,
                            window_size=args.window_size, kv_dtype=args.kv_dtype, kv_layout=args.kv_layout)
    tokenizer = load_tokenizer(args.model_name)
    eot_1, eot_2 = _eot(tokenizer)
    print_(f"eot_1: {eot_1}, eot_2: {eot_2}")
    dataset = _dataset(args, tokenizer, engine.model.tok_embeddings.weight.shape[0])
    dataloader = DataLoader(dataset, batch_size=BATCH_SIZE, shuffle=False, drop_last=True)
    num_eval_steps = min(10, len(dataloader))
    total_time, num_gen_tokens, target_steps = 0.0, 0, 0
    timers = harness.PhaseTimers(DEVICE) if args.benchmark else None
    for step, batch in enumerate(dataloader):
        if step >= num_eval_steps:
            break
        input_ids = batch[0].to(DEVICE)
        st, dt = harness.run_selfspec_batch(engine, input_ids, args.gamma, MAX_LEN_TARGET, eot_1, eot_2, streaming,
                                            timers=timers)
        total_time += dt
        target_steps += st.iters
        num_gen_tokens += int(st.num_nodes.sum() - (input_ids.shape[1] + 1) * BATCH_SIZE)
        if args.printoutput:
            for i in range(BATCH_SIZE):
                print_("Sequence ", i)
                print_(tokenizer.decode(st.output[i, args.prefix_len:st.num_nodes[i]]))
        print_("total time :{:.5f}s, time per iter :{:.5f}s, decoding step: {}, large model step: {}".format(
            total_time, total_time / target_steps, num_gen_tokens, target_steps))
        if args.benchmark:     # tests/SnapKV/selfspec_benchmark.py (same line as the longspec script)
            print_("target time :{:.5f}s, draft time :{:.5f}s, verify loop : {}, avg generate len per sentence: {}".format(
                timers.target / target_steps, timers.draft / target_steps, timers.verify_loop / target_steps,
                _avg_len(num_gen_tokens, target_steps, BATCH_SIZE)))
        if step < 5:
            total_time, num_gen_tokens, target_steps = 0.0, 0, 0
            if timers is not None:
                timers.reset()
        if use_tp:
            dist.barrier()
    print_(f"Final tokens per second :{num_gen_tokens / total_time}")
    return num_gen_tokens / total_time


def baseline_main(argv=None):
    """tests/baseline_benchmark.py: the autoregressive denominator of every speedup figure."""
    args = baseline_parser().parse_args(argv)
    assert args.prefix_len < args.max_len          # tests/baseline_benchmark.py:30-31
    assert args.max_len % 128 == 0
    DEVICE = _device()
    use_tp = len(args.rank_group) > 1
    global_group = None
    rank = 0
    print_ = builtins.print
    if use_tp:
        from .Engine.tp import init_dist
        rank, global_group = init_dist()
        if rank != args.rank_group[0]:
            print_ = lambda *a, **k: None
        DEVICE = f"cuda:{rank}" if torch.cuda.is_available() else "cpu"
    setup_seed(args.seed)
    print_(f"Using device={DEVICE}")
    from .Engine.SnapKV.backend import LMBackend
    engine = LMBackend(dtype=torch.bfloat16, device=DEVICE)
    engine.load_model(args.model, use_tp=use_tp, rank_group=args.rank_group, group=global_group)
    if args.compile:
        engine.compile()
    engine.setup_caches(max_batch_size=args.B, max_seq_length=args.max_len, kv_dtype=args.kv_dtype,
                        kv_layout=args.kv_layout)
    tokenizer = load_tokenizer(args.model_name)
    eot_1, eot_2 = _eot(tokenizer)
    print_(f"eot_1: {eot_1}, eot_2: {eot_2}")
    dataset = _dataset(args, tokenizer, engine.model.tok_embeddings.weight.shape[0])
    dataloader = DataLoader(dataset, batch_size=args.B, shuffle=False, drop_last=True)
    num_eval_steps = min(10, len(dataloader))
    total_time, model_steps = 0.0, 0
    for step, batch in enumerate(dataloader):
        if step >= num_eval_steps:
            break
        input_ids = batch[0].to(DEVICE)
        output, steps, dt = harness.run_baseline_batch(engine, input_ids, args.max_len, eot_1, eot_2)
        total_time += dt
        model_steps += steps
        if args.printoutput:
            for i in range(args.B):
                print_(tokenizer.decode(output[i, args.prefix_len:]))
        print_(f"Tokens per second :{args.B * (model_steps / total_time)}")
        if step < 5:
            total_time, model_steps = 0.0, 0
        if use_tp:
            dist.barrier()
    print_(f"Final tokens per second :{args.B * (model_steps / total_time)}")
    return args.B * (model_steps / total_time)
