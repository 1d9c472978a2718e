"""Tensor parallelism of the decode path: KV-head sharding over the GPUs of one node (Engine/tp.py of the
reference, same function names), one process per GPU, RCCL ("nccl" backend on ROCm) over xGMI.

Partitioning (Engine/tp.py:36-52,67-207): rank r owns kv heads [start,end) -- remainder heads go to the lowest
ranks -- and the matching q heads; wqkv is sliced by those head ranges, wo by the same q-column range, w1/w3 are
row-chunked, w2 column-chunked, the lm head vocab-chunked; embedding, norms and every page table are
replicated.  No KV ever crosses GPUs; the only data-path collectives are the two sum-all-reduces per layer
(bf16 [B,n,dim]) and the two small ones of the argmax merge.
"""
import os
from itertools import accumulate

import torch
import torch.distributed as dist
from torch import nn


def _get_global_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def _get_world_size() -> int:
    return int(os.environ.get("LOCAL_WORLD_SIZE", "1"))


def _select_kv_heads(num_kv_heads, rank_group: list):
    """Head range of this process inside `rank_group` (Engine/tp.py:36-52)."""
    rank = rank_group.index(_get_global_rank())
    world_size = len(rank_group)
    base, rem = divmod(num_kv_heads, world_size)
    cum = list(accumulate(base + (1 if i < rem else 0) for i in range(world_size)))
    return (0 if rank == 0 else cum[rank - 1]), cum[rank]


def init_dist(draft_ranks=None):
    """World group over all local ranks (+ an optional draft sub-group), Engine/tp.py:54-64.  RCCL when a GPU
    is present; gloo otherwise (CPU tests)."""
    global_rank = _get_global_rank()
    world_size = _get_world_size()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if torch.cuda.is_available() and os.environ.get("MAGICDEC_TP_SINGLE_GPU", "0") == "1":
        # development / test hook: every rank uses GPU 0 (a 1-GPU box), gloo as the bootstrap and fallback
        # transport (RCCL refuses two ranks on one device); with MAGICDEC_ONESHOT_AR=1 the per-layer all-reduces
        # still run through the IPC one-shot kernel.  tests/test_gpu_engine.py::test_tp2_on_one_gpu
        torch.cuda.set_device(0)
        dist.init_process_group(backend="gloo", rank=global_rank, world_size=world_size)
    elif torch.cuda.is_available():
        torch.cuda.set_device(global_rank)
        dist.init_process_group(backend="nccl", rank=global_rank, world_size=world_size,
                                device_id=torch.device(f"cuda:{global_rank}"))
    else:
        dist.init_process_group(backend="gloo", rank=global_rank, world_size=world_size)
    global_group = dist.group.WORLD
    if draft_ranks is not None:
        return global_rank, global_group, dist.new_group(draft_ranks)
    return global_rank, global_group


def _slice_param(linear: nn.Linear, weight, size_attr, scales=None):
    """`scales`: the matching slice of a WeightOnlyInt8Linear's per-row scales for column-parallel (output-sliced)
    layers (Engine/tp.py:105-110,141-142 of the reference); row-parallel layers keep all their scales."""
    weight = weight.clone()      # own storage: the full tensor is released, GEMM operands stay contiguous
    if weight.dtype == torch.int8:
        linear.weight = weight
        if scales is not None:
            linear.scales = scales.clone()
    else:
        linear.weight = nn.Parameter(weight, requires_grad=False)
    setattr(linear, size_attr, weight.shape[0 if size_attr == "out_features" else 1])


def apply_tp(model, rank_group, group) -> None:
    """Shards `model` (a magicdec_amd Transformer) in place for this process and rewrites config.n_head /
    n_local_heads / dim to LOCAL values, as Engine/tp.py:184-207."""
    cfg = model.config
    D = cfg.head_dim
    g = cfg.n_head // cfg.n_local_heads
    s, e = _select_kv_heads(cfg.n_local_heads, rank_group)
    qs, qe, ks, ke = s * g * D, e * g * D, s * D, e * D
    q_size, kv_size = cfg.n_head * D, cfg.n_local_heads * D
    world = len(rank_group)
    rank = rank_group.index(_get_global_rank())
    for block in model.layers:
        att, ff = block.attention, block.feed_forward
        q, k, v = att.wqkv.weight.split([q_size, kv_size, kv_size], dim=0)
        sc = getattr(att.wqkv, "scales", None)
        if sc is not None:
            sq, sk, sv = sc.split([q_size, kv_size, kv_size], dim=0)
            sc = torch.cat((sq[qs:qe], sk[ks:ke], sv[ks:ke]))
        _slice_param(att.wqkv, torch.cat((q[qs:qe], k[ks:ke], v[ks:ke]), dim=0), "out_features", sc)
        if att.wqkv.bias is not None:
            bq, bk, bv = att.wqkv.bias.split([q_size, kv_size, kv_size], dim=0)
            att.wqkv.bias = nn.Parameter(torch.cat((bq[qs:qe], bk[ks:ke], bv[ks:ke])), requires_grad=False)
        _slice_param(att.wo, att.wo.weight[:, qs:qe], "in_features")
        chunk_sc = lambda lin: torch.chunk(lin.scales, world, dim=0)[rank] if hasattr(lin, "scales") else None
        _slice_param(ff.w1, torch.chunk(ff.w1.weight, world, dim=0)[rank], "out_features", chunk_sc(ff.w1))
        _slice_param(ff.w3, torch.chunk(ff.w3.weight, world, dim=0)[rank], "out_features", chunk_sc(ff.w3))
        _slice_param(ff.w2, torch.chunk(ff.w2.weight, world, dim=1)[rank], "in_features")
        att.process_group = group
        ff.process_group = group
    _slice_param(model.output, torch.chunk(model.output.weight, world, dim=0)[rank], "out_features",
                 torch.chunk(model.output.scales, world, dim=0)[rank] if hasattr(model.output, "scales") else None)
    lkh = e - s
    cfg.dim = cfg.dim * lkh // cfg.n_local_heads
    cfg.n_head = lkh * g
    cfg.n_local_heads = lkh
    for block in model.layers:
        block.attention.n_head, block.attention.n_local_heads, block.attention.dim = cfg.n_head, lkh, cfg.dim
    model.process_group = group
    model.world_size = dist.get_world_size(group)
    model.rank = dist.get_rank(group)
    from . import oneshot
    if oneshot.enabled():
        model._oneshot = oneshot.try_create(group)       # None (-> RCCL) unless every rank validated it
