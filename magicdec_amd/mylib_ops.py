"""The reference's seven ``mylib::*`` torch.library operators, bound to the C ABI (libmagicdec_hip.so).

This is the executable form of the drop-in boundary: a maintainer of
the reference replaces the bodies of its custom ops -- which call flashinfer -- by these registrations and keeps
every call site (``torch.ops.mylib.update_kv(...)``, ``self.rope(q, k, indptr, offsets)``,
``self.attn_decode(q, kv_cache)`` ...) unchanged.  Schemas are character-for-character the reference's:

    mylib::update_kv       Engine/utils.py:31-34
    mylib::rope            Engine/SnapKV/model.py:134-137     (mylib::draft_rope: Engine/SnapKV/model_draft.py, same schema)
    mylib::target_decode   Engine/SnapKV/backend.py:56-59     mylib::target_prefill  :68-71
    mylib::draft_decode    Engine/SnapKV/backend.py:96-99     mylib::draft_prefill   Engine/SnapKV/backend_draft.py:81-84

The four attention ops take only ``(q, kv_cache)``: in the reference the page table, head counts and the causal flag
live in a flashinfer wrapper object filled by a preceding host-side ``plan()`` (Engine/SnapKV/backend.py:148-159).
`PagedAttentionPlan` is that holder here: same constructor / ``plan`` / ``run`` surface as
``flashinfer.BatchPrefillWithPagedKVCacheWrapper`` as the reference uses it, backed by ``md_paged_attn``.

``register()`` defines the ops once per process (the reference defines them inside ``setup_caches`` and therefore
raises on a second call; here re-registration only re-binds the holders) for the "cuda" dispatch key -- on ROCm
builds of PyTorch that key is the HIP device -- plus shape-only fake kernels so that the ops stay opaque to tracing.
There is no CPU kernel: calling them on CPU tensors raises, like the reference.
"""
from __future__ import annotations

import torch

from . import ops

SCHEMAS = {
    "update_kv": "(Tensor k, Tensor v, Tensor kv_append_indptr, Tensor(a!) kv_cache, Tensor kv_page_indices, "
                 "Tensor kv_page_indptr, Tensor cachelen) -> ()",
    "rope": "(Tensor q, Tensor k, Tensor indptr, Tensor offsets) -> (Tensor ropeq, Tensor ropek)",
    "draft_rope": "(Tensor q, Tensor k, Tensor indptr, Tensor offsets) -> (Tensor ropeq, Tensor ropek)",
    "target_decode": "(Tensor q, Tensor kv_cache) -> Tensor",
    "target_prefill": "(Tensor q, Tensor kv_cache) -> Tensor",
    "draft_decode": "(Tensor q, Tensor kv_cache) -> Tensor",
    "draft_prefill": "(Tensor q, Tensor kv_cache) -> Tensor",
}
ATTENTION_OPS = ("target_decode", "target_prefill", "draft_decode", "draft_prefill")


class PagedAttentionPlan:
    """Stand-in for ``flashinfer.BatchPrefillWithPagedKVCacheWrapper`` as the reference drives it
    (Engine/SnapKV/backend.py:49-55 ctor, :148-159 plan, :60-64 run).  ``plan`` records the page table (device
    tensors, read by the kernel at run time) and the two host-side bounds the launch needs; ``run`` is one
    ``md_paged_attn`` launch on the current stream."""

    def __init__(self, float_workspace_buffer=None, kv_layout="NHD", use_cuda_graph=False, qo_indptr_buf=None,
                 paged_kv_indptr_buf=None, paged_kv_indices_buf=None, paged_kv_last_page_len_buf=None):
        if kv_layout not in ops.KV_LAYOUTS:
            raise ValueError(f"kv_layout must be one of {ops.KV_LAYOUTS} (flashinfer's two page layouts)")
        self.kv_layout = kv_layout
        self._ws = None
        self._plan = None
        self.kv_scales = None      # (k_scale, v_scale) for an fp8 cache

    def plan(self, qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len, num_qo_heads, num_kv_heads,
             head_dim, page_size, q_data_type=None, causal=True, sm_scale=None, **_):
        if q_data_type not in (None, torch.bfloat16):
            raise TypeError("the gfx950 attention kernels are bf16")
        qo = qo_indptr.to(torch.int32)
        ip = paged_kv_indptr.to(torch.int32)
        # plan() is the host-side step (flashinfer's plan synchronises too): bounds for the grid
        n_max = max(int((qo[1:] - qo[:-1]).max()), 1)
        max_pages = max(int((ip[1:] - ip[:-1]).max()), 1)
        self._plan = dict(qo=qo, indptr=ip, indices=paged_kv_indices.to(torch.int32),
                          last=paged_kv_last_page_len.to(torch.int32), H=num_qo_heads, KH=num_kv_heads, D=head_dim,
                          page_size=page_size, causal=causal, sm_scale=sm_scale, n_max=n_max, max_pages=max_pages)

    def run(self, q, kv_cache):
        p = self._plan
        if p is None:
            raise RuntimeError("PagedAttentionPlan.run before plan()")
        page_size, KH, _ = ops._kv_geom(kv_cache, self.kv_layout)
        if q.shape[1] != p["H"] or q.shape[2] != p["D"] or KH != p["KH"] or page_size != p["page_size"]:
            raise ValueError("q / kv_cache do not match the planned head geometry")
        if self._ws is None:
            self._ws = ops.AttnWorkspace(q.device)
        return ops.paged_attention(q, kv_cache, p["qo"], p["indices"], p["indptr"], p["last"], p["n_max"],
                                   p["max_pages"], self._ws, causal=p["causal"], sm_scale=p["sm_scale"],
                                   kv_scales=self.kv_scales, kv_layout=self.kv_layout)


class _Registry:
    lib = None
    plans = {name: None for name in ATTENTION_OPS}
    rope_tables = {"rope": None, "draft_rope": None}


def _rope_impl(which):
    def rope(q, k, indptr, offsets):
        tab = _Registry.rope_tables[which]
        if tab is None:
            raise RuntimeError(f"mylib::{which}: no RoPE table bound (mylib_ops.bind_rope)")
        B = indptr.numel() - 1
        rows = q.shape[0]
        n_max = rows if B == 0 else -(-rows // B)       # the reference always passes equal counts per request
        return ops.rope(q, k, indptr.to(torch.int32), offsets.to(torch.int32), tab, n_max=max(n_max, 1))
    return rope


def _attn_impl(which):
    def attn(q, kv_cache):
        plan = _Registry.plans[which]
        if plan is None:
            raise RuntimeError(f"mylib::{which}: no attention plan bound (mylib_ops.bind_plan)")
        return plan.run(q, kv_cache)
    return attn


def _update_kv(k, v, kv_append_indptr, kv_cache, kv_page_indices, kv_page_indptr, cachelen):
    """`cachelen` is the reference's name for kv_page_last_len (Engine/utils.py:33 vs :43)."""
    ops.update_kv(k, v, kv_append_indptr.to(torch.int32), kv_cache, kv_page_indices.to(torch.int32),
                  kv_page_indptr.to(torch.int32), cachelen.to(torch.int32))


def register():
    """Defines the seven schemas under the ``mylib`` namespace and installs the HIP-backed kernels.  Idempotent."""
    if _Registry.lib is not None:
        return
    lib = torch.library.Library("mylib", "FRAGMENT")
    for name, schema in SCHEMAS.items():
        lib.define(name + schema)
    lib.impl("update_kv", _update_kv, "CUDA")
    for which in ("rope", "draft_rope"):
        lib.impl(which, _rope_impl(which), "CUDA")
    for which in ATTENTION_OPS:
        lib.impl(which, _attn_impl(which), "CUDA")

    # shape-only kernels (the reference's register_fake blocks): opaque to torch.compile / fake tensors
    torch.library.register_fake("mylib::update_kv", lambda *a: None, lib=lib)
    for which in ("rope", "draft_rope"):
        torch.library.register_fake("mylib::" + which, lambda q, k, indptr, offsets: (torch.empty_like(q),
                                                                                      torch.empty_like(k)), lib=lib)
    for which in ATTENTION_OPS:
        torch.library.register_fake("mylib::" + which, lambda q, kv_cache: torch.empty_like(q), lib=lib)
    _Registry.lib = lib


def bind_plan(which: str, plan: PagedAttentionPlan):
    """Binds the wrapper object whose plan() state ``torch.ops.mylib.<which>(q, kv_cache)`` uses (the reference
    captures `self.decode_wrapper` etc. by closure, Engine/SnapKV/backend.py:60-64)."""
    if which not in ATTENTION_OPS:
        raise KeyError(which)
    _Registry.plans[which] = plan


def bind_rope(which: str, config, max_positions: int, device="cuda"):
    """Binds the RoPE constants ``mylib::rope`` / ``mylib::draft_rope`` close over in the reference
    (Engine/SnapKV/model.py:133-156: llama-3.1 smoothing iff both frequency factors are set)."""
    if which not in ("rope", "draft_rope"):
        raise KeyError(which)
    llama31 = config.high_freq_factor is not None and config.low_freq_factor is not None
    head_dim = config.dim // config.n_head
    _Registry.rope_tables[which] = ops.RopeTable(
        int(max_positions), head_dim, config.rope_base, config.scaling_factor,
        config.low_freq_factor if llama31 else None, config.high_freq_factor if llama31 else None,
        config.original_max_position_embeddings if llama31 else None, device=device)
    return _Registry.rope_tables[which]
