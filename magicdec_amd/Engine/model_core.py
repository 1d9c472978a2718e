"""The transformer of the draft/verify decode path, MI355X-native.

One implementation serves the four `Transformer` classes of the reference
(Engine/SnapKV/model.py, Engine/SnapKV/model_draft.py, Engine/StreamingLLM/model.py,
Engine/StreamingLLM/model_draft.py): same constructor, parameter names (so a
reference `model.pth` loads unchanged), `config` object, `setup_caches` and the
forward / verify / draft_forward / prefill / draft_prefill methods returning
token ids [B, n].  What differs from the reference is everything underneath:

* no flashinfer, no torch.library ops, no torch.compile: the attention, RoPE,
  KV append, SnapKV select, StreamingLLM eviction, norms, SiLU*mul and argmax
  are hand-written gfx950 kernels reached through the C ABI (magicdec_amd.ops);
* per layer and step: 4 linears (w1|w3 fused into one) -- each on the kernel the
  measured policy of Engine/gemm_policy.py picks: md_linear_fused (launch-bound small
  products: the product AND its consumer -- rope+append, residual add, SiLU*mul, the
  RMSNorm in front -- in one launch: a 1B draft layer is 5-7 launches), md_linear (long
  weight streams; its split-K combine also adds the residual and normalises) or
  hipBLASLt via F.linear (prefill-sized M, the M = 256 verify GEMMs) -- plus the
  attention and whatever the chosen linears did not absorb (rope+append, add+rmsnorm,
  silu*mul kernels);
* the page table is read on the device by the attention kernel -- there is no
  host-side plan() and no host<->device sync anywhere in a step, so a step can be
  captured into a hipGraph (Engine/graph.py);
* KV pages live in one slab per layer sized for 288 GB HBM (no page migration).

bf16 rounding points follow the reference (see oracle/magicdec_ref.py): linear
outputs bf16, fp32 RMSNorm -> bf16 -> *weight in bf16, residual adds in bf16,
SiLU in bf16 then the product in bf16, fp32 softmax, argmax on bf16 logits with
lowest-index tie-break.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn import functional as F

from .. import ops


def find_multiple(n: int, k: int) -> int:
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class ModelArgs:
    """Same fields and defaults as the reference's ModelArgs (Engine/SnapKV/model.py:17-43)."""
    block_size: int = 2048
    vocab_size: int = 32000
    n_layer: int = 32
    n_head: int = 32
    dim: int = 4096
    intermediate_size: int = None
    n_local_heads: int = -1
    head_dim: int = 64
    rope_base: float = 10000
    norm_eps: float = 1e-5
    scaling_factor: float = 1.0
    low_freq_factor: int = None
    high_freq_factor: int = None
    original_max_position_embeddings: int = None
    qkv_bias: bool = False

    def __post_init__(self):
        if self.n_local_heads == -1:
            self.n_local_heads = self.n_head
        if self.intermediate_size is None:
            self.intermediate_size = find_multiple(int(2 * 4 * self.dim / 3), 256)
        self.head_dim = self.dim // self.n_head

    @classmethod
    def from_name(cls, name: str):
        """Exact key, else the longest config key contained in `name` (checkpoint directory name),
        as Engine/SnapKV/model.py:45-57."""
        if name in transformer_configs:
            return cls(**transformer_configs[name])
        hits = [c for c in transformer_configs if c.lower() in str(name).lower()]
        if not hits:       # the reference indexes an empty list here (IndexError); same exception type, with a message
            raise IndexError(f"no transformer config matches '{name}' (known: {', '.join(transformer_configs)})")
        hits.sort(key=len, reverse=True)
        if len(hits) > 1:
            assert len(hits[0]) != len(hits[1]), name
        return cls(**transformer_configs[hits[0]])


# model zoo of the reference (Engine/SnapKV/model.py:60-79)
transformer_configs = {
    "llama-2-7b": dict(block_size=4096, n_layer=32, n_head=32, dim=4096),
    "llama-2-7b-32k": dict(block_size=32768, n_layer=32, dim=4096, vocab_size=32000, scaling_factor=8),
    "llama-2-13b": dict(block_size=4096, n_layer=40, n_head=40, dim=5120),
    "llama-2-70b": dict(block_size=4096, n_layer=80, n_head=64, dim=8192, n_local_heads=8, intermediate_size=28672),
    "llama-3-8b": dict(block_size=8192, n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336,
                       vocab_size=128256, rope_base=500000),
    "llama-3-70b": dict(block_size=8192, n_layer=80, n_head=64, n_local_heads=8, dim=8192, intermediate_size=28672,
                        vocab_size=128256, rope_base=500000),
    "68m": dict(block_size=2048, n_layer=2, n_head=12, n_local_heads=12, dim=768, intermediate_size=3072,
                vocab_size=32000),
    "tinyllama": dict(block_size=2048, n_layer=22, n_head=32, n_local_heads=4, dim=2048, intermediate_size=5632,
                      vocab_size=32000),
    "llama-3.1-8b": dict(block_size=131072, n_layer=32, n_head=32, n_local_heads=8, dim=4096,
                         intermediate_size=14336, vocab_size=128256, rope_base=500000.0, scaling_factor=8,
                         high_freq_factor=4, low_freq_factor=1, original_max_position_embeddings=8192),
    "llama-3.1-70b": dict(block_size=131072, n_layer=80, n_head=64, n_local_heads=8, dim=8192,
                          intermediate_size=28672, vocab_size=128256, rope_base=500000.0, scaling_factor=8,
                          high_freq_factor=4, low_freq_factor=1, original_max_position_embeddings=8192),
    "llama-3.2-1b": dict(block_size=131072, n_layer=16, n_head=32, n_local_heads=8, dim=2048, intermediate_size=8192,
                         vocab_size=128256, rope_base=500000.0, scaling_factor=32, high_freq_factor=4,
                         low_freq_factor=1, original_max_position_embeddings=8192),
    "Qwen2.5-7b": dict(block_size=131072, n_layer=28, n_head=28, n_local_heads=4, dim=3584, intermediate_size=18944,
                       vocab_size=152064, rope_base=1000000.0, qkv_bias=True, norm_eps=1e-6),
    "Qwen2.5-14b": dict(block_size=131072, n_layer=48, n_head=40, n_local_heads=8, dim=5120, intermediate_size=13824,
                        vocab_size=152064, rope_base=1000000.0, qkv_bias=True, norm_eps=1e-6),
    "Qwen2.5-32b": dict(block_size=131072, n_layer=64, n_head=40, n_local_heads=8, dim=5120, intermediate_size=27648,
                        vocab_size=152064, rope_base=1000000.0, qkv_bias=True, norm_eps=1e-6),
    "Yi-1.5-6b": dict(block_size=4096, n_layer=32, n_head=32, n_local_heads=4, dim=4096, intermediate_size=11008,
                      vocab_size=64000, rope_base=500000.0),
    "Yi-1.5-34b-32k": dict(block_size=32768, n_layer=60, n_head=56, n_local_heads=8, dim=7168,
                           intermediate_size=20480, vocab_size=64000, rope_base=500000.0),
    "Mistral-7B-v0.1": dict(n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336,
                            vocab_size=32000),
    "Mistral-7B-v0.3": dict(n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336,
                            vocab_size=32768, rope_base=1000000.0),
}

SINK = 16          # StreamingLLM sink tokens (Engine/StreamingLLM/model_draft.py:124)
POOL_KERNEL = 5    # SnapKV avg-pool width (Engine/SnapKV/model.py:169)
KV_DTYPES = {"bf16": torch.bfloat16, "fp8": ops.FP8_DTYPE}   # fp8 = OCP e4m3fn full-context cache (BASELINE cfg5)
FP8_MAX = 448.0
FP8_MARGIN = 1.5


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return ops.rmsnorm(x, self.weight, self.eps)


class KVCache(nn.Module):
    """Paged caches of one layer: `kv_cache` [pages, 2, 128, KH, D] and, for speculation, `draft_cache`."""

    def __init__(self, max_num_pages, page_size, n_heads, head_dim, dtype=torch.bfloat16, draft_max_num_pages=0,
                 kv_len=0, kv_dtype=torch.bfloat16, kv_layout="NHD"):
        """kv_dtype=float8_e4m3fn stores `kv_cache` as OCP e4m3fn bytes with static per-kv-head scales
        (`k_scale`, `v_scale`, float32 [KH]; x ~ byte * scale); `draft_cache` is always bf16.
        kv_layout="HND" stores `kv_cache` as [pages, 2, KH, 128, D] (rows of a kv head contiguous: the streaming
        reads of the verify step become 128-row runs); `draft_cache` is always NHD."""
        super().__init__()
        if kv_layout not in ops.KV_LAYOUTS:
            raise ValueError(f"kv_layout must be one of {ops.KV_LAYOUTS}")
        self.layout = kv_layout
        if max_num_pages > 0:
            shape = ((max_num_pages, 2, n_heads, page_size, head_dim) if kv_layout == "HND"
                     else (max_num_pages, 2, page_size, n_heads, head_dim))
            self.register_buffer("kv_cache", torch.zeros(shape, dtype=kv_dtype), persistent=False)
        self.fp8 = kv_dtype == ops.FP8_DTYPE
        if self.fp8:
            self.register_buffer("k_scale", torch.ones(n_heads, dtype=torch.float32), persistent=False)
            self.register_buffer("v_scale", torch.ones(n_heads, dtype=torch.float32), persistent=False)
        self.calibrated = False
        if draft_max_num_pages > 0:
            self.register_buffer("draft_cache",
                                 torch.zeros((draft_max_num_pages, 2, page_size, n_heads, head_dim), dtype=dtype),
                                 persistent=False)
        self.kv_len = kv_len
        self.page_size = page_size

    def scales(self, which="kv_cache"):
        return (self.k_scale, self.v_scale) if (self.fp8 and which == "kv_cache") else None

    def layout_of(self, which="kv_cache"):
        return self.layout if which == "kv_cache" else "NHD"

    def calibrate(self, k, v, margin=FP8_MARGIN):
        """Static scales from the first prefill chunk: K's bound is the largest rotary-pair norm (what any
        rotation angle can turn into one component), V's the largest |v|; `margin` leaves headroom for later
        tokens (values beyond it saturate at +-448 in the quantiser)."""
        rows, KH, D = k.shape
        kb = k.float().view(rows, KH, D // 2, 2).square().sum(-1).amax(dim=(0, 2)).sqrt()
        vb = v.float().abs().amax(dim=(0, 2))
        self.k_scale.copy_((kb * (margin / FP8_MAX)).clamp_min(2.0 ** -20))
        self.v_scale.copy_((vb * (margin / FP8_MAX)).clamp_min(2.0 ** -20))
        self.calibrated = True


class Attention(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        assert config.dim % config.n_head == 0
        total_head_dim = (config.n_head + 2 * config.n_local_heads) * config.head_dim
        self.wqkv = nn.Linear(config.dim, total_head_dim, bias=config.qkv_bias)
        self.wo = nn.Linear(config.dim, config.dim, bias=False)
        self.kv_cache: Optional[KVCache] = None
        self.process_group = None
        self.n_head, self.head_dim = config.n_head, config.head_dim
        self.n_local_heads, self.dim = config.n_local_heads, config.dim
        self._register_load_state_dict_pre_hook(self.load_hook)

    def load_hook(self, state_dict, prefix, *args):
        """Accept un-fused wq/wk/wv checkpoints (Engine/SnapKV/model.py:309-320)."""
        if prefix + "wq.weight" in state_dict:
            state_dict[prefix + "wqkv.weight"] = torch.cat([state_dict.pop(prefix + "wq.weight"),
                                                            state_dict.pop(prefix + "wk.weight"),
                                                            state_dict.pop(prefix + "wv.weight")])
        if prefix + "wq.bias" in state_dict:
            state_dict[prefix + "wqkv.bias"] = torch.cat([state_dict.pop(prefix + "wq.bias"),
                                                          state_dict.pop(prefix + "wk.bias"),
                                                          state_dict.pop(prefix + "wv.bias")])


class FeedForward(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.w1 = nn.Linear(config.dim, config.intermediate_size, bias=False)
        self.w3 = nn.Linear(config.dim, config.intermediate_size, bias=False)
        self.w2 = nn.Linear(config.intermediate_size, config.dim, bias=False)
        self.process_group = None


class TransformerBlock(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.attention = Attention(config)
        self.feed_forward = FeedForward(config)
        self.ffn_norm = RMSNorm(config.dim, config.norm_eps)
        self.attention_norm = RMSNorm(config.dim, config.norm_eps)


@dataclass
class PageTable:
    """(indices, indptr, last_page_len) of one paged cache + host-known bounds for the kernels."""
    indices: torch.Tensor
    indptr: torch.Tensor
    last_page_len: torch.Tensor
    max_pages: int  # host upper bound of pages per request


class Transformer(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.config = config
        self.tok_embeddings = nn.Embedding(config.vocab_size, config.dim)
        self.layers = nn.ModuleList(TransformerBlock(config) for _ in range(config.n_layer))
        self.norm = RMSNorm(config.dim, eps=config.norm_eps)
        self.output = nn.Linear(config.dim, config.vocab_size, bias=False)
        self.world_size = None
        self.rank = None
        self.process_group = None
        self._ready = False
        self.skip_head = False
        # optional static fp8 KV scales [(k_scale[KH], v_scale[KH]) per layer], e.g. from an offline calibration;
        # when set they replace the first-chunk calibration of every encode()
        self.kv_scale_override = None

    @classmethod
    def from_name(cls, name: str):
        return cls(ModelArgs.from_name(name))

    # ------------------------------------------------------------------ setup
    def setup_caches(self, num_pages, page_size=128, spec=False, draft_num_pages=0, draft_budget=0, window_size=32,
                     max_positions=None, streaming=False, kv_dtype="bf16", kv_layout="NHD", decode_rows=None):
        """Allocates the per-layer KV slabs and the device-side constants of the step
        (Engine/SnapKV/model.py:127-169 / StreamingLLM/model_draft.py:157-189 without the op registration)."""
        c = self.config
        dev = self.output.weight.device
        dtype = self.output.weight.dtype if self.output.weight.dtype == torch.float16 else torch.bfloat16
        if dtype != torch.bfloat16:
            raise NotImplementedError("the gfx950 kernels are bf16")
        if kv_dtype not in KV_DTYPES:
            raise ValueError(f"kv_dtype must be one of {sorted(KV_DTYPES)}")
        if (kv_dtype == "fp8" or kv_layout != "NHD") and streaming and not spec:
            raise NotImplementedError("the StreamingLLM ring cache is bf16 NHD only (in-place shift + re-rotation)")
        head_dim = c.dim // c.n_head
        self.page_size = page_size
        self.spec, self.streaming = spec, streaming
        self.draft_budget, self.window_size = draft_budget, window_size
        for b in self.layers:
            b.attention.kv_cache = KVCache(num_pages, page_size, c.n_local_heads, head_dim, dtype,
                                           draft_num_pages if spec else 0, kv_len=draft_budget,
                                           kv_dtype=KV_DTYPES[kv_dtype], kv_layout=kv_layout).to(dev)
        if max_positions is None:
            max_positions = max(num_pages, draft_num_pages, 1) * page_size + 256
        llama31 = c.high_freq_factor is not None and c.low_freq_factor is not None
        self.rope_table = ops.RopeTable(int(max_positions), head_dim, c.rope_base, c.scaling_factor,
                                        c.low_freq_factor if llama31 else None, c.high_freq_factor if llama31 else None,
                                        c.original_max_position_embeddings if llama31 else None, device=dev)
        self.workspace = ops.AttnWorkspace(dev)
        self._rot_scratch = None
        # row counts of the decode / verify steps this engine will run (the back-end knows: batch x {1, 2, dec_len}); None =
        # unknown: every count 1..256 is assumed (the streaming-layout copies and the row-major tensors are both kept)
        self.decode_rows = tuple(sorted(set(int(r) for r in decode_rows))) if decode_rows else None
        self._fuse_weights()
        self._ready = True

    def _fuse_weights(self):
        """w1|w3 -> one [2I, dim] GEMM operand; the original parameters become views of it (no extra HBM).
        Weight-only int8 models (Engine/quantize.py) fuse the int8 rows and their per-row scales the same way."""
        self._w13, self._s13 = [], []
        for b in self.layers:
            ff = b.feed_forward
            w13 = torch.cat([ff.w1.weight.data, ff.w3.weight.data], dim=0).contiguous()
            inter = ff.w1.weight.shape[0]
            if w13.dtype == torch.int8:
                ff.w1.weight, ff.w3.weight = w13[:inter], w13[inter:]
                self._s13.append(torch.cat([ff.w1.scales, ff.w3.scales]).contiguous())
            else:
                ff.w1.weight = nn.Parameter(w13[:inter], requires_grad=False)
                ff.w3.weight = nn.Parameter(w13[inter:], requires_grad=False)
                self._s13.append(None)
            self._w13.append(w13)
        self._pack_weights()

    def _pack_weights(self, force=False):
        """Streaming-layout copies (ops.PackedWeight) of the weights the hand-written GEMMs serve in decode / verify
        steps -- md_linear for the long weight streams, md_linear_fused for the launch-bound small products
        (Engine/gemm_policy.py); keyed by the id of the row-major tensor the step would otherwise use.  The row-major
        tensors stay for the prefill-sized library GEMMs; the extra bytes are reported once (`packed_bytes`)."""
        from .gemm_policy import want_packed
        self._packed = {}
        self._released, self._by_id, self._roles = set(), {}, {}
        if not self.output.weight.is_cuda and not force:      # force: host tests of the bookkeeping (no kernel runs)
            return

        todo = []
        self._roles = {}                                # id(weight) -> (role, layer index | None)
        for i, b in enumerate(self.layers):
            for w, sw, role in ((self._w13[i], True, "w13"), (b.attention.wqkv.weight, False, "wqkv"),
                                (b.attention.wo.weight, False, "wo"), (b.feed_forward.w2.weight, False, "w2")):
                todo.append((w, sw))
                self._roles[id(w)] = (role, i)
        todo.append((self.output.weight, False))
        self._roles[id(self.output.weight)] = ("head", None)
        todo = [(w, sw) for w, sw in todo if w.shape[1] % 16 == 0 and self._wants_packed(self._roles[id(w)][0], w)]
        # headroom check: the copies double the served weights' footprint (17 GB for the 8B + 1B pair)
        # and are made AFTER the KV slabs exist; if they do not fit beside them with a margin for the step's workspaces,
        # serve everything from the row-major tensors (library GEMMs / row-major md_linear) instead of failing later.
        # MAGICDEC_PACKED_COPIES=0 switches them off outright (Engine/gemm_policy.py).
        need = sum(w.numel() * w.element_size() for w, _ in todo)
        try:
            torch.cuda.empty_cache()       # mem_get_info does not see blocks the caching allocator holds but no longer uses
            free = torch.cuda.mem_get_info(self.output.weight.device)[0]
        except Exception:              # not a HIP device (tests with stand-in ops): nothing to check
            free = None
        margin = 6 << 30
        fits = free is None or need + margin <= free
        # ONE decision for all ranks that must run the same kernels: the ranks of a TP group -- and the
        # ranks running a REPLICATED draft, which stay in lock-step only because they compute bit-identical tokens -- must
        # not choose per rank from their own free memory: a rank that falls back to the library GEMMs rounds differently,
        # can flip a near-tie argmax, and then verifies another draft than its peers (diverging accept lengths, a hang in
        # the next collective).  Everyone packs or nobody does.
        grp = self.process_group if self.process_group is not None else getattr(self, "replica_group", None)
        if grp is not None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                t = torch.tensor([1 if fits else 0], dtype=torch.int32,
                                 device=self.output.weight.device if dist.get_backend(grp) == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=grp)
                if fits and int(t.item()) == 0:
                    print("[magicdec_amd] streaming-layout weight copies fit on this rank but not on every rank of its "
                          "group: not made anywhere (all ranks must run the same kernels)")
                fits = bool(int(t.item()))
        if not fits:
            if free is not None and need + margin > free:
                print(f"[magicdec_amd] streaming-layout weight copies need {need / 2**30:.1f} GiB but only {free / 2**30:.1f} GiB "
                      f"are free beside the KV cache (margin {margin >> 30} GiB): not made -- the decode / verify linears run on "
                      f"the row-major weights (slower); shrink the cache or set MAGICDEC_PACKED_COPIES=0 to silence this")
            return
        for w, sw in todo:
            pw = ops.PackedWeight(w.data if isinstance(w, nn.Parameter) else w, swiglu=sw)
            self._packed[id(w)] = pw
        self._released = set()          # ids of weights whose row-major tensor is currently released (release_rowmajor)
        self._by_id = {id(w): w for w, _ in todo}

    @property
    def packed_bytes(self):
        """Bytes of weights held TWICE right now: a streaming-layout copy beside a resident row-major tensor.  After
        release_rowmajor() that is only what decode reads in both layouts (at different row counts)."""
        pk = getattr(self, "_packed", None)
        if not pk:
            return 0
        rel = getattr(self, "_released", set())
        return sum(p.data.numel() * p.data.element_size() for k, p in pk.items() if k not in rel)

    @property
    def released_bytes(self):
        """Bytes of row-major weight tensors currently released (their streaming-layout copy is the one resident copy)."""
        return sum(self._by_id[k].shape[0] * self._by_id[k].shape[1] * self._by_id[k].element_size()
                   for k in getattr(self, "_released", ()))

    @packed_bytes.setter
    def packed_bytes(self, v):          # legacy assignment in _pack_weights: ignored (the property is computed)
        pass

    # ------------------------------------------------------------------ one resident copy per weight (round 6)
    def _decode_hows(self, role, N, K, int8, M, packed=True):
        """The implementations a decode / verify step of M rows may run this weight on ("fused" | "split" | "block" |
        "skinny" | "lib"), assuming a streaming-layout copy exists iff `packed` -- the decisions of _std_step /
        _proj_add_norm / _linear, through the same policy and shape-support functions, without running anything."""
        from .gemm_policy import choose, use_split
        pk = True if packed else None
        tp = self.process_group is not None
        l0 = self.layers[0]

        def resid_fused(w):          # is the residual epilogue (-> a DEFERRED norm for its consumer) taken for this projection?
            n, k = w.shape
            return not tp and w.dtype != torch.int8 and self._fused_ok(M, n, k, True, False, "resid", False)
        if role == "w13":
            pros = {resid_fused(l0.attention.wo.weight)}                       # its input comes from the wo sub-layer
            return {self._linear_how(M, N, K, True, int8, pk, "swiglu", pro) for pro in pros}
        if role == "wqkv":
            pros = {False, resid_fused(l0.feed_forward.w2.weight)}             # layer 0 reads a materialised norm
            hows = set()
            for pro in pros:
                fused = self.config.head_dim in (64, 128) and self._fused_ok(M, N, K, pk, int8, "qkv", pro)
                hows.add("fused" if fused else self._linear_how(M, N, K, False, int8, pk, "plain", False))
            return hows
        if role in ("wo", "w2"):
            if not tp and self._fused_ok(M, N, K, pk, int8, "resid", False):
                return {"fused"}
            if not tp and pk is not None and not int8 and use_split(M, N, K, "resid") and ops.fused_split_supported(M, N, K):
                return {"split"}
            if not tp:
                how = choose(M, N, K, False, int8, pk is not None, "resid")
                if how == "block" and ops.linear_block_supported(M, N, K):
                    return {"block"}
                if how in ("skinny", "block") and ops.linear_add_rmsnorm_supported(M, N, K):
                    return {"skinny"}
        return {self._linear_how(M, N, K, False, int8, pk, "plain", False)}          # wo / w2 fall-through, lm head

    def _wants_packed(self, role, w):
        """A streaming-layout copy of this weight?  With known decode row counts: iff some step of this engine runs it on a
        hand-written kernel; unknown: gemm_policy.want_packed (any row count 1..256)."""
        from .gemm_policy import want_packed, _MODE, _PACKED
        N, K = w.shape
        int8 = w.dtype == torch.int8
        if self.decode_rows is None or _MODE != "auto" or int8:
            return want_packed(N, K, role == "w13", int8)
        if _PACKED == "0" or K % 64:
            return False
        return any(self._decode_hows(role, N, K, int8, M) - {"lib"} for M in self.decode_rows if M <= 256)

    def _rowmajor_in_decode(self, w):
        """Does some decode / verify step of this engine hand the ROW-MAJOR tensor of `w` to a kernel (the library GEMM)?"""
        role, _ = self._roles[id(w)]
        if self._packed.get(id(w)) is None or w.dtype == torch.int8 or self.decode_rows is None:
            return True
        N, K = w.shape
        return any(M > 256 or "lib" in self._decode_hows(role, N, K, False, M) for M in self.decode_rows)

    def release_rowmajor(self):
        """After prefill: free the row-major tensor of every weight that decode only ever reads in the streaming layout
        (every packed weight used to be held twice, 16.9 GB at configs[2]).  The Parameter objects stay
        (same id, shape and dtype: a zero-stride view of one element) so that every look-up keyed on them keeps working;
        restore_rowmajor() re-materialises them from the streaming copy before the next prefill.  MAGICDEC_KEEP_ROWMAJOR=1
        switches this off."""
        import os
        if not getattr(self, "_packed", None) or os.environ.get("MAGICDEC_KEEP_ROWMAJOR", "0") == "1":
            return 0
        freed = 0
        for k, w in self._by_id.items():
            if k in self._released or self._rowmajor_in_decode(w):
                continue
            N, K = w.shape
            fake = torch.empty(1, dtype=w.dtype, device=w.device).expand(N, K)
            role, i = self._roles[k]
            freed += N * K * w.element_size()
            if role == "w13":
                ff = self.layers[i].feed_forward
                inter = N // 2
                w.set_(fake)
                ff.w1.weight.data = fake[:inter]
                ff.w3.weight.data = fake[inter:]
            else:
                w.data = fake
            self._released.add(k)
        return freed

    def restore_rowmajor(self):
        """Before a prefill: every released row-major tensor back from its streaming-layout copy (one permute each)."""
        for k in list(getattr(self, "_released", ())):
            w = self._by_id[k]
            rows = self._packed[k].unpack().contiguous()
            role, i = self._roles[k]
            if role == "w13":
                ff = self.layers[i].feed_forward
                inter = rows.shape[0] // 2
                w.set_(rows)
                ff.w1.weight.data = w[:inter]
                ff.w3.weight.data = w[inter:]
            else:
                w.data = rows
            self._released.discard(k)

    def _resident(self, w):
        """The row-major tensor of `w` for a library GEMM; a released one is unpacked for this call only (a decode step the
        policy prediction did not foresee: correct, slow, and reported once)."""
        if id(w) in getattr(self, "_released", ()):
            if not getattr(self, "_warned_transient", False):
                self._warned_transient = True
                print(f"[magicdec_amd] a library GEMM needs the released row-major weight {tuple(w.shape)} in a decode step "
                      f"of {getattr(self, '_last_M', '?')} rows (not among decode_rows={self.decode_rows}): unpacked per call")
            return self._packed[id(w)].unpack()
        return w

    # ------------------------------------------------------------------ building blocks
    def _reduce(self, y, group):
        """The per-layer sum-all-reduce (C1).  RCCL by default; the one-shot xGMI kernel when tp.apply_tp attached
        one (MAGICDEC_ONESHOT_AR=1) and the message fits its registered buffer (decode steps do, prefill chunks
        do not)."""
        if group is not None:
            ar = getattr(self, "_oneshot", None)
            if ar is not None and ar.fits(y):
                ar.all_reduce_(y)
            else:
                dist.all_reduce(y, group=group)
        return y

    def _linear(self, x2d, lin, swiglu_w13=None, resid=None, want_ssq=False):
        """One linear of a step.  Four implementations, chosen per shape by the measured rules of
        Engine/gemm_policy.py: md_linear_fused (csrc/tilegemm.hip: the launch-bound small products -- draft-model
        linears, tensor-parallel shards -- in one launch together with their epilogue), md_linear_block
        (csrc/blockgemm.hip: the wide 129..256-row products of a verify step), md_linear (csrc/gemm.hip: the long weight
        streams at <= 128 rows, split-K + combine) or the library GEMM (prefill-sized M, and whatever the A/B gave it).
        `swiglu_w13 = (w13, s13)`: the fused w1|w3 product with the SiLU*mul epilogue.
        `resid`: return bf16(resid + linear) instead (the residual add of the block; only honoured by the fused
        kernel -- callers check `_fused_here` first); with `want_ssq` also the per-row partial sums of squares of the
        result (the producer half of a deferred RMSNorm).  `x2d` may be an `ops.DeferredNorm`: the fused w1|w3 kernel
        applies it on the fly, every other path materialises it first."""
        from .gemm_policy import choose, skinny_absorbs_norm
        if swiglu_w13 is not None:
            w, scales, bias = swiglu_w13[0], swiglu_w13[1], None
        else:
            w, scales, bias = lin.weight, getattr(lin, "scales", None), lin.bias
        pro = x2d if isinstance(x2d, ops.DeferredNorm) else None      # a norm still to be applied to the input
        if pro is not None:
            x2d = pro.h
        M, K = x2d.shape
        N = w.shape[0]
        self._last_M = M
        swiglu = swiglu_w13 is not None
        pk = self._packed.get(id(w))
        kind = "swiglu" if swiglu else ("resid" if resid is not None else "plain")
        how = (choose(M, N, K, swiglu, w.dtype == torch.int8, pk is not None, kind, pro is not None and swiglu)
               if x2d.is_cuda else "lib")
        if how == "fused" and ops.fused_linear_supported(M, N, K):
            if pro is not None and not swiglu:
                x2d, pro = pro.materialize(), None
            return ops.fused_linear(x2d, pk, bias, swiglu=swiglu, resid=resid, want_ssq=want_ssq, pro=pro)
        assert resid is None, "the residual epilogue exists on the fused kernel only"
        skinny = how in ("fused", "skinny", "block") and ops.linear_supported(M, N, K, swiglu)
        if how == "block" and ops.linear_block_supported(M, N, K, swiglu):
            skinny = False
        if pro is not None and skinny and w.dtype == torch.bfloat16 and skinny_absorbs_norm(M):
            # the weight-streaming kernel normalises its activation slabs on the way to LDS (md_linear_normed)
            return ops.linear(x2d, pk if pk is not None else w, bias, None, swiglu, self.workspace, pro=pro)
        if pro is not None:
            x2d = pro.materialize()
        if how == "block" and ops.linear_block_supported(M, N, K, swiglu):
            return ops.linear_block(x2d, pk, bias, swiglu, self.workspace)
        if skinny:
            return ops.linear(x2d, pk if pk is not None else w, bias, scales, swiglu, self.workspace)
        if w.dtype == torch.int8:      # WeightOnlyInt8Linear.forward (Engine/quantize.py:84-86), dequantised on the fly
            h = F.linear(x2d, w.to(dtype=x2d.dtype)) * scales
        else:
            h = F.linear(x2d, self._resident(w), bias)
        if swiglu:
            inter = N // 2
            return ops.silu_mul(h[:, :inter], h[:, inter:])
        return h

    def _linear_how(self, M, N, K, swiglu, int8, pk, kind, has_pro):
        """Which implementation `_linear` ends up on for this shape: "fused" | "block" | "skinny" | "lib" -- the same
        sequence of policy and shape-support checks, without running anything (release_rowmajor's prediction)."""
        from .gemm_policy import choose
        how = choose(M, N, K, swiglu, int8, pk is not None, kind, has_pro and swiglu)
        if how == "fused" and ops.fused_linear_supported(M, N, K):
            return "fused"
        if how == "block" and ops.linear_block_supported(M, N, K, swiglu):
            return "block"
        if how in ("fused", "skinny", "block") and ops.linear_supported(M, N, K, swiglu):
            return "skinny"
        return "lib"

    def _fused_ok(self, M, N, K, pk, int8, kind, absorbs_norm):
        from .gemm_policy import choose
        return (pk is not None and not int8 and choose(M, N, K, False, False, True, kind, absorbs_norm) == "fused"
                and ops.fused_linear_supported(M, N, K))

    def _fused_here(self, x2d, w, kind, absorbs_norm=False):
        """Does the fused kernel serve the linear `w` with epilogue `kind` ("qkv" | "resid") for these rows?
        (policy + shape support + a packed bf16 copy)"""
        from .gemm_policy import choose
        M, K = x2d.shape
        N = w.shape[0]
        pk = self._packed.get(id(w))
        return (x2d.is_cuda and pk is not None and w.dtype != torch.int8
                and choose(M, N, K, False, False, True, kind, absorbs_norm) == "fused"
                and ops.fused_linear_supported(M, N, K))

    def _proj_add_norm(self, inp, lin, x, norm, group):
        """Output projection of a sub-layer (wo / w2), the residual add and the RMSNorm for the next sub-layer:
        (h, y) = (x + sum_ranks(inp . W^T), rmsnorm(h) * w).  Without tensor parallelism and on a launch-bound shape the
        projection and the residual add are one launch (md_linear_fused, MD_FL_RESID) and y is returned as an
        `ops.DeferredNorm`; otherwise the projection, then `_reduce_add_norm` (collective + fused add + norm)."""
        if group is None and x.is_contiguous() and self._fused_here(inp, lin.weight, "resid"):
            # the norm is DEFERRED: the residual epilogue leaves the per-row partial sums of squares, and the linear that
            # consumes the normalised rows (w1|w3, the next layer's wqkv) applies rmsnorm on the fly; a consumer that
            # cannot (library GEMM, lm head) materialises it with the stand-alone kernel
            h, ssq = self._linear(inp, lin, resid=x, want_ssq=True)
            return h, ops.DeferredNorm(h, ssq, norm.weight, norm.eps)
        if group is None and inp.is_cuda and x.stride(-1) == 1:
            # the weight-streaming kernel's split-K combine launch also adds the residual and normalises
            from .gemm_policy import choose, use_split
            w = lin.weight
            M, K = inp.shape
            pk = self._packed.get(id(w))
            if (pk is not None and w.dtype == torch.bfloat16 and inp.stride(-1) == 1
                    and use_split(M, w.shape[0], K, "resid") and ops.fused_split_supported(M, w.shape[0], K)):
                # deep narrow projection of a draft step (the 1B w2): the tile kernel with K split over workgroups; its
                # combine launch IS the residual add + RMSNorm launch (md_linear_fused_split_add_rmsnorm)
                return ops.fused_split_linear_add_rmsnorm(inp, pk, x, norm.weight, norm.eps, lin.bias, self.workspace)
            how = choose(M, w.shape[0], K, False, w.dtype == torch.int8, pk is not None, "resid")
            if how == "block" and ops.linear_block_supported(M, w.shape[0], K) and inp.stride(-1) == 1:
                # 129..256 rows: the block-tile GEMM, same combine launch (residual add + RMSNorm)
                return ops.linear_block_add_rmsnorm(inp, pk, x, norm.weight, norm.eps, lin.bias, self.workspace)
            if (how in ("skinny", "block")
                    and ops.linear_add_rmsnorm_supported(M, w.shape[0], K)):
                return ops.linear_add_rmsnorm(inp, pk if pk is not None else w, x, norm.weight, norm.eps, lin.bias,
                                              getattr(lin, "scales", None), self.workspace)
        return self._reduce_add_norm(self._linear(inp, lin), x, norm, group)

    def _reduce_add_norm(self, partial, x, norm, group):
        """all-reduce of a sub-layer's partial output (C1), residual add, RMSNorm for the next sub-layer:
        (h, y) = (x + sum_ranks(partial), rmsnorm(h) * w).  With the xGMI all-reduce attached (tp.apply_tp,
        MAGICDEC_ONESHOT_AR=1) and a decode-sized message this is ONE launch (md_allreduce_add_rmsnorm); otherwise
        the collective (RCCL) followed by the fused add + norm kernel."""
        if group is not None:
            ar = getattr(self, "_oneshot", None)
            if ar is not None and ar.fits_fused(partial, norm.weight) and x.is_contiguous():
                return ar.all_reduce_add_rmsnorm(partial, x, norm.weight, norm.eps)
            partial = self._reduce(partial, group)
        return ops.add_rmsnorm(x, partial, norm.weight, norm.eps)

    def _qkv(self, layer, y2d):
        c = self.config
        att = layer.attention
        qkv = self._linear(y2d, att.wqkv)                              # [rows, (H+2KH)*D]
        H, KH, D = c.n_head, c.n_local_heads, c.head_dim
        rows = qkv.shape[0]
        q = qkv[:, :H * D].unflatten(1, (H, D))
        k = qkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
        v = qkv[:, (H + KH) * D:].unflatten(1, (KH, D))
        return q, k, v, rows

    def _run(self, idx, attn_fn):
        """embed -> L x (norm, attention, +res, norm, mlp, +res) -> norm -> head -> argmax.
        attn_fn(i, layer, y, n) -> attention output [rows, H, D]: the qkv projection, RoPE, the KV append and the
        attention of layer i over the normalised hidden states y [rows, dim].
        With self.skip_head set (non-final prefill chunks, whose tokens the reference computes and discards,
        Engine/SnapKV/backend.py:239-263) the lm head is skipped and None is returned."""
        assert self._ready, "call setup_caches first"
        B, n = idx.shape
        x = self.tok_embeddings(idx).view(B * n, -1)
        layers = self.layers
        y = ops.rmsnorm(x, layers[0].attention_norm.weight, layers[0].attention_norm.eps)
        for i, layer in enumerate(layers):
            o = attn_fn(i, layer, y, n)
            x, y = self._proj_add_norm(o.view(o.shape[0], -1), layer.attention.wo, x, layer.ffn_norm,
                                       layer.attention.process_group)
            act = self._linear(y, None, swiglu_w13=(self._w13[i], self._s13[i]))
            nxt = layers[i + 1].attention_norm if i + 1 < len(layers) else self.norm
            x, y = self._proj_add_norm(act, layer.feed_forward.w2, x, nxt, layer.feed_forward.process_group)
        if self.skip_head:
            return None
        logits = self._linear(y, self.output)                         # [rows, vocab / tp]; materialises a deferred norm
        self._last_logits = logits
        return self._argmax(logits).view(B, n)

    def _argmax(self, logits):
        """argmax, or the TP merge of per-rank maxima (Engine/SnapKV/model.py:178-188): two sum-all-reduces of
        one-hot-slot tensors, then the lowest rank among equal maxima."""
        if self.process_group is None:
            return ops.argmax(logits)
        # the one-hot-slot tensors of the reference (zeros + this rank's column), written by the argmax launch itself
        all_v, all_i = ops.argmax_tp_slots(logits, self.rank, self.world_size, index_offset=self.rank * logits.shape[1])
        dist.all_reduce(all_v, group=self.process_group)
        dist.all_reduce(all_i, group=self.process_group)
        return ops.tp_argmax_merge(all_v, all_i)

    def _attend(self, q_rot, cache, qo_indptr, tab: PageTable, n, kv_scales=None, kv_layout="NHD"):
        return ops.paged_attention(q_rot, cache, qo_indptr, tab.indices, tab.indptr, tab.last_page_len, n,
                                   tab.max_pages, self.workspace, causal=True, kv_scales=kv_scales,
                                   kv_layout=kv_layout)

    # ------------------------------------------------------------------ step variants
    def _std_step(self, idx, offsets, qo_indptr, tab: PageTable, which="kv_cache", tab2: PageTable = None,
                  snap_tab: PageTable = None, calibrate=False):
        """rope(offsets) -> append (-> 2nd cache) -> attention (-> SnapKV select): Attention.forward / verify /
        draft_forward / prefill of Engine/SnapKV/model.py:322-387."""
        c = self.config

        def fn(i, layer, y, n):
            kvc = layer.attention.kv_cache
            att = layer.attention
            cache = getattr(kvc, which)
            cache2 = kvc.draft_cache if tab2 is not None else None
            scales = kvc.scales(which)
            layout = kvc.layout_of(which)
            need_calib = scales is not None and calibrate and not kvc.calibrated
            pro = y if isinstance(y, ops.DeferredNorm) else None
            yin = pro.h if pro is not None else y
            # the fused kernel derives (request, row-in-request) from the row index: every request must own exactly n
            # consecutive rows (what every back-end passes: qo_indptr = arange * n); anything else takes rope_append, which
            # honours qo_indptr
            uniform_rows = yin.shape[0] == (qo_indptr.numel() - 1) * n
            if (not need_calib and uniform_rows and self._fused_here(yin, att.wqkv.weight, "qkv", pro is not None)
                    and c.head_dim in (64, 128)):
                # (deferred RMSNorm +) wqkv + bias + RoPE + paged append (both caches of a self-speculation verify):
                # ONE launch
                q_rot = ops.fused_qkv_rope_append(
                    yin, self._packed[id(att.wqkv.weight)], att.wqkv.bias, c.n_head, c.n_local_heads, c.head_dim, n,
                    offsets, self.rope_table, cache, tab.indices, tab.indptr, tab.last_page_len, cache2,
                    tab2.indices if tab2 else None, tab2.indptr if tab2 else None,
                    tab2.last_page_len if tab2 else None, kv_scales=scales, kv_layout=layout, pro=pro)
            else:
                if pro is not None:
                    y = pro.materialize()
                q, k, v, _ = self._qkv(layer, y)
                if need_calib:
                    if self.kv_scale_override is not None:
                        kvc.k_scale.copy_(self.kv_scale_override[i][0])
                        kvc.v_scale.copy_(self.kv_scale_override[i][1])
                        kvc.calibrated = True
                    else:
                        kvc.calibrate(k, v)
                q_rot = ops.rope_append(q, k, v, qo_indptr, offsets, self.rope_table, cache, tab.indices, tab.indptr,
                                        tab.last_page_len, cache2, tab2.indices if tab2 else None,
                                        tab2.indptr if tab2 else None, tab2.last_page_len if tab2 else None, n_max=n,
                                        kv_scales=scales, kv_layout=layout)
            o = self._attend(q_rot, cache, qo_indptr, tab, n, scales, layout)
            if snap_tab is not None:
                ops.snapkv_select(q_rot, cache, tab.indices, tab.indptr, self._snap_ctx_len, self.window_size,
                                  self.draft_budget, POOL_KERNEL, kvc.draft_cache, snap_tab.indices, snap_tab.indptr,
                                  snap_tab.last_page_len, self.workspace, kv_scales=scales, kv_layout=layout)
            return o
        return self._run(idx, fn)

    def forward(self, idx, input_pos, kv_append_indptr, tab: PageTable):
        return self._std_step(idx, input_pos, kv_append_indptr, tab)

    def verify(self, idx, input_pos, kv_append_indptr, tab: PageTable, draft_tab: PageTable = None):
        """Self-spec SnapKV verify also appends the gamma+1 rows to the draft cache (model.py:338-353)."""
        return self._std_step(idx, input_pos, kv_append_indptr, tab, tab2=draft_tab)

    def draft_forward(self, idx, input_pos, kv_append_indptr, draft_tab: PageTable):
        return self._std_step(idx, input_pos, kv_append_indptr, draft_tab, which="draft_cache")

    def prefill(self, idx, input_pos, kv_append_indptr, tab: PageTable, is_last=False, draft_tab: PageTable = None,
                ctx_len=None):
        """Chunked prefill; on the last chunk of a SnapKV engine also runs the select (model.py:371-387).
        ctx_len = offsets[0]+seqlen is passed by the back-end (host-known, no device read)."""
        snap = draft_tab if (is_last and self.spec and not self.streaming and draft_tab is not None) else None
        if snap is not None and idx.shape[1] != self.window_size:
            # the reference takes the whole last chunk as the observation window (Engine/SnapKV/model.py:389-395),
            # which is `window_size` rows only under its CLI assert (prefix_len - window_size) % 128 == 0
            raise ValueError(f"SnapKV select needs the last prefill chunk to be exactly window_size="
                             f"{self.window_size} tokens, got {idx.shape[1]} ((prefix_len - window_size) % 128 != 0)")
        self._snap_ctx_len = ctx_len
        return self._std_step(idx, input_pos, kv_append_indptr, tab, snap_tab=snap, calibrate=True)

    def stream_prefill(self, idx, ctx, kv_append_indptr, tab: PageTable, is_last, which, B):
        """StreamingLLM draft prefill of one chunk (Attention.prefill + KVCache.prefill,
        Engine/StreamingLLM/model_draft.py:102-143,309-326; draft_prefill / prefill_draft of StreamingLLM/model.py).
        ctx = tokens already in the (capped) cache, host int."""
        kv_len = self.draft_budget
        dev = idx.device

        def fn(i, layer, y, n):
            if isinstance(y, ops.DeferredNorm):
                y = y.materialize()
            q, k, v, _ = self._qkv(layer, y)
            cache = getattr(layer.attention.kv_cache, which)
            ppr = cache.shape[0] // B
            overflow = ctx + n > kv_len
            off = torch.full((B,), (kv_len - n) if overflow else ctx, dtype=torch.int32, device=dev)
            q_rot, _ = ops.rope(q, None, kv_append_indptr, off, self.rope_table, n_max=n)
            if not overflow:
                ops.update_kv(k, v, kv_append_indptr, cache, tab.indices, tab.indptr, tab.last_page_len, n_max=n)
                valid = ctx + n
            else:
                ops.streaming_shift_append(k, v, cache, n, kv_len, SINK, ppr)
                valid = kv_len
            if overflow and is_last:
                rot = cache                           # persistent cache becomes rotated (model_draft.py:141-142)
            else:
                if self._rot_scratch is None or self._rot_scratch.shape != cache.shape:
                    self._rot_scratch = torch.empty_like(cache)
                rot = self._rot_scratch
            ops.streaming_rotate(cache, rot, B, valid, ppr, self.rope_table)
            return self._attend(q_rot, rot, kv_append_indptr, tab, n)
        return self._run(idx, fn)
