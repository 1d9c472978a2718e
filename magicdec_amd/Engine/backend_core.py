"""Host side of the four MagicDec back-ends (L4 of SURVEY.md section 1), MI355X-native.

Public surface = the reference's: classes `LMBackend` / `LMBackend_Draft` with
`load_model, setup_caches, compile, encode, inference, verify, speculate,
draft_encode, clear_kv` and the tensors the harness reads AND REBINDS between
calls (`cachelens, paged_kv_last_page_len, draft_cachelens,
draft_paged_kv_last_page_len`, tests/SnapKV/longspec_benchmark.py:228-256) --
every method therefore re-reads those attributes at call time.

What is different underneath:
  * no flashinfer wrappers and no plan(): the page table goes to the kernels as
    device pointers; nothing in a step reads device memory on the host;
  * request b owns pages [b*P, (b+1)*P) (as in the reference, backend.py:273) so the
    page-index vector is built once per chunk by one vectorised op, not B aranges;
  * the lm head runs only on the last prefill chunk (the reference computes and
    discards 128 x vocab logits per chunk);
  * `compile()` captures the decode step into a hipGraph (there is no Inductor).
"""
from __future__ import annotations

import torch

from .model_core import PageTable

PAGE_SIZE = 128    # Engine/SnapKV/backend.py:31
CHUNK = 128        # Engine/SnapKV/backend.py:236


def default_kv_layout():
    """Page layout of the full-context cache when setup_caches() is not told one: "HND" (rows of one kv head contiguous
    inside a page: the verify step's stream is read in 128-row runs, 85 % instead of 82 % of the HBM peak in bf16, 81 %
    instead of 71 % in fp8 -- DESIGN.md section 3.1) unless MAGICDEC_KV_LAYOUT=NHD asks for the reference's flashinfer
    layout (Engine/SnapKV/backend.py:30).  The layout is internal to the Engine API (the harness never touches the
    cache tensors) and moves bytes, not arithmetic: every kernel's output is bit-identical between the two
    (tests/test_gpu_hnd.py), and the whole GPU suite is green under either default (profiles/r03_hnd_suite_summary.txt).
    The mylib::* operator boundary (magicdec_amd/mylib_ops.py) keeps the reference's NHD signature default."""
    import os
    v = os.environ.get("MAGICDEC_KV_LAYOUT", "HND").upper()
    if v not in ("NHD", "HND"):
        raise ValueError(f"MAGICDEC_KV_LAYOUT must be NHD or HND, got {v!r}")
    return v


def _pages_per_request(max_batch_size, max_seq_length, page_size=PAGE_SIZE):
    """Engine/SnapKV/backend.py:32-35."""
    n = max_batch_size * max_seq_length // page_size
    if n * page_size < max_batch_size * max_seq_length:
        n += max_batch_size
    return n, n // max_batch_size


class _PagedState:
    """One paged cache's bookkeeping under a name prefix ('' or 'draft_') on the owning back-end."""

    def __init__(self, owner, prefix, ppr):
        self.o, self.p, self.ppr = owner, prefix, ppr

    def _get(self, name):
        return getattr(self.o, self.p + name)

    def _set(self, name, val):
        setattr(self.o, self.p + name, val)

    def reset(self, last_page_len_init=0, full_table=False, indptr_stride=1):
        o, B, dev = self.o, self.o.batch_size, self.o.device
        ppr = self.ppr
        self._set("num_pages_per_request", torch.zeros(B, dtype=torch.int32, device=dev))
        self.host_pages = 0
        if full_table:   # SnapKV draft cache: all pages mapped from the start (backend.py:87-90)
            self._set("paged_kv_indptr", torch.arange(B + 1, dtype=torch.int32, device=dev) * ppr)
            self._set("paged_kv_indices", torch.arange(B * ppr, dtype=torch.int32, device=dev))
            self.host_pages = ppr
        else:
            self._set("paged_kv_indptr", torch.arange(B + 1, dtype=torch.int32, device=dev) * indptr_stride)
            self._set("paged_kv_indices", torch.zeros(B * ppr, dtype=torch.int32, device=dev))
        self._set("paged_kv_last_page_len", torch.full((B,), last_page_len_init, dtype=torch.int32, device=dev))

    def map_pages(self, n_pages, last_page_len):
        """Request b maps pages b*ppr .. b*ppr+n_pages-1; last page holds `last_page_len` rows
        (pre_encode, Engine/SnapKV/backend.py:270-275)."""
        o, B, dev = self.o, self.o.batch_size, self.o.device
        self.host_pages = n_pages
        self._get("num_pages_per_request").fill_(n_pages)
        base = torch.arange(B, dtype=torch.int32, device=dev).view(-1, 1) * self.ppr
        self._set("paged_kv_indices", (base + torch.arange(n_pages, dtype=torch.int32, device=dev).view(1, -1)).flatten())
        self._get("paged_kv_indptr").copy_(torch.arange(B + 1, dtype=torch.int32, device=dev) * n_pages)
        self._set("paged_kv_last_page_len", torch.full((B,), last_page_len, dtype=torch.int32, device=dev))

    def table(self):
        return PageTable(self._get("paged_kv_indices"), self._get("paged_kv_indptr"),
                         self._get("paged_kv_last_page_len"), max(self.host_pages, 1))


class _BackendBase:
    def __init__(self, dtype=torch.bfloat16, device="cuda:0"):
        self.dtype, self.device = dtype, device
        self.model = None
        self.cachelens = None
        self._graphs = {}
        self._use_graphs = False

    # -- to be provided by the wrapper modules (which loader / which Transformer class)
    _loader = None

    def load_model(self, checkpoints, use_tp: bool, rank_group=None, group=None):
        from .utils import enable_tuned_gemms
        enable_tuned_gemms()
        self.model = type(self)._loader(checkpoint_path=checkpoints, device=self.device, precision=self.dtype,
                                        use_tp=use_tp, rank_group=rank_group, group=group)

    def compile(self):
        """The reference compiles with Inductor/Triton + CUDA graphs (Engine/SnapKV/backend.py:116-125); here the
        decode steps are captured into hipGraphs on first use (see Engine/graph.py)."""
        self._use_graphs = True

    def _qo(self, n):
        return self.qo_indptr * n

    def _decode_rows(self, *per_request, two_token_step=True):
        """Row counts of this engine's decode / verify steps: batch x (1, dec_len ..., and 2 -- the two-token step after an
        all-accept iteration -- for the engines that DRAFT with it: the stand-alone draft models and the StreamingLLM
        self-speculation engine; a longspec target and the SnapKV self-speculation engine never see it); the model packs /
        releases weight copies for exactly these (Transformer.setup_caches(decode_rows=...))."""
        B = self.batch_size
        return sorted({B * int(n) for n in (1,) + ((2,) if two_token_step else ()) + tuple(per_request) if n})

    def _prefill_begin(self):
        """Prefill-sized products run on the library GEMM over the row-major weights: re-materialise the ones released
        after the previous prefill (Transformer.restore_rowmajor)."""
        if hasattr(self.model, "restore_rowmajor"):
            self.model.restore_rowmajor()

    def _prefill_end(self):
        """Decode reads most weights in the streaming layout only: free their row-major tensors (one resident copy)."""
        if hasattr(self.model, "release_rowmajor"):
            self.model.release_rowmajor()

    def _reset_kv_calibration(self):
        """fp8 caches re-calibrate their static scales on the first prefill chunk of every encode()."""
        for b in self.model.layers:
            b.attention.kv_cache.calibrated = False

    def _run_step(self, key, fn, *tensors):
        if self._use_graphs:
            from .graph import run_captured
            return run_captured(self, key, fn, *tensors)
        return fn(*tensors)


# ======================================================================================= target (+ SnapKV self-spec)
class SnapKVTargetBackend(_BackendBase):
    """Engine/SnapKV/backend.py LMBackend: the paged-KV target engine of every longspec / baseline run and,
    with draft_dec_len set, the SnapKV self-speculation engine (two caches, one set of weights)."""

    def __init__(self, dtype=torch.bfloat16, device: str = "cuda:0", dec_len: int = 1, draft_dec_len: int = None):
        super().__init__(dtype, device)
        self.dec_len = dec_len
        self.is_spec = draft_dec_len is not None
        self.draft_dec_len = draft_dec_len
        self.draft_cachelens = None

    @torch.no_grad()
    def setup_caches(self, max_batch_size: int = 1, max_seq_length: int = 2048, draft_budget=0, window_size=32,
                     kv_dtype="bf16", kv_layout=None):
        """kv_dtype="fp8": the full-context cache is OCP e4m3fn with static per-head scales calibrated on the
        first prefill chunk (not in the reference; BASELINE.json configs[4]).  The compressed draft cache stays bf16.
        kv_layout="HND": the full-context cache keeps the rows of one kv head contiguous inside a page (the
        reference's flashinfer wrappers are planned "NHD": Engine/SnapKV/backend.py:30); same results, longer
        contiguous runs for the verify step's stream (matters for fp8 rows of 128 bytes).  None = default_kv_layout()."""
        kv_layout = default_kv_layout() if kv_layout is None else kv_layout
        self.max_length, self.batch_size = max_seq_length, max_batch_size
        dev = self.device
        self.page_size = PAGE_SIZE
        self.max_num_pages, self.max_num_pages_per_request = _pages_per_request(max_batch_size, max_seq_length)
        self.cachelens = torch.zeros(max_batch_size, dtype=torch.int32, device=dev)
        self.qo_indptr = torch.arange(max_batch_size + 1, dtype=torch.int32, device=dev)
        self._t = _PagedState(self, "", self.max_num_pages_per_request)
        self._t.reset()
        if self.is_spec:
            self.draft_budget, self.window_size = draft_budget, window_size
            self.draft_pages_per_request = draft_budget // PAGE_SIZE + 1
            self.draft_num_pages = self.draft_pages_per_request * max_batch_size
            self.draft_cachelens = torch.zeros(max_batch_size, dtype=torch.int32, device=dev)
            self._d = _PagedState(self, "draft_", self.draft_pages_per_request)
            self._d.reset(last_page_len_init=1, full_table=True)
            self.model.setup_caches(num_pages=self.max_num_pages, page_size=PAGE_SIZE, spec=True,
                                    draft_num_pages=self.draft_num_pages, draft_budget=draft_budget,
                                    window_size=window_size, max_positions=max_seq_length + 256, kv_dtype=kv_dtype,
                                    kv_layout=kv_layout,
                                    decode_rows=self._decode_rows(self.dec_len, getattr(self, "draft_dec_len", None),
                                                                 two_token_step=False))
        else:
            self.model.setup_caches(num_pages=self.max_num_pages, page_size=PAGE_SIZE,
                                    max_positions=max_seq_length + 256, kv_dtype=kv_dtype, kv_layout=kv_layout,
                                    decode_rows=self._decode_rows(self.dec_len, two_token_step=False))

    @torch.no_grad()
    def clear_kv(self):
        for b in self.model.layers:
            b.attention.kv_cache.kv_cache.zero_()
            if self.is_spec:
                b.attention.kv_cache.draft_cache.zero_()
        self._reset_kv_calibration()
        self.cachelens.zero_()
        self.qo_indptr = torch.arange(self.batch_size + 1, dtype=torch.int32, device=self.device)
        self._t.reset()
        if self.is_spec:
            self.draft_cachelens.zero_()
            self._d.reset(last_page_len_init=1, full_table=True)

    @torch.no_grad()
    def encode(self, input_ids: torch.LongTensor, benchmark=False):
        """Chunked prefill (backend.py:232-268)."""
        self.clear_kv()
        self._prefill_begin()
        seq_len = input_ids.shape[1]
        tokens = None
        num_chunks = (seq_len + CHUNK - 1) // CHUNK
        is_last = False
        done = 0
        for i in range(num_chunks):
            ids = input_ids[:, i * CHUNK:min((i + 1) * CHUNK, seq_len)]
            n = ids.shape[1]
            if n != CHUNK:
                is_last = True
            self._t.map_pages(i + 1, n)                       # pre_encode
            self.model.skip_head = i != num_chunks - 1
            tokens = self.model.prefill(ids, self.cachelens, self._qo(n), self._t.table(), is_last=is_last,
                                        draft_tab=self._d.table() if (self.is_spec and is_last) else None,
                                        ctx_len=done + n)
            self.cachelens += n
            done += n
        self.model.skip_head = False
        if self.is_spec:
            if not is_last:
                raise ValueError("SnapKV self-speculation: prefix_len is a multiple of 128, so no last chunk ran the "
                                 "select and the draft cache is empty (need (prefix_len - window_size) % 128 == 0)")
            self.draft_cachelens.copy_(self.cachelens)
        self._prefill_end()
        return tokens

    @torch.no_grad()
    def inference(self, input_ids: torch.LongTensor, benchmark=False):
        """Autoregressive step / longspec verification (backend.py:129-159)."""
        n = input_ids.shape[1]
        self.paged_kv_last_page_len += n                      # pre_decode
        out = self._run_step(("fwd", n), lambda ids: self.model.forward(ids, self.cachelens, self._qo(n),
                                                                          self._t.table()), input_ids)
        self.cachelens += n
        if benchmark:
            self.cachelens -= n
            self.paged_kv_last_page_len -= n
        return out

    @torch.no_grad()
    def verify(self, input_ids: torch.LongTensor, benchmark=False):
        """Self-spec verification: also appends the gamma+1 rows to the draft cache (backend.py:163-197)."""
        n = input_ids.shape[1]
        self.paged_kv_last_page_len += n                      # pre_verify
        self.draft_paged_kv_last_page_len += 1
        self.draft_cachelens += 1
        out = self._run_step(("verify", n), lambda ids: self.model.verify(ids, self.cachelens, self._qo(n),
                                                                            self._t.table(), self._d.table()),
                             input_ids)
        self.cachelens += n
        if benchmark:
            self.cachelens -= n
            self.paged_kv_last_page_len -= n
        return out

    @torch.no_grad()
    def speculate(self, input_ids: torch.LongTensor, benchmark=False):
        """Self-spec draft step over the SnapKV draft cache (backend.py:200-229)."""
        n = input_ids.shape[1]
        self.draft_paged_kv_last_page_len += n                # pre_spec
        out = self._run_step(("spec", n), lambda ids: self.model.draft_forward(ids, self.draft_cachelens, self._qo(n),
                                                                                 self._d.table()), input_ids)
        self.draft_cachelens += n
        if benchmark:
            self.draft_cachelens -= n
            self.draft_paged_kv_last_page_len -= n
        return out


# ======================================================================================= SnapKV stand-alone draft
class SnapKVDraftBackend(_BackendBase):
    """Engine/SnapKV/backend_draft.py LMBackend_Draft: small draft model, full-KV prefill, SnapKV select on the
    last chunk into `draft_cache`, decode over the draft cache only (draft_budget == -1: full-KV draft)."""

    def __init__(self, dtype=torch.bfloat16, device: str = "cuda:0", dec_len: list = [1], draft_budget: int = None):
        super().__init__(dtype, device)
        self.dec_len = dec_len
        self.is_compress = draft_budget != -1

    @torch.no_grad()
    def setup_caches(self, max_batch_size: int = 1, max_seq_length: int = 2048, draft_budget=0, window_size=32):
        self.batch_size = max_batch_size
        dev = self.device
        self.page_size = PAGE_SIZE
        self.max_num_pages, self.max_num_pages_per_request = _pages_per_request(max_batch_size, max_seq_length)
        self.cachelens = torch.zeros(max_batch_size, dtype=torch.int32, device=dev)
        self.qo_indptr = torch.arange(max_batch_size + 1, dtype=torch.int32, device=dev)
        self._t = _PagedState(self, "", self.max_num_pages_per_request)
        self._t.reset()
        if self.is_compress:
            self.draft_budget, self.window_size = draft_budget, window_size
            self.draft_pages_per_request = draft_budget // PAGE_SIZE + 1
            self.draft_num_pages = self.draft_pages_per_request * max_batch_size
            self._d = _PagedState(self, "draft_", self.draft_pages_per_request)
            self._d.reset(last_page_len_init=1, full_table=True)
            self.model.setup_caches(num_pages=self.max_num_pages, page_size=PAGE_SIZE, spec=True,
                                    draft_num_pages=self.draft_num_pages, draft_budget=draft_budget,
                                    window_size=window_size, max_positions=max_seq_length + 256,
                                    decode_rows=self._decode_rows())
        else:
            self.model.setup_caches(num_pages=self.max_num_pages, page_size=PAGE_SIZE,
                                    max_positions=max_seq_length + 256, decode_rows=self._decode_rows())

    @torch.no_grad()
    def clear_kv(self):
        for b in self.model.layers:
            b.attention.kv_cache.kv_cache.zero_()
            if self.is_compress:
                b.attention.kv_cache.draft_cache.zero_()
        self._reset_kv_calibration()
        self.cachelens.zero_()
        self.qo_indptr = torch.arange(self.batch_size + 1, dtype=torch.int32, device=self.device)
        self._t.reset()
        if self.is_compress:
            self._d.reset(last_page_len_init=1, full_table=True)

    @torch.no_grad()
    def encode(self, input_ids: torch.LongTensor, benchmark=False):
        """backend_draft.py:176-209."""
        self.clear_kv()
        self._prefill_begin()
        seq_len = input_ids.shape[1]
        tokens = None
        num_chunks = (seq_len + CHUNK - 1) // CHUNK
        is_last = False
        done = 0
        for i in range(num_chunks):
            ids = input_ids[:, i * CHUNK:min((i + 1) * CHUNK, seq_len)]
            n = ids.shape[1]
            if n != CHUNK:
                is_last = True
            self._t.map_pages(i + 1, n)
            self.model.skip_head = i != num_chunks - 1
            tokens = self.model.prefill(ids, self.cachelens, self._qo(n), self._t.table(), is_last=is_last,
                                        draft_tab=self._d.table() if (self.is_compress and is_last) else None,
                                        ctx_len=done + n)
            self.cachelens += n
            done += n
        self.model.skip_head = False
        if self.is_compress and not is_last:
            raise ValueError("SnapKV draft: prefix_len is a multiple of 128, so no last chunk ran the select and the "
                             "draft cache is empty (need (prefix_len - window_size) % 128 == 0)")
        self._prefill_end()
        return tokens

    @torch.no_grad()
    def inference(self, input_ids: torch.LongTensor, benchmark=False, cachelen_update=None):
        """One (or, after an all-accept iteration, a two-token) draft step (backend_draft.py:113-173)."""
        n = input_ids.shape[1]
        if self.is_compress:
            self.draft_paged_kv_last_page_len += n            # pre_decode
            out = self._run_step(("draft", n), lambda ids: self.model.draft_forward(ids, self.cachelens, self._qo(n),
                                                                                      self._d.table()), input_ids)
        else:
            self.paged_kv_last_page_len += n
            out = self._run_step(("fwd", n), lambda ids: self.model.forward(ids, self.cachelens, self._qo(n),
                                                                              self._t.table()), input_ids)
        if cachelen_update is None:
            self.cachelens += n
        else:
            cu = cachelen_update.to(torch.int32).flatten()
            self.cachelens += cu
            if self.is_compress:
                self.draft_paged_kv_last_page_len = self.draft_paged_kv_last_page_len - n + cu
            else:
                self.paged_kv_last_page_len = self.paged_kv_last_page_len - n + cu
        if benchmark:   # as the reference: always the un-prefixed table (backend_draft.py:139-142)
            self.cachelens -= n
            self.paged_kv_last_page_len -= n
        return out


# ======================================================================================= StreamingLLM drafts
class _StreamingMixin:
    """Sink(16)+window prefill shared by the stand-alone StreamingLLM draft and the self-spec engine."""

    def _stream_encode(self, input_ids, state: _PagedState, lens_attr, which):
        self._prefill_begin()
        seq_len = input_ids.shape[1]
        tokens = None
        num_chunks = (seq_len + CHUNK - 1) // CHUNK
        is_last = False
        ctx = 0                                       # host mirror of the (capped) cache length
        for i in range(num_chunks):
            ids = input_ids[:, i * CHUNK:min((i + 1) * CHUNK, seq_len)]
            n = ids.shape[1]
            if n != CHUNK:
                is_last = True
            if ctx + n <= self.draft_budget:          # pre_encode (StreamingLLM/backend_draft.py:155-192)
                state.map_pages(state.host_pages + 1, n)
            else:
                state.map_pages(state.ppr, self.draft_budget % PAGE_SIZE)
            self.model.skip_head = i != num_chunks - 1
            tokens = self.model.stream_prefill(ids, ctx, self._qo(n), state.table(), is_last, which, self.batch_size)
            ctx += n
            lens = getattr(self, lens_attr)
            lens += n
            if ctx >= self.draft_budget:
                ctx = self.draft_budget
                lens.fill_(self.draft_budget)
        self.model.skip_head = False
        self._prefill_end()
        return tokens


class StreamingDraftBackend(_BackendBase, _StreamingMixin):
    """Engine/StreamingLLM/backend_draft.py LMBackend_Draft: the draft's only cache holds 16 sink tokens plus the
    most recent budget-16; RoPE positions are cache slots."""

    def __init__(self, dtype=torch.bfloat16, device: str = "cuda:0"):
        super().__init__(dtype, device)

    @torch.no_grad()
    def setup_caches(self, max_batch_size: int = 1, draft_budget=0):
        self.draft_budget, self.batch_size = draft_budget, max_batch_size
        dev = self.device
        self.page_size = PAGE_SIZE
        self.max_num_pages_per_request = draft_budget // PAGE_SIZE + 1
        self.max_num_pages = self.max_num_pages_per_request * max_batch_size
        self.cachelens = torch.zeros(max_batch_size, dtype=torch.int32, device=dev)
        self.qo_indptr = torch.arange(max_batch_size + 1, dtype=torch.int32, device=dev)
        self._t = _PagedState(self, "", self.max_num_pages_per_request)
        self._t.reset()
        self.model.setup_caches(num_pages=self.max_num_pages, page_size=PAGE_SIZE, draft_budget=draft_budget,
                                streaming=True, max_positions=self.max_num_pages_per_request * PAGE_SIZE + 256,
                                decode_rows=self._decode_rows())

    @torch.no_grad()
    def clear_kv(self):
        for b in self.model.layers:
            b.attention.kv_cache.kv_cache.zero_()
        self._reset_kv_calibration()
        self.cachelens.zero_()
        self.qo_indptr = torch.arange(self.batch_size + 1, dtype=torch.int32, device=self.device)
        self._t.reset()

    @torch.no_grad()
    def encode(self, input_ids: torch.LongTensor, benchmark=False):
        self.clear_kv()
        return self._stream_encode(input_ids, self._t, "cachelens", "kv_cache")

    @torch.no_grad()
    def inference(self, input_ids: torch.LongTensor, benchmark=False, cachelen_update=None):
        """StreamingLLM/backend_draft.py:89-124."""
        n = input_ids.shape[1]
        self.paged_kv_last_page_len += n
        out = self._run_step(("fwd", n), lambda ids: self.model.forward(ids, self.cachelens, self._qo(n),
                                                                          self._t.table()), input_ids)
        if cachelen_update is None:
            self.cachelens += n
        else:
            cu = cachelen_update.to(torch.int32).flatten()
            self.cachelens += cu
            self.paged_kv_last_page_len = self.paged_kv_last_page_len - n + cu
        if benchmark:
            self.cachelens -= n
            self.paged_kv_last_page_len -= n
        return out


class StreamingSelfSpecBackend(_BackendBase, _StreamingMixin):
    """Engine/StreamingLLM/backend.py LMBackend: one set of weights, the full target cache and a streaming
    draft cache (config 2 of BASELINE.json)."""

    def __init__(self, dtype=torch.bfloat16, device: str = "cuda:0", dec_len: int = 1):
        super().__init__(dtype, device)
        self.dec_len = dec_len
        self.draft_cachelens = None

    @torch.no_grad()
    def setup_caches(self, max_batch_size: int = 1, max_seq_length: int = 2048, draft_budget=0, kv_dtype="bf16",
                     kv_layout=None):
        kv_layout = default_kv_layout() if kv_layout is None else kv_layout
        self.draft_budget, self.batch_size = draft_budget, max_batch_size
        dev = self.device
        self.page_size = PAGE_SIZE
        self.max_num_pages, self.max_num_pages_per_request = _pages_per_request(max_batch_size, max_seq_length)
        self.cachelens = torch.zeros(max_batch_size, dtype=torch.int32, device=dev)
        self.draft_cachelens = torch.zeros(max_batch_size, dtype=torch.int32, device=dev)
        self.qo_indptr = torch.arange(max_batch_size + 1, dtype=torch.int32, device=dev)
        self.draft_max_num_pages_per_request = draft_budget // PAGE_SIZE + 1
        self.draft_max_num_pages = self.draft_max_num_pages_per_request * max_batch_size
        self._t = _PagedState(self, "", self.max_num_pages_per_request)
        self._d = _PagedState(self, "draft_", self.draft_max_num_pages_per_request)
        self._t.reset()
        self._d.reset(indptr_stride=self.draft_max_num_pages_per_request)
        self.model.setup_caches(num_pages=self.max_num_pages, page_size=PAGE_SIZE, spec=True,
                                draft_num_pages=self.draft_max_num_pages, draft_budget=draft_budget, streaming=True,
                                max_positions=max_seq_length + 256, kv_dtype=kv_dtype, kv_layout=kv_layout,
                                decode_rows=self._decode_rows(self.dec_len))

    @torch.no_grad()
    def clear_kv(self):
        for b in self.model.layers:
            b.attention.kv_cache.kv_cache.zero_()
            b.attention.kv_cache.draft_cache.zero_()
        self._reset_kv_calibration()
        self.cachelens.zero_()
        self.draft_cachelens.zero_()
        self.qo_indptr = torch.arange(self.batch_size + 1, dtype=torch.int32, device=self.device)
        self._t.reset()
        self._d.reset(indptr_stride=self.draft_max_num_pages_per_request)   # StreamingLLM/backend.py:316

    @torch.no_grad()
    def encode(self, input_ids: torch.LongTensor, benchmark=False):
        """Target prefill (StreamingLLM/backend.py:190-211)."""
        self.clear_kv()
        self._prefill_begin()
        seq_len = input_ids.shape[1]
        tokens = None
        num_chunks = (seq_len + CHUNK - 1) // CHUNK
        for i in range(num_chunks):
            ids = input_ids[:, i * CHUNK:min((i + 1) * CHUNK, seq_len)]
            n = ids.shape[1]
            self._t.map_pages(i + 1, n)
            self.model.skip_head = i != num_chunks - 1
            tokens = self.model.prefill(ids, self.cachelens, self._qo(n), self._t.table())
            self.cachelens += n
        self.model.skip_head = False
        self._prefill_end()
        return tokens

    @torch.no_grad()
    def draft_encode(self, input_ids: torch.LongTensor, benchmark=False):
        """Second pass filling the streaming draft cache (StreamingLLM/backend.py:234-258)."""
        return self._stream_encode(input_ids, self._d, "draft_cachelens", "draft_cache")

    @torch.no_grad()
    def verify(self, input_ids: torch.LongTensor, benchmark=False):
        n = input_ids.shape[1]
        self.paged_kv_last_page_len += n
        out = self._run_step(("verify", n), lambda ids: self.model.verify(ids, self.cachelens, self._qo(n),
                                                                            self._t.table()), input_ids)
        self.cachelens += n
        if benchmark:
            self.cachelens -= n
            self.paged_kv_last_page_len -= n
        return out

    @torch.no_grad()
    def speculate(self, input_ids: torch.LongTensor, benchmark=False, cachelen_update=None):
        n = input_ids.shape[1]
        self.draft_paged_kv_last_page_len += n
        out = self._run_step(("spec", n), lambda ids: self.model.draft_forward(ids, self.draft_cachelens,
                                                                                 self._qo(n), self._d.table()),
                             input_ids)
        if cachelen_update is None:
            self.draft_cachelens += n
        else:
            cu = cachelen_update.to(torch.int32).flatten()
            self.draft_cachelens += cu
            self.draft_paged_kv_last_page_len = self.draft_paged_kv_last_page_len - n + cu
        if benchmark:
            self.draft_cachelens -= n
            self.draft_paged_kv_last_page_len -= n
        return out
