"""The executable inner boundary: magicdec_amd.mylib_ops registers the reference's seven mylib::* operators on the C ABI.

CPU: the registered schemas are the ones the REAL reference registers (fixture tests/golden/mylib_schemas.json, read
back from the dispatcher after constructing the reference's back-ends, oracle/gen_golden.py:scen_mylib_schemas), the
fake kernels give the right shapes, CPU tensors are rejected.
GPU (-m gpu): every op called as the reference calls it (torch.ops.mylib.*, plan() on a wrapper object, then
run(q, kv_cache)) against the oracle: append / RoPE bit-exact, attention within the measured forward bound.
"""
import pytest
import torch

from oracle import flashinfer_ref as fr
from tests import golden_cfg as gc

BF = torch.bfloat16


def test_registered_schemas_equal_the_reference():
    from magicdec_amd import mylib_ops
    mylib_ops.register()
    mylib_ops.register()          # idempotent (the reference raises on a second setup_caches)
    want = gc.load_json("mylib_schemas.json")
    assert set(want) == set(mylib_ops.SCHEMAS)
    for name, schema in want.items():
        assert str(getattr(torch.ops.mylib, name).default._schema) == schema


def test_fake_kernels_and_cpu_rejection():
    from torch._subclasses.fake_tensor import FakeTensorMode
    from magicdec_amd import mylib_ops
    mylib_ops.register()
    with FakeTensorMode():
        q = torch.empty(8, 4, 64, dtype=BF)
        k = torch.empty(8, 2, 64, dtype=BF)
        kv = torch.empty(3, 2, 128, 2, 64, dtype=BF)
        ip = torch.empty(3, dtype=torch.int32)
        assert torch.ops.mylib.target_decode(q, kv).shape == q.shape
        rq, rk = torch.ops.mylib.rope(q, k, ip, ip)
        assert rq.shape == q.shape and rk.shape == k.shape
        assert torch.ops.mylib.update_kv(k, k, ip, kv, ip, ip, ip) is None
    with pytest.raises(NotImplementedError):
        torch.ops.mylib.draft_decode(torch.empty(8, 4, 64, dtype=BF), torch.empty(3, 2, 128, 2, 64, dtype=BF))


@pytest.mark.gpu
def test_mylib_ops_on_gpu_match_the_oracle():
    from magicdec_amd import mylib_ops
    from magicdec_amd.Engine.model_core import ModelArgs
    from tests.parity_util import check_attention, dense_attention_f64
    from tests.test_gpu_ops import bits, make_paged
    mylib_ops.register()
    dev = "cuda"
    B, n, H, KH, D = 3, 4, 8, 2, 128
    lens = [300, 131, 4]          # lengths AFTER the append
    cache, indices, indptr, last, max_pages = make_paged(B, lens, KH, D, seed=21, scatter=True)
    cfg = ModelArgs(n_layer=1, n_head=H, n_local_heads=KH, dim=H * D, vocab_size=128, rope_base=500000.0,
                    scaling_factor=8, high_freq_factor=4, low_freq_factor=1, original_max_position_embeddings=8192)
    mylib_ops.bind_rope("rope", cfg, 4096, device=dev)
    tab_ref = fr.rope_table(4096, D, 500000.0, 8.0, low_freq_factor=1.0, high_freq_factor=4.0, old_context_len=8192)
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * n, (H + 2 * KH) * D, generator=g).to(BF)
    q = qkv[:, :H * D].unflatten(1, (H, D)).contiguous()
    k = qkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D)).contiguous()
    v = qkv[:, (H + KH) * D:].unflatten(1, (KH, D)).contiguous()
    ip = torch.arange(B + 1, dtype=torch.int32) * n
    offsets = torch.tensor([l - n for l in lens], dtype=torch.int32)
    d = lambda t: t.to(dev)
    # mylib::rope  (Engine/SnapKV/model.py:329: q, k = self.rope(q, k, kv_append_indptr, offsets))
    rq, rk = fr.apply_rope(q, k, ip, offsets, tab_ref)
    oq, ok = torch.ops.mylib.rope(d(q), d(k), d(ip), d(offsets))
    assert torch.equal(bits(oq.cpu()), bits(rq)) and torch.equal(bits(ok.cpu()), bits(rk))
    # mylib::update_kv  (Engine/SnapKV/model.py:90-112 via KVCache.update)
    ref_cache = cache.clone()
    fr.append_paged_kv_cache(rk, v, ip, ref_cache, indices, indptr, last)
    dcache = d(cache)
    assert torch.ops.mylib.update_kv(ok, d(v), d(ip), dcache, d(indices), d(indptr), d(last)) is None
    assert torch.equal(bits(dcache.cpu()), bits(ref_cache))
    # mylib::target_decode after a host-side plan()  (Engine/SnapKV/backend.py:148-159, model.py:331)
    wrapper = mylib_ops.PagedAttentionPlan(torch.empty(1, dtype=torch.uint8, device=dev), "NHD", use_cuda_graph=True)
    for which in mylib_ops.ATTENTION_OPS:
        mylib_ops.bind_plan(which, wrapper)
    wrapper.plan(qo_indptr=d(ip), paged_kv_indptr=d(indptr), paged_kv_indices=d(indices),
                 paged_kv_last_page_len=d(last), num_qo_heads=H, num_kv_heads=KH, head_dim=D, page_size=128,
                 q_data_type=BF, causal=True)
    oracle = fr.batch_prefill_paged(rq, ref_cache, ip, indices, indptr, last, H, KH, D, causal=True)
    ref64, bnd = dense_attention_f64(rq, ref_cache, ip, indices, indptr, last, H, KH, D, causal=True)
    for which in mylib_ops.ATTENTION_OPS:
        out = getattr(torch.ops.mylib, which)(oq, dcache)
        check_attention("mylib::" + which, out, oracle, ref64, bnd)
    with pytest.raises(ValueError):
        torch.ops.mylib.target_decode(oq[:, :4], dcache)      # head geometry differs from the plan
