"""CPU tests (no GPU): the C-ABI surface, and the product's HOST logic -- back-end page-table state machines,
model wiring, decode loops, CLI entry points, TP sharding over gloo (world_size 2) -- run with the device ops
replaced by oracle stand-ins (tests/cpu_ops.py) and compared bit-exactly with the traces recorded from the real
reference scripts (tests/golden/run_*.json)."""
import os
import re
import subprocess
import sys
import tempfile
from pathlib import Path

import pytest
import torch

from tests import cpu_ops
from tests import golden_cfg as gc

ROOT = Path(__file__).resolve().parents[1]


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    """include/magicdec_hip.h <-> libmagicdec_hip.so <-> magicdec_amd/_lib.py agree (no compute calls)."""
    header = (ROOT / "include" / "magicdec_hip.h").read_text()
    declared = set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", header))
    from magicdec_amd import _lib
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    # the tuning knobs live in their own header and only in a -DMD_DEV_KNOBS build (the in-tree default)
    assert not any(n.startswith("md_debug_set") for n in declared)
    dev = set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", (ROOT / "include" / "magicdec_hip_dev.h").read_text()))
    assert dev == set(_lib.DEV_SYMBOLS) and all(n.startswith("md_debug_set") for n in dev)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.md_abi_version() == _lib.ABI_VERSION
    assert lib.md_last_error_string() is not None


def test_host_side_rope_table_matches_oracle():
    import ctypes
    from magicdec_amd import _lib
    from oracle import flashinfer_ref as fr
    lib = _lib.load()
    for (mp, D, theta, scale, lo, hi, old) in [(512, 64, 10000.0, 1.0, None, None, None),
                                               (2048, 128, 500000.0, 8.0, 1.0, 4.0, 8192)]:
        host = torch.empty(mp, D // 2, 2)
        rc = lib.md_rope_fill_table_host(ctypes.c_void_p(host.data_ptr()), mp, D, theta, scale, lo or 0.0, hi or 0.0,
                                         float(old or 0))
        assert rc == 0
        assert torch.equal(host, fr.rope_table(mp, D, theta, scale, lo, hi, old))


def test_ops_refuse_cpu_tensors_and_bad_arguments():
    """No CPU fallback: device ops raise on CPU tensors; the C side validates shapes and reports a message."""
    import ctypes
    from magicdec_amd import _lib, ops
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.rmsnorm(x, torch.ones(64, dtype=torch.bfloat16), 1e-5)
    lib = _lib.load()
    rc = lib.md_paged_attn(None, 0, None, None, None, None, None, None, 1, 1, 8, 2, 128, 128, 1, 1.0, 1, 0, None, None,
                           None, 0, None)
    assert rc == -1 and b"null pointer" in lib.md_last_error_string()
    p = ctypes.c_void_p(256)
    rc = lib.md_paged_attn(p, 1024, p, p, p, p, p, p, 1, 1, 8, 2, 96, 128, 1, 1.0, 1, 0, None, None, None, 0, None)
    assert rc == -2 and b"head_dim" in lib.md_last_error_string()
    comm = ctypes.c_void_p()
    assert lib.md_ar_create(0, 9, 1 << 20, ctypes.byref(comm)) == -1 and b"world" in lib.md_last_error_string()
    assert lib.md_ar_create(2, 2, 1 << 20, ctypes.byref(comm)) == -1
    assert lib.md_allreduce_oneshot(None, p, p, 8, None) == -1


def test_product_fails_loudly_without_the_hip_library(monkeypatch):
    """No CPU fallback: with the shared library missing every device op raises MagicDecHipError naming the build
    command (not a silent PyTorch path), and the error is sticky."""
    from magicdec_amd import _lib, ops
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_err", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmagicdec_hip.so")
    with pytest.raises(_lib.MagicDecHipError, match="no CPU fallback"):
        _lib.load()
    with pytest.raises(_lib.MagicDecHipError, match="__graft_entry__"):
        ops.RopeTable(64, 64, 10000.0, 1.0, device="cpu")
    # nothing under the product imports the oracle
    import subprocess
    out = subprocess.run(["grep", "-rln", "-E", r"^\s*(from|import) oracle", str(ROOT / "magicdec_amd")],
                         capture_output=True, text=True).stdout.strip()
    assert out == "", out


def test_every_entry_point_rejects_null_arguments():
    """Error behaviour of the C ABI: every entry point called with all-zero / NULL arguments returns a negative
    MD_ERR_* code (or 0 bytes for the size queries) and leaves a message naming itself -- validation happens before
    anything touches the device, so this runs without a GPU."""
    import ctypes
    from magicdec_amd import _lib
    lib = _lib.load()
    skip = {"md_abi_version", "md_last_error_string", "md_ar_destroy", "md_debug_attn_timing", "md_debug_attn_timing_read",
            "md_clear_last_hip_error"}      # (no arguments: returns the HIP runtime's sticky error code, e.g. "no device" here)
    assert lib.md_linear_supported(0, 0, 0, 0) == 0 and lib.md_linear_supported(64, 128, 256, 0) == 1
    assert lib.md_linear_fused_supported(0, 0, 0, 0) == 0 and lib.md_linear_fused_supported(64, 768, 2048, 3) == 1
    assert lib.md_linear_fused_supported(64, 770, 2048, 0) == 0 and lib.md_linear_fused_supported(300, 768, 2048, 0) == 0
    assert lib.md_linear_add_rmsnorm_supported(0, 0, 0) == 0 and lib.md_linear_add_rmsnorm_supported(256, 4096, 14336) == 1
    assert lib.md_linear_block_supported(0, 0, 0, 0) == 0 and lib.md_linear_block_supported(256, 4096, 4096, 0) == 1
    assert lib.md_linear_block_supported(256, 4100, 4096, 0) == 0 and lib.md_linear_block_supported(257, 4096, 4096, 0) == 0
    assert lib.md_linear_block_workspace_bytes(256, 4096, 4096, 0) > 0 and lib.md_linear_block_workspace_bytes(256, 28672, 4096, 0) == 0
    skip |= {"md_linear_supported", "md_linear_fused_supported", "md_linear_add_rmsnorm_supported", "md_linear_block_supported"}
    # md_linear_fused takes a struct: NULL struct, then a zeroed struct (null tensors), then tensors but a bad shape
    assert lib.md_linear_fused(None, None) < 0 and "md_linear_fused" in lib.md_last_error_string().decode()
    fa = _lib.FusedLinearArgs()
    assert lib.md_linear_fused(ctypes.byref(fa), None) < 0 and "md_linear_fused" in lib.md_last_error_string().decode()
    fa.x = fa.w_packed = fa.out = 4096
    fa.M, fa.N, fa.K = 64, 100, 256
    assert lib.md_linear_fused(ctypes.byref(fa), None) < 0 and "unsupported shape" in lib.md_last_error_string().decode()
    skip.add("md_linear_fused")
    for name, (restype, argtypes) in _lib._SIGNATURES.items():
        if name in skip:
            continue
        args = []
        for t in argtypes:
            args.append(None if t is ctypes.c_void_p else (0.0 if t in (ctypes.c_float, ctypes.c_double) else 0))
        rc = getattr(lib, name)(*args)
        if restype is ctypes.c_size_t:
            assert rc == 0, name
        else:
            assert rc < 0, (name, rc)
            msg = lib.md_last_error_string().decode()
            assert name.replace("md_allreduce_oneshot", "md_allreduce") in msg or name in msg, (name, msg)
    assert lib.md_ar_destroy(None) == 0          # destroying nothing is fine


# ------------------------------------------------------------------ host logic vs the reference's own traces
@pytest.fixture()
def cpu_ops_patched(monkeypatch):
    cpu_ops.install(monkeypatch)
    cpu_ops.TOPK_REPLAY.update(table=None, pos=0)
    yield
    cpu_ops.TOPK_REPLAY.update(table=None, pos=0)


@pytest.fixture(scope="module")
def ckpt_dir():
    from magicdec_amd.Engine import model_core
    d = tempfile.mkdtemp(prefix="md_ckpt_")
    for name in gc.TINY:
        cfg, sd = gc.tiny(name)
        os.makedirs(os.path.join(d, name), exist_ok=True)
        torch.save(sd, os.path.join(d, name, "model.pth"))
        model_core.transformer_configs[name] = gc.config_kwargs(cfg)
    return Path(d)


class Tracer:
    """Records the product back-end calls in the same format as oracle/gen_golden.py's tracer."""
    ATTRS = ["cachelens", "paged_kv_last_page_len", "paged_kv_indptr", "draft_cachelens", "draft_paged_kv_last_page_len",
             "draft_paged_kv_indptr"]

    def __init__(self, eng, cls, log, fns):
        self.__dict__.update(eng=eng, cls=cls, log=log, fns=fns)

    def __getattr__(self, name):
        attr = getattr(self.eng, name)
        if name not in self.fns:
            return attr

        def call(*a, **kw):
            ids = a[0] if a else kw["input_ids"]
            out = attr(*a, **kw)
            rec = dict(cls=self.cls, fn=name, inp=ids.tolist() if ids.shape[1] <= 8 else [int(ids.shape[1])],
                       out=out.tolist() if out.shape[1] <= 8 else out[:, -1:].tolist())
            if kw.get("cachelen_update") is not None:
                rec["cachelen_update"] = kw["cachelen_update"].flatten().tolist()
            for at in self.ATTRS:
                if getattr(self.eng, at, None) is not None:
                    rec[at] = getattr(self.eng, at).tolist()
            self.log.append(rec)
            return out
        return call

    def __setattr__(self, k, v):
        setattr(self.eng, k, v)


def _compare(trace, ref):
    assert len(trace) == len(ref), (len(trace), len(ref))
    for a, b in zip(trace, ref):
        assert (a["cls"], a["fn"]) == (b["cls"], b["fn"])
        for key in ("inp", "out", "cachelen_update", *Tracer.ATTRS):
            if key in b:
                assert a.get(key) == b[key], (a["cls"], a["fn"], key, a.get(key), b[key])


@pytest.mark.parametrize("kind,gamma", [("longspec_snapkv", 3), ("longspec_snapkv_rej", 1), ("longspec_stream", 3),
                                        ("longspec_stream_70b", 3), ("longspec_snapkv_fullkv", 3),
                                        ("longspec_snapkv_b257", 3), ("longspec_stream_noevict", 3)])
def test_product_longspec_host_logic_matches_reference_trace(kind, gamma, cpu_ops_patched, ckpt_dir):
    from magicdec_amd import harness
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    j = gc.load_json(f"run_{kind}.json")
    eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gamma + 1)
    eng.load_model(ckpt_dir / ("tiny70b" if kind.endswith("70b") else "tinytgt") / "model.pth", use_tp=False)
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
    if "snapkv" in kind:
        from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
        budget = int(j["argv"][j["argv"].index("--draft_budget") + 1])    # 129, 257 (3 draft pages) or -1 = the draft
        # decodes over its full KV (the script default)
        drf = LMBackend_Draft(dtype=torch.bfloat16, device="cpu", draft_budget=budget)
        drf.load_model(ckpt_dir / ("tinydrf" if kind.endswith(("rej", "fullkv")) else "tinytgt") / "model.pth",
                       use_tp=False)
        drf.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=budget)
        cpu_ops.TOPK_REPLAY.update(table=j["snapkv_topk"], pos=0)
        dcls = "SnapKV.LMBackend_Draft"
    else:
        from magicdec_amd.Engine.StreamingLLM.backend_draft import LMBackend_Draft
        drf = LMBackend_Draft(dtype=torch.bfloat16, device="cpu")
        drf.load_model(ckpt_dir / ("tinydrf" if kind.endswith("70b") else "tinytgt") / "model.pth", use_tp=False)
        # 129, or 513 > prefix + generated tokens: the StreamingLLM cache never evicts
        drf.setup_caches(max_batch_size=gc.B, draft_budget=int(j["argv"][j["argv"].index("--draft_budget") + 1]))
        dcls = "StreamingLLM.LMBackend_Draft"
    log = []
    te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
    td = Tracer(drf, dcls, log, ("encode", "inference"))
    last = None
    for ids in gc.synthetic_batches():
        last, _ = harness.run_longspec_batch(te, td, ids, gamma, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
    _compare(log, j["trace"])
    assert last.output.tolist() == j["final"]["output"]
    assert last.num_nodes.tolist() == j["final"]["num_nodes"]


@pytest.mark.parametrize("kind", ["selfspec_snapkv", "selfspec_stream", "selfspec_snapkv_qwen", "selfspec_snapkv_70b",
                                  "selfspec_stream_b257", "selfspec_snapkv_b257", "selfspec_stream_g5", "selfspec_snapkv_g5"])
def test_product_selfspec_host_logic_matches_reference_trace(kind, cpu_ops_patched, ckpt_dir):
    """_b257: the BASELINE configs[1] / configs[4] draft budget (3 draft pages per request); _g5: the scripts' default
    speculation length gamma = 5."""
    from magicdec_amd import harness
    j = gc.load_json(f"run_{kind}.json")
    model = "tinyqwen" if kind.endswith("qwen") else "tiny70b" if kind.endswith("70b") else "tinytgt"
    streaming = "selfspec_stream" in kind
    gamma = int(j["argv"][j["argv"].index("--gamma") + 1])
    budget = int(j["argv"][j["argv"].index("--draft_budget") + 1])
    if streaming:
        from magicdec_amd.Engine.StreamingLLM.backend import LMBackend
        eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gamma + 1)
        cls = "StreamingLLM.LMBackend"
    else:
        from magicdec_amd.Engine.SnapKV.backend import LMBackend
        eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gamma + 1, draft_dec_len=1)
        cls = "SnapKV.LMBackend"
        cpu_ops.TOPK_REPLAY.update(table=j["snapkv_topk"], pos=0)
    eng.load_model(ckpt_dir / model / "model.pth", use_tp=False)
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=budget)
    log = []
    te = Tracer(eng, cls, log, ("encode", "draft_encode", "speculate", "verify"))
    last = None
    for ids in gc.synthetic_batches():
        last, _ = harness.run_selfspec_batch(te, ids, gamma, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, streaming)
    _compare(log, j["trace"])
    assert last.output.tolist() == j["final"]["output"]
    assert last.num_nodes.tolist() == j["final"]["num_nodes"]


def test_product_baseline_host_logic_matches_reference_trace(cpu_ops_patched, ckpt_dir):
    from magicdec_amd import harness
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    j = gc.load_json("run_baseline.json")
    eng = LMBackend(dtype=torch.bfloat16, device="cpu")
    eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
    log = []
    te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
    out = None
    for ids in gc.synthetic_batches():
        out, _, _ = harness.run_baseline_batch(te, ids, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
    _compare(log, j["trace"])
    assert out.tolist() == j["final"]["output"]


def test_model_config_lookup_and_cli_asserts():
    from magicdec_amd.Engine.model_core import ModelArgs
    a = ModelArgs.from_name("Meta-Llama-3.1-8B")
    assert (a.n_layer, a.n_head, a.n_local_heads, a.dim, a.head_dim, a.scaling_factor) == (32, 32, 8, 4096, 128, 8)
    b = ModelArgs.from_name("Llama-3.2-1B")
    assert (b.n_layer, b.n_local_heads, b.head_dim, b.scaling_factor) == (16, 8, 64, 32)
    assert ModelArgs.from_name("llama-68m").dim == 768
    from magicdec_amd.cli import longspec_main
    with pytest.raises(AssertionError):     # prefix that is not window + k*128 (tests/SnapKV/longspec_benchmark.py:45)
        longspec_main("SnapKV", ["--B", "1", "--prefix_len", "400", "--max_len", "512", "--draft_budget", "129",
                                 "--rank_group", "0", "--draft_rank_group", "0"])


# ------------------------------------------------------------------ tensor parallel over gloo, world_size 2
TP_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["MD_ROOT"])
import torch.distributed as dist
from pathlib import Path
from tests import cpu_ops, golden_cfg as gc
from oracle import magicdec_ref as mr, harness_ref as hr
cpu_ops.install()
from magicdec_amd import harness
from magicdec_amd.Engine import model_core
from magicdec_amd.Engine.tp import init_dist
from magicdec_amd.Engine.SnapKV.backend import LMBackend
from magicdec_amd.Engine.StreamingLLM.backend_draft import LMBackend_Draft
ck = Path(os.environ["MD_CKPT"])
gc.register_tiny(model_core)
draft_ranks = [int(x) for x in os.environ["MD_DRAFT_RANKS"].split(",")]
rank, group, dgroup = init_dist(draft_ranks)
world = dist.get_world_size()
eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1)
eng.load_model(ck / "tinytgt" / "model.pth", use_tp=True, rank_group=list(range(world)), group=group)
eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
drf = None
if rank in draft_ranks:
    drf = LMBackend_Draft(dtype=torch.bfloat16, device="cpu")
    drf.load_model(ck / "tinytgt" / "model.pth", use_tp=len(draft_ranks) > 1, rank_group=draft_ranks, group=dgroup)
    drf.setup_caches(max_batch_size=gc.B, draft_budget=gc.BUDGET)
bcast = (draft_ranks[0], group) if len(draft_ranks) != world else None
ids = gc.synthetic_batches()[0]
st, _ = harness.run_longspec_batch(eng, drf, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, bcast=bcast, barrier=dist.barrier)
# oracle with the same sharding and the same gloo all-reduce
cfg, sd = gc.tiny("tinytgt")
ssd, lcfg = mr.shard_state_dict(sd, cfg, rank, world)
o_t = mr.RefEngine("target", lcfg, ssd, gc.B, gc.MAX_LEN, group=group, rank=rank, world=world)
if rank in draft_ranks:
    if len(draft_ranks) > 1:
        dsd, dcfg = mr.shard_state_dict(sd, cfg, draft_ranks.index(rank), len(draft_ranks))
        o_d = mr.RefEngine("stream_draft", dcfg, dsd, gc.B, 0, gc.BUDGET, group=dgroup, rank=draft_ranks.index(rank), world=len(draft_ranks))
    else:
        o_d = mr.RefEngine("stream_draft", cfg, sd, gc.B, 0, gc.BUDGET)
else:
    o_d = None
res = dict(rank=rank, output=st.output.tolist(), num_nodes=st.num_nodes.tolist(), iters=st.iters,
           cachelens=eng.cachelens.tolist(), local_heads=[eng.model.config.n_head, eng.model.config.n_local_heads])
if o_d is not None and len(draft_ranks) == world:
    ref = hr.longspec_batch(o_t, o_d, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
    res["oracle_output"] = ref["output"].tolist()
    res["oracle_num_nodes"] = ref["num_nodes"].tolist()
if os.environ.get("MD_GOLDEN") == "1":
    # all batches, traced like the reference's own TP=2 run (tests/golden/run_longspec_stream_tp2.json)
    from tests.test_host_cpu import Tracer
    log = []
    te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
    td = Tracer(drf, "StreamingLLM.LMBackend_Draft", log, ("encode", "inference"))
    last = None
    for b_ids in gc.synthetic_batches():
        last, _ = harness.run_longspec_batch(te, td, b_ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, barrier=dist.barrier)
    res["golden_trace"] = log
    res["golden_final"] = dict(output=last.output.tolist(), num_nodes=last.num_nodes.tolist())
    # the oracle, sharded the same way, over the same batches
    o_t2 = mr.RefEngine("target", lcfg, ssd, gc.B, gc.MAX_LEN, group=group, rank=rank, world=world)
    dsd2, dcfg2 = mr.shard_state_dict(sd, cfg, rank, world)
    o_d2 = mr.RefEngine("stream_draft", dcfg2, dsd2, gc.B, 0, gc.BUDGET, group=dgroup, rank=rank, world=world)
    ref_last = None
    for b_ids in gc.synthetic_batches():
        ref_last = hr.longspec_batch(o_t2, o_d2, b_ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
    res["oracle_golden_final"] = dict(output=ref_last["output"].tolist(), num_nodes=ref_last["num_nodes"].tolist())
json.dump(res, open(os.path.join(os.environ["MD_OUT"], f"rank{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("draft_ranks", ["0,1", "0"])
def test_tensor_parallel_gloo_world2(draft_ranks, ckpt_dir):
    """TP=2 over gloo: both ranks end with identical replicated state; with draft TP == target TP the run equals the
    oracle sharded the same way (same gloo bf16 sum all-reduce) bit for bit, and both equal the trace of the real
    reference run at TP=2 (golden fixture); with a 1-rank draft sub-group the
    gamma tokens are broadcast (tests/SnapKV/longspec_benchmark.py:189)."""
    import json
    out = tempfile.mkdtemp(prefix="md_tp_")
    script = os.path.join(out, "worker.py")
    Path(script).write_text(TP_WORKER)
    port = 29500 + (os.getpid() % 2000) + (7 if draft_ranks == "0" else 0)
    procs = []
    for r in range(2):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_ROOT=str(ROOT), MD_CKPT=str(ckpt_dir), MD_OUT=out,
                   MD_DRAFT_RANKS=draft_ranks, MD_GOLDEN="1" if draft_ranks == "0,1" else "0", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = (json.load(open(os.path.join(out, f"rank{r}.json"))) for r in range(2))
    assert r0["local_heads"] == [4, 1] and r1["local_heads"] == [4, 1]
    assert r0["output"] == r1["output"] and r0["num_nodes"] == r1["num_nodes"] and r0["cachelens"] == r1["cachelens"]
    assert r0["iters"] == r1["iters"] and r0["iters"] > 3
    if draft_ranks == "0,1":
        assert r0["output"] == r0["oracle_output"] and r0["num_nodes"] == r0["oracle_num_nodes"]
        # ... and both equal the REAL reference run at TP=2 over gloo (oracle/gen_golden.py run_longspec_stream_tp2):
        # every Engine call's tokens and page-table state, and the final output, bit for bit
        j = gc.load_json("run_longspec_stream_tp2.json")
        _compare(r0["golden_trace"], j["trace"])
        assert r0["golden_final"] == j["final"] and r0["oracle_golden_final"] == j["final"]
        assert r1["golden_final"] == j["final"]


TP_SNAPKV_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["MD_ROOT"])
import torch.distributed as dist
from pathlib import Path
from tests import cpu_ops, golden_cfg as gc
cpu_ops.install()
from magicdec_amd import harness
from magicdec_amd.Engine import model_core
from magicdec_amd.Engine.tp import init_dist
from magicdec_amd.Engine.SnapKV.backend import LMBackend
from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
from tests.test_host_cpu import Tracer
ck = Path(os.environ["MD_CKPT"])
gc.register_tiny(model_core)
draft_ranks = [int(x) for x in os.environ.get("MD_DRAFT_RANKS", "0,1").split(",")]
model = os.environ.get("MD_MODEL", "tinytgt")
rank, group, dgroup = init_dist(draft_ranks)
world = dist.get_world_size()
ranks = list(range(world))
kind = os.environ["MD_KIND"]                      # fixture name: run_longspec_snapkv_tp2 | run_selfspec_snapkv_tp2 | ...
# each rank replays the reference's tie resolution for ITS kv heads (torch.topk's tie order is implementation-defined);
# in the longspec layout only the draft sub-group runs the SnapKV select
if os.environ.get("MD_REPLICATE_DRAFT", "0") == "1":
    pass                                          # no reference run to replay: torch.topk's own tie order
elif "snapkv" in kind and (rank in draft_ranks or not kind.startswith("run_longspec")):
    name = f"{kind}.json" if rank == 0 else f"{kind}_topk_rank{rank}.json"
    cpu_ops.TOPK_REPLAY.update(table=gc.load_json(name)["snapkv_topk"], pos=0)
log = []
last = None
if kind.startswith("run_longspec"):
    eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1)
    eng.load_model(ck / model / "model.pth", use_tp=True, rank_group=ranks, group=group)
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
    td = None
    replicate = os.environ.get("MD_REPLICATE_DRAFT", "0") == "1"      # --replicate_draft: the whole draft on every rank
    if rank in draft_ranks or replicate:          # tests/SnapKV/longspec_benchmark.py:96-103
        drf = LMBackend_Draft(dtype=torch.bfloat16, device="cpu", draft_budget=gc.BUDGET)
        drf.load_model(ck / model / "model.pth", use_tp=len(draft_ranks) > 1 and not replicate, rank_group=draft_ranks,
                       group=dgroup)
        drf.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
        td = Tracer(drf, "SnapKV.LMBackend_Draft", log, ("encode", "inference"))
    te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
    bcast = (draft_ranks[0], group) if (len(draft_ranks) != world and not replicate) else None   # longspec_benchmark.py:189
    for b_ids in gc.synthetic_batches():
        last, _ = harness.run_longspec_batch(te, td, b_ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, bcast=bcast,
                                             barrier=dist.barrier)
elif "snapkv" in kind:
    eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1, draft_dec_len=1)
    eng.load_model(ck / model / "model.pth", use_tp=True, rank_group=ranks, group=group)
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
    te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "draft_encode", "speculate", "verify"))
    for b_ids in gc.synthetic_batches():
        last, _ = harness.run_selfspec_batch(te, b_ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, False)
else:
    from magicdec_amd.Engine.StreamingLLM.backend import LMBackend as StreamSelf
    eng = StreamSelf(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1)
    eng.load_model(ck / model / "model.pth", use_tp=True, rank_group=ranks, group=group)
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
    te = Tracer(eng, "StreamingLLM.LMBackend", log, ("encode", "draft_encode", "speculate", "verify"))
    for b_ids in gc.synthetic_batches():
        last, _ = harness.run_selfspec_batch(te, b_ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, True)
json.dump(dict(trace=log, final=dict(output=last.output.tolist(), num_nodes=last.num_nodes.tolist()),
               local_heads=[eng.model.config.n_head, eng.model.config.n_local_heads],
               draft_heads=([drf.model.config.n_head, drf.model.config.n_local_heads]
                            if kind.startswith("run_longspec") and td is not None else None)),
          open(os.path.join(os.environ["MD_OUT"], f"rank{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("kind", ["run_longspec_snapkv_tp2", "run_selfspec_snapkv_tp2", "run_selfspec_stream_tp2"])
def test_tensor_parallel_snapkv_matches_reference_tp2_trace(kind, ckpt_dir):
    """The TP layouts of BASELINE configs[2] and configs[4] in miniature -- target TP2 + SnapKV draft TP2, and TP2
    self-speculation with the SnapKV cache (kv-head-sharded select/gather) -- over gloo: every Engine call's tokens
    and page-table state and the final output equal the REAL reference's TP=2 runs (oracle/gen_golden.py
    run_{longspec,selfspec}_snapkv_tp2) bit for bit, on both ranks."""
    import json
    out = tempfile.mkdtemp(prefix="md_tp_snap_")
    script = os.path.join(out, "worker.py")
    Path(script).write_text(TP_SNAPKV_WORKER)
    port = 29400 + (os.getpid() % 500) + ["run_longspec_snapkv_tp2", "run_selfspec_snapkv_tp2", "run_selfspec_stream_tp2"].index(kind)
    procs = []
    for r in range(2):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_ROOT=str(ROOT), MD_CKPT=str(ckpt_dir), MD_OUT=out,
                   MD_KIND=kind, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    j = gc.load_json(f"{kind}.json")
    for r in range(2):
        got = json.load(open(os.path.join(out, f"rank{r}.json")))
        _compare(got["trace"], j["trace"])
        assert got["final"] == j["final"]


def test_tensor_parallel_uneven_kv_head_shards_match_reference_tp3_trace(ckpt_dir):
    """KH % tp != 0 (SURVEY 8e "Constraint"): four kv heads over three ranks -> shards of 2, 1, 1 kv heads
    (Engine/tp.py:36-52), i.e. differently sized wqkv / wo slices and KV caches per rank.  StreamingLLM self-speculation
    over gloo, world_size 3: every rank's Engine-call trace and final output equal the REAL reference's TP=3 run
    (oracle/gen_golden.py run_selfspec_stream_tp3) bit for bit."""
    import json
    kind = "run_selfspec_stream_tp3"
    out = tempfile.mkdtemp(prefix="md_tp3_")
    script = os.path.join(out, "worker.py")
    Path(script).write_text(TP_SNAPKV_WORKER)
    port = 29200 + (os.getpid() % 500)
    procs = []
    for r in range(3):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="3", RANK=str(r), WORLD_SIZE="3",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_ROOT=str(ROOT), MD_CKPT=str(ckpt_dir), MD_OUT=out,
                   MD_KIND=kind, MD_DRAFT_RANKS="0,1,2", MD_MODEL="tinykh4", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=1500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    j = gc.load_json(f"{kind}.json")
    got = [json.load(open(os.path.join(out, f"rank{r}.json"))) for r in range(3)]
    assert [g["local_heads"] for g in got] == [[8, 2], [4, 1], [4, 1]]
    for g in got:
        _compare(g["trace"], j["trace"])
        assert g["final"] == j["final"]


def test_tensor_parallel_selfspec_snapkv_one_kv_head_per_rank_matches_reference_tp4_trace(ckpt_dir):
    """BASELINE configs[4]'s per-rank layout exactly -- ONE kv head per rank (four-kv-head model over 4 ranks), every
    rank running the SnapKV select / gather of its own head -- over gloo, world_size 4: all four product ranks reproduce
    the REAL reference's TP=4 trace (oracle/gen_golden.py run_selfspec_snapkv_tp4; each rank replays the reference's
    torch.topk tie resolution for its head)."""
    import json
    kind = "run_selfspec_snapkv_tp4"
    out = tempfile.mkdtemp(prefix="md_tp4s_")
    script = os.path.join(out, "worker.py")
    Path(script).write_text(TP_SNAPKV_WORKER)
    port = 29050 + (os.getpid() % 40)
    procs = []
    for r in range(4):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="4", RANK=str(r), WORLD_SIZE="4",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_ROOT=str(ROOT), MD_CKPT=str(ckpt_dir), MD_OUT=out,
                   MD_KIND=kind, MD_DRAFT_RANKS="0,1,2,3", MD_MODEL="tinykh4", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=1500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    j = gc.load_json(f"{kind}.json")
    for r in range(4):
        got = json.load(open(os.path.join(out, f"rank{r}.json")))
        assert got["local_heads"] == [4, 1]
        _compare(got["trace"], j["trace"])
        assert got["final"] == j["final"]


def test_tensor_parallel_tp4_target_with_tp2_draft_subgroup_matches_reference_trace(ckpt_dir):
    """The reference README's topology (README.md:69: target on all ranks, draft on a sub-group) in miniature: target
    TP4 + SnapKV draft TP2 on ranks {0, 1} over gloo, world_size 4, four-kv-head model.  Ranks 2 and 3 hold no draft
    model and receive the gamma draft tokens by broadcast (tests/SnapKV/longspec_benchmark.py:176-189).  Rank 0's
    Engine-call trace (target + draft) and rank 3's (target only) and every rank's final output equal the REAL
    reference's run of the same command (oracle/gen_golden.py run_longspec_snapkv_tp4d2) bit for bit."""
    import json
    kind = "run_longspec_snapkv_tp4d2"
    out = tempfile.mkdtemp(prefix="md_tp4_")
    script = os.path.join(out, "worker.py")
    Path(script).write_text(TP_SNAPKV_WORKER)
    port = 29300 + (os.getpid() % 500)
    procs = []
    for r in range(4):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="4", RANK=str(r), WORLD_SIZE="4",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_ROOT=str(ROOT), MD_CKPT=str(ckpt_dir), MD_OUT=out,
                   MD_KIND=kind, MD_DRAFT_RANKS="0,1", MD_MODEL="tinykh4", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=1500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    j = gc.load_json(f"{kind}.json")
    j3 = gc.load_json(f"{kind}_trace_rank3.json")
    got = [json.load(open(os.path.join(out, f"rank{r}.json"))) for r in range(4)]
    _compare(got[0]["trace"], j["trace"])
    _compare(got[3]["trace"], j3["trace"])
    assert sum(1 for r in j["trace"] if "cachelen_update" in r) > 10          # two-token draft steps occurred
    assert not any(r["cls"].endswith("LMBackend_Draft") for r in got[2]["trace"] + got[3]["trace"])
    for r in range(4):
        assert got[r]["final"] == j["final"]


def test_tensor_parallel_target_with_replicated_draft(ckpt_dir):
    """`--replicate_draft` (round 4, not in the reference): target TP2 over gloo, the WHOLE SnapKV draft model on both
    ranks -- no draft-side collective, no per-iteration token broadcast.  Greedy drafting is deterministic, so both
    ranks must stay in lock-step on their own: identical Engine-call traces (target and draft) and identical outputs;
    the draft is unsharded on both; and because greedy speculative decoding returns the TARGET's greedy continuation
    whatever the draft proposes, the final output equals the reference's TP2 run with its sharded draft
    (tests/golden/run_longspec_snapkv_tp2.json)."""
    import json
    kind = "run_longspec_snapkv_tp2"
    out = tempfile.mkdtemp(prefix="md_tp_repl_")
    script = os.path.join(out, "worker.py")
    Path(script).write_text(TP_SNAPKV_WORKER)
    port = 29250 + (os.getpid() % 40)
    procs = []
    for r in range(2):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_ROOT=str(ROOT), MD_CKPT=str(ckpt_dir), MD_OUT=out,
                   MD_KIND=kind, MD_REPLICATE_DRAFT="1", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = [json.load(open(os.path.join(out, f"rank{r}.json"))) for r in range(2)]
    assert got[0]["trace"] == got[1]["trace"] and got[0]["final"] == got[1]["final"]
    assert any(r["cls"].endswith("LMBackend_Draft") for r in got[1]["trace"])          # rank 1 drafts too
    assert got[0]["draft_heads"] == got[1]["draft_heads"]
    assert got[0]["draft_heads"][1] == 2 * got[0]["local_heads"][1]                     # draft: all kv heads; target: half
    assert got[0]["final"] == gc.load_json(f"{kind}.json")["final"]


# ------------------------------------------------------------------ checkpoint ingestion
def test_hf_checkpoint_conversion_roundtrip():
    """An HF-layout (half-split RoPE rows, separate q/k/v, sharded safetensors) copy of a tiny model converts to
    exactly the state dict the Engine loads; layout facts as convert_hf_checkpoint.py:103-114,147-161."""
    import json as _json
    from safetensors.torch import save_file
    from magicdec_amd.Engine import model_core
    from magicdec_amd.convert_hf_checkpoint import convert_hf_checkpoint
    cfg, sd = gc.tiny("tinydrf")
    model_core.transformer_configs["tinydrf"] = dict(block_size=4096, n_layer=cfg.n_layer, n_head=cfg.n_head,
                                                     n_local_heads=cfg.n_local_heads, dim=cfg.dim,
                                                     intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size)
    D, H, KH = cfg.head_dim, cfg.n_head, cfg.n_local_heads

    def to_hf_rows(w, nh):      # inverse of the interleave: rows (h, r, half) -> (h, half, r)
        return w.reshape(nh, D // 2, 2, -1).transpose(1, 2).reshape(nh * D, -1)
    hf = {"model.embed_tokens.weight": sd["tok_embeddings.weight"], "model.norm.weight": sd["norm.weight"],
          "lm_head.weight": sd["output.weight"]}
    for i in range(cfg.n_layer):
        p, h = f"layers.{i}.", f"model.layers.{i}."
        q, k, v = sd[p + "attention.wqkv.weight"].split([H * D, KH * D, KH * D])
        hf[h + "self_attn.q_proj.weight"] = to_hf_rows(q, H).contiguous()
        hf[h + "self_attn.k_proj.weight"] = to_hf_rows(k, KH).contiguous()
        hf[h + "self_attn.v_proj.weight"] = v.contiguous()
        hf[h + "self_attn.o_proj.weight"] = sd[p + "attention.wo.weight"]
        hf[h + "mlp.gate_proj.weight"] = sd[p + "feed_forward.w1.weight"]
        hf[h + "mlp.up_proj.weight"] = sd[p + "feed_forward.w3.weight"]
        hf[h + "mlp.down_proj.weight"] = sd[p + "feed_forward.w2.weight"]
        hf[h + "input_layernorm.weight"] = sd[p + "attention_norm.weight"]
        hf[h + "post_attention_layernorm.weight"] = sd[p + "ffn_norm.weight"]
    d = Path(tempfile.mkdtemp(prefix="md_hf_")) / "tinydrf"
    d.mkdir()
    names = sorted(hf)
    shards = {"model-00001-of-00002.safetensors": names[:len(names) // 2], "model-00002-of-00002.safetensors": names[len(names) // 2:]}
    for fn, ks in shards.items():
        save_file({k: hf[k].contiguous() for k in ks}, str(d / fn))
    (d / "model.safetensors.index.json").write_text(_json.dumps({"weight_map": {k: fn for fn, ks in shards.items() for k in ks}}))
    out = torch.load(convert_hf_checkpoint(d), weights_only=True)
    assert set(out) == set(sd)
    for k in sd:
        assert torch.equal(out[k], sd[k]), k


@pytest.mark.parametrize("case", ["llama_sharded", "qwen_bias_tied_single"])
def test_hf_checkpoint_conversion_equals_the_reference_converter(case):
    """Our converter against what the REFERENCE's convert_hf_checkpoint.py:79-163 wrote for the same seeded HF-layout
    checkpoint (fixture tests/golden/convert_hf.json, generated by oracle/gen_golden.py:scen_convert_hf running the
    real converter): same key set, shapes, dtypes, and every tensor byte-identical (sha256) -- sharded safetensors +
    index for the Llama-style case; q/k/v biases (:94-99) and a tied lm head (:147-149) for the Qwen-style one."""
    from magicdec_amd.Engine import model_core
    from magicdec_amd.convert_hf_checkpoint import convert_hf_checkpoint
    from tests import hf_fixture
    name, tied, sharded = hf_fixture.CASES[case]
    cfg, _ = gc.tiny(name)
    model_core.transformer_configs[name] = gc.config_kwargs(cfg)
    d = hf_fixture.write_hf_checkpoint(tempfile.mkdtemp(prefix="md_hf_"), case, cfg)
    got = hf_fixture.describe(torch.load(convert_hf_checkpoint(Path(d), model_name=name), weights_only=True))
    want = gc.load_json("convert_hf.json")[case]
    assert set(got) == set(want)
    for k in want:
        assert got[k] == want[k], k


@pytest.mark.parametrize("tag,bos", [("bos", True), ("no_bos", False)])
def test_pg19_adapter_equals_the_reference(tag, bos):
    """magicdec_amd.data.tokenize_pg19 against the tensor the REFERENCE's Data/data_converter.py:44-58 produced for the
    same seeded corpus and stub tokenizer (fixture tests/golden/pg19.json, oracle/gen_golden.py:scen_pg19): 50 of the
    52 books, 8000-token skip, last chunk of every book dropped (also when full), BOS (or EOS) in column 0,
    repeat(end): bit-identical int64 tensor (sha256)."""
    import hashlib
    from magicdec_amd.data import convert_pg19_dataset, tokenize_pg19
    from tests import pg19_fixture as pf
    root = tempfile.mkdtemp(prefix="md_pg19_")
    d = pf.write_corpus(root)
    t = tokenize_pg19(pf.WordTokenizer(bos), seq_len=pf.SEQ_LEN, end=pf.END, data_dir=d + "/")
    want = gc.load_json("pg19.json")[tag]
    assert list(t.shape) == want["shape"] and str(t.dtype) == want["dtype"]
    assert t[0].tolist() == want["first_row"] and sorted(set(t[:, 0].tolist())) == want["col0"]
    assert hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest() == want["sha256"]
    # the dataset entry point picks the corpus up when it is there, and says so when it is not
    ds = convert_pg19_dataset(tokenizer=pf.WordTokenizer(bos), seq_len=pf.SEQ_LEN, end=pf.END, data_dir=d + "/")
    assert torch.equal(ds.tensors[0], t)
    with pytest.raises(IndexError):
        tokenize_pg19(pf.WordTokenizer(bos), seq_len=pf.SEQ_LEN, end=1, data_dir=d + "/", n_books=60)


def test_int8_quantiser_and_linear_equal_the_reference():
    """magicdec_amd.Engine.quantize against the REFERENCE's Engine/quantize.py on the same seeded inputs (fixture
    tests/golden/int8_quant.json, oracle/gen_golden.py:scen_int8_quant): int8 weights, scales and the
    WeightOnlyInt8Linear output (CPU path: F.linear(x, w.to(bf16)) * scales) byte-identical; the oracle's int8 linear
    (the checker of the GPU path) gives the same bytes."""
    import hashlib
    from magicdec_amd.Engine import quantize as Q
    from oracle import magicdec_ref as mr
    want = gc.load_json("int8_quant.json")
    g = torch.Generator().manual_seed(31)
    w = torch.randn(96, 256, generator=g) * 0.05
    w[3] = 0.0
    w[5] = -w[5].abs()
    w[7, 11] = 40.0
    q, sc, zp = Q.dynamically_quantize_per_channel(w.float(), -128, 127, torch.int8)
    h = lambda t: hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()
    assert h(q) == want["q_sha"] and h(sc) == want["scales_sha"] and bool((zp == 0).all())
    assert q[7].tolist() == want["q_row7"]
    lin = Q.WeightOnlyInt8Linear(256, 96)
    lin.weight, lin.scales = q, sc.to(torch.bfloat16)
    x = torch.randn(5, 256, generator=g).to(torch.bfloat16)
    assert h(lin(x)) == want["y_sha"]
    assert h(mr.linear(x, q, None, sc.to(torch.bfloat16))) == want["y_sha"]
    # handler: state dict keys / dtypes as the reference's create_quantized_state_dict
    m = torch.nn.Sequential(torch.nn.Linear(256, 96, bias=False)).to(torch.bfloat16)
    sd = Q.WeightOnlyInt8QuantHandler(m).create_quantized_state_dict()
    assert sd["0.weight"].dtype == torch.int8 and sd["0.scales"].dtype == torch.bfloat16
    m2 = Q.WeightOnlyInt8QuantHandler(m).convert_for_runtime()
    m2.load_state_dict(sd)
    assert isinstance(m2[0], Q.WeightOnlyInt8Linear)


def test_fp8_kv_cache_host_logic(cpu_ops_patched, ckpt_dir):
    """kv_dtype="fp8" (BASELINE configs[4], not in the reference): the full cache is e4m3fn, scales are calibrated on
    the first prefill chunk, the compressed draft cache stays bf16, and teacher-forced logits stay close to the
    bf16-cache engine's (gate: relative L2 error of the verify logits <= 5%)."""
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    ids = next(iter(gc.synthetic_batches()))
    logits = {}
    for kvd in ("bf16", "fp8"):
        eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1, draft_dec_len=1)
        eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET, kv_dtype=kvd)
        eng.encode(ids)
        kvc = eng.model.layers[0].attention.kv_cache
        assert kvc.kv_cache.dtype == (torch.float8_e4m3fn if kvd == "fp8" else torch.bfloat16)
        assert kvc.draft_cache.dtype == torch.bfloat16
        if kvd == "fp8":
            assert kvc.calibrated and (kvc.k_scale != 1).all() and (kvc.v_scale > 0).all()
            assert kvc.draft_cache.float().abs().sum() > 0          # SnapKV gather dequantised into the draft cache
        probe = ids[:, :gc.GAMMA + 1].clone()
        eng.verify(probe)
        logits[kvd] = eng.model._last_logits.float()
    rel = (logits["fp8"] - logits["bf16"]).norm() / logits["bf16"].norm()
    assert rel <= 5e-2, rel
    with pytest.raises(NotImplementedError):
        from magicdec_amd.Engine.StreamingLLM.backend_draft import LMBackend_Draft
        d = LMBackend_Draft(dtype=torch.bfloat16, device="cpu")
        d.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        d.model.setup_caches(num_pages=4, streaming=True, draft_budget=gc.BUDGET, kv_dtype="fp8")


def test_hnd_page_layout_host_logic(cpu_ops_patched, ckpt_dir):
    """kv_layout="HND" (flashinfer's other page layout; the reference plans "NHD"): the full-context cache is allocated
    [pages, 2, KH, 128, D], the compressed draft cache stays NHD, every op on the full cache is told the layout, and the
    engine's tokens / lengths / cache contents equal the NHD engine's (the cache up to the permutation)."""
    from magicdec_amd import harness, ops
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    ids = next(iter(gc.synthetic_batches()))
    res = {}
    for layout in ops.KV_LAYOUTS:
        eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1, draft_dec_len=1)
        eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET, kv_layout=layout)
        kvc = eng.model.layers[0].attention.kv_cache
        assert kvc.layout_of("kv_cache") == layout and kvc.layout_of("draft_cache") == "NHD"
        assert tuple(kvc.kv_cache.shape[2:4]) == ((eng.model.config.n_local_heads, 128) if layout == "HND"
                                                  else (128, eng.model.config.n_local_heads))
        assert kvc.draft_cache.shape[2] == 128
        st, _ = harness.run_selfspec_batch(eng, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, False)
        res[layout] = (st.output.clone(), st.num_nodes.clone(), eng.cachelens.clone(),
                       [b.attention.kv_cache.kv_cache.clone() for b in eng.model.layers])
    for a, b in zip(res["NHD"][:3], res["HND"][:3]):
        assert torch.equal(a, b)
    for a, b in zip(res["NHD"][3], res["HND"][3]):
        assert torch.equal(a.permute(0, 1, 3, 2, 4), b)
    # argument plumbing of the C-ABI flag
    c = torch.zeros(2, 2, 4, 128, 64, dtype=torch.bfloat16)
    assert ops._kv_geom(c, "HND") == (128, 4, 64) and ops._kv_geom(c.permute(0, 1, 3, 2, 4).contiguous(), "NHD") == (128, 4, 64)
    assert ops._kv_args(c, None, "HND")[0] == ops.MD_KV_BF16 | ops.MD_KV_LAYOUT_HND == 0x100
    assert ops._kv_args(c, None)[0] == ops.MD_KV_BF16
    with pytest.raises(ValueError):
        ops._kv_geom(c, "DNH")
    with pytest.raises(NotImplementedError):
        from magicdec_amd.Engine.StreamingLLM.backend_draft import LMBackend_Draft
        d = LMBackend_Draft(dtype=torch.bfloat16, device="cpu")
        d.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        d.model.setup_caches(num_pages=4, streaming=True, draft_budget=gc.BUDGET, kv_layout="HND")


def test_c_abi_rejects_unknown_kv_dtype_flags():
    """kv_dtype carries the storage type in its low byte and MD_KV_LAYOUT_HND (0x100) as a flag; anything else is
    refused before a kernel is launched (host pointers are enough to reach the check)."""
    import ctypes
    from magicdec_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(1 << 12, dtype=torch.uint8)
    p = ctypes.c_void_p(buf.data_ptr())
    assert buf.data_ptr() % 16 == 0
    for bad in (0x200, 0x102 | 0x400, 2, 0x102):
        rc = lib.md_append_paged_kv(p, p, 64, 64, p, p, p, p, p, 1, 1, 1, 64, 128, bad, p, p, None)
        assert rc < 0, hex(bad)
        assert "md_append_paged_kv" in lib.md_last_error_string().decode()
        rc = lib.md_paged_attn(p, 64, p, p, p, p, p, p, 1, 1, 1, 1, 64, 128, 1, 0.125, 1, bad, p, p, p, 256, None)
        assert rc < 0 and "md_paged_attn" in lib.md_last_error_string().decode(), hex(bad)


def test_default_kv_layout_env(monkeypatch, cpu_ops_patched, ckpt_dir):
    """setup_caches() without kv_layout: HND (round 3: the layout bench.py measures is the Engine API's default), or
    what MAGICDEC_KV_LAYOUT says (NHD = the reference's flashinfer layout)."""
    from magicdec_amd.Engine import backend_core
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    monkeypatch.setenv("MAGICDEC_KV_LAYOUT", "nhd")
    assert backend_core.default_kv_layout() == "NHD"
    monkeypatch.delenv("MAGICDEC_KV_LAYOUT", raising=False)
    assert backend_core.default_kv_layout() == "HND"
    eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1, draft_dec_len=1)
    eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
    assert eng.model.layers[0].attention.kv_cache.layout == "HND"
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET, kv_layout="NHD")
    assert eng.model.layers[0].attention.kv_cache.layout == "NHD"
    monkeypatch.setenv("MAGICDEC_KV_LAYOUT", "DNH")
    with pytest.raises(ValueError):
        backend_core.default_kv_layout()


def test_scripts_without_a_layout_flag_build_the_cache_layout_bench_measures(monkeypatch):
    """tests/SnapKV/longspec_benchmark.py with no --kv_layout must run
    the configuration bench.py reports -- every script parser leaves the layout to the Engine default
    (backend_core.default_kv_layout() = what bench.py's --kv-layout defaults to), and --kv_layout NHD stays selectable."""
    import bench
    from magicdec_amd import cli
    from magicdec_amd.Engine import backend_core
    monkeypatch.delenv("MAGICDEC_KV_LAYOUT", raising=False)
    bench_default = bench.parse([]).kv_layout
    assert bench_default == backend_core.default_kv_layout() == "HND"
    for parser in (cli.longspec_parser("SnapKV"), cli.longspec_parser("StreamingLLM"), cli.selfspec_parser("SnapKV"),
                   cli.selfspec_parser("StreamingLLM"), cli.baseline_parser()):
        req = [a for a in parser._actions if a.required]
        argv = []
        for a in req:
            argv += [a.option_strings[0], "0"]
        args = parser.parse_args(argv)
        assert args.kv_layout is None                     # -> setup_caches(kv_layout=None) -> default_kv_layout()
        layout = backend_core.default_kv_layout() if args.kv_layout is None else args.kv_layout
        assert layout == bench_default
        assert parser.parse_args(argv + ["--kv_layout", "NHD"]).kv_layout == "NHD"


@pytest.mark.parametrize("tag", ["benchflag_snapkv_self", "benchflag_longspec_snapkv", "benchflag_longspec_stream",
                                 "benchflag_stream_self"])
def test_benchmark_flag_runs_then_undoes_the_length_updates_like_the_reference(tag, cpu_ops_patched, ckpt_dir):
    """`benchmark=True` of every Engine method ("run, then undo the cache-length / page-table updates",
    Engine/SnapKV/backend.py:140-143, backend_draft.py:139-142 and the StreamingLLM twins) -- no reference script passes
    it, so the fixtures were made by driving the REAL reference engines with a fixed program of calls (normal and
    benchmark calls interleaved, one- and two-token draft steps with cachelen_update); the product's engines run the
    same program: tokens and every length / page-table vector after each call are identical."""
    import importlib
    j = gc.load_json(f"{tag}.json")
    if j["snapkv_topk"]:      # the reference's resolution of torch.topk ties (row ORDER in the draft cache)
        cpu_ops.TOPK_REPLAY.update(table=j["snapkv_topk"], pos=0)
    engines = {}
    for key, spec in j["engines"].items():
        cls = getattr(importlib.import_module("magicdec_amd." + spec["module"]), spec["cls"])
        e = cls(dtype=torch.bfloat16, device="cpu", **spec["ctor"])
        e.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        e.setup_caches(max_batch_size=gc.B, **spec["caches"])
        engines[key] = e
    attrs = ["cachelens", "paged_kv_last_page_len", "paged_kv_indptr", "draft_cachelens", "draft_paged_kv_last_page_len",
             "draft_paged_kv_indptr"]
    n_bench = 0
    for i, (step, want) in enumerate(zip(j["program"], j["trace"])):
        e = engines[step["key"]]
        kw = dict(step["kwargs"])
        if "cachelen_update" in kw:
            kw["cachelen_update"] = torch.tensor(kw["cachelen_update"])
        r = getattr(e, step["fn"])(gc.benchflag_inputs(step["ncols"], i), **kw)
        got = r.tolist() if r.shape[1] <= 8 else r[:, -1:].tolist()
        assert got == want["out"], (i, step, got, want["out"])
        for at in attrs:
            if at in want:
                assert getattr(e, at).tolist() == want[at], (i, step, at, getattr(e, at).tolist(), want[at])
        n_bench += int(want["benchmark"])
    assert n_bench >= 2


def test_baseline_configs0_llama68m_batch1_matches_reference_trace(cpu_ops_patched):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case): baseline_benchmark.py on the reference's "68m"
    table entry (MHA: 12 heads, dim 768, vocab 32000), B = 1, prefix_len 129, max_len 256, seeded random weights.  The
    product's autoregressive loop reproduces the REAL reference's trace (oracle/gen_golden.py run_baseline_68m_b1):
    every encode / inference call's token and cache lengths over 7 sequences, and the final output."""
    from magicdec_amd import harness
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    from oracle.magicdec_ref import RefConfig, init_state_dict
    j = gc.load_json("run_baseline_68m_b1.json")
    cfg = RefConfig(n_layer=2, n_head=12, n_local_heads=12, dim=768, intermediate_size=3072, vocab_size=32000)
    d = Path(tempfile.mkdtemp(prefix="md_68m_")) / "llama-68m"          # the directory name selects the "68m" entry
    d.mkdir(parents=True)
    torch.save(init_state_dict(cfg, 68, wo_scale=0.1), d / "model.pth")
    eng = LMBackend(dtype=torch.bfloat16, device="cpu")
    eng.load_model(d / "model.pth", use_tp=False)
    c = eng.model.config
    assert (c.n_head, c.n_local_heads, c.dim, c.vocab_size, c.head_dim) == (12, 12, 768, 32000, 64)
    eng.setup_caches(max_batch_size=1, max_seq_length=256)
    g = torch.Generator().manual_seed(123)
    ids = torch.randint(4, 32000, (7, 129), generator=g)
    ids[:, 0] = 1
    log = []
    te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
    out = None
    for b in range(7):
        out, _, _ = harness.run_baseline_batch(te, ids[b:b + 1], 256, gc.EOT_1, gc.EOT_2)
    _compare(log, j["trace"])
    assert out.tolist() == j["final"]["output"]


@pytest.mark.parametrize("kind", ["longspec_snapkv_b1", "selfspec_stream_b1"])
def test_batch_size_one_matches_reference_trace(kind, cpu_ops_patched, ckpt_dir):
    """B = 1, the default batch size of the reference's scripts (their loop bodies' squeeze()s yield 0-d tensors there,
    tests/SnapKV/longspec_benchmark.py:273-279): longspec with a SnapKV draft (gamma = 1: accept_nums in {1, 2}, two-token
    draft steps) and StreamingLLM self-speculation (gamma = 3) over 8 sequences equal the REAL reference's traces."""
    from magicdec_amd import harness
    j = gc.load_json(f"run_{kind}.json")
    gamma = int(j["argv"][j["argv"].index("--gamma") + 1])
    g = torch.Generator().manual_seed(123)
    ids = torch.randint(4, 2048, (8, gc.S), generator=g)
    ids[:, 0] = 1
    log = []
    last = None
    if kind.startswith("longspec"):
        from magicdec_amd.Engine.SnapKV.backend import LMBackend
        from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
        eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gamma + 1)
        eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        eng.setup_caches(max_batch_size=1, max_seq_length=gc.MAX_LEN)
        drf = LMBackend_Draft(dtype=torch.bfloat16, device="cpu", draft_budget=gc.BUDGET)
        drf.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        drf.setup_caches(max_batch_size=1, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
        cpu_ops.TOPK_REPLAY.update(table=j["snapkv_topk"], pos=0)
        te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
        td = Tracer(drf, "SnapKV.LMBackend_Draft", log, ("encode", "inference"))
        for b in range(8):
            last, _ = harness.run_longspec_batch(te, td, ids[b:b + 1], gamma, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
    else:
        from magicdec_amd.Engine.StreamingLLM.backend import LMBackend
        eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gamma + 1)
        eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        eng.setup_caches(max_batch_size=1, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
        te = Tracer(eng, "StreamingLLM.LMBackend", log, ("encode", "draft_encode", "speculate", "verify"))
        for b in range(8):
            last, _ = harness.run_selfspec_batch(te, ids[b:b + 1], gamma, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, True)
    _compare(log, j["trace"])
    assert last.output.tolist() == j["final"]["output"]
    assert last.num_nodes.tolist() == j["final"]["num_nodes"]
    assert sum(1 for r in j["trace"] if "cachelen_update" in r) > 50


def test_product_apply_tp_shards_equal_the_reference(monkeypatch, ckpt_dir):
    """magicdec_amd.Engine.tp.apply_tp against the reference's (tests/golden/tp_shapes_kh4.json: four kv heads over 2,
    3 -- uneven -- and 4 ranks): every parameter of every rank has the reference's shape and sum, the config is
    rewritten to the same local head counts / dim."""
    import torch.distributed as dist
    from magicdec_amd.Engine import model_core, tp
    j = gc.load_json("tp_shapes_kh4.json")
    gc.register_tiny(model_core)
    _, sd = gc.tiny("tinykh4")
    for rec in j["shard"]:
        world, r = rec["world"], rec["rank"]
        monkeypatch.setenv("LOCAL_RANK", str(r))
        monkeypatch.setenv("RANK", str(r))
        monkeypatch.setattr(dist, "get_world_size", lambda g=None, w=world: w)
        monkeypatch.setattr(dist, "get_rank", lambda g=None, rr=r: rr)
        model = model_core.Transformer.from_name("tinykh4")
        model.load_state_dict(sd, assign=True)
        tp.apply_tp(model, list(range(world)), group="G")
        c = model.config
        assert [c.n_head, c.n_local_heads, c.dim] == rec["cfg"], (world, r)
        got = model.state_dict()
        for name, shape in rec["shapes"].items():
            assert list(got[name].shape) == shape, (world, r, name, list(got[name].shape), shape)
            assert abs(float(got[name].float().sum()) - rec["sums"][name]) <= 1e-3 * (1 + abs(rec["sums"][name])), name


def test_model_zoo_config_table_and_name_lookup_equal_the_reference(capsys):
    """Every entry of the reference's transformer_configs resolved through ModelArgs (derived intermediate_size, head_dim,
    n_local_heads, RoPE scaling fields, qkv_bias), and ModelArgs.from_name's fuzzy lookup of 20 checkpoint directory
    names (incl. the two that match no entry and must fail the same way): tests/golden/model_configs.json, recorded
    from the reference's four model modules (which hold one identical table)."""
    from magicdec_amd.Engine.model_core import ModelArgs, transformer_configs
    j = gc.load_json("model_configs.json")
    test_only = set(gc.TINY) | {"llama-68m-gqa"}       # registered by other tests / bench.py in this process
    assert set(transformer_configs) - test_only == set(j["table"])
    for name, want in j["table"].items():
        a = ModelArgs.from_name(name)
        got = {f: getattr(a, f, None) for f in want}
        assert got == want, (name, {f: (got[f], want[f]) for f in want if got[f] != want[f]})
    n_err = 0
    for path, want in j["lookup"].items():
        if "error" in want:
            with pytest.raises(Exception) as ei:
                ModelArgs.from_name(path)
            assert type(ei.value).__name__ == want["error"], (path, type(ei.value).__name__, want["error"])
            n_err += 1
            continue
        a = ModelArgs.from_name(path)
        got = {f: getattr(a, f, None) for f in want}
        assert got == want, (path, {f: (got[f], want[f]) for f in want if got[f] != want[f]})
    assert n_err == 2 and len(j["lookup"]) == 20


class _StubTokenizer:
    """What oracle/gen_golden.py gave the reference scripts as AutoTokenizer (ids printed as decimal numbers)."""
    eos_token, eos_token_id, unk_token_id, bos_token_id, pad_token = "</s>", 2, 0, 1, None

    def decode(self, ids, **k):
        return " ".join(str(int(i)) for i in ids)

    def encode(self, s, **k):
        return [3]


@pytest.mark.parametrize("tag", ["cli_longspec_snapkv", "cli_longspec_stream", "cli_selfspec_snapkv", "cli_selfspec_stream",
                                 "cli_baseline"])
def test_entry_points_print_the_reference_scripts_report(tag, cpu_ops_patched, ckpt_dir, monkeypatch, capsys):
    """The five entry points as a user runs them (tests/SnapKV|StreamingLLM/{longspec,selfspec}_benchmark.py,
    tests/baseline_benchmark.py; --printoutput, --benchmark): the product's command line prints the REAL reference
    scripts' report line for line -- device, EOT ids, every decoded sequence, the per-batch summaries with their token /
    step counts and tokens-per-sentence average, the final line -- wall-clock numbers masked
    (oracle/gen_golden.py cli_*: the scripts run unmodified with a stub tokenizer and a seeded synthetic dataset)."""
    import re
    from torch.utils.data import TensorDataset
    from magicdec_amd import cli
    j = gc.load_json(f"{tag}.json")
    g = torch.Generator().manual_seed(123)
    ids = torch.randint(4, 2048, (7 * gc.B, gc.S), generator=g)
    ids[:, 0] = 1
    monkeypatch.setattr(cli, "load_tokenizer", lambda name: _StubTokenizer())
    monkeypatch.setattr(cli, "convert_pg19_dataset", lambda **kw: TensorDataset(ids))
    monkeypatch.setattr(cli, "_device", lambda: "cpu")
    if j["snapkv_topk"]:
        cpu_ops.TOPK_REPLAY.update(table=j["snapkv_topk"], pos=0)
    argv, models = [], list(j["models"])
    for a in j["args"]:
        argv.append(a)
        if a in ("--target", "--model"):
            argv.append(str(ckpt_dir / models.pop(0) / "model.pth"))
    if "longspec" in tag:
        cli.longspec_main("SnapKV" if "snapkv" in tag else "StreamingLLM", argv)
    elif "selfspec" in tag:
        cli.selfspec_main("SnapKV" if "snapkv" in tag else "StreamingLLM", argv)
    else:
        cli.baseline_main(argv)
    keep = ("Using device", "eot_1", "Sequence", "total time", "target time", "Final tokens per second", "Tokens per second")
    got = []
    for ln in capsys.readouterr().out.splitlines():
        ln = ln.rstrip()
        if not (ln.startswith(keep) or re.fullmatch(r"[0-9 ]+", ln)):
            continue
        ln = re.sub(r"(total time :|time per iter :|target time :|draft time :)[0-9.eE+-]+s", r"\1<t>s", ln)
        ln = re.sub(r"(avg latency: |verify loop : |Final tokens per second :|Tokens per second :)[0-9.eE+-]+", r"\1<t>", ln)
        got.append(ln)
    assert len(got) == len(j["stdout"]), (len(got), len(j["stdout"]), got[-3:], j["stdout"][-3:])
    for i, (a, b) in enumerate(zip(got, j["stdout"])):
        assert a == b, (i, a, b)


@pytest.mark.parametrize("kind", ["longspec_snapkv_eot", "selfspec_stream_eot", "baseline_eot"])
def test_eot_driven_termination_matches_reference_trace(kind, cpu_ops_patched, ckpt_dir):
    """Termination by end-of-text tokens (tests/SnapKV/longspec_benchmark.py:212-226,262-264; baseline_benchmark.py:88):
    the reference scripts were run with a tokenizer whose eos / unk ids (866, 1410) are tokens the tiny model emits, so
    batches end on an accepted EOT draft token, an EOT bonus token or an EOT baseline step; the product's loops, given
    the same ids, reproduce every Engine call and the final output."""
    from magicdec_amd import harness
    j = gc.load_json(f"run_{kind}.json")
    e1, e2 = 866, 1410
    log = []
    if kind.startswith("longspec"):
        from magicdec_amd.Engine.SnapKV.backend import LMBackend
        from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
        eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1)
        eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
        drf = LMBackend_Draft(dtype=torch.bfloat16, device="cpu", draft_budget=gc.BUDGET)
        drf.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        drf.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
        cpu_ops.TOPK_REPLAY.update(table=j["snapkv_topk"], pos=0)
        te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
        td = Tracer(drf, "SnapKV.LMBackend_Draft", log, ("encode", "inference"))
        last = None
        for ids in gc.synthetic_batches():
            last, _ = harness.run_longspec_batch(te, td, ids, gc.GAMMA, gc.MAX_LEN, e1, e2)
        final = dict(output=last.output.tolist(), num_nodes=last.num_nodes.tolist())
    elif kind.startswith("selfspec"):
        from magicdec_amd.Engine.StreamingLLM.backend import LMBackend
        eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1)
        eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
        te = Tracer(eng, "StreamingLLM.LMBackend", log, ("encode", "draft_encode", "speculate", "verify"))
        last = None
        for ids in gc.synthetic_batches():
            last, _ = harness.run_selfspec_batch(te, ids, gc.GAMMA, gc.MAX_LEN, e1, e2, True)
        final = dict(output=last.output.tolist(), num_nodes=last.num_nodes.tolist())
    else:
        from magicdec_amd.Engine.SnapKV.backend import LMBackend
        eng = LMBackend(dtype=torch.bfloat16, device="cpu")
        eng.load_model(ckpt_dir / "tinytgt" / "model.pth", use_tp=False)
        eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
        te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
        out = None
        for ids in gc.synthetic_batches():
            out, _, _ = harness.run_baseline_batch(te, ids, gc.MAX_LEN, e1, e2)
        final = dict(output=out.tolist())
    _compare(log, j["trace"])
    for k, v in j["final"].items():
        assert final[k] == v, k
    # the EOT ids really ended batches early: fewer Engine calls than the same run with the default ids
    assert len(j["trace"]) < len(gc.load_json(f"run_{kind[:-4]}.json")["trace"])


TP_QWEN_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["MD_ROOT"])
import torch.distributed as dist
from pathlib import Path
from tests import cpu_ops, golden_cfg as gc
from oracle import magicdec_ref as mr, harness_ref as hr
cpu_ops.install()
from magicdec_amd import harness
from magicdec_amd.Engine import model_core
from magicdec_amd.Engine.tp import init_dist
from magicdec_amd.Engine.SnapKV.backend import LMBackend
ck = Path(os.environ["MD_CKPT"])
gc.register_tiny(model_core)
rank, group = init_dist()
world = dist.get_world_size()
eng = LMBackend(dtype=torch.bfloat16, device="cpu", dec_len=gc.GAMMA + 1, draft_dec_len=1)
eng.load_model(ck / "tinyqwen" / "model.pth", use_tp=True, rank_group=list(range(world)), group=group)
eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
bias = eng.model.layers[0].attention.wqkv.bias
cfg, sd = gc.tiny("tinyqwen")
ssd, lcfg = mr.shard_state_dict(sd, cfg, rank, world)
ora = mr.RefEngine("snapkv_self", lcfg, ssd, gc.B, gc.MAX_LEN, gc.BUDGET, group=group, rank=rank, world=world)
res = dict(rank=rank, bias_shape=list(bias.shape), bias_equal=bool(torch.equal(bias, ssd["layers.0.attention.wqkv.bias"])),
           batches=[])
for ids in gc.synthetic_batches()[:2]:
    st, _ = harness.run_selfspec_batch(eng, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, False)
    ref = hr.selfspec_batch(ora, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, False)
    res["batches"].append(dict(output=st.output.tolist(), num_nodes=st.num_nodes.tolist(),
                               oracle_output=ref["output"].tolist(), oracle_num_nodes=ref["num_nodes"].tolist()))
json.dump(res, open(os.path.join(os.environ["MD_OUT"], f"rank{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_tensor_parallel_qkv_bias_model_shards_the_bias(ckpt_dir):
    """BASELINE configs[4] is a qkv-bias model (Qwen2.5-32B) at TP = 8 -- which the reference cannot run: Engine/tp.py
    slices wqkv.weight (and int8 scales) by kv-head range but leaves wqkv.bias whole, so F.linear raises
    ("The expanded size of the tensor (448) must match the existing size (896)": reproduced with the tiny Qwen-like
    model at TP = 2).  The product slices the bias with the weight's head ranges; with no reference run to compare
    with, the yardstick is the oracle sharded the same way over the same gloo all-reduce: identical tokens on both ranks."""
    import json
    out = tempfile.mkdtemp(prefix="md_tpq_")
    script = os.path.join(out, "worker.py")
    Path(script).write_text(TP_QWEN_WORKER)
    port = 29100 + (os.getpid() % 500)
    procs = []
    for r in range(2):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_ROOT=str(ROOT), MD_CKPT=str(ckpt_dir), MD_OUT=out,
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=1500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = [json.load(open(os.path.join(out, f"rank{r}.json"))) for r in range(2)]
    for g in got:
        assert g["bias_shape"] == [(5 + 2) * 64] and g["bias_equal"]          # one kv head per rank: 5 q heads + k + v
        for b in g["batches"]:
            assert b["output"] == b["oracle_output"] and b["num_nodes"] == b["oracle_num_nodes"]
    assert got[0]["batches"] == got[1]["batches"]


def test_int8_weight_only_end_to_end_matches_reference_trace(cpu_ops_patched, ckpt_dir):
    """Weight-only int8 end to end: quantise the tiny checkpoint (Engine/quantize.py), load `model_int8.pth` through the
    loader's "int8 in the path" switch (Engine/utils.py:201-205), decode with the baseline loop -- every Engine call's
    tokens equal the REAL reference doing the same with its own quantiser and loader (run_baseline_int8)."""
    from magicdec_amd import harness
    from magicdec_amd.Engine import model_core, quantize as Q
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    j = gc.load_json("run_baseline_int8.json")
    _, sd = gc.tiny("tinytgt")
    m = model_core.Transformer.from_name("tinytgt")
    m.load_state_dict(sd, assign=True)
    d = Path(tempfile.mkdtemp(prefix="md_int8_")) / "tinytgt"
    d.mkdir(parents=True)
    torch.save(Q.WeightOnlyInt8QuantHandler(m).create_quantized_state_dict(), d / "model_int8.pth")
    eng = LMBackend(dtype=torch.bfloat16, device="cpu")
    eng.load_model(d / "model_int8.pth", use_tp=False)
    assert isinstance(eng.model.layers[0].attention.wqkv, Q.WeightOnlyInt8Linear)
    assert eng.model.layers[0].feed_forward.w2.weight.dtype == torch.int8
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
    log = []
    te = Tracer(eng, "SnapKV.LMBackend", log, ("encode", "inference"))
    out = None
    for ids in gc.synthetic_batches():
        out, _, _ = harness.run_baseline_batch(te, ids, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
    _compare(log, j["trace"])
    assert out.tolist() == j["final"]["output"]
    # and it is not the bf16 model's trace
    assert [r["out"] for r in j["trace"]] != [r["out"] for r in gc.load_json("run_baseline.json")["trace"]]


def test_command_line_flags_and_defaults_equal_the_reference_scripts():
    """Every parser.add_argument(...) of the reference's five scripts (tests/golden/cli_args.json: read from their source
    with ast) exists in the product's parser for that script with the same type, nargs, action and DEFAULT -- the
    scripts differ (self-speculation defaults to B = 45, prefix 100000, gamma 7, budget 4097; StreamingLLM longspec to
    budget 1025; the baseline to B = 16, prefix 8065).  Deliberate differences: checkpoint paths are relative
    ("checkpoints/...", not "/scratch/models/..."), and the product adds --kv_dtype / --kv_layout (for the baseline
    --dataset / --benchmark, for the two-model scripts --replicate_draft)."""
    from magicdec_amd import cli
    j = gc.load_json("cli_args.json")
    parsers = {"tests/SnapKV/longspec_benchmark.py": cli.longspec_parser("SnapKV"),
               "tests/StreamingLLM/longspec_benchmark.py": cli.longspec_parser("StreamingLLM"),
               "tests/SnapKV/selfspec_benchmark.py": cli.selfspec_parser("SnapKV"),
               "tests/StreamingLLM/selfspec_benchmark.py": cli.selfspec_parser("StreamingLLM"),
               "tests/baseline_benchmark.py": cli.baseline_parser()}
    extras = {"--kv_dtype", "--kv_layout", "-h", "--help"}
    for script, ref_args in j.items():
        acts = {a.option_strings[0]: a for a in parsers[script]._actions if a.option_strings}
        for ra in ref_args:
            flag = ra["flags"][0]
            assert flag in acts, (script, flag)
            a = acts[flag]
            if ra.get("action") == "'store_true'":
                assert a.const is True and a.default is False and a.nargs == 0, (script, flag)
                continue
            assert a.type.__name__ == ra["type"], (script, flag, a.type, ra["type"])
            assert repr(a.nargs) == ra.get("nargs", "None"), (script, flag, a.nargs)
            if ra["type"] == "Path":
                assert a.default.name == "model.pth" and a.default.parent.name in ra["default"], (script, flag, a.default)
            elif "default" in ra:
                assert repr(a.default) == ra["default"], (script, flag, a.default, ra["default"])
            else:
                assert a.default is None, (script, flag, a.default)
        allowed = extras | ({"--dataset", "--benchmark"} if "baseline" in script else set()) \
            | ({"--replicate_draft"} if "longspec" in script else set())
        assert set(acts) - {r["flags"][0] for r in ref_args} <= allowed, (script, set(acts) - {r["flags"][0] for r in ref_args})


def test_gemm_policy_encodes_the_measured_ab_table():
    """Engine/gemm_policy.choose is a pure function of the shape: pin the decisions that profiles/r03_fused_ab.txt and
    profiles/r02_gemm_ab.txt measured (which kernel serves which linear of the BASELINE models), so that a threshold
    edit cannot silently move a production shape to a slower kernel."""
    from magicdec_amd.Engine import gemm_policy as g
    if g.mode() != "auto" or g.fused_mode() != "auto" or g.block_mode() != "auto":
        pytest.skip("MAGICDEC_GEMM / MAGICDEC_FUSED / MAGICDEC_BLOCK override the policy in this environment")
    c = lambda M, N, K, kind, norm=False: g.choose(M, N, K, kind == "swiglu", False, True, kind, norm)
    # 1B draft model at TP1 (M = 64; two-token step M = 128)
    assert c(64, 3072, 2048, "qkv") == "fused" and c(64, 2048, 2048, "resid") == "fused"
    # round 6 (profiles/r06_fused_pro22_ab.txt): 2 x 2 tiles re-read the weights once per PAIR of M tiles -- the 1B w1|w3 is
    # fused with and without the absorbed norm at 64 rows (18.4-18.9 / 20.3 us against md_linear's 22.8-23.2 / 24.8) and,
    # with the norm, at the 128 rows of the two-token step (32.0 against ~31 for library + rmsnorm + SiLU*mul launches:
    # a tie that frees the weight's row-major copy); without the norm 128 rows stay on the library
    assert c(64, 16384, 2048, "swiglu") == "fused" and c(64, 16384, 2048, "swiglu", True) == "fused"
    assert c(128, 16384, 2048, "swiglu", True) == "fused" and c(128, 16384, 2048, "swiglu") == "lib"
    assert c(64, 2048, 8192, "resid") == "lib" and c(128, 3072, 2048, "qkv") == "fused"
    assert c(64, 128256, 2048, "plain") == "skinny"
    # its TP4 shards and the 8B model's TP8 shards at M = 64: everything fused
    for N, K, kind in [(768, 2048, "qkv"), (2048, 512, "plain"), (4096, 2048, "swiglu"), (2048, 2048, "plain"),
                       (768, 4096, "qkv"), (4096, 512, "plain"), (3584, 4096, "swiglu"), (4096, 1792, "plain")]:
        assert c(64, N, K, kind) == "fused", (N, K, kind)
    # verify pass (M = 256): only the sharded qkv projection goes to the fused kernel; the 8B model's w2 streams
    assert c(256, 768, 4096, "qkv") == "fused" and c(256, 1536, 4096, "qkv") == "fused"
    assert c(256, 6144, 4096, "qkv") == "lib"
    # round 5 (profiles/r05_fused_tp8_shard_tiles.txt): two more TP8 shards of the verify pass -- w1|w3 and the K = 512 wo
    assert c(256, 4096, 512, "plain") == "fused" and c(256, 3584, 4096, "swiglu") == "fused"
    assert c(256, 7168, 4096, "swiglu") == "lib" and c(256, 4096, 1024, "plain") == "lib"       # the TP4 shards
    # round 4 (profiles/r04_block_ab_final.txt): the wide 129..256-row products and the K = 14336 down projection run
    # on the block-tile GEMM; the narrow projections and every TP shard stay where they were
    assert c(256, 28672, 4096, "swiglu") == "block" and c(256, 4096, 14336, "resid") == "block"
    assert c(256, 128256, 4096, "plain") == "block" and c(128, 28672, 4096, "swiglu") == "block"
    # round 6 (profiles/r06_fused_pro22_ab.txt, r06_ab_wo256_*): an output projection WITH its residual add whose 64 x 64 tile
    # groups fill the chip (the 8B wo: 4 x 64 = 256) is a measured tie between the library + add/norm launch and the tile
    # kernel (24.9-25.0 vs 25.2-25.9 us; in the trace 20.9 + 5.0 vs 23.3 + 5.0) and takes the tile kernel, which leaves no
    # weight of configs[2] in two layouts during decode; the collective-bound ("plain") shards and K > 4096 stay
    assert c(128, 4096, 14336, "resid") == "skinny" and c(256, 4096, 1792, "plain") == "lib"
    if os.environ.get("MAGICDEC_WO256", "1") != "0":
        assert c(256, 4096, 4096, "resid") == "fused" and c(256, 4096, 1792, "resid") == "fused"
    assert c(256, 4096, 4096, "plain") == "lib" and c(256, 2048, 8192, "resid") == "lib"
    # round 4 (profiles/r04_skinny_norm_ab.txt): md_linear absorbs a deferred norm up to 64 rows only
    if os.environ.get("MAGICDEC_SKINNY_NORM", "auto") == "auto":
        assert g.skinny_absorbs_norm(1) and g.skinny_absorbs_norm(32) and g.skinny_absorbs_norm(64)
        assert not g.skinny_absorbs_norm(65) and not g.skinny_absorbs_norm(128) and not g.skinny_absorbs_norm(256)
    # a streaming-layout copy only for weights some hand-written kernel can be chosen for
    assert g.want_packed(28672, 4096, True) and g.want_packed(4096, 14336) and g.want_packed(3072, 2048)
    assert not g.want_packed(1000, 16384) and not g.want_packed(4100, 4096)
    # round 6 (profiles/r06_split_ab.txt): the deep narrow output projection of a draft step -- the 1B w2 -- runs on the tile
    # kernel with K split over workgroups (its combine launch is the add + norm launch); not the 8B w2 (a long stream:
    # md_linear), not a verify-sized M, not a K <= 4096 projection (the single-launch fused kernel)
    if os.environ.get("MAGICDEC_SPLIT", "auto") == "auto":
        assert g.use_split(64, 2048, 8192, "resid") and g.use_split(128, 2048, 8192, "resid")
        assert not g.use_split(64, 4096, 14336, "resid") and not g.use_split(256, 2048, 8192, "resid")
        assert not g.use_split(64, 2048, 2048, "resid") and not g.use_split(64, 2048, 8192, "plain")
        assert g.want_packed(2048, 8192)
    # round 6 (profiles/r06_shard70b_ab.txt): the 70B model's TP-8 shards at the 128 rows of configs[3]'s verify -- w2 on
    # 2 x 2 tiles and the deep narrow wqkv go to the tile kernel; at 32 rows (40 workgroups) the qkv shard stays on the library
    assert c(128, 8192, 3584, "plain") == "fused" and c(128, 1280, 8192, "qkv") == "fused"
    assert c(32, 1280, 8192, "qkv") == "lib" and c(128, 7168, 8192, "swiglu") == "skinny"
    # ... and Qwen2.5-32B's (dim 5120) at the 128 rows of configs[4]'s draft steps (profiles/r06_shard_ab.txt): wqkv 896 x 5120
    # (112 tile workgroups) and w1|w3 6912 x 5120 on 2 x 2 tiles go to the tile kernel, w2 5120 x 3456 stays on the library;
    # the UNSHARDED 1B w1|w3 at 128 rows (K = 2048) is not touched by the deep-shard rule
    assert c(128, 896, 5120, "qkv") == "fused" and c(128, 6912, 5120, "swiglu") == "fused"
    assert c(128, 5120, 3456, "plain") == "lib" and c(128, 5120, 640, "plain") == "fused"
    assert c(128, 16384, 2048, "swiglu") == "lib" and c(512, 6912, 5120, "swiglu") == "lib"
    # autoregressive 8B steps (M = 64)
    assert c(64, 6144, 4096, "qkv") == "lib" and c(64, 4096, 4096, "resid") == "fused"
    assert c(64, 28672, 4096, "swiglu") == "skinny" and c(64, 4096, 14336, "resid") == "skinny"
    # never without a packed copy, never for int8 rows, never for shapes the kernel does not take
    assert g.choose(64, 3072, 2048, False, False, False, "qkv") == "lib"
    assert g.choose(64, 3072, 2048, False, True, True, "qkv") == "skinny"
    assert c(64, 3080, 2048, "plain") == "lib" and c(300, 768, 2048, "qkv") == "lib"


def test_iteration_hook_runs_after_the_body_and_before_the_host_read():
    """LoopState.before_host_read: device-side statistics are queued behind the accept kernel and in front of the
    iteration's one host read (bench.py counts its tokens there)."""
    from magicdec_amd import harness
    st = harness.new_state(2, 3, 8, "cpu")
    order = []
    st.before_host_read = lambda s: order.append(("hook", int(s.flags[0]), s.iters))

    def body(forced):
        st.flags[0] = 1
        order.append(("body", forced))

    term, nd = harness._iterate(None, None, st, body, None)
    assert order == [("body", None), ("hook", 1, 1)]
    assert term is True and nd is False and st.iters == 1
    st.before_host_read = None
    st.flags.zero_()
    assert harness._iterate(None, None, st, lambda f: None, None) == (False, False)


def test_one_resident_copy_per_weight_release_and_restore():
    """every weight a hand-written kernel may serve used to be held TWICE (streaming layout +
    row-major: 16.9 GB at configs[2]).  Now (a) only weights that some step of THIS engine runs on a hand-written kernel
    are packed (the back-end passes its decode row counts), (b) after prefill the row-major tensor of every weight that
    decode reads in the streaming layout only is released (Transformer.release_rowmajor; the Parameter objects stay), and
    (c) the next prefill gets them back bit-exactly from the streaming copy (restore_rowmajor).  Host test of the
    bookkeeping on CPU tensors (`_pack_weights(force=True)`; no kernel runs): the 1B draft's shapes end with ZERO
    duplicated bytes, an 8B layer keeps exactly its wo (library GEMM at 256 rows, fused kernel at 64)."""
    import torch
    from magicdec_amd.Engine import gemm_policy as g
    from magicdec_amd.Engine import model_core
    if g.mode() != "auto" or g.fused_mode() != "auto" or g.block_mode() != "auto" or os.environ.get("MAGICDEC_SPLIT", "auto") != "auto":
        pytest.skip("policy overridden by the environment")

    def build(name, rows, values=False, **cfg):
        model_core.transformer_configs[name] = dict(block_size=4096, n_layer=1, vocab_size=4096, rope_base=500000.0, **cfg)
        try:
            torch.manual_seed(0)
            m = model_core.Transformer.from_name(name).to(torch.bfloat16)
            if values:                       # only the restore check below compares values; the rest is bookkeeping
                for p_ in m.parameters():
                    p_.data.normal_(0, 0.02)
            m.setup_caches(num_pages=2, decode_rows=rows)
            m._pack_weights(force=True)
            return m
        finally:
            model_core.transformer_configs.pop(name, None)

    # ---- the 1B draft's layer shapes, decode rows B = 64 and the two-token step
    m = build("dedupe1b", (64, 128), values=True, n_head=32, n_local_heads=8, dim=2048, intermediate_size=8192)
    lay = m.layers[0]
    ws = {"wqkv": lay.attention.wqkv.weight, "wo": lay.attention.wo.weight, "w13": m._w13[0],
          "w2": lay.feed_forward.w2.weight, "head": m.output.weight}
    assert all(id(w) in m._packed for w in ws.values())
    before = {k: w.detach().clone() for k, w in ws.items()}
    w1_before = lay.feed_forward.w1.weight.detach().clone()
    total = sum(p.data.numel() * 2 for p in m._packed.values())
    assert m.packed_bytes == total
    freed = m.release_rowmajor()
    assert m.packed_bytes == 0 and freed == total and len(m._released) == 5       # nothing is held twice in decode
    for k, w in ws.items():
        assert w.shape == before[k].shape and w.stride() == (0, 0) and id(w) in m._packed, k     # same object, no storage
    assert lay.feed_forward.w1.weight.stride() == (0, 0)
    # a library call on a released weight (a row count nobody announced) still computes the right thing
    x = torch.randn(3, 2048).to(torch.bfloat16)
    assert torch.equal(torch.nn.functional.linear(x, m._resident(ws["wo"])), torch.nn.functional.linear(x, before["wo"]))
    m.restore_rowmajor()
    assert not m._released and m.packed_bytes == total
    for k, w in ws.items():
        assert torch.equal(w.detach(), before[k]), k
    assert torch.equal(lay.feed_forward.w1.weight.detach(), w1_before)
    assert lay.feed_forward.w1.weight.data_ptr() == m._w13[0].data_ptr()            # w1 / w3 are views of w13 again
    assert m.release_rowmajor() == total and m.release_rowmajor() == 0              # idempotent

    # ---- an 8B target layer: 64-row autoregressive steps, 256-row verify (a longspec target has no two-token step)
    m = build("dedupe8b", (64, 256), n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336)
    lay = m.layers[0]
    assert id(lay.attention.wqkv.weight) not in m._packed       # library GEMM at 64 and 256 rows: never packed
    assert id(lay.attention.wo.weight) in m._packed and id(m._w13[0]) in m._packed
    m.release_rowmajor()
    assert id(m._w13[0]) in m._released and id(lay.feed_forward.w2.weight) in m._released
    # what decode reads in BOTH layouts stays held twice: only this test's 4096-row head (library at 256 rows; the real
    # 128 256-row head runs on md_linear / md_linear_block at every row count).  wo: fused kernel at 64 AND at 256 rows
    assert set(m._packed) - m._released == {id(m.output.weight)} and m.packed_bytes == 4096 * 4096 * 2
    # with the 128 rows of a two-token step announced, wo runs on the library there and must keep its row-major tensor
    m3 = build("dedupe8b_128", (64, 128, 256), n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336)
    m3.release_rowmajor()
    assert id(m3.layers[0].attention.wo.weight) not in m3._released
    # without announced row counts nothing is released (any row count may come)
    m2 = build("dedupe_any", None, n_head=32, n_local_heads=8, dim=2048, intermediate_size=8192)
    assert m2.release_rowmajor() == 0 and m2.packed_bytes > 0
