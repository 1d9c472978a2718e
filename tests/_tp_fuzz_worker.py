"""Worker of tests/test_gpu_engine_fuzz.py::test_fuzz_tensor_parallel_free_running_loop: one of WORLD_SIZE tensor-parallel
ranks, all on GPU 0 (MAGICDEC_TP_SINGLE_GPU=1: gloo as bootstrap transport, the per-layer all-reduces through the one-shot
IPC kernel with MAGICDEC_ONESHOT_AR=1), on the random geometry of seed MD_SEED.  Runs the product's free-running loop and
writes its final state as JSON."""
import json
import os
import sys
import tempfile

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_cfg as gc  # noqa: E402
from tests.test_gpu_engine_fuzz import _register, draw, peaked  # noqa: E402


def main():
    from magicdec_amd import harness
    from magicdec_amd.Engine.tp import init_dist
    seed, world = int(os.environ["MD_SEED"]), int(os.environ["WORLD_SIZE"])
    c = draw(seed, tp=world)
    mode, cfg_t, cfg_d, B, S, max_len, gamma, budget = (c[k] for k in ("mode", "cfg_t", "cfg_d", "B", "S", "max_len", "gamma",
                                                                      "budget"))
    sd_t, sd_d = peaked(cfg_t, cfg_d, c["wseed"], c["miss_every"])
    tmp = tempfile.mkdtemp(prefix="md_tpfuzz_")
    ck_t = _register(tmp, f"tpfuzz{seed}t", cfg_t, sd_t)
    ck_d = _register(tmp, f"tpfuzz{seed}d", cfg_d, sd_d)
    ranks = list(range(world))
    rank, group, dgroup = init_dist(ranks)
    dev = "cuda:0"
    g = torch.Generator().manual_seed(c["wseed"] + 3)
    ids = torch.randint(4, cfg_t.vocab_size, (B, S), generator=g)
    ids[:, 0] = 1
    ids = ids.to(dev)
    tp = dict(use_tp=True, rank_group=ranks)
    if mode.startswith("longspec"):
        from magicdec_amd.Engine.SnapKV.backend import LMBackend
        eng = LMBackend(dtype=torch.bfloat16, device=dev, dec_len=gamma + 1)
        eng.load_model(ck_t, group=group, **tp)
        eng.setup_caches(max_batch_size=B, max_seq_length=max_len)
        if mode.endswith("snapkv"):
            from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
            drf = LMBackend_Draft(dtype=torch.bfloat16, device=dev, draft_budget=budget)
            drf.load_model(ck_d, group=dgroup, **tp)
            drf.setup_caches(max_batch_size=B, max_seq_length=max_len, draft_budget=budget)
        else:
            from magicdec_amd.Engine.StreamingLLM.backend_draft import LMBackend_Draft
            drf = LMBackend_Draft(dtype=torch.bfloat16, device=dev)
            drf.load_model(ck_d, group=dgroup, **tp)
            drf.setup_caches(max_batch_size=B, draft_budget=budget)
        st, _ = harness.run_longspec_batch(eng, drf, ids, gamma, max_len, gc.EOT_1, gc.EOT_2, barrier=dist.barrier)
        ars = [eng.model, drf.model]
    else:
        streaming = mode.endswith("stream")
        if streaming:
            from magicdec_amd.Engine.StreamingLLM.backend import LMBackend
            eng = LMBackend(dtype=torch.bfloat16, device=dev, dec_len=gamma + 1)
        else:
            from magicdec_amd.Engine.SnapKV.backend import LMBackend
            eng = LMBackend(dtype=torch.bfloat16, device=dev, dec_len=gamma + 1, draft_dec_len=1)
        eng.load_model(ck_t, group=group, **tp)
        eng.setup_caches(max_batch_size=B, max_seq_length=max_len, draft_budget=budget)
        st, _ = harness.run_selfspec_batch(eng, ids, gamma, max_len, gc.EOT_1, gc.EOT_2, streaming)
        ars = [eng.model]
    res = dict(rank=rank, output=st.output.cpu().tolist(), num_nodes=st.num_nodes.cpu().tolist(), iters=st.iters,
               local_heads=[eng.model.config.n_head, eng.model.config.n_local_heads],
               oneshot=[getattr(m, "_oneshot", None) is not None for m in ars],
               ar_status=[m._oneshot.status() if getattr(m, "_oneshot", None) is not None else None for m in ars])
    json.dump(res, open(os.path.join(os.environ["MD_OUT"], f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: OK")


if __name__ == "__main__":
    main()
