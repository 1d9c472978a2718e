"""Worker of tests/test_gpu_engine.py::test_tp2_on_one_gpu: one of two TP ranks, both on GPU 0
(MAGICDEC_TP_SINGLE_GPU=1), HIP kernels on KV-head shards, per-layer all-reduces through the one-shot IPC kernel
(MAGICDEC_ONESHOT_AR=1), everything else over gloo.  Writes its final state as JSON."""
import json
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_cfg as gc  # noqa: E402


def main():
    from magicdec_amd import harness
    from magicdec_amd.Engine import model_core
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
    from magicdec_amd.Engine.tp import init_dist
    ck = Path(os.environ["MD_CKPT"])
    gc.register_tiny(model_core)
    rank, group, dgroup = init_dist([0, 1])
    dev = "cuda:0"
    use_graphs = os.environ.get("MD_GRAPHS", "0") == "1"
    eng = LMBackend(dtype=torch.bfloat16, device=dev, dec_len=gc.GAMMA + 1)
    eng.load_model(ck / "tinytgt" / "model.pth", use_tp=True, rank_group=[0, 1], group=group)
    eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
    drf = LMBackend_Draft(dtype=torch.bfloat16, device=dev, draft_budget=gc.BUDGET)
    drf.load_model(ck / "tinytgt" / "model.pth", use_tp=True, rank_group=[0, 1], group=dgroup)
    drf.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
    if use_graphs:
        eng.compile()
        drf.compile()
    assert getattr(eng.model, "_oneshot", None) is not None, "one-shot all-reduce not attached"
    ids = gc.synthetic_batches()[0].to(dev)
    # teacher-forced probe: prefill + one (gamma+1)-token target step; this rank's vocab shard of the logits
    eng.encode(ids)
    eng.inference(ids[:, :gc.GAMMA + 1].clone())
    torch.save(eng.model._last_logits.float().cpu(), os.path.join(os.environ["MD_OUT"], f"logits_rank{rank}.pt"))
    st, _ = harness.run_longspec_batch(eng, drf, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, barrier=dist.barrier)
    res = dict(rank=rank, output=st.output.cpu().tolist(), num_nodes=st.num_nodes.cpu().tolist(), iters=st.iters,
               cachelens=eng.cachelens.cpu().tolist(),
               local_heads=[eng.model.config.n_head, eng.model.config.n_local_heads],
               ar_status=[eng.model._oneshot.status(), drf.model._oneshot.status()])
    # bench.py's per-collective report (run by the driver's N > 1 benches): RCCL path here = gloo, xGMI kernels = IPC
    import bench
    coll = bench.collective_microbench(group, [("probe", 8, 512)], dev, iters=3)
    res["collectives"] = coll
    json.dump(res, open(os.path.join(os.environ["MD_OUT"], f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: OK")


if __name__ == "__main__":
    main()
