"""GPU test of the one-shot all-reduce (C1, csrc/allreduce.hip) with 2 and 3 processes sharing the box's single GPU.
Bar: bit-exact against "sum in rank order, fp32 accumulate, round once" (integer-exact determinism is what the
replicated page tables need), eager and under hipGraph replay; no peer time-outs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,publish", [(2, "wt"), (3, "wt"), (2, "fence"), (3, "fence")])
def test_oneshot_allreduce_multiprocess_one_gpu(world, publish):
    """publish = "fence": the release-fence arm of the hand-off (md_ar_set_publish / MAGICDEC_AR_PUBLISH=fence, the fallback
    a multi-GPU run selects when the write-through publish fails on real links) through the SAME worker: every size,
    algorithm, fused form and graph replay bit-exact."""
    port = 29640 + world + (10 if publish == "fence" else 0)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", MAGICDEC_AR_PUBLISH=publish)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ar_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r}: OK" in out, f"rank {r} failed:\n{out[-3000:]}"


def test_mid_run_peer_loss_raises_at_that_iteration():
    """A peer that stops calling the collective in the middle of a run (tests/_ar_timeout_worker.py): the surviving rank
    raises AllReduceTimeout from harness._read_flags in the iteration it happened in -- after the kernel's 2 s bounded
    spin, with NaN-poisoned output -- instead of hanging or iterating on garbage."""
    world, port = 2, 29655
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ar_timeout_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=180)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r}: OK" in out, f"rank {r} failed:\n{out[-3000:]}"
    assert "AllReduceTimeout raised in iteration 5" in outs[0], outs[0][-2000:]
    from tests.conftest import parity_report
    parity_report("[allreduce] mid-run peer loss: " + [l for l in outs[0].splitlines() if "AllReduceTimeout raised" in l][0])


def test_oneshot_allreduce_single_rank_is_identity():
    import torch
    import torch.distributed as dist
    from magicdec_amd.Engine.oneshot import OneShotAllReduce
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29639"
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ar = OneShotAllReduce(dist.group.WORLD, max_bytes=1 << 20)
        x = torch.randn(4096, device="cuda").to(torch.bfloat16)
        y = x.clone()
        ar.all_reduce_(y)
        assert torch.equal(x, y) and ar.status() == 0
        ar.close()
    finally:
        dist.destroy_process_group()
