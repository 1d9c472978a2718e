"""Child process of bench.py's per-collective report (one per rank, started by bench.collective_microbench_isolated).

The xGMI all-reduce kernels (csrc/allreduce.hip) map peer memory through HIP IPC and spin on flags in it; they had
never run over real links when the first multi-GPU bench was launched.  Measuring them inside the benchmark's own
processes would put the headline JSON line at the mercy of that code (a GPU fault aborts the process, the line is
lost), so the measurement runs here: same ranks, same GPUs, its own rendezvous port and its own RCCL communicator, while
the parents wait on the host.  Rank 0 writes the result (or the error) as JSON to --out.

    RANK=r WORLD_SIZE=n LOCAL_RANK=r MASTER_ADDR=127.0.0.1 MASTER_PORT=p \
        python tools/collective_bench.py --shapes verify:256:4096,autoregressive:64:4096 --iters 30 --out /tmp/x.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", required=True, help="name:rows:dim[,name:rows:dim...]")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--out", required=True)
    ap.add_argument("--dry", action="store_true", help="rendezvous + barrier only (CPU test of the orchestration)")
    a = ap.parse_args()
    shapes = [(n, int(r), int(d)) for n, r, d in (s.split(":") for s in a.shapes.split(","))]
    import torch
    import torch.distributed as dist
    from magicdec_amd.Engine.tp import init_dist
    rank, group = init_dist()
    res = None
    try:
        if a.dry:
            dist.barrier()
            res = {"dry": True, "world": dist.get_world_size(), "shapes": [s[0] for s in shapes]}
        else:
            import bench
            dev = f"cuda:{torch.cuda.current_device()}"
            res = bench.collective_microbench(group, shapes, dev, iters=a.iters)
    except Exception as e:  # noqa: BLE001 -- reported to the parent, which reports it in the JSON line
        res = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        tmp = a.out + ".tmp"
        with open(tmp, "w") as f:
            json.dump(res, f)
        os.replace(tmp, a.out)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001 -- the result is already on disk
        pass


if __name__ == "__main__":
    main()
