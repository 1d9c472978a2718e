"""Worker of tests/test_gpu_allreduce.py::test_mid_run_peer_loss_raises_at_that_iteration: WORLD_SIZE = 2 processes on
GPU 0.  Both run `good` iterations of (xGMI all-reduce -> the iteration's host read, harness._read_flags); in the next
iteration rank 1 SKIPS its md_allreduce call -- a peer that went missing in the middle of a run.  Rank 0's kernel must
give up after its bounded spin (2 s), poison its output with NaN, and harness._read_flags must raise AllReduceTimeout
in THAT iteration (the reference has only dist.barrier(), Engine/tp.py:54-64)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from magicdec_amd import harness
    from magicdec_amd.Engine.oneshot import AllReduceTimeout, OneShotAllReduce
    dev = "cuda:0"
    ar = OneShotAllReduce(dist.group.WORLD, max_bytes=4 << 20)
    st = harness.new_state(4, 3, 64, dev)
    good, n = 5, 64 * 2048
    for it in range(good + 1):
        st.iters = it
        x = torch.full((n,), float(rank + 1), dtype=torch.bfloat16, device=dev)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.time()
        if it == good and rank == 1:
            break                                   # the missing peer: no call, no flag for rank 0
        ar.all_reduce_(x)
        try:
            harness._read_flags(st, (ar,))
        except AllReduceTimeout as e:
            dt = time.time() - t0
            assert rank == 0 and it == good, (rank, it, "time-out in a healthy iteration")
            assert f"iteration {good}" in str(e), str(e)
            assert 1.0 < dt < 6.0, f"gave up after {dt:.2f} s (spin bound 2 s)"
            assert bool(torch.isnan(x.float()).any()), "rows that could not be completed must be NaN-poisoned"
            assert ar.status() != 0
            print(f"rank 0: AllReduceTimeout raised in iteration {it} after {dt:.2f} s, "
                  f"{int(torch.isnan(x.float()).sum())} of {n} elements NaN", flush=True)
            break
        assert it < good, "rank 0 did not notice the missing peer"
        assert bool((x.float() == 3.0).all()), (it, x[:4])
    dist.barrier()              # rank 1 waits here (host side) while rank 0's kernel spins
    print(f"rank {rank}: OK", flush=True)
    sys.stdout.flush()
    os._exit(0)                 # the communicators' sequence numbers disagree now: no orderly teardown of the IPC state


if __name__ == "__main__":
    main()
