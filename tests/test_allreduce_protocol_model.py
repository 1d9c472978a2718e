"""Model check (randomised schedules) of the synchronisation protocol of csrc/allreduce.hip -- CPU only.

The kernel's correctness argument (double-buffered data, one monotonic flag per (block, source rank), no closing
barrier, stream order between calls) is easy to get subtly wrong and cannot be exercised across real GPUs on the
development box, so the protocol is restated here as interleaved state machines and run under thousands of random
schedules.  Every rank r executes a sequence of calls; per call every launched block b

    1. f = ++flag[r][b]; half = f & 1
    2. writes its slice of the input into data[r][half]        (one micro-step per element)
    3. stores f into start[p][b][r] of every peer p
    4. waits until start[r][b][p] >= f for every p
    5. reads its slice from data[p][half] of every rank p       (one micro-step per element per rank)

and a rank starts call k+1 only when all blocks of call k have finished (kernels of one stream run in order).
Invariant: every element read in call k was written in call k.  The model also shows WHY the launcher always starts
the full grid: with a message-size-dependent grid the per-block counters drift apart and a small call overwrites a
region a slower peer is still reading (found by this model, see test_size_dependent_grid_is_unsafe)."""
import random

import pytest

MAXB = 4          # blocks of the model (kMaxBlocks = 64 in the kernel)


def simulate(n_ranks, sizes, seed, full_grid):
    """sizes: elements of each call (same on all ranks).  Returns None, or a description of the first violation."""
    rng = random.Random(seed)
    cap = max(sizes)
    data = [[[None] * cap, [None] * cap] for _ in range(n_ranks)]            # data[r][half][i] = call id written
    start = [[[0] * n_ranks for _ in range(MAXB)] for _ in range(n_ranks)]   # start[owner][b][src]
    flag = [[0] * MAXB for _ in range(n_ranks)]

    def blocks_of(n):
        # size-dependent variant: ~2 elements per block up to the block cap, like the first version of the launcher
        # (ceil(n / 1024 vectors) capped at 64): once the cap is hit the per-block regions grow and overlap the
        # regions other blocks use in smaller calls
        return MAXB if full_grid else max(1, min(MAXB, (n + 1) // 2))

    class Block:
        def __init__(self, r, k, b, nb):
            self.r, self.k, self.b = r, k, b
            n = sizes[k]
            per = (n + nb - 1) // nb
            self.lo, self.hi = b * per, min(n, (b + 1) * per)
            self.pc, self.i, self.p = 0, self.lo, 0
            self.f = self.half = None
            self.done = False

        def runnable(self):
            if self.pc == 3:      # spinning on the peers' flags
                return all(start[self.r][self.b][p] >= self.f for p in range(n_ranks))
            return True

        def step(self):
            r, b = self.r, self.b
            if self.pc == 0:
                flag[r][b] += 1
                self.f, self.half = flag[r][b], flag[r][b] & 1
                self.pc = 1 if self.lo < self.hi else 2
            elif self.pc == 1:                                   # copy one element of my slice
                data[r][self.half][self.i] = self.k
                self.i += 1
                if self.i == self.hi:
                    self.pc = 2
            elif self.pc == 2:                                   # raise my flag at one peer per micro-step
                start[self.p][b][r] = self.f
                self.p += 1
                if self.p == n_ranks:
                    self.pc, self.p, self.i = 3, 0, self.lo
            elif self.pc == 3:
                self.pc = 4 if self.lo < self.hi else 5
            elif self.pc == 4:                                   # read one element of one rank's buffer
                got = data[self.p][self.half][self.i]
                if got != self.k:
                    return (f"rank {r} call {self.k} block {b}: read element {self.i} of rank {self.p} half "
                            f"{self.half} written by call {got}")
                self.i += 1
                if self.i == self.hi:
                    self.i, self.p = self.lo, self.p + 1
                    if self.p == n_ranks:
                        self.pc = 5
            if self.pc == 5:
                self.done = True
            return None

    call = [0] * n_ranks
    live = [[] for _ in range(n_ranks)]
    for r in range(n_ranks):
        nb = blocks_of(sizes[0])
        live[r] = [Block(r, 0, b, nb) for b in range(nb)]
    steps = 0
    while any(call[r] < len(sizes) for r in range(n_ranks)):
        cands = [blk for r in range(n_ranks) for blk in live[r] if not blk.done and blk.runnable()]
        assert cands, "deadlock"
        # bias the schedule: now and then let one rank sprint ahead, the interesting interleavings
        blk = rng.choice(cands)
        for _ in range(rng.choice((1, 1, 1, 4, 16))):
            err = blk.step()
            steps += 1
            if err:
                return err
            if blk.done or not blk.runnable():
                break
        r = blk.r
        if all(x.done for x in live[r]):
            call[r] += 1
            if call[r] < len(sizes):
                nb = blocks_of(sizes[call[r]])
                live[r] = [Block(r, call[r], b, nb) for b in range(nb)]
            else:
                live[r] = []
    return None


SIZES = [16, 4, 16, 2, 6, 16, 3, 8, 16, 7, 16, 16, 1, 12, 4, 16]      # alternating large / small messages


@pytest.mark.parametrize("n_ranks", [2, 3, 4])
def test_protocol_invariant_under_random_schedules(n_ranks):
    for seed in range(400):
        err = simulate(n_ranks, SIZES, seed, full_grid=True)
        assert err is None, (seed, err)


def test_size_dependent_grid_is_unsafe():
    """The variant the kernel does NOT use: launching only as many blocks as the message needs.  The model finds a
    schedule in which a rank reads data of the wrong call (the reason md_allreduce_oneshot always launches the full
    grid)."""
    assert any(simulate(2, SIZES, seed, full_grid=False) is not None for seed in range(400))
