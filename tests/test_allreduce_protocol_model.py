"""Model check (randomised schedules) of the synchronisation protocol of csrc/allreduce.hip -- CPU only.

The kernel's correctness argument (double-buffered data and result buffers, one monotonic flag per (block, source
rank) and hop, a per-call sequence number, a grid sized to the message, no closing barrier, stream order between
calls) is easy to get subtly wrong and cannot be exercised across real GPUs on the development box, so the protocol
is restated here as interleaved state machines and run under thousands of random schedules.  Every rank r executes the
same sequence of calls; call k launches G(k) = min(MAXB, ceil(rows / N)) blocks, and block b

    0. reads the rank's call counter (k = counter + 1), THEN draws a ticket from the rank's `started` count; the block
       that draws ticket G - 1 -- every block of the call has read the counter by then -- resets `started` and stores k
       into the counter at some LATER point of its run (round 5; rounds 2-4: the last block to FINISH advanced it).
       Blocks of one launch start at different times, so a late block may read the counter long after an early one has
       finished: it must still see the same k.  Checked as a second invariant.

    1. writes the rows of block b into data[r][k & 1]               (rows are dealt by (row / N) % G)
    2. stores k into start[p][b][r] of every peer p;  waits until start[r][b][p] >= k for every p
    3. one-shot: reads every row of block b from data[p][k & 1] of every rank p
       two-shot: reads ITS rows (row % N == r) of block b from every rank, writes them to res[r][k & 1];
                 stores k into start2[p][b][r] of every p;  waits until start2[r][b][p] >= k;
                 reads the other owners' rows of block b from res[owner][k & 1]

and a rank starts call k+1 only when all blocks of call k have finished (kernels of one stream run in order).
Invariant: every element read in call k was written in call k.  Why a message-size-dependent grid is safe here (it
was not with the per-block call counters of the first version of the kernel, whose blocks could disagree on the
buffer half): a rank overwrites half (k & 1) in call k+2, i.e. after ALL its blocks finished call k+1, and at least
one of them (block 0 exists in every call) waited for every peer's call-(k+1) flag -- so every peer had entered call
k+1 and therefore finished reading in call k."""
import random

MAXB = 4          # blocks of the model (kMaxBlocks = 64 in the kernel)


def simulate(n_ranks, calls, seed, handoff="last_reader"):
    """calls: list of (rows, two_shot).  Returns None, or a description of the first violation.
    handoff: "last_reader" = the kernel's rule (the block that drew the last ticket stores k when it ends);
    "first_finisher" = a deliberately BROKEN rule (the first block to finish stores k) -- the negative control."""
    rng = random.Random(seed)
    N = n_ranks
    cap = max(c[0] for c in calls)
    data = [[[None] * cap, [None] * cap] for _ in range(N)]          # data[r][half][row] = call id written
    res = [[[None] * cap, [None] * cap] for _ in range(N)]
    start = [[[0] * N for _ in range(MAXB)] for _ in range(N)]       # start[owner][b][src]
    start2 = [[[0] * N for _ in range(MAXB)] for _ in range(N)]
    counter = [0] * N            # ctrl[0]: the call counter blocks read
    started = [0] * N            # ctrl[1]: blocks of the running call that have read it
    launch_k = [1] * N           # the k every block of the rank's running launch must obtain

    def grid_of(rows):
        return max(1, min(MAXB, (rows + N - 1) // N))

    class Block:
        def __init__(self, r, ci, b, G):
            self.r, self.ci, self.b, self.G = r, ci, b, G
            self.rows_n, self.two = calls[ci]
            self.k = None                                       # read in step -2 (blocks of a launch start at different times)
            self.rows = [i for i in range(self.rows_n) if (i // N) % G == b]
            self.mine = [i for i in self.rows if i % N == r]
            self.others = [i for i in self.rows if i % N != r]
            self.pc, self.i, self.p = -2, 0, 0
            self.hand_on = False                                # drew the last ticket: stores k into the counter when it ends
            self.late = rng.random() < 0.5                      # the ticket is drawn at once (the last wave has no row) or
            #                                                     after the block's rows (it has): both occur in the kernel
            self.done = False

        def draw(self):
            t = started[self.r]
            started[self.r] += 1
            if t == self.G - 1:
                started[self.r] = 0
                self.hand_on = True

        def runnable(self):
            if self.pc == 3:
                return all(start[self.r][self.b][p] >= self.k for p in range(N))
            if self.pc == 6:
                return all(start2[self.r][self.b][p] >= self.k for p in range(N))
            return True

        def step(self):
            r, b = self.r, self.b
            if self.pc == -2:                                    # read the call counter
                self.k = counter[r] + 1
                self.half = self.k & 1
                if self.k != launch_k[r]:
                    return f"rank {r} block {b}: read call number {self.k}, its launch is call {launch_k[r]}"
                self.pc = 0 if self.late else -1
                return None
            if self.pc == -1:                                    # draw a ticket (the read above has returned: barrier)
                self.draw()
                self.pc = 0
                return None
            k, h = self.k, self.half
            if self.pc == 0:
                self.pc, self.i = (1 if self.rows else 2), 0
            elif self.pc == 1:                                   # publish one row
                data[r][h][self.rows[self.i]] = k
                self.i += 1
                if self.i == len(self.rows):
                    self.pc, self.p = 2, 0
            elif self.pc == 2:                                   # raise my flag at one peer per micro-step
                start[self.p][b][r] = k
                self.p += 1
                if self.p == N:
                    self.pc = 3
            elif self.pc == 3:
                self.pc, self.i, self.p = 4, 0, 0
                if not (self.mine if self.two else self.rows):
                    self.pc = 5 if self.two else 9
            elif self.pc == 4:                                   # read one row of one rank's data buffer
                rows = self.mine if self.two else self.rows
                row = rows[self.i]
                got = data[self.p][h][row]
                if got != k:
                    return f"rank {r} call {k} block {b}: read row {row} of rank {self.p} data half {h} written by call {got}"
                self.p += 1
                if self.p == N:
                    self.p = 0
                    if self.two:
                        res[r][h][row] = k
                    self.i += 1
                    if self.i == len(rows):
                        self.pc, self.p = (5 if self.two else 9), 0
            elif self.pc == 5:                                   # second hop flags
                start2[self.p][b][r] = k
                self.p += 1
                if self.p == N:
                    self.pc = 6
            elif self.pc == 6:
                self.pc, self.i = (7 if self.others else 9), 0
            elif self.pc == 7:                                   # gather one row from its owner's result buffer
                row = self.others[self.i]
                got = res[row % N][h][row]
                if got != k:
                    return f"rank {r} call {k} block {b}: gathered row {row} of owner {row % N} half {h} from call {got}"
                self.i += 1
                if self.i == len(self.others):
                    self.pc = 9
            if self.pc == 9:
                if self.late:
                    self.draw()
                if (self.hand_on if handoff == "last_reader" else True):
                    counter[r] = self.k                          # at the END of the block that drew the last ticket
                self.done = True
            return None

    ci = [0] * N
    live = [[Block(r, 0, b, grid_of(calls[0][0])) for b in range(grid_of(calls[0][0]))] for r in range(N)]
    while any(ci[r] < len(calls) for r in range(N)):
        cands = [blk for r in range(N) for blk in live[r] if not blk.done and blk.runnable()]
        assert cands, "deadlock"
        blk = rng.choice(cands)
        for _ in range(rng.choice((1, 1, 1, 4, 16))):           # now and then let one block sprint ahead
            err = blk.step()
            if err:
                return err
            if blk.done or not blk.runnable():
                break
        r = blk.r
        if all(x.done for x in live[r]):
            assert counter[r] == launch_k[r] and started[r] == 0, (r, counter[r], launch_k[r], started[r])
            launch_k[r] += 1                                     # kernels of one stream run in order
            ci[r] += 1
            if ci[r] < len(calls):
                G = grid_of(calls[ci[r]][0])
                live[r] = [Block(r, ci[r], b, G) for b in range(G)]
            else:
                live[r] = []
    return None


ROWS = [16, 4, 16, 2, 6, 16, 3, 8, 16, 7, 16, 16, 1, 12, 4, 16]      # alternating large / small messages


def _calls(mode):
    if mode == "oneshot":
        return [(n, False) for n in ROWS]
    if mode == "twoshot":
        return [(n, True) for n in ROWS]
    return [(n, i % 3 == 1) for i, n in enumerate(ROWS)]             # mixed, as the auto policy produces


def test_protocol_invariant_under_random_schedules():
    for n_ranks in (2, 3, 4):
        for mode in ("oneshot", "twoshot", "mixed"):
            for seed in range(150):
                err = simulate(n_ranks, _calls(mode), seed)
                assert err is None, (n_ranks, mode, seed, err)


def test_model_catches_a_counter_handed_on_too_early():
    """Negative control: if the FIRST block to finish advanced the call counter (instead of the block that drew the
    last ticket), a block of the same launch that starts late reads k + 1 -- the model must find that schedule."""
    found = None
    for seed in range(300):
        found = simulate(3, _calls("mixed"), seed, handoff="first_finisher")
        if found:
            break
    assert found is not None and "read call number" in found, found
