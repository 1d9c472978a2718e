"""Protocol model of the tile ring of prefill32p_attn_kernel (csrc/attn.hip): two wave groups run one phase apart
(waves 4..7 execute one extra barrier), tile x is written in phase P_b(x - 2) and read in P_a(x - 1) (K) and P_a(x) (V).
The model walks the hardware-barrier intervals and checks that every read sees the tile it expects, completely written
by BOTH groups in earlier intervals, and that no group overwrites a slot another group still reads in the same
interval -- for every (tiles of the workgroup, tiles of group 0, tiles of group 1).  A ring of three slots must fail
(that is why the kernel uses four).  CPU only; the kernel itself is checked on the GPU in tests/test_gpu_ops.py."""
import pytest


def intervals(n, my, g, ring):
    """Per hardware-barrier interval, the LDS accesses ('r' | 'w', slot, tile) of wave group g."""
    pro = [("w", x % ring, x) for x in range(min(n, 2))]            # before the barrier that ends the prologue staging
    s0 = [("r", 0, 0)] if my > 0 else []                            # S(0), all waves in step
    phases = []
    for t in range(n):
        phases.append([("w", (t + 2) % ring, t + 2)] if t + 2 < n else [])                                  # P_b(t)
        phases.append(([("r", (t + 1) % ring, t + 1)] if t + 1 < my else []) +
                      ([("r", t % ring, t)] if t < my else []))                                            # P_a(t)
    if g == 0:      # no barrier between S(0) and P_b(0); one balancing barrier after the loop, then the epilogue
        body = ([s0 + phases[0]] + phases[1:] + [[], []]) if phases else [s0, []]
    else:           # the extra barrier sits between S(0) and P_b(0); then the loop, then the epilogue
        body = [s0] + phases + [[]]
    return [pro] + body


def check(n, my0, my1, ring):
    seq = [intervals(n, my0, 0, ring), intervals(n, my1, 1, ring)]
    assert len(seq[0]) == len(seq[1]), "both groups execute the same number of barriers"
    holds, halves = {}, {}
    for i in range(len(seq[0])):
        acts = [(g, a) for g in (0, 1) for a in seq[g][i]]
        for g, (kind, slot, tile) in acts:
            if kind != "r":
                continue
            assert holds.get(slot) == tile, (n, my0, my1, i, g, slot, tile, holds.get(slot))
            assert not any(k == "w" and s == slot for _, (k, s, _) in acts), (n, my0, my1, i, "write during read", slot)
        for g, (kind, slot, tile) in acts:
            if kind == "w":
                halves.setdefault((slot, tile), set()).add(g)
                holds[slot] = tile if halves[(slot, tile)] == {0, 1} else None


def test_ring_of_four_is_race_free_for_every_tile_count():
    for n in range(0, 13):
        for my0 in range(0, n + 1):
            for my1 in range(0, n + 1):
                check(n, my0, my1, 4)


def test_ring_of_three_is_not():
    with pytest.raises(AssertionError):
        for n in range(0, 13):
            check(n, n, n, 3)
