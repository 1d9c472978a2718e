"""What the GPU does around the iteration boundary of the speculative loop, from a rocprofv3 rocpd database (--kernel-trace).

`accept_kernel` (md_accept_rollback) closes an iteration; the host then reads the iteration's two flags and launches the next
one.  For the steady-state iterations this prints the kernels from the accept kernel on -- name, duration, and the idle gap in
front of each -- averaged position by position over the iterations that show the most common sequence of names, so that the
host-bound stretch (before the host is ahead of the GPU again) can be read off directly.

usage: python tools/iter_boundary.py <db> [n_after=14] [n_before=3]"""
import sqlite3
import sys
from collections import Counter


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return n.split("(")[0][:70]


def main(db_path, n_after=14, n_before=3):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "accept_kernel" in r[0]]
    marks = [m for m in marks if m - n_before >= 0 and m + n_after < len(rows)]
    if len(marks) < 6:
        raise SystemExit("fewer than 6 accept_kernel dispatches in the trace")
    marks = marks[2:-1]                                     # drop the warm-up iterations and the last one
    seqs = Counter(tuple(short(rows[m + d][0]) for d in range(-n_before, n_after + 1)) for m in marks)
    seq, cnt = seqs.most_common(1)[0]
    same = [m for m in marks if tuple(short(rows[m + d][0]) for d in range(-n_before, n_after + 1)) == seq]
    print(f"# {len(marks)} iteration boundaries, {cnt} with the most common kernel sequence ({len(seqs)} distinct sequences)")
    print(f"# {'pos':>4s} {'gap before us':>13s} {'duration us':>11s}  kernel")
    tot_gap = 0.0
    for j, d in enumerate(range(-n_before, n_after + 1)):
        gap = sum(rows[m + d][1] - rows[m + d - 1][2] for m in same) / len(same) / 1e3
        dur = sum(rows[m + d][2] - rows[m + d][1] for m in same) / len(same) / 1e3
        if d > 0:
            tot_gap += max(gap, 0.0)
        print(f"  {d:>+4d} {gap:13.2f} {dur:11.2f}  {seq[j]}")
    print(f"# idle in the {n_after} launches after the accept kernel: {tot_gap:.1f} us per iteration")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 14, int(a[3]) if len(a) > 3 else 3)
