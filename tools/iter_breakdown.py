"""Per-iteration kernel breakdown of the speculative loop from a rocprofv3 rocpd database (--kernel-trace).

One `accept_kernel` (md_accept_rollback) dispatch closes every speculative iteration, so consecutive dispatches of it delimit
iterations.  For the steady-state iterations (the most common interval length +-20 %) this prints, averaged per
iteration: wall time, GPU-busy time, idle gap, and time / calls per kernel name.

usage: python tools/iter_breakdown.py <db> [out.csv]"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    """Kernel name without namespaces, template arguments and parameter lists."""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0][:48]


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "accept_kernel" in r[0]]
    if len(marks) < 4:
        raise SystemExit("fewer than 4 accept_kernel dispatches in the trace")
    spans = [(marks[i], marks[i + 1], rows[marks[i + 1]][2] - rows[marks[i]][2]) for i in range(len(marks) - 1)]
    lens = sorted(s[2] for s in spans)
    med = lens[len(lens) // 2]
    steady = [s for s in spans if abs(s[2] - med) <= 0.2 * med]
    agg = defaultdict(lambda: [0, 0])
    wall = busy = 0
    for a, b, ln in steady:
        wall += ln
        for n, st, en in rows[a + 1:b + 1]:
            agg[n][0] += en - st
            agg[n][1] += 1
            busy += en - st
    k = len(steady)
    # where the idle time sits: gaps between consecutive kernels, by (kernel before, kernel after)
    gaps = defaultdict(lambda: [0, 0])
    for a, b, ln in steady:
        for (n0, _, e0), (n1, s1, _) in zip(rows[a:b], rows[a + 1:b + 1]):
            if s1 > e0:
                key = (short(n0), short(n1))
                gaps[key][0] += s1 - e0
                gaps[key][1] += 1
    lines = [f"# {k} steady-state iterations (median {med / 1e6:.3f} ms) of {len(spans)} intervals",
             f"# per iteration: wall {wall / k / 1e6:.3f} ms, gpu busy {busy / k / 1e6:.3f} ms, "
             f"idle {(wall - busy) / k / 1e6:.3f} ms",
             "name,calls_per_iter,ms_per_iter,avg_us,pct_of_wall"]
    for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
        lines.append(f"\"{n.replace(',', ';')[:150]}\",{c / k:.2f},{t / k / 1e6:.4f},{t / c / 1e3:.2f},"
                     f"{100.0 * t / wall:.2f}")
    lines.append("# idle gaps by (kernel before -> kernel after): gaps_per_iter, us_per_iter, avg_us")
    for (n0, n1), (t, c) in sorted(gaps.items(), key=lambda x: -x[1][0])[:25]:
        lines.append(f"# gap,\"{n0} -> {n1}\",{c / k:.2f},{t / k / 1e3:.1f},{t / c / 1e3:.2f}")
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
