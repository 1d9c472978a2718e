"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) into a per-kernel stats CSV
(name, calls, total_ms, avg_us, min_us, max_us, pct) -- the same columns as `rocprofv3 --stats`' kernel_stats."""
import sqlite3
import sys


def main(db_path, out_path=None, top=40):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            f"from kernels group by {name_col} order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_ms,avg_us,min_us,max_us,pct"]
    for n, c, s, a, mn, mx in rows[:top]:
        n = n.replace(",", ";")
        lines.append(f"\"{n[:160]}\",{c},{s / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * s / tot:.2f}")
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
