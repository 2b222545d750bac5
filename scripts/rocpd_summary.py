#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (the default output format of rocprofv3 on ROCm 7.2) into the
per-kernel table `--stats` would print: calls, total / average / min / max duration, share of GPU time; and, when
the run collected PMC counters, the per-kernel counter sums per dispatch.
Usage: rocpd_summary.py <results.db> [--pmc]"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# {path}")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>12s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>6s}")
    for name, n, s, a, mn, mx in rows:
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:70s} {n:6d} {s / 1e6:12.3f} {a / 1e3:12.2f} {mn / 1e3:12.2f} {mx / 1e3:12.2f} {100.0 * s / tot:6.2f}")
    print(f"{'TOTAL':70s} {sum(r[1] for r in rows):6d} {tot / 1e6:12.3f}")
    if "--pmc" in sys.argv:
        try:
            q = ("select k.name, p.counter_name, count(*), sum(p.counter_value) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
                 "group by k.name, p.counter_name order by k.name")
            for name, cname, n, v in cur.execute(q):
                short = name.split("(")[0][:120]          # the whole template argument list (rocprofv3 prints it before the parameters)
                print(f"PMC {short:60s} {cname:16s} dispatches={n} sum={v:.6g} per_dispatch={v / n:.6g}")
        except sqlite3.Error as e:
            cols = [r[1] for r in cur.execute("pragma table_info('pmc_events')")]
            print("pmc_events columns:", cols, "error:", e)


if __name__ == "__main__":
    main()
