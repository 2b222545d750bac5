#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from a rocprofv3 rocpd database collected with
  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace
MFMA_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the fraction of SIMD cycles the matrix pipe is busy at the
clock the kernel actually ran at; clock_GHz = (GRBM_GUI_ACTIVE / 8) / kernel duration (DVFS: the bf16 kernels run below 2.4 GHz, and the
roofline fractions of bench.py / DESIGN.md are against the 2.4 GHz peak, so frac ~= MFMA_util x clock / 2.4).
Usage: mfma_util.py <results.db> <tag>"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tag = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = db.cursor()
    dur = {n: (c, s) for n, c, s in cur.execute("select name, count(*), sum(duration) from kernels group by name")}
    pmc = {}
    q = ("select k.name, p.counter_name, sum(p.counter_value) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
         "group by k.name, p.counter_name")
    for name, cname, v in cur.execute(q):
        pmc.setdefault(name, {})[cname] = v
    rows = sorted(pmc.items(), key=lambda kv: -dur.get(kv[0], (0, 0))[1])
    for name, c in rows:
        mf, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        if not mf or not gui:
            continue
        n, ns = dur[name]
        short = name.split("(")[0]
        print(f"{tag:10s} {short:58s} launches={n:4d} avg_us={ns / n / 1e3:9.1f} clock_GHz={gui / 8 / ns:5.2f} "
              f"MFMA_util={mf / (1024.0 * gui / 8):.3f} VALU_per_MFMA={c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_INSTS_MFMA', 0), 1):.2f}")


if __name__ == "__main__":
    main()
