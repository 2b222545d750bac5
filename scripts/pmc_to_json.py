#!/usr/bin/env python3
"""scripts/evidence.sh, stage pmc: the per-kernel FETCH_SIZE / WRITE_SIZE sums of the separate rocprofv3 passes (pmc_<mode>_<counter>.txt,
lines `PMC <kernel> <counter> dispatches=N sum=S per_dispatch=P` from scripts/rocpd_summary.py) -> the JSON bench.py reads for
roofline.traffic: HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB of 1024 B; MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports
half of a wide streaming read).  Keyed by the kernel SYMBOL as rocprofv3 prints it; attention symbols also get the algorithmic bytes of the
uniform configs[1] launch (q, k, v read once + out written once, averaged over the per-part and per-sample launches: identical)."""
import json
import os
import re
import sys

out_dir = sys.argv[1]
MODES = {"float32": 4, "float32x2": 4, "bfloat16": 2, "float16": 2}      # bytes per value of q / k / v / out
TOK, D = 32 * 2 * 4096, 512
kernels = {}
for mode, elem in MODES.items():
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(out_dir, f"pmc_{mode}_{counter}.txt")
        if not os.path.exists(path):
            continue
        for line in open(path):
            m = re.match(r"PMC (.+?)\s+(\S+)\s+dispatches=(\d+) sum=(\S+) per_dispatch=(\S+)", line)
            if m and m.group(2) == counter:
                per.setdefault(m.group(1).strip(), {})[counter] = (int(m.group(3)), float(m.group(5)))
    for name, c in per.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or not ("attention" in name or "gemm" in name or "layernorm" in name):
            continue
        sym = name[5:] if name.startswith("void ") else name
        n, f = c["FETCH_SIZE"]; _, w = c["WRITE_SIZE"]
        rec = {"measured_on": f"{mode} model path, one flow step of the configs[1] batch", "dispatches": n, "fetch_size_kb_per_launch": f,
               "fetch_correction": 2.0, "write_size_kb_per_launch": w, "hbm_bytes_per_launch": int(round((2.0 * f + w) * 1024))}
        if "attention" in sym and "combine" not in sym:
            rec["algorithmic_bytes_per_launch"] = 4 * elem * TOK * D
        kernels[sym] = rec
# ragged reference-regime batch (evidence.sh stage "ragged"): attention symbols only; algorithmic bytes of the 262 011-point batch
RAGGED_TOK = 262011
ragged = {}
for mode, elem in MODES.items():
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(out_dir, f"pmc_ragged_{mode}_{counter}.txt")
        if not os.path.exists(path):
            continue
        for line in open(path):
            m = re.match(r"PMC (.+?)\s+(\S+)\s+dispatches=(\d+) sum=(\S+) per_dispatch=(\S+)", line)
            if m and m.group(2) == counter:
                per.setdefault(m.group(1).strip(), {})[counter] = (int(m.group(3)), float(m.group(5)))
    for name, c in per.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or "combine" in name:
            continue
        sym = name[5:] if name.startswith("void ") else name
        n, f = c["FETCH_SIZE"]; _, w = c["WRITE_SIZE"]
        ragged[sym] = {"measured_on": f"{mode} model path, one flow step of the ragged reference-regime batch (bench.py --workload ragged)", "dispatches": n,
                       "fetch_size_kb_per_launch": f, "fetch_correction": 2.0, "write_size_kb_per_launch": w,
                       "hbm_bytes_per_launch": int(round((2.0 * f + w) * 1024)), "algorithmic_bytes_per_launch": 4 * elem * RAGGED_TOK * D}
old = {}
try:
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")) as fh:
        old = json.load(fh)
except (OSError, ValueError):
    pass
doc = {"source": f"{out_dir}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only (scripts/evidence.sh, stage pmc) "
                 "of ONE flow step of the configs[1] batch through the model path (bench.py --flow-steps 1), per arithmetic mode",
       "unit": "KB per dispatch as rocprofv3 reports them (1 KB = 1024 B); FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide streaming read)",
       "kernels": kernels}
if ragged:
    note = (old.get("ragged_kernels") or {}).get("note")
    doc["ragged_kernels"] = dict(({"note": note} if note else {}), **ragged)
    doc["ragged_source"] = f"{out_dir}: the same two passes on the ragged batch (scripts/evidence.sh, stage ragged)"
elif "ragged_kernels" in old:      # not re-measured in this run: carried over unchanged
    doc["ragged_kernels"] = old["ragged_kernels"]
    doc["ragged_source"] = old.get("ragged_source", "rounds 4-5 ragged passes (scripts/pmc_ragged.sh); every entry names its file in measured_on")
print(json.dumps(doc, indent=1))
