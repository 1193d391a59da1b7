#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch.
usage: pmc_summary.py counter_collection.csv [name-substring ...] -> JSON on stdout"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"]
    if want and not any(w in name for w in want):
        continue
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, counters in acc.items():
    ent = {c: sum(v) / len(v) for c, v in counters.items()}
    ent["dispatches"] = len(next(iter(counters.values())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and ent.get("SQ_BUSY_CU_CYCLES"):
        # MFMA_BUSY: cycles summed over SIMDs; BUSY_CU: cycles summed over CUs (4 SIMDs each)
        ent["mfma_pipe_busy_frac_of_busy_simd_cycles"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * ent["SQ_BUSY_CU_CYCLES"])
    if ent.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in ent:
                ent[c + "_frac_of_wave_cycles"] = ent[c] / ent["SQ_WAVE_CYCLES"]
    out[name[:150]] = ent
json.dump(out, sys.stdout, indent=1)
