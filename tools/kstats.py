#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: per-step time by kernel (steps = calls of k_mape)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = next((int(r["Calls"]) for r in rows if "k_adamw" in r["Name"] or "k_readout_fused" in r["Name"] or "k_mape(" in r["Name"]), 1)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"steps={steps} kernel time per step = {tot / steps / 1e3:.1f} us")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{r['Name'][:86]:86s} calls/step={int(r['Calls']) / steps:5.1f} avg_us={float(r['AverageNs']) / 1e3:8.1f} "
          f"us/step={float(r['TotalDurationNs']) / steps / 1e3:8.1f}")
