#!/usr/bin/env python3
"""Per-step timeline of a rocprofv3 --kernel-trace CSV (kernel_trace.csv): a step ends at each k_mape.
Prints, for the median step (of the gap-free ones, i.e. the hipGraph replays, when there are any): wall time, summed kernel time, busy (union) time, idle gaps, and every kernel
with its start offset, duration and the gap to the previous kernel's end.
usage: timeline.py kernel_trace.csv [step_index_from_end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
steps, cur = [], []
for k in ks:
    cur.append(k)
    if ("k_mape" in k[2] or "k_readout_fused" in k[2]) and "bwd" not in k[2]:
        steps.append(cur)
        cur = []
if not steps:
    sys.exit("no k_mape kernel in the trace")
lens = [len(s) for s in steps]
common = max(set(lens), key=lens.count)
steps = [s for s in steps if len(s) == common]
def idle_of(s):
    busy, end = 0, s[0][0]
    for a, b, _ in s:
        if b > end:
            busy += b - max(a, end)
            end = b
    return (s[-1][1] - s[0][0]) - busy


# the median step among the back-to-back ones (hipGraph replays: idle < 1 % of the wall time) when the trace has any --
# a trace that also holds eager / instrumented steps would otherwise put one of those at the median
tight = [i for i, s in enumerate(steps) if idle_of(s) * 100 < (s[-1][1] - s[0][0])]
pool = tight if tight else list(range(len(steps)))
walls = sorted((steps[i][-1][1] - steps[i][0][0], i) for i in pool)
pick = walls[len(walls) // 2][1] if len(sys.argv) < 3 else len(steps) - 1 - int(sys.argv[2])
s = steps[pick]
t0 = s[0][0]
busy, end = 0, t0
for a, b, _ in s:
    if b > end:
        busy += b - max(a, end)
        end = b
wall = s[-1][1] - t0
print(f"steps with {common} kernels: {len(steps)}; step {pick}: wall {wall / 1e3:.1f} us, kernel sum "
      f"{sum(b - a for a, b, _ in s) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us")
prev_end = t0
for a, b, n in s:
    print(f"{(a - t0) / 1e3:8.1f} +{(b - a) / 1e3:7.1f}  gap {(a - prev_end) / 1e3:6.1f}  {n[:100]}")
    prev_end = max(prev_end, b)
