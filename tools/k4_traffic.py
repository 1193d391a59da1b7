#!/usr/bin/env python3
"""HBM traffic of the aggregation kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected in
separate runs: the TCC block cannot hold both, MI355X_MICROARCH.md "HBM / rocprofv3" section).

usage: k4_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <C2|C3> <nodes> <rows E'> <hidden>
                     [kernel-substring] > profiles/r02_k4_hbm_traffic_C2.json

gfx950 corrections (same section of the guide): FETCH_SIZE counts the 128-B requests of wide (16 B / lane)
coalesced reads at 64 B -> doubled; WRITE_SIZE is exact for 16 B / lane streaming stores.  Both are in KiB."""
import csv
import json
import sys


def mean_kb(path, counter, needle):
    vals, names = [], set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"].split("(")[0])
    if not vals:
        raise SystemExit(f"no {counter} rows for a kernel matching {needle!r} in {path}")
    return sum(vals) / len(vals), len(vals), sorted(names)


def main():
    fetch_csv, write_csv, cfg, n, ep, h = sys.argv[1:7]
    needle = sys.argv[7] if len(sys.argv) > 7 else "k_pna_aggregate<2"
    n, ep, h = int(n), int(ep), int(h)
    f_kb, f_n, f_names = mean_kb(fetch_csv, "FETCH_SIZE", needle)
    w_kb, w_n, w_names = mean_kb(write_csv, "WRITE_SIZE", needle)
    assert f_names == w_names and len(f_names) == 1, (f_names, w_names)
    rd, wr = 2.0 * f_kb * 1024.0, w_kb * 1024.0
    alg = 8 * h * (ep + 4 * n) + 8 * ep
    json.dump({
        "kernel": f_names[0], "config": cfg, "launches_measured": min(f_n, w_n),
        "FETCH_SIZE_avg_kb": f_kb, "WRITE_SIZE_avg_kb": w_kb,
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for 16 B/lane coalesced reads); "
                      "WRITE_SIZE exact for 16 B/lane streaming stores; separate --pmc passes",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
        "note": "counted at the L2's fabric side: Infinity-Cache hits are included, so at C2 (working set < 256 MiB "
                "L3) this is L2-miss traffic, not DRAM traffic",
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py "
                   f"--config {cfg[1:]} --steps 10 --warmup 2 --graph 0 --train-steps 0 --no-cpu-baseline",
    }, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
