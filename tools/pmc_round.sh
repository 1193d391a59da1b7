#!/bin/bash
# generic counter round: pmc_round.sh <tag> <kernel-name-substring> -- <python script and args>
set -e -o pipefail
TAG=$1; FILT=$2; shift 3
OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  D=$OUT/${TAG}_pmc_$(echo $SET | tr ' ' '_')
  rm -rf $D
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -- python3 "$@" > /dev/null 2> $D.err || (tail -5 $D.err; true)
  F=$(find $D -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python3 tools/pmc_summary.py $F $(echo $FILT | tr "," " ") > $OUT/${TAG}_pmc_$(echo $SET | tr ' ' '_').json; fi
  K=$(find $D -name "*kernel_trace.csv" | head -1)
  if [ -n "$K" ] && [ ! -f $OUT/${TAG}_pmc_kernel_trace.csv ]; then cp $K $OUT/${TAG}_pmc_kernel_trace.csv; fi
  rm -rf $D $D.err
  echo "$SET done"
done
python3 - <<PY
import json, glob, collections, csv
merged = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/${TAG}_pmc_*.json")):
    for k, v in json.load(open(f)).items():
        merged[k].update(v)
dur = collections.defaultdict(list)
try:
    for r in csv.DictReader(open("$OUT/${TAG}_pmc_kernel_trace.csv")):
        dur[r["Kernel_Name"][:150]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
except Exception as e:
    print("no kernel trace:", e)
for k, v in merged.items():
    d = dur.get(k)
    if d:
        v["us_under_profiler"] = sorted(d)[len(d) // 2]
        if "GRBM_GUI_ACTIVE" in v:
            v["clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8 / v["us_under_profiler"] / 1e3
json.dump(merged, open("$OUT/${TAG}_pmc_merged.json", "w"), indent=1)
for k, v in merged.items():
    print(k[:120])
    print("   ", {a: (round(b, 3) if b < 1000 else int(b)) for a, b in v.items()})
PY
