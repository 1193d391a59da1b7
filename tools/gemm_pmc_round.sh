#!/bin/bash
# MFMA / wait counters of the three per-layer GEMMs at C2 and C3 shapes (separate --pmc passes).  usage: gemm_pmc_round.sh <tag>
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
for CFG in C2 C3; do
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE"; do
    D=$OUT/${TAG}_gemmpmc_${CFG}_$(echo $SET | tr ' ' '_')
    rm -rf $D
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -- python3 tools/gemm_pmc.py $CFG > /dev/null 2> $D.err || (tail -5 $D.err; true)
    F=$(find $D -name "*counter_collection.csv" | head -1)
    if [ -n "$F" ]; then python3 tools/pmc_summary.py $F k_gemm_f32 > $OUT/${TAG}_gemmpmc_${CFG}_$(echo $SET | tr ' ' '_').json; fi
    rm -rf $D
    echo "$CFG $SET done"
  done
done
python3 - <<PY
import json, glob, collections
for cfg in ("C2", "C3"):
    merged = collections.defaultdict(dict)
    for f in sorted(glob.glob("$OUT/${TAG}_gemmpmc_%s_*.json" % cfg)):
        for k, v in json.load(open(f)).items():
            merged[k].update(v)
    json.dump(merged, open("$OUT/${TAG}_gemm_mfma_pmc_%s.json" % cfg, "w"), indent=1)
    for k, v in merged.items():
        print(cfg, k[:90])
        print("   ", {a: (round(b, 3) if b < 10 else int(b)) for a, b in v.items()})
PY
