#!/bin/bash
# counters of ONE GEMM shape: pmc_one.sh M N_OUT K CFG "CTR1 CTR2" ...
set -e -o pipefail
export TMPDIR=/tmp
M=$1; N=$2; K=$3; C=$4; shift 4
for SET in "$@"; do
  D=gpurun_out/pmc_one_tmp
  rm -rf $D
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -- python3 tools/gemm_one.py $M $N $K $C > /dev/null 2> $D.err || (tail -3 $D.err; true)
  F=$(find $D -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py $F k_gemm_f32 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print({a:(int(b) if b>10 else round(b,3)) for a,b in v.items()})"
  rm -rf $D
done
