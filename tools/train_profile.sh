#!/bin/bash
# kernel stats of full training steps (one stream) at C2 and C3.  usage: train_profile.sh <tag>
set -e -o pipefail
TAG=${1:-x}
OUT=gpurun_out
export TMPDIR=/tmp
export GNNSAFT_BACKWARD_SIDE_STREAM=0
for CFG in 2 3; do
  rm -rf $OUT/${TAG}_prof_train_c${CFG}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train_c${CFG} -- python3 tools/train_step.py $CFG 20 > $OUT/${TAG}_train_c${CFG}.log 2>&1
  cp $(find $OUT/${TAG}_prof_train_c${CFG} -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c${CFG}_train_step_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof_train_c${CFG}
  echo "== C$CFG"; tail -2 $OUT/${TAG}_train_c${CFG}.log
  python3 tools/kstats.py $OUT/${TAG}_c${CFG}_train_step_kernel_stats.csv 24
done
