#!/bin/bash
# One GPU-box session: the default bench line, rocprofv3 kernel stats of the bench command at C2 and C3, and of
# training steps (tools/train_step.py, one stream so that kernel durations are not stretched by overlap) at C2 / C3.
# usage (through gpurun): bash tools/gpu_profile_train.sh <tag>   -> gpurun_out/<tag>_*
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
python3 bench.py --steps 100 --warmup 10 > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench_c2.err
echo "bench c2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c2 -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-c3 > $OUT/${TAG}_bench_c2_under_rocprof.json 2> $OUT/${TAG}_prof_c2.err
cp $(find $OUT/${TAG}_prof_c2 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c2_kernel_stats.csv
echo "rocprof c2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c3 -- python3 bench.py --config 3 --steps 10 --warmup 3 --train-steps 3 --no-cpu-baseline > $OUT/${TAG}_bench_c3_under_rocprof.json 2> $OUT/${TAG}_prof_c3.err
cp $(find $OUT/${TAG}_prof_c3 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c3_kernel_stats.csv
echo "rocprof c3 done"
export GNNSAFT_BACKWARD_SIDE_STREAM=0
for CFG in 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train_c${CFG} -- python3 tools/train_step.py $CFG 20 > $OUT/${TAG}_train_c${CFG}.log 2>&1
  cp $(find $OUT/${TAG}_prof_train_c${CFG} -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c${CFG}_train_step_kernel_stats.csv
  echo "rocprof train c$CFG done"
done
