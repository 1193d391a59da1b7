#!/bin/bash
# One GPU-box session: rocprofv3 kernel stats of the default bench command + the two PMC passes of K4 at C2 and C3.
# usage (through gpurun): bash tools/gpu_profile_round.sh <tag>   -> gpurun_out/<tag>_*
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
python3 bench.py --steps 100 --warmup 10 > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench_c2.err
echo "bench c2 done"; tail -c 600 $OUT/${TAG}_bench_c2.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c2 -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-c3 > $OUT/${TAG}_bench_c2_under_rocprof.json 2> $OUT/${TAG}_prof_c2.err
cp $(find $OUT/${TAG}_prof_c2 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c2_kernel_stats.csv
echo "rocprof c2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c3 -- python3 bench.py --config 3 --steps 10 --warmup 3 --train-steps 3 --no-cpu-baseline > $OUT/${TAG}_bench_c3_under_rocprof.json 2> $OUT/${TAG}_prof_c3.err
cp $(find $OUT/${TAG}_prof_c3 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c3_kernel_stats.csv
echo "rocprof c3 done"
for CFG in 2 3; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_c${CFG}_${CTR} -- python3 bench.py --config $CFG --steps 10 --warmup 2 --graph 0 --train-steps 0 --no-cpu-baseline --no-c3 > $OUT/${TAG}_pmc_c${CFG}_${CTR}.json 2> $OUT/${TAG}_pmc_c${CFG}_${CTR}.err
    echo "pmc c$CFG $CTR done"
  done
done
