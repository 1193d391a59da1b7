#!/bin/bash
# Round-4 evidence in one GPU-box session -> gpurun_out/r04_*  (copied to profiles/ afterwards)
set -e -o pipefail
TAG=r04
OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
# 1. the default bench line (C2 headline + c3 block + train_step + cpu_baseline)
python3 bench.py --steps 100 --warmup 10 > $OUT/${TAG}_c2_bench_line.json 2> $OUT/${TAG}_c2_bench.err
echo "bench c2 done"; python3 -c "
import json; d=json.loads(open('$OUT/${TAG}_c2_bench_line.json').read().strip().splitlines()[-1])
print('C2', d['value'], d['ms_per_step'], 'K4', d['roofline']['frac'], 'gemm', d['roofline_gemm']['frac'], 'c3', d['c3']['ms_per_step'], d['c3']['graphs_per_s'], 'cpu', d['cpu_baseline']['value'], 'flags', d['input_error_flags'])"
# 2. rocprofv3 kernel stats of the same command, C2 and C3
rm -rf $OUT/${TAG}_prof_c2 $OUT/${TAG}_prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c2 -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-c3 > $OUT/${TAG}_c2_bench_line_under_rocprof.json 2> $OUT/${TAG}_prof_c2.err
cp $(find $OUT/${TAG}_prof_c2 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c2_kernel_stats.csv
python3 tools/timeline.py $(find $OUT/${TAG}_prof_c2 -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_c2_graph_replay_timeline.txt || true
rm -rf $OUT/${TAG}_prof_c2
echo "rocprof c2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_c3 -- python3 bench.py --config 3 --steps 10 --warmup 3 --train-steps 3 --no-cpu-baseline > $OUT/${TAG}_c3_bench_line_under_rocprof.json 2> $OUT/${TAG}_prof_c3.err
cp $(find $OUT/${TAG}_prof_c3 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c3_kernel_stats.csv
rm -rf $OUT/${TAG}_prof_c3
echo "rocprof c3 done"
# 3. K4 HBM traffic (separate PMC passes)
for CFG in 2 3; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    D=$OUT/${TAG}_pmc_c${CFG}_${CTR}
    rm -rf $D
    rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $D -- python3 bench.py --config $CFG --steps 10 --warmup 2 --graph 0 --train-steps 0 --no-cpu-baseline --no-c3 > $D.json 2> $D.err
    echo "pmc c$CFG $CTR done"
  done
  NODES=$(python3 -c "import json;d=json.loads(open('$OUT/${TAG}_pmc_c${CFG}_FETCH_SIZE.json').read().strip().splitlines()[-1]);print(d['config']['nodes'], d['config']['edges_with_self_loops'])")
  H=$([ $CFG = 2 ] && echo 128 || echo 256)
  python3 tools/k4_traffic.py $(find $OUT/${TAG}_pmc_c${CFG}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/${TAG}_pmc_c${CFG}_WRITE_SIZE -name "*counter_collection.csv" | head -1) C${CFG} $NODES $H > $OUT/${TAG}_k4_hbm_traffic_C${CFG}.json || true
  rm -rf $OUT/${TAG}_pmc_c${CFG}_FETCH_SIZE $OUT/${TAG}_pmc_c${CFG}_WRITE_SIZE
done
# 4. training-step kernel stats (one stream)
bash tools/train_profile.sh ${TAG} > $OUT/${TAG}_train_profile.txt 2>&1 || true
# 5. MFMA / VALU / LDS / wait counters of the GEMM kernels the forward actually launches (incl. the fused aggregation + update)
for CFG in 2 3; do
  rm -f $OUT/${TAG}gemm${CFG}_pmc_*.json $OUT/${TAG}gemm${CFG}_pmc_kernel_trace.csv
  bash tools/pmc_round.sh ${TAG}gemm${CFG} k_gemm,k_update_agg -- bench.py --config $CFG --steps 6 --warmup 2 --graph 0 --train-steps 0 --no-cpu-baseline --no-c3 > $OUT/${TAG}_gemm_pmc_c${CFG}.txt 2>&1 || true
  cp $OUT/${TAG}gemm${CFG}_pmc_merged.json $OUT/${TAG}_gemm_mfma_pmc_C${CFG}.json || true
done
# 6. two-rank rehearsal of the N > 1 line (gloo, both ranks on this GPU: control flow only)
GNNSAFT_BENCH_REHEARSAL=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --train-steps 5 --no-cpu-baseline > $OUT/${TAG}_two_rank_rehearsal_bench_line.json 2> $OUT/${TAG}_two_rank.err || true
# 7. C5 stand-in
python3 bench.py --config 5 --steps 200 --warmup 20 > $OUT/${TAG}_c5_bench_line.json 2> $OUT/${TAG}_c5.err || true
echo "all done"
