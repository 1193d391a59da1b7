#!/bin/bash
# bench (C2 forward, hipGraph replay) + launch timeline of one replayed step.  usage: bash tools/quick_timeline.sh <tag>
set -e -o pipefail
TAG=${1:-x}
OUT=gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
python3 bench.py --steps 200 --warmup 20 --train-steps 10 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || (tail -20 $OUT/${TAG}_bench.err; exit 1)
python3 - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
print("C2 value", round(d["value"]), "graphs/s", "ms/step", round(d["ms_per_step"], 4), "eager", round(d["eager_ms_per_step"], 4))
print("K4", d["roofline"]["avg_launch_ms"], "frac", round(d["roofline"]["frac"], 3))
print("gemm", {k: (round(v["avg_ms"] * 1e3, 1), round(v["executed_tflops"], 1)) for k, v in d["roofline_gemm"]["per_kernel"].items()}, "frac", round(d["roofline_gemm"]["frac"], 3))
print("train", d["train_step"])
if "c3" in d:
    c = d["c3"]
    print("C3 ms/step", round(c["ms_per_step"], 3), "graphs/s", round(c["graphs_per_s"]), "K4 frac", round(c["roofline"]["frac"], 3))
    print("C3 train", c.get("train_step"))
    print("C3 gemm", {k: (round(v["avg_ms"] * 1e3, 1), round(v["executed_tflops"], 1)) for k, v in c["roofline_gemm"]["per_kernel"].items()}, "frac", round(c["roofline_gemm"]["frac"], 3))
print("flags", d.get("input_error_flags"))
PY
rm -rf $OUT/${TAG}_tl
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_tl -- python3 bench.py --steps 100 --warmup 10 --train-steps 0 --no-cpu-baseline --no-c3 > $OUT/${TAG}_tl.json 2> $OUT/${TAG}_tl.err
python3 tools/timeline.py $(find $OUT/${TAG}_tl -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_timeline.txt
cat $OUT/${TAG}_timeline.txt
rm -rf $OUT/${TAG}_tl
