#!/bin/bash
# Package power and shader clock while bench.py's timed region runs (rocm-smi samples every 2 s), then the large GEMMs of the call on random,
# post-ReLU and all-zero activations (operand toggling -> power -> clock).   -> gpurun_out/power_trace.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/power_trace.txt
mkdir -p $R/gpurun_out
{
  echo "# bash tools/power_trace.sh   (1x MI355X)"
  rocm-smi --showmaxpower 2>&1 | grep -i "Power (W)"
  echo "# idle:"
  rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk"
  echo "# python3 bench.py --no-legs --steps 900 --warmup 5 in the background; samples from t = 18 s:"
} > $O
(python3 $R/bench.py --no-legs --steps 900 --warmup 5 > /tmp/power_bench.json 2>/dev/null &)
sleep 18
for i in 1 2 3 4 5 6 7; do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | tr '\n' ' ' >> $O; echo >> $O; sleep 2; done
sleep 14
{ echo "# the bench line of that run:"; cut -c1-220 /tmp/power_bench.json; } >> $O
for d in randn relu zeros; do
  echo "# SUO_BENCH_DATA=$d python3 tools/bench_gemm_x3_shapes.py 256" >> $O
  SUO_BENCH_DATA=$d python3 $R/tools/bench_gemm_x3_shapes.py 256 2>&1 | grep "M=" >> $O
done
cat $O
