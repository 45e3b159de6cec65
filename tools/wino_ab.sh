# A/B of kernel variant builds (tools/build_variant.sh <name> -DFLAG ...) on the dominant kernel's launch shape: fused Residual tail and 3x3 alone, on
# both matrix pipes.   bash tools/wino_ab.sh <variant> [<variant> ...]      (inside a gpurun call)
cd $GRAFT_REPO_ROOT
show='
import sys, ast
r = ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); sp = r["same_process"]
f32 = [v for k, v in sp.items() if k.startswith("wino3x3_kernel<true>")]
print(sys.argv[1], "fused (as launched)", r["avg_launch_us"], "| fp32-pipe fused", f32[0]["avg_launch_us"] if f32 else None,
      "| 3x3 alone: fp32", sp["wino3x3_kernel<false> (fp32 pipe, 3x3 alone)"]["avg_launch_us"], "bf16x3", sp["wino3x3_x3_kernel<false> (bf16x3, 3x3 alone)"]["avg_launch_us"])'
python tools/bench_dominant.py 40 256 2>/dev/null | python -c "$show" base
for v in "$@"; do
  SUO_HIP_LIB=$PWD/suo_slam_amd/variants/libsuo_hip_$v.so python tools/bench_dominant.py 40 256 2>/dev/null | python -c "$show" $v
done
