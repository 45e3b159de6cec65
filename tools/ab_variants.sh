# A/B of kernel variant builds (tools/build_variant.sh <name> -DFLAG ...) on the dominant kernel's launch shape (256 crops @64x64), fp16 form: the fused Residual tail and
# the 3x3 alone, HIP events.   bash tools/ab_variants.sh <variant> [<variant> ...]      (inside a gpurun call; "base" = the shipped library)
cd $GRAFT_REPO_ROOT
show='
import sys, ast
r = ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); sp = r["same_process"]
print("%-12s fused tail %8.1f us | 3x3 alone %8.1f us" % (sys.argv[1], r["avg_launch_us"], sp["wino3x3_x3_kernel<false,false,false,4,2> (f16x2, 3x3 alone)"]["avg_launch_us"]))'
for v in base "$@" base; do
  if [ $v = base ]; then unset SUO_HIP_LIB; else export SUO_HIP_LIB=$PWD/suo_slam_amd/variants/libsuo_hip_$v.so; fi
  python tools/bench_dominant.py 40 256 2>/dev/null | python -c "$show" $v
done
