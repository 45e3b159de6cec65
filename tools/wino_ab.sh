cd $GRAFT_REPO_ROOT
python tools/bench_dominant.py 40 256 2>/dev/null | python -c "
import sys,ast; r=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print('base  fused', r['avg_launch_us'], 'plain', r['same_process']['wino3x3_kernel<false> (3x3 alone)']['avg_launch_us'])"
for v in "$@"; do
SUO_HIP_LIB=$PWD/suo_slam_amd/variants/libsuo_hip_$v.so python tools/bench_dominant.py 40 256 2>/dev/null | python -c "
import sys,ast; r=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print('$v fused', r['avg_launch_us'], 'plain', r['same_process']['wino3x3_kernel<false> (3x3 alone)']['avg_launch_us'])"
done
