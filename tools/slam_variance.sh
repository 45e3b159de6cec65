cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -c "
from bench_legs.path import slam_leg
r = slam_leg()
print('leg alone', r['tracking_ms_per_view'], r['host_debug_route']['tracking_ms_per_view'])" 2>/dev/null; done
python bench.py --no-cpu-baseline --no-tless-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench default flags: slam', d['slam']['tracking_ms_per_view'], d['slam']['host_debug_route']['tracking_ms_per_view'], d['latency']['network_ms_per_frame'])"
python bench.py --no-cpu-baseline --no-tless-leg --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench driver flags: slam', d['slam']['tracking_ms_per_view'], d['slam']['host_debug_route']['tracking_ms_per_view'], d['latency']['network_ms_per_frame'])"
python bench.py --no-cpu-baseline --no-tless-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench default flags: slam', d['slam']['tracking_ms_per_view'], d['slam']['host_debug_route']['tracking_ms_per_view'], d['latency']['network_ms_per_frame'])"
