#!/bin/bash
# bench.py's latency leg (one / four frames in flight, 8 crops per call) under dispatch settings given as arguments ("VAR=VAL VAR=VAL" per case)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for c in "$@"; do
  echo "== $c"
  env $c python3 $R/bench.py --no-cpu-baseline --no-slam-leg --no-global-ba-leg --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print(json.dumps(d.get('latency')))"
done
