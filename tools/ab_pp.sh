# round 6: the two-tiles-per-workgroup form (SUO_WINO_PP) against the one-tile form on one box: dominant kernel, the forms beside it, and the headline
cd $GRAFT_REPO_ROOT
for pp in 1 0 1 0; do
  echo "SUO_WINO_PP=$pp"
  SUO_WINO_PP=$pp python tools/bench_f16x2.py 256 2>&1 | grep -i "fused\|3x3"
done
for pp in 1 0 1 0; do
  echo "SUO_WINO_PP=$pp headline: $(SUO_WINO_PP=$pp python bench.py --no-legs --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
