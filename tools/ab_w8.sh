cd $GRAFT_REPO_ROOT
ms() { python3 bench.py --no-legs --only cnn --depth 1 --objects $1 --frames-per-step 1 --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for L in 3 5 8; do echo "crops $L: W8 $(ms $L) ms | SUO_WINO_W8=0 $(SUO_WINO_W8=0 ms $L) ms | W8 $(ms $L) | off $(SUO_WINO_W8=0 ms $L)"; done
