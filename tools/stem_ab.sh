#!/bin/bash
# A/B of the prior-less pass's first two launches: fused RoIAlign + stem on the bf16 pipe (SUO_STEM_X3=1, default) against
# roi_align_concat + fp32-pipe stem (SUO_STEM_X3=0); same box, alternating.  Output: gpurun_out/stem_ab.txt
mkdir -p gpurun_out
out=gpurun_out/stem_ab.txt
: > $out
for rep in 1 2; do
  for m in 1 0; do
    echo "== SUO_STEM_X3=$m rep $rep (batched, driver flags)" >> $out
    SUO_STEM_X3=$m python3 bench.py --no-legs --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" >> $out
  done
done
for m in 1 0; do
  echo "== SUO_STEM_X3=$m one frame per call" >> $out
  SUO_STEM_X3=$m python3 tools/time_frame_chain.py 2>&1 | tail -n 12 >> $out
done
cat $out
