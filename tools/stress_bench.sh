#!/bin/bash
# Run bench.py N times under a watchdog (faulthandler dumps every thread's stack if a run exceeds 150 s): hang hunting.
#   bash tools/stress_bench.sh N [bench flags...]
N=${1:-6}; shift
FLAGS=${@:---steps 20 --warmup 5}
mkdir -p gpurun_out/stress
for i in $(seq 1 $N); do
  timeout 220 python -c "
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(150, exit=True)
sys.argv = ['bench.py'] + '$FLAGS'.split()
runpy.run_path('bench.py', run_name='__main__')
" > gpurun_out/stress/run_$i.log 2>&1
  echo "run $i rc=$? $(grep -c metric gpurun_out/stress/run_$i.log) line(s) $(grep -o '"value": [0-9.]*' gpurun_out/stress/run_$i.log | head -1)"
  grep -B2 -A40 "Timeout (" gpurun_out/stress/run_$i.log | head -80
done
