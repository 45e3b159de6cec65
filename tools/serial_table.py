"""Per-network-call table from tools/profile_bench_serial.sh output: python tools/serial_table.py [F] [calls]
(kernel x grid rows of gpurun_out/bench_serial_F<F>/stats.txt divided by the number of network calls in the run)."""
import os
import re
import sys

F = sys.argv[1] if len(sys.argv) > 1 else "32"
ncall = int(sys.argv[2]) if len(sys.argv) > 2 else 8
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for line in open(os.path.join(root, "gpurun_out", f"bench_serial_F{F}", "stats.txt")):
    m = re.match(r"(.{84})\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)", line)
    if m:
        rows.append((m.group(1).strip()[:78], int(m.group(2)), float(m.group(3)), float(m.group(4)), int(m.group(8))))
tot = 0.0
for name, calls, total, avg, wgs in rows:
    roof = calls == 41 and ("convk_kernel<3" in name or "wino3x3" in name)          # bench.py's roofline loop (41 launches each)
    if "at::native" in name or "copyBuffer" in name or roof:
        continue
    tot += total / ncall
    if total / ncall > 20:
        print(f"{name:80s} {calls / ncall:6.1f}/call  avg {avg:8.1f} us   {total / ncall:8.1f} us/call   wgs {wgs}")
print(f"sum {tot / 1e3:.2f} ms per call (rows of the roofline loop with exactly 41 launches skipped; mixed rows include it)")
