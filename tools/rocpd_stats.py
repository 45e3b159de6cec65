"""Summarise a rocprofv3 rocpd sqlite database: per-(kernel, grid) count / total / avg / share (like --stats)."""
import sqlite3
import sys

db = sys.argv[1]
by_grid = len(sys.argv) > 2 and sys.argv[2] == "grid"
c = sqlite3.connect(db)
grp = "name, grid_x, grid_y" if by_grid else "name"
rows = c.execute(f"select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                 f"max(grid_x/workgroup_x), max(grid_y), max(vgpr_count), max(accum_vgpr_count), max(lds_size) "
                 f"from kernels group by {grp} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':84s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} {'wgs':>7s} {'vgpr':>5s} {'lds':>6s}")
for r in rows:
    print(f"{r[0][:84]:84s} {r[1]:6d} {r[2]/1e3:10.1f} {r[3]/1e3:9.2f} {r[4]/1e3:9.2f} {r[5]/1e3:9.2f} {100*r[2]/tot:6.2f} "
          f"{r[6]*r[7]:7d} {r[8]+r[9]:5d} {r[10]:6d}")
print(f"total kernel time {tot/1e3:.1f} us")
