"""per (kernel, grid) launch-duration table from a rocprofv3 rocpd database.  usage: kernel_by_grid.py <db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = f"%{sys.argv[2]}%" if len(sys.argv) > 2 else "%"
q = ("select name, grid_x, grid_y, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like ? "
     "group by name, grid_x, grid_y order by 1, 2")
for r in db.execute(q, (pat,)):
    print(f"n={r[3]:5d} avg {r[4]:8.1f} us min {r[5]:8.1f} grid {r[1]:8d}x{r[2]:<4d} {r[0][:90]}")
