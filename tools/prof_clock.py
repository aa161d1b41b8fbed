import sqlite3,glob,sys
db=sqlite3.connect(glob.glob(sys.argv[1]+"/**/*.db",recursive=True)[0])
cur=db.cursor()
rows=list(cur.execute("select k.name, k.grid_x, k.grid_y, avg(k.end-k.start), avg(p.counter_value), count(*) from pmc_events p join kernels k on k.dispatch_id=p.dispatch_id where p.counter_name='GRBM_GUI_ACTIVE' group by k.name,k.grid_x,k.grid_y order by 4 desc limit 14"))
for n,gx,gy,ns,cyc,c in rows:
    print(f"{n[:60]:60s} grid {gx}x{gy} n={c} {ns/1e3:9.1f} us  GUI_ACTIVE {cyc:12.0f} cycles -> {cyc/ns:.3f} GHz")
