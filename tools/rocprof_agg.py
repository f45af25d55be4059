"""Per-kernel average / minimum duration and grid of the mca:: kernels in a rocprofv3 results database.
usage: rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 bench.py ... ; python tools/rocprof_agg.py DIR/NAME_results.db"""
import sqlite3,re,collections,sys
db=sqlite3.connect(sys.argv[1])
rows=list(db.execute("select name, duration, grid_x, grid_y, workgroup_x from kernels"))
agg=collections.defaultdict(list)
for n,d,gx,gy,wx in rows:
    if 'mca' in n: agg[(re.sub(r'\(.*','',n)[-50:],gx//wx,gy)].append(d)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])): print("%-52s grid %6d x %3d  n %4d  avg %7.1f us  min %7.1f" % (k[0],k[1],k[2],len(v),sum(v)/len(v)/1e3, min(v)/1e3))
