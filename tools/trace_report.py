"""Per-kernel totals and a stretch of the timeline from a rocprofv3 --kernel-trace database (rocpd .db).
    python tools/trace_report.py <results.db> [position 0..1 = 0.75] [milliseconds = 30] [rows = 80]
(Development aid.)"""
import sqlite3, re, collections, sys
db=sqlite3.connect(sys.argv[1])
c=db.cursor()
rows=list(c.execute("select name, start, end, grid_x, workgroup_x, stream_id, lds_size from kernels order by start"))
def short(n):
    m=re.search(r'(k_\w+|__amd\w+)', n)
    return m.group(1) if m else n[:40]
agg=collections.defaultdict(lambda:[0,0.0])
for n,s,e,g,w,st,l in rows:
    agg[short(n)][0]+=1; agg[short(n)][1]+=(e-s)/1e6
for k,v in sorted(agg.items(), key=lambda x:-x[1][1])[:12]: print("%-40s %6d launches %9.2f ms total %8.3f ms avg"%(k,v[0],v[1],v[1]/v[0]))
frac=float(sys.argv[2]) if len(sys.argv)>2 else 0.75
span=float(sys.argv[3]) if len(sys.argv)>3 else 30
mid=rows[int(len(rows)*frac)][1]
sel=[r for r in rows if mid<=r[1]<mid+span*1e6]
t0=sel[0][1]
last=None
for n,s,e,g,w,st,l in sel[:int(sys.argv[4]) if len(sys.argv)>4 else 80]:
    print("%9.3f %9.3f %7.3f  %-26s wgs=%d x %d lds=%d s=%s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,short(n),g//w,w,l,st))
