import sqlite3, sys, subprocess, re
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
cols=[r[1] for r in cur.execute("pragma table_info(kernels)")]
rows=cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s}")
for n,c,t,a,mn,mx in rows:
    n=re.sub(r'\(.*','',n)
    n=re.sub(r'sdpb::Solver<(\d+)>::','S\\1::',n)
    n=n.replace('void ','').replace('sdpb::','')
    print(f"{n[:70]:70s} {c:6d} {t/1e6:10.3f} {a/1e6:10.4f} {mn/1e6:9.4f} {mx/1e6:9.4f} {100*t/tot:6.2f}")
print(f"{'TOTAL':70s} {sum(r[1] for r in rows):6d} {tot/1e6:10.3f}")
