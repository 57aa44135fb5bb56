"""Per-kernel statistics (and optional overlap of two kernel families) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace` writes <pid>_results.db on this ROCm build instead of CSV files).
    python profiles/tools/rocpd_stats.py X_results.db [--top 40] [--overlap PATTERN_A PATTERN_B]
"""
import argparse, sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--overlap", nargs=2, default=None)
a = ap.parse_args()
c = sqlite3.connect(a.db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"# {a.db}: {sum(r[1] for r in rows)} dispatches of {len(rows)} kernels, {tot/1e6:.3f} ms of kernel time")
print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'%':>6}  name")
for n, k, s, av, mn, mx in rows[:a.top]:
    print(f"{k:7d} {s/1e6:10.3f} {av/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {100*s/tot:6.2f}  {n[:120]}")
if a.overlap:
    pa, pb = a.overlap
    A = c.execute("select start, end, stream_id from kernels where name like ? order by start", (f"%{pa}%",)).fetchall()
    B = c.execute("select start, end, stream_id from kernels where name like ? order by start", (f"%{pb}%",)).fetchall()
    ov, pairs, j0 = 0, 0, 0
    for s, e, _ in A:
        hit = False
        for t, u, _ in B:
            if u <= s:
                continue
            if t >= e:
                break
            ov += min(e, u) - max(s, t)
            hit = True
        pairs += hit
    ta, tb = sum(e - s for s, e, _ in A), sum(e - s for s, e, _ in B)
    print(f"# overlap: {len(A)} x '{pa}' ({ta/1e6:.3f} ms, streams {sorted(set(x[2] for x in A))}) vs {len(B)} x '{pb}' ({tb/1e6:.3f} ms, streams "
          f"{sorted(set(x[2] for x in B))}): {pairs} of the '{pa}' dispatches ran while a '{pb}' dispatch was in flight, {ov/1e6:.3f} ms of shared time")
