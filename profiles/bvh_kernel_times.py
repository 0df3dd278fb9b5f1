"""Per-kernel totals and per-level durations of one GPU BVH build from a rocprofv3 kernel trace (rocpd database).

    rocprofv3 --kernel-trace -d OUT -o bvh -- python3 profiles/bvh_build_time.py --sizes 20000000 --repeats 1
    python3 profiles/bvh_kernel_times.py OUT/bvh_results.db
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
for name, calls, ms, avg in rows:
    short = name.replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{short:28s} calls {calls:4d}  total {ms:8.3f} ms  avg {avg:9.1f} us  {100 * ms / total:5.1f} %")
print(f"all kernels {total:.3f} ms")
for k in sys.argv[2:]:
    d = c.execute("select (end-start)/1e3 from kernels where name like ? order by start", (f"%{k}%",)).fetchall()
    print(k, "per launch, us:", " ".join(f"{x[0]:.0f}" for x in d))
