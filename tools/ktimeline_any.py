"""Kernel timeline of one steady-state iteration of ANY looped program from a rocprofv3 rocpd database: anchors on
the k_preprocess_fwd launches.  usage: ktimeline_any.py <db> [frames_from_end=40]"""
import sqlite3
import sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = sqlite3.connect(db)
rows = list(c.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
idx = [i for i, r in enumerate(rows) if "k_preprocess_fwd" in r[0]]
i0, i1 = idx[-back], idx[-back + 1]
t0 = rows[i0][1]
print(f"period: {(rows[i1][1] - t0) / 1000.0:.1f} us, {i1 - i0} kernels")
prev = None
for name, st, en in rows[i0:i1]:
    gap = (st - prev) / 1000.0 if prev is not None else 0.0
    short = name.replace("_ZN2fr", "").replace("_ZN2at6native", "at::")[:70]
    print(f"{(st - t0) / 1000.0:8.1f} us  dur {(en - st) / 1000.0:6.1f}  gap {gap:6.1f}  {short}")
    prev = en
