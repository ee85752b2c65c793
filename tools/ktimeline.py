"""Print the kernel timeline of one steady-state frame from a rocprofv3 rocpd database: start offset, duration
and the idle gap before each kernel.  usage: ktimeline.py <db> [frame_index_from_end=60]"""
import sqlite3
import sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 60
c = sqlite3.connect(db)
q = """select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""
rows = list(c.execute(q))
# frames start at k_preprocess_fwd's predecessor zero kernel; use k_preprocess_fwd as the anchor
idx = [i for i, r in enumerate(rows) if "k_preprocess_fwd" in r[0]]
i0, i1 = idx[-back], idx[-back + 1]
t0 = rows[i0][1]
prev_end = None
print(f"frame period: {(rows[i1][1] - t0) / 1000.0:.1f} us")
busy = 0.0
for name, st, en in rows[i0:i1]:
    gap = (st - prev_end) / 1000.0 if prev_end is not None else 0.0
    busy += (en - st) / 1000.0
    short = name.replace("_ZN2fr", "").replace("_ZN2at6native", "at::")[:44]
    print(f"{(st - t0) / 1000.0:8.1f} us  dur {(en - st) / 1000.0:6.1f}  gap {gap:6.1f}  {short}")
    prev_end = en
print(f"busy {busy:.1f} us")
