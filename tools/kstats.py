"""Print per-kernel statistics from a rocprofv3 rocpd database (kernel-trace)."""
import sqlite3
import sys

db = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = sqlite3.connect(db)
q = """select s.kernel_name, d.end-d.start, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d
join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""
rows = list(c.execute(q))
stats = {}
for name, dur, gx, wx in rows:
    st = stats.setdefault(name, [])
    st.append(dur / 1000.0)
print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_us':>10s}")
tot = 0
for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1][skip:] or kv[1])):
    w = v[skip:] or v
    tot += sum(w)
    print(f"{name[:72]:72s} {len(w):6d} {sum(w)/len(w):9.1f} {min(w):9.1f} {max(w):9.1f} {sum(w):10.1f}")
print("total kernel time us:", round(tot, 1))
