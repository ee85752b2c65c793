"""Per-kernel average of PMC counters from a rocprofv3 rocpd database."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
names = {r[0]: r[1] for r in c.execute("select id, name from rocpd_info_pmc")}
kern = {r[0]: r[1] for r in c.execute("select d.event_id, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id")}
acc = defaultdict(lambda: defaultdict(list))
for ev, pmc, val in c.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
    if ev in kern:
        acc[kern[ev]][names.get(pmc, str(pmc))].append(val)
for k, d in acc.items():
    print(k[:90])
    for n, v in sorted(d.items()):
        print(f"    {n:28s} avg={sum(v)/len(v):16.1f} n={len(v)}")
