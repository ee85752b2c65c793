"""Per-kernel PMC counters from a rocprofv3 rocpd database: for every counter the average SAMPLE (one per counter
instance: shader engine / XCD / channel) and the average PER-LAUNCH TOTAL (samples of one dispatch added up)."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
names = {r[0]: r[1] for r in c.execute("select id, name from rocpd_info_pmc")}
kern = {r[0]: r[1] for r in c.execute("select d.event_id, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id")}
acc = defaultdict(lambda: defaultdict(list))
per_ev = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for ev, pmc, val in c.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
    if ev in kern:
        n = names.get(pmc, str(pmc))
        acc[kern[ev]][n].append(val)
        per_ev[kern[ev]][n][ev] += val
for k, d in acc.items():
    print(k[:90])
    for n, v in sorted(d.items()):
        tot = list(per_ev[k][n].values())
        print(f"    {n:28s} avg={sum(v)/len(v):16.1f} n={len(v)}   per_launch_total={sum(tot)/len(tot):18.1f} launches={len(tot)}")
