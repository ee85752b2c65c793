"""profiles/blend_bwd_counters.json from the PMC summaries of tools/pmc.sh (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU passes).

    python tools/make_counters_json.py <tag> <fetch pmc.txt> <write pmc.txt> <sq pmc.txt> [P res]

HBM bytes per launch as MI355X_MICROARCH.md prescribes: counters in their own passes; FETCH_SIZE (KB) doubled for this
kernel's wide (16 B per lane) coalesced reads — gfx950's rocprofv3 reports half of those; WRITE_SIZE (KB) as reported."""
import json
import re
import sys

tag, fpath, wpath, spath = sys.argv[1:5]
P, res = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (100000, 512)
KERNEL = "k_unit_blend_bwd_sparse"


def per_launch(path, counter):
    cur = None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        elif cur and KERNEL in cur and line.split()[0] == counter:
            m = re.search(r"per_launch_total=\s*([0-9.]+)", line)
            return float(m.group(1))
    raise SystemExit(f"{counter} of {KERNEL} not found in {path}")


f_kb, w_kb = per_launch(fpath, "FETCH_SIZE"), per_launch(wpath, "WRITE_SIZE")
valu = per_launch(spath, "SQ_INSTS_VALU")
out = {"P": P, "res": res, "kernel": KERNEL, "collected": tag,
       "FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
       "hbm_bytes_per_launch": int(round((2 * f_kb + w_kb) * 1024)), "sq_insts_valu_per_launch": int(round(valu)),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU in separate passes over tools/probe.py (per-launch totals = "
               "all counter instances of a dispatch added up); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half "
               "of wide coalesced reads), WRITE_SIZE as reported"}
json.dump(out, open("profiles/blend_bwd_counters.json", "w"), indent=1)
print(json.dumps(out))
