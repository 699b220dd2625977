#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_round.sh into the summaries committed under profiles/:
   <tag>_kernel_stats.csv  per-kernel calls / total / avg / min / max / %  (from the --kernel-trace CSV)
   <tag>_hbm_traffic.json  per-kernel average FETCH_SIZE / WRITE_SIZE bytes per launch (separate --pmc passes), with the gfx950
                           correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B: doubled for wide reads).
usage: profile_summarise.py <gpurun_out/prof_tag dir> <tag>"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
dst = Path(__file__).resolve().parent.parent / "profiles"
dst.mkdir(exist_ok=True)


def short(name: str) -> str:
    return name.split("(")[0][:120]


# ---- kernel trace -> stats
agg = defaultdict(list)
for f in glob.glob(str(src / "stats" / "**" / "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values()) or 1.0
with open(dst / f"{tag}_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), round(sum(v) / 1e3, 3), round(sum(v) / len(v), 2), round(min(v), 2), round(max(v), 2), round(100 * sum(v) / tot, 2)])
print(f"kernel stats: {len(agg)} kernels, {tot / 1e3:.1f} ms of GPU kernel time")

# ---- PMC passes -> bytes per launch
traffic = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(str(src / f"pmc_{c}" / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            a = per[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in per.items():
        traffic.setdefault(k, {"launches": n})[c + "_KB_per_launch"] = v / n
for k, t in traffic.items():
    fk, wk = t.get("FETCH_SIZE_KB_per_launch", 0.0), t.get("WRITE_SIZE_KB_per_launch", 0.0)
    t["hbm_bytes_per_launch"] = (2.0 * fk + wk) * 1024.0       # FETCH doubled (gfx950: 128-B requests tallied at 64 B)
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1`; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); Infinity-Cache hits are counted, so this is an upper bound "
                   "on true HBM bytes", "kernels": traffic}, open(dst / f"{tag}_hbm_traffic.json", "w"), indent=1)
print("traffic:", {k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]})
