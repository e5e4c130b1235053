"""Condense the rocprofv3 CSVs written by tools/profile_bench.sh: kernel stats (top kernels) and per-kernel PMC means."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total Duration (ns)", 0)) or 0))
    res["kernel_stats_top"] = rows[:14]
    with open(os.path.join(out, "kernel_stats.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
pmc = defaultdict(lambda: defaultdict(list))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d) or d.endswith("pmc_generic"):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (did, k, c), v in per.items():
            pmc[k][c].append(v)
short = lambda k: (re.search(r"(k_\w+(<[^>]*>)?)", k) or re.search(r"(\w+)", k)).group(1)
res["pmc_mean_per_dispatch"] = {short(k): {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())}
                                for k, cs in pmc.items() if "hgp" in k or "k_" in k}
gen = defaultdict(lambda: defaultdict(list))          # the forced-generic pass (HGP_PAIRS_GENERIC=1): k_pairs<8, false> on the headline pairs
for f in glob.glob(os.path.join(out, "pmc_generic", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (did, k, c), v in per.items():
        gen[k][c].append(v)
res["pmc_generic_mean_per_dispatch"] = {short(k): {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in gen.items() if "k_pairs" in k}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, cs in res["pmc_mean_per_dispatch"].items():
    print(k, {c: (round(v) if isinstance(v, float) else v) for c, v in cs.items()})
