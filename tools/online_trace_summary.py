"""Summarise a rocprofv3 --kernel-trace --stats run of tools/time_online.py (two runs of the fixture: warm-up + timed) into
profiles/<tag>_online_kernel_stats.{csv,json}:  python tools/online_trace_summary.py gpurun_out/prof_DIR TAG BEATS"""
import csv
import json
import os
import shutil
import subprocess
import sys

src, tag, beats = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
f = os.path.join(src, "kt_kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
n = sum(int(r["Calls"]) for r in rows)
ours = [r for r in rows if "anonymous namespace" in r["Name"] and "at::native" not in r["Name"]]
out = {"what": "rocprofv3 --kernel-trace --stats of tools/time_online.py (fixture run twice: warm-up + timed run)",
       "commit": subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
       "beats_traced": 2 * beats, "launches_per_beat": n / (2 * beats), "kernel_ms_per_beat": tot_ns / 1e6 / (2 * beats),
       "library_kernel_launches_per_beat": sum(int(r["Calls"]) for r in ours) / (2 * beats),
       "top": [{"kernel": r["Name"].replace("(anonymous namespace)::", "")[:80], "calls_per_beat": int(r["Calls"]) / (2 * beats),
                "avg_us": float(r["AverageNs"]) / 1e3, "share": float(r["Percentage"])} for r in rows[:12]]}
shutil.copy(f, os.path.join(root, "profiles", f"{tag}_online_kernel_stats.csv"))
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_online_kernel_stats.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "top"}))
