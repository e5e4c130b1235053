"""Copy the summaries of a tools/profile_bench.sh run (gpurun_out/prof_TAG) into profiles/<round>_* and refresh pairs_traffic.json:
    python tools/bank_profiles.py TAG [ROUND=r04]"""
import datetime, json, shutil, subprocess, sys
tag = sys.argv[1]
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"
P = f"gpurun_out/prof_{tag}"
shutil.copy(P + "/bench.json", f"profiles/{RND}_bench.json")
shutil.copy(P + "/kernel_stats.csv", f"profiles/{RND}_kernel_stats.csv")
shutil.copy(P + "/offline_kernel_stats.csv", f"profiles/{RND}_offline600_kernel_stats.csv")
shutil.copy(P + "/summary.json", f"profiles/{RND}_pmc_summary.json")
s = json.load(open(P + "/summary.json"))
p = s["pmc_mean_per_dispatch"]["k_pairs<8, true>"]
t = json.load(open("profiles/pairs_traffic.json"))
t["FETCH_SIZE_KB_raw"] = p["FETCH_SIZE"]
t["WRITE_SIZE_KB"] = p["WRITE_SIZE"]
t["hbm_bytes_per_launch"] = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
t["sq"] = {k: v for k, v in p.items() if k.startswith("SQ_")}
t["mfma_f64_instructions_per_pair"] = round(p["SQ_INSTS_VALU_MFMA_F64"] / 16384)
t["source"] = f"profiles/{RND}_pmc_summary.json (tools/profile_bench.sh {tag}: one rocprofv3 --pmc pass per counter group)"
# bench.py copies these two into its JSON line: the counters are NOT measured in the driver's run, the line says which code they belong to
t["measured_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() + " (+ working tree)"
t["measured_date"] = datetime.date.today().isoformat()
json.dump(t, open("profiles/pairs_traffic.json", "w"), indent=1)
b = json.load(open(P + "/bench.json"))
n = 16384
print("value", b["value"], "ms/step", b["ms_per_step"], "frac", b["roofline"]["frac"], "kernel_ms", b["roofline"]["kernel_ms"])
for k in ["secondary", "secondary_pairs_T90", "secondary_large_T", "strong_scaling_base", "secondary_rank1"]:
    print(k, b[k].get("value"), b[k].get("kernel_ms", b[k].get("ms_per_step")), b[k]["roofline"]["frac"])
m, M = b["secondary_matrix_terms"], b["secondary_matrix_terms_T256"]
print("a8/a9 T90", m["a8"]["value"], m["a9"]["value"], "T256", M["a8"]["value"], M["a8"]["roofline"]["frac"], M["a9"]["value"], M["a9"]["roofline"]["frac"])
print("offline", b["offline_r100"]["offline_r100_s"], "cpu", b["cpu_baseline"]["value"])
print("mfma/pair", p["SQ_INSTS_VALU_MFMA_F64"] / n, "valu", (p["SQ_INSTS_VALU"] - p["SQ_INSTS_VALU_MFMA_F64"]) / n, "busy",
      p["SQ_VALU_MFMA_BUSY_CYCLES"] / (p["SQ_WAVE_CYCLES"] * 4), "wavecyc/pair", p["SQ_WAVE_CYCLES"] * 4 / n, "hbm", t["hbm_bytes_per_launch"])
