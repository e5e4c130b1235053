"""Register / spill / scratch table of one translation unit:  python tools/kernel_resources.py hgp_pairs [-DFLAG...]"""
import re
import subprocess
import sys

f, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-mllvm", "-pragma-unroll-threshold=1048576", "-fPIC", "-Wno-unused-result",
       *extra, "-Rpass-analysis=kernel-resource-usage", "-c", "-o", f"/tmp/kr_{f}_{len(extra)}.o", f"hdpgpc_amd/csrc/{f}.hip"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.splitlines():
    if "error" in l:
        print(l.strip())
    m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", l)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    g = lambda k: r.get(k, "?")  # noqa: E731
    print(f"{r['name'][:64]:64s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>4} SGPRspill {g('SGPRs Spill'):>4} VGPRspill {g('VGPRs Spill'):>4} "
          f"scratch {g('ScratchSize [bytes/lane]'):>5} occ {g('Occupancy [waves/SIMD]')}")
