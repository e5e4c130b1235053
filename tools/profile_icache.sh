#!/bin/bash
# Run on the GPU box: instruction-cache counters of bench.py's kernels (k_pairs<8> is 164 KiB of straight-line code; the
# instruction cache is 64 KiB per two CUs).  Output: gpurun_out/prof_icache/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  echo "[profile_icache] pmc pass $i: $grp  $(date +%T)"
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_$i.log 2>&1 || { echo "pass $i failed"; grep -v "^    @" $OUT/pmc_$i.log | tail -4; }
done
python3 $R/tools/summarize_profiles.py $OUT
