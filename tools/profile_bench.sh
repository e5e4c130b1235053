#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/profile_bench.sh TAG): rocprofv3 kernel trace + PMC passes of bench.py.
# Summaries land in gpurun_out/prof_TAG/; copy the ones to be judged into profiles/.
set -o pipefail
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "[profile_bench] kernel trace $(date +%T)"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-offline > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
# one rocprofv3 run per counter group: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 ("exceeds the capabilities")
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  echo "[profile_bench] pmc pass $i: $grp  $(date +%T)"
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_$i.log 2>&1 || { echo "pass $i failed"; grep -v "^    @" $OUT/pmc_$i.log | tail -4; }
done
# the mask-driven pair kernel on the same pairs (HGP_PAIRS_GENERIC=1): its executed MFMAs and VALU instructions
export HGP_PAIRS_GENERIC=1
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $OUT/pmc_generic -o pmc -- python3 $R/tools/time_pairs.py 128 2048 8 > $OUT/pmc_generic.log 2>&1 || echo "generic pass failed"
unset HGP_PAIRS_GENERIC
python3 $R/tools/summarize_profiles.py $OUT
# the producer (member chains of the offline loop) on the first 600 beats of record 100: kernel trace
echo "[profile_bench] offline loop kernel trace $(date +%T)"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_offline -o kt -- python3 $R/tools/time_offline.py 600 > $OUT/kt_offline.log 2>&1 || tail -5 $OUT/kt_offline.log
f=$(find $OUT/kt_offline -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/offline_kernel_stats.csv
tail -60 $OUT/kt_offline.log | grep "include_batch on" || true
# the online step (configs[4]): kernel trace of the 24-beat T = 256 fixture, run twice (warm-up + timed)
echo "[profile_bench] online step kernel trace $(date +%T)"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_online -o kt -- python3 $R/tools/time_online.py > $OUT/kt_online.log 2>&1 || tail -5 $OUT/kt_online.log
grep "ms per beat" $OUT/kt_online.log || true
