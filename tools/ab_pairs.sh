#!/bin/bash
# one GPU call: test suite + A/B timing of the pair kernels against a baseline build (round 3)
set -o pipefail
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gpu_tests_f.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03_gpu_tests_f.log
{
for lib in "" build/probe/libhgp_diag0.so; do
  timeout -k 5 120 python tools/time_pairs.py 128 2048 8 $lib
  timeout -k 5 120 python tools/time_pairs.py 90 2048 8 $lib
  timeout -k 5 120 python tools/time_pairs.py 128 2048 32 $lib
done
HGP_PAIRS_GENERIC=1 timeout -k 5 120 python tools/time_pairs.py 128 2048 8
HGP_PAIRS_GENERIC=1 timeout -k 5 120 python tools/time_pairs.py 90 2048 8
} > $O/r03_ab_pairs.txt 2>&1
grep -v amdgpu.ids $O/r03_ab_pairs.txt
