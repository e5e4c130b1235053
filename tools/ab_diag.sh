#!/bin/bash
# one GPU call: probe + test suite + A/B timings of the diag16 variants (round 3)
set -o pipefail
O=gpurun_out
timeout -k 5 60 ./tools/probe_diag16 > $O/r03_probe_diag16.txt 2>&1
cat $O/r03_probe_diag16.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r03_gpu_tests_d.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03_gpu_tests_d.log
{
for lib in "" build/probe/libhgp_diag0.so build/probe/libhgp_diag1.so; do
  timeout -k 5 120 python tools/time_pairs.py 128 2048 8 $lib
  timeout -k 5 120 python tools/time_pairs.py 90 2048 8 $lib
  timeout -k 5 120 python tools/time_pairs.py 256 1024 16 $lib
done
for lib in "" build/probe/libhgp_diag0.so; do
  echo "== quick_time lib=$lib"; HGP_LIB=$lib timeout -k 5 300 python tools/quick_time.py
  echo "== time_matlik lib=$lib"; HGP_LIB=$lib timeout -k 5 300 python tools/time_matlik.py
done
} > $O/r03_ab_diag.txt 2>&1
cat $O/r03_ab_diag.txt
