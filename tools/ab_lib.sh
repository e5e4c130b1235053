#!/bin/bash
# A/B of a variant library (arg 1) against the shipped one: whole GPU suite on the variant, then timings of both
set -o pipefail
LIBV=$1
O=gpurun_out
HGP_LIB=$LIBV timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/ab_lib_tests.log 2>&1; echo "pytest(variant) rc=$?"; tail -3 $O/ab_lib_tests.log
{
for lib in "" $LIBV; do
  for a in "128 2048 8" "90 2048 8" "256 1024 16" "64 2048 8"; do timeout -k 5 120 python tools/time_pairs.py $a $lib; done
  HGP_LIB=$lib timeout -k 5 200 python tools/time_matlik.py 2>&1 | tail -12
done
} > $O/ab_lib.txt 2>&1
grep -v amdgpu.ids $O/ab_lib.txt
