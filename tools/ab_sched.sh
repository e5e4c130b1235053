#!/bin/bash
# A/B of the diag16_acc instruction orders (HGP_DIAG_SCHED 0 / 1 / shipped) in one GPU call: per-pair kernels and the member paths
for lib in "" hdpgpc_amd/lib/ab/libhgp_sched0.so hdpgpc_amd/lib/ab/libhgp_sched1.so; do
  for rep in 1 2; do
    for a in "128 2048 8" "90 2048 8" "256 2048 16"; do timeout -k 5 120 python tools/time_pairs.py $a $lib 2>&1 | tail -1; done
  done
  HGP_LIB=$lib timeout -k 5 200 python tools/time_matlik.py 2>&1 | grep -E "T=90|T=256" | head -8
done
