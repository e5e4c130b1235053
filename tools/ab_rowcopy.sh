#!/bin/bash
# A/B of diag16_acc's row copies (HGP_DIAG_ROWCOPY 0 = ds_bpermute, shipped / 1 = through LDS memory) in one GPU call
for lib in "" hdpgpc_amd/lib/ab/libhgp_rowcopy1.so; do
  echo "== lib: ${lib:-shipped}"
  for rep in 1 2; do
    for a in "128 2048 8" "90 2048 8" "256 2048 16"; do timeout -k 5 120 python tools/time_pairs.py $a $lib 2>&1 | tail -1 || exit 1; done
  done
  HGP_LIB=$lib timeout -k 5 200 python tools/time_matlik.py 2>&1 | grep -E "T=90|T=128|T=256" | head -12
done
