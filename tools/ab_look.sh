#!/bin/bash
# A/B in one GPU call: shipped library against variant libraries (tools/build_variant.sh):  tools/ab_look.sh NAME...
for name in "" "$@"; do
  lib=${name:+hdpgpc_amd/lib/ab/libhgp_$name.so}
  echo "== lib: ${lib:-shipped}"
  for rep in 1 2; do
    for a in "128 2048 8" "90 2048 8" "64 2048 8" "256 2048 16"; do timeout -k 5 120 python tools/time_pairs.py $a $lib 2>&1 | tail -1 || exit 1; done
  done
  HGP_LIB=$lib timeout -k 5 200 python tools/time_matlik.py 2>&1 | grep -E "b=16384 T=90|b=2272 T=90|T=128|T=256" | head -12
done
