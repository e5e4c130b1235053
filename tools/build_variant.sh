#!/bin/bash
# Build an A/B variant of libhdpgpc_hip.so:  tools/build_variant.sh NAME -DFLAG...  ->  build/probe/libhgp_NAME.so
set -e
name=$1; shift
od=build/probe/obj_$name
mkdir -p $od
FLAGS="-O3 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1048576 -fPIC -Wno-unused-result"
for f in hgp_kernels hgp_pairs hgp_pairs_acc hgp_matlik hgp_assign hgp_warp hgp_chain; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $od/$f.o hdpgpc_amd/csrc/$f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/probe/libhgp_$name.so $od/*.o
echo built build/probe/libhgp_$name.so
