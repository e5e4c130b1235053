#!/bin/bash
# Build an A/B variant of libhdpgpc_hip.so:  tools/build_variant.sh NAME -DFLAG...  ->  hdpgpc_amd/lib/ab/libhgp_NAME.so
# (hdpgpc_amd/lib/ab/ travels to the GPU box, build/ does not; both are git-ignored)
set -e
name=$1; shift
od=build/probe/obj_$name
mkdir -p $od hdpgpc_amd/lib/ab
FLAGS="-O3 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1048576 -fPIC -Wno-unused-result"
for f in hgp_kernels hgp_pairs hgp_pairs_acc hgp_matlik hgp_matlik_coop hgp_assign hgp_warp hgp_chain; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $od/$f.o hdpgpc_amd/csrc/$f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hdpgpc_amd/lib/ab/libhgp_$name.so $od/*.o
echo built hdpgpc_amd/lib/ab/libhgp_$name.so
