#!/bin/bash
# quick A/B of the pair kernels on the GPU box: edge-case tests (band == generic bit for bit) + timings
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for a in "128 2048 8" "90 2048 8" "128 2048 32"; do timeout -k 5 120 python tools/time_pairs.py $a $1 2>&1 | grep -v amdgpu.ids; done
