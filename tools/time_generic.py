"""k_pairs<8, true> against k_pairs<8, false> (bench.py secondary_pairs_generic alone):  python tools/time_generic.py"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import synthetic_workload as synth
from hdpgpc_amd import ops
r = bench.secondary_pairs_generic(torch.device("cuda", 0), ops, synth)
print(json.dumps({k: v for k, v in r.items() if k != "workload"}, indent=1))
