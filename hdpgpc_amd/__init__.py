"""hdpgpc_amd: MI355X-native GP-emission hot path of HDP-GPC (kernel Gram, Cholesky factor/solve,
Gaussian log-likelihood for every (segment, cluster) pair) behind the reference's Python API names.

HIP kernels (hdpgpc_amd/csrc) are reached through the C-ABI of include/hdpgpc_hip.h via ctypes.
There is no CPU fallback: importing `hdpgpc_amd.ops` without the built library raises ImportError.
"""
__version__ = "0.1.0"
