"""CPU-side checks of the boundary: the shared library loads and exports every symbol the header declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "hdpgpc_amd", "lib", "libhdpgpc_hip.so")


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "hdpgpc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hgp_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_path():
    syms = header_symbols()
    for s in ("hgp_gram_rbf_f64", "hgp_potrf_batched_f64", "hgp_score_groups_f64", "hgp_loglik_pairs_f64",
              "hgp_pairs_plan_create", "hgp_pairs_plan_update"):
        assert s in syms


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (hgp_[a-z0-9_]+)", out))
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, missing


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built (run __graft_entry__.build())")
def test_ctypes_binding_loads_without_a_gpu():
    from hdpgpc_amd import _ffi
    assert _ffi.lib.hgp_abi_version() == 6
    assert set(_ffi.EXPORTS) == set(header_symbols())
    # argument validation happens before any HIP call
    assert _ffi.lib.hgp_gram_rbf_f64(None, 4, None, 4, 1.0, 1.0, 0.0, None, None) == -1
    assert _ffi.lib.hgp_pairs_plan_device_bytes(0, 8, 2) == 0
    assert _ffi.lib.hgp_pairs_plan_device_bytes(128, 128, 8) > 8 * 7 * 128 * 128 * 8
