"""Host logic of GPI_HDP.include_batch (SURVEY.md 8b Face 1) on CPU: the control flow of the offline variational loop -
proposals, accept / reject decisions, bookkeeping, HDP terms - driven through tests/cpu_double.py (a NumPy / oracle stand-in
for the kernels, test infrastructure only) and compared with the trace of the REFERENCE's own run on the first 80 beats of
MIT-BIH record 100 (tests/golden/include_batch_r100_n80.npz, written by tests/golden/make_golden.py ib80).  The same trace is
replayed on the GPU, through the HIP kernels, in tests/test_gpu_include_batch.py."""
import numpy as np
import pytest
import torch

from conftest import golden

import cpu_double
from offline_trace import run_traced, compare_trace


def test_include_batch_trace_cpu(monkeypatch):
    cpu_double.install(monkeypatch)
    g = golden("include_batch_r100_n80.npz")
    sw, tr = run_traced(g, g["y"])
    compare_trace(g, sw, tr, q_tol=1e-8)


def test_hdp_elbo_terms_finite():
    from hdpgpc_amd import hdp_global as hg
    rho = hg.create_initrho(3)
    omega = 2.0 * np.ones(3)
    tc = np.array([[5.0, 1, 0, 0], [1, 3, 1, 0], [0, 1, 2, 0], [0, 0, 0, 0]])
    sc = np.array([1.0, 0, 0, 0])
    tt, st = hg.calc_theta_full(tc, sc, 4, rho, 1.0, 0.1, 0.0)
    v = hg.elbo_linear_terms(rho, omega, 1.0, 0.1, 0.0, 1.0, tt, st, sc, tc)
    assert np.isfinite(v)
    resp = np.eye(3)[[0, 0, 1, 2, 0]]
    pair = np.zeros((5, 3, 3))
    pair[np.arange(1, 5), [0, 0, 1, 2], [0, 1, 2, 0]] = 1.0
    assert abs(hg.elbo_entropy(resp, pair)) < 1e-20


def test_include_sample_trace_cpu(monkeypatch):
    """Host logic of the online step (GPI_HDP.include_sample) against the reference's own 40-beat run on record 102."""
    from online_trace import compare_online, run_online
    cpu_double.install(monkeypatch)
    g = golden("include_sample_r102_n40.npz")
    _, tr = run_online(g)
    compare_online(g, tr, 1e-9)


def test_cluster_new_batch_learning_two_leads_cpu(monkeypatch):
    """Host logic with two leads (SNR-weighted combination, one model per (lead, cluster)): cluster_new_batch(learning=True)
    against the reference's trace; tolerance as in tests/test_gpu_include_batch.py."""
    from offline_trace import run_cluster_learning
    cpu_double.install(monkeypatch)
    g = golden("cluster_learning_r102_2leads.npz")
    sw, tr = run_cluster_learning(g)
    compare_trace(g, sw, tr, q_tol=max(1e-8, 50.0 * float(g["ref_sens"])))


def test_include_batch_two_leads_cpu(monkeypatch):
    """Host logic of include_batch with two leads against the reference's trace (record 102, 100 beats)."""
    cpu_double.install(monkeypatch)
    g = golden("include_batch_r102_2leads_n100.npz")
    sw, tr = run_traced(g, g["y"])
    compare_trace(g, sw, tr, q_tol=max(1e-8, 50.0 * float(g["ref_sens"])))
