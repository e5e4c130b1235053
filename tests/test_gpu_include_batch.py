"""SURVEY.md 8b Face 1 / BASELINE configs[0]: GPI_HDP.include_batch - the offline variational loop - on the HIP kernels,
driven as hdpgpc/tests/test_offline.py:32-79 drives the reference and compared with the trace of the reference's own run
(tests/golden/include_batch_*.npz, written by tests/golden/make_golden.py ib80 / ib100): the loop must take the SAME decisions
(identical sequence of proposals, identical assignments after every estimate_q_all and every EM iteration) and compute the same
numbers (every bound evaluation, every full_pass_weighted, q and q_lat of every EM iteration)."""
import time

import numpy as np
import pytest
import torch

from conftest import golden
from offline_trace import compare_trace, run_traced

pytestmark = pytest.mark.gpu


def test_include_batch_r100_first80():
    g = golden("include_batch_r100_n80.npz")
    sw, tr = run_traced(g, g["y"])
    worst = compare_trace(g, sw, tr, q_tol=1e-8)
    print(f"include_batch, 80 beats: {len(tr['order'])} traced calls, worst relative error {worst:.2e}")


def test_include_batch_warp_first80():
    """include_batch(warp=True) (GPI_HDP.py:3412-3525 inside the loop): every segment is warped onto the representative of every
    cluster column by the batched Adam fit (hgp_warp_batch_f64) before it is scored.  The reference fits the same warps with
    torch.optim.Adam one segment at a time (fitted warps agree to 1e-6, tests/test_gpu_warp_batch.py); observed on the scores: 1.8e-9;
    the decisions must be identical."""
    g = golden("include_batch_r100_n80_warp.npz")
    assert bool(g["warp"])
    sw, tr = run_traced(g, g["y"], warp=True)
    worst = compare_trace(g, sw, tr, q_tol=1e-7)
    print(f"include_batch(warp=True), 80 beats: {len(tr['order'])} traced calls, worst relative error {worst:.2e}")


def test_include_batch_r100_full():
    """BASELINE configs[0]: the whole record (2 272 beats) - final counts [2271, 1] as the reference."""
    g = golden("include_batch_r100.npz")
    y = golden("mitbih100_lead0.npz")["y"]
    torch.cuda.synchronize()
    t0 = time.time()
    sw, tr = run_traced(g, y)
    torch.cuda.synchronize()
    wall = time.time() - t0
    worst = compare_trace(g, sw, tr, q_tol=1e-7)
    print(f"include_batch, record 100: {wall:.1f} s (reference on 8 vCPU: {float(g['wall_s']):.0f} s), worst relative error {worst:.2e}")


def test_include_batch_r102_full():
    """BASELINE configs[2]: the whole of record 102, lead 0 (2 187 beats, T = 90) as hdpgpc/tests/test_offline.py:32-79 drives it
    (make_golden.py ib102: 486 traced calls, 10 EM iterations) - final counts [2009, 113, 37, 14, 6, 4, 3, 1] as SURVEY.md section 6
    reports for the reference; every decision identical."""
    g = golden("include_batch_r102.npz")
    y = golden("mitbih102_lead0.npz")["y"]
    torch.cuda.synchronize()
    t0 = time.time()
    sw, tr = run_traced(g, y)
    torch.cuda.synchronize()
    wall = time.time() - t0
    worst = compare_trace(g, sw, tr, q_tol=1e-7)
    assert [len(m.indexes) for m in sw.gpmodels[0]] == [2009, 113, 37, 14, 6, 4, 3, 1]
    print(f"include_batch, record 102: {wall:.1f} s (reference on 8 vCPU: {float(g['wall_s']):.0f} s, SURVEY: 348 s), "
          f"{len(tr['order'])} traced calls, worst relative error {worst:.2e}")


def test_print_results_after_include_batch(capsys):
    from hdpgpc.util_plots import print_results
    g = golden("include_batch_r100_n80.npz")
    labels = golden("mitbih100_lead0.npz")["labels"]
    sw, _ = run_traced(g, g["y"])
    main_model = print_results(sw, labels, 0, error=False)
    assert len(main_model) == int(g["M_final"]) and sw.selected_gpmodels() == list(range(int(g["M_final"])))


def test_cluster_new_batch_learning_two_leads():
    """cluster_new_batch(learning=True) as hdpgpc/tests/test_offline_multi_output_load.py:85 calls it, BOTH leads: 300 beats of
    record 102 rebuilt from their labels, 20 new beats classified and the EM loop re-entered on all 320 (the reference's run
    ends in a NameError at its stop condition, GPI_HDP.py:3139; its state at that point is the fixture).  Decisions identical;
    numbers within 50 x the reference's OWN change under a sub-ulp perturbation of the inputs (stored in the fixture: 1.9e-7 -
    lead 1 is ill-conditioned, see test_gpu_dropin.py)."""
    from offline_trace import run_cluster_learning
    g = golden("cluster_learning_r102_2leads.npz")
    assert bool(g["ref_pert_same_decisions"])
    sw, tr = run_cluster_learning(g)
    worst = compare_trace(g, sw, tr, q_tol=max(1e-8, 50.0 * float(g["ref_sens"])))
    print(f"cluster_new_batch(learning=True), 2 leads: {len(tr['order'])} traced calls, worst relative error {worst:.2e}")


def test_include_batch_two_leads():
    """hdpgpc/tests/test_offline_multi_output.py's configuration (both leads, one model per (lead, cluster), SNR-weighted
    combination of the leads) on the first 100 beats of record 102: 334 traced calls, 8 EM iterations, six clusters.  Gate as
    for the other two-lead fixtures: 50 x the reference's own change under a sub-ulp perturbation of the inputs."""
    g = golden("include_batch_r102_2leads_n100.npz")
    assert bool(g["ref_pert_same_decisions"])
    sw, tr = run_traced(g, g["y"])
    worst = compare_trace(g, sw, tr, q_tol=max(1e-8, 50.0 * float(g["ref_sens"])))
    print(f"include_batch, two leads: {len(tr['order'])} traced calls, worst relative error {worst:.2e}")
