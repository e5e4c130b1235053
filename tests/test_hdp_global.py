"""Host-side HDP pseudo-counts of the drop-in surface (hdpgpc_amd/hdp_global.py: the bnpy surrogate bound on (rho, omega),
L-BFGS-B) against what the reference computed in reload_model_from_labels on MIT-BIH record 102
(tests/golden/reload_r102.npz, reload_r102_2leads.npz): counts from the labels -> theta -> two rounds of optimisation.
Pure NumPy / SciPy: runs without a GPU."""
import numpy as np
import pytest
from scipy.special import digamma

from conftest import golden


@pytest.mark.parametrize("fixture", ["reload_r102.npz", "reload_r102_2leads.npz"])
def test_rho_omega_and_theta_match_the_reference(fixture):
    from hdpgpc_amd import hdp_global
    g = golden(fixture)
    lab, M = g["labels"], int(g["M"])
    start = np.zeros(M)
    start[lab[0]] = 1.0
    trans = np.zeros((M, M))
    np.add.at(trans, (lab[:-1], lab[1:]), 1.0)
    gamma, trans_alpha, start_alpha, kappa = 1.0, 1.0, 0.1, 0.0          # hdp_hyp = 'balanced', the constructor default (GPI_HDP.py:274-291)
    rho = hdp_global.create_initrho(M)                                   # reinit_global_params, then two rounds (GPI_HDP.py:4006-4011)
    omega = (1.0 + gamma) * np.ones(M)
    for _ in range(2):
        tt, st = hdp_global.calc_theta_full(trans, start, M + 1, rho, trans_alpha, start_alpha, kappa)
        e_log_pi = digamma(tt) - np.log(np.sum(np.exp(digamma(tt)), axis=1) + 1e-5)[:, None]
        s_log_pi = digamma(st) - np.log(np.sum(np.exp(digamma(st))) + 1e-5)
        rho, omega, _ = hdp_global.find_optimum_rho_omega(np.sum(e_log_pi, axis=0), start_alpha * s_log_pi, M + 1, gamma,
                                                          trans_alpha, kappa, rho, omega)
    assert np.allclose(tt, g["transTheta"], rtol=1e-9) and np.allclose(st, g["startTheta"], rtol=1e-9)
    assert np.allclose(rho, g["rho"], rtol=1e-9) and np.allclose(omega, g["omega"], rtol=1e-9)


def test_lean_lbfgsb_driver_is_scipys_fmin_l_bfgs_b_bit_for_bit():
    """hdp_global._fmin_l_bfgs_b calls the same L-BFGS-B routine with the same arguments as scipy.optimize.fmin_l_bfgs_b (no bounds,
    fun returns (f, g)): iterates, value, evaluation and iteration counts and the warning flag must be identical."""
    import scipy.optimize
    from hdpgpc_amd import hdp_global as H
    rng = np.random.default_rng(0)
    for trial in range(40):
        K = int(rng.integers(2, 30))
        a, b = rng.uniform(0.5, 3, K), rng.normal(size=K)
        Q = rng.normal(size=(K, K))
        Q = Q @ Q.T / K + np.eye(K)

        def fun(c):
            e = np.exp(0.3 * c)
            return 0.5 * c @ Q @ c + b @ c + a @ e + np.sum(np.log1p(c * c)), Q @ c + b + 0.3 * a * e + 2 * c / (1 + c * c)

        x0 = rng.normal(size=K)
        for factr in (1e5, 1e7, 1e10):
            x1, f1, d1 = scipy.optimize.fmin_l_bfgs_b(fun, x0, factr=factr)
            x2, f2, d2 = H._fmin_l_bfgs_b(fun, x0, factr)
            assert np.array_equal(x1, x2) and f1 == f2
            assert (d1["warnflag"], d1["funcalls"], d1["nit"]) == (d2["warnflag"], d2["funcalls"], d2["nit"])
