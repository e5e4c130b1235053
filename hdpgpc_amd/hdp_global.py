"""Top level of the HDP-HMM (host side, O(K) work): the stick-breaking weights q(u_k) = Beta(rho_k omega_k, (1 - rho_k) omega_k)
and the Dirichlet pseudo-counts of the transition rows.

The reference takes this step from bnpy (hdpgpc/hdpgpc/OptimizerRhoOmega.py, "adapted from bnpy"; GPI_HDP.py:360-430,
2752-2828); it is control plane, not part of the GP-emission hot path, and only enters the drop-in surface through the
transition / start pseudo-counts that `cluster_new_batch` reads.  What is restated here is the published objective
(Hughes, Stephenson & Sudderth, "Scalable adaptation of state complexity for nonparametric hidden Markov models", 2015,
surrogate bound for E[log Dir]), minimised with scipy's L-BFGS-B in the unconstrained variables
(logit rho, log omega), with the tolerance schedule of the reference's driver of that optimiser.
"""
import warnings

import numpy as np
import scipy.optimize
from scipy.special import digamma, gammaln, zeta

EPS = 1e-8


def _fmin_l_bfgs_b(fun, x0, factr):
    """scipy.optimize.fmin_l_bfgs_b(fun, x0, factr=factr) for an unbounded problem whose `fun` returns (f, g) - the same calls of the
    same L-BFGS-B routine (scipy.optimize._lbfgsb.setulb, m = 10, pgtol = 1e-5, maxls = 20, 15 000 evaluations / iterations), hence
    the same iterates bit for bit, without the per-evaluation wrappers of scipy's driver (ScalarFunction, memoisation, array_equal
    checks, OptimizeResult / inverse-Hessian objects: ~20 us per evaluation and ~0.2 ms per run - the online step runs this four
    times per beat with ~20 evaluations each, tools/time_online.py --profile).  Any surprise in the private interface (another
    SciPy) falls back to the public function."""
    try:
        from scipy.optimize import _lbfgsb
        x = np.array(x0, dtype=np.float64).ravel()
        n, m = x.size, 10
        nbd, low, up = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
        f, g = fun(x.copy())                              # scipy evaluates the start point when it wraps the function
        nfev = 1
        f, g = float(f), np.asarray(g, dtype=np.float64)
        wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        iwa = np.zeros(3 * n, np.int32)
        task, ln_task = np.zeros(2, np.int32), np.zeros(2, np.int32)
        lsave, isave, dsave = np.zeros(4, np.int32), np.zeros(44, np.int32), np.zeros(29, np.float64)
        nit, first = 0, True
        while True:
            g = g.astype(np.float64)
            _lbfgsb.setulb(m, x, low, up, nbd, f, g, factr, 1e-5, wa, iwa, task, lsave, isave, dsave, 20, ln_task)
            if task[0] == 3:
                if first:                                  # the routine's first request is the start point: already evaluated
                    first = False
                else:
                    f, g = fun(x.copy())
                    nfev += 1
                    f, g = float(f), np.asarray(g, dtype=np.float64)
            elif task[0] == 1:
                nit += 1
                if nit >= 15000:
                    task[0], task[1] = 5, 504
                elif nfev > 15000:
                    task[0], task[1] = 5, 502
            else:
                break
        warn = 0 if task[0] == 4 else (1 if (nfev > 15000 or nit >= 15000) else 2)
        return x, f, {"warnflag": warn, "task": f"status {int(task[0])}, message {int(task[1])}", "funcalls": nfev, "nit": nit}
    except (ImportError, TypeError, AttributeError):
        x, f, d = scipy.optimize.fmin_l_bfgs_b(fun, x0, factr=factr)
        return x, f, d


def create_initrho(K):
    """rho whose implied E[beta] is nearly uniform with a small leftover mass (GPI_HDP.py:380-384)."""
    rem = min(0.1, 1.0 / (K * K))
    delta = (-1.0 + rem) * np.arange(0.0, K, 1.0)
    return (1.0 - rem) / (K + delta)


def rho2beta(rho, size="K+1"):
    """E[beta] from the stick fractions (GPI_HDP.py:432-440): K active weights (+ the leftover mass)."""
    rho = np.asarray(rho, dtype=np.float64)
    if size == "K":
        beta = rho.copy()
        beta[1:] *= np.cumprod(1.0 - rho[:-1])
        return beta
    beta = np.append(rho, 1.0)
    beta[1:] *= np.cumprod(1.0 - rho)
    return beta


def calc_theta_full(trans_count, start_count, M, rho, trans_alpha, start_alpha, kappa):
    """Dirichlet pseudo-counts of the M x M transition table and the start vector (GPI_HDP.py:400-421): prior mass
    alpha E[beta] on every column, observed counts (+ sticky kappa) on the leading (M-1) x (M-1) block."""
    rho = np.asarray(rho, dtype=np.float64)
    ebeta = rho2beta(rho, "K" if M == rho.shape[0] else "K+1")
    trans = np.zeros((M, M)) + trans_alpha * ebeta[None, :]
    tc = np.asarray(trans_count, dtype=np.float64)
    trans[:M - 1, :M - 1] += tc[:M - 1, :M - 1] + kappa * np.eye(M - 1)
    start = start_alpha * ebeta
    start[:M - 1] += np.asarray(start_count, dtype=np.float64)[:M - 1]
    return trans, start


def _objective(rhoomega, T_vec, n_doc, gamma, kappa, K):
    """Negative surrogate ELBO in (rho, omega) and its gradient.  (One call of each special function on the concatenated
    arguments, no Python loop over the sticks: the online step runs this optimisation four times per beat and ~20 evaluations
    per run - at K = 20 the evaluation was 0.1 ms of NumPy call overhead.)"""
    rho, omega = rhoomega[:K], rhoomega[K:]
    g1, g0 = rho * omega, (1.0 - rho) * omega
    args = np.concatenate((omega, g1, g0))
    dg = digamma(args)
    tri = zeta(2.0, args)                       # polygamma(1, x) = zeta(2, x): what scipy.special.polygamma evaluates
    dg_omega = dg[:K]
    e_log_u, e_log_1mu = dg[K:2 * K] - dg_omega, dg[2 * K:] - dg_omega
    kv = np.arange(K, 0, -1, dtype=np.float64)
    if kappa > 0:
        scale = 1.0
        on, off = K + 1.0 - g1, K * kv + 1.0 + gamma - g0
    else:
        scale = float(n_doc)
        on, off = 1.0 + (1.0 - g1) / scale, kv + (gamma - g0) / scale
    one_m = 1.0 - rho
    ebeta = np.empty(K + 1)
    ebeta[:K] = rho
    ebeta[K] = 1.0
    ebeta[1:] *= np.cumprod(one_m)
    gl = gammaln(np.concatenate((g1 + g0, g1, g0)))
    c_beta = np.sum(gl[:K] - gl[K:2 * K] - gl[2 * K:])
    w = ebeta * T_vec
    elbo = -c_beta / scale + on @ e_log_u + off @ e_log_1mu + np.sum(w)
    tri_o, tri_1, tri_0 = tri[:K], tri[K:2 * K], tri[2 * K:]
    # d E[beta_j] / d rho_k: E[beta_k] / rho_k on the diagonal, -E[beta_j] / (1 - rho_k) for j > k
    tail = np.cumsum(w[::-1])[::-1]             # tail[k] = sum_{j >= k} E[beta_j] T_j
    g_rho = on * omega * tri_1 - off * omega * tri_0 + (w[:K] / rho - tail[1:] / one_m)
    g_omega = on * (rho * tri_1 - tri_o) + off * (one_m * tri_0 - tri_o)
    return -elbo, -np.concatenate([g_rho, g_omega])


def find_optimum_rho_omega(sum_log_pi, start_alpha_log_pi, n_doc, gamma, alpha, kappa, init_rho=None, init_omega=None):
    """argmax of the surrogate bound over (rho, omega): K = len(sum_log_pi) - 1 sticks.

    sum_log_pi [K+1]: column sums of E[log pi] over the transition rows; start_alpha_log_pi [K+1]: startAlpha E[log pi_0]."""
    sum_log_pi = np.asarray(sum_log_pi, dtype=np.float64).reshape(-1)
    K = sum_log_pi.size - 1
    if kappa > 0:
        t_vec = alpha * sum_log_pi + start_alpha_log_pi
        t_vec[:-1] += np.log(alpha + kappa) - np.log(kappa)
    else:
        t_vec = alpha * sum_log_pi / n_doc + np.asarray(start_alpha_log_pi, dtype=np.float64) / n_doc
    rho0 = create_initrho(K) if init_rho is None else np.asarray(init_rho, dtype=np.float64)
    rho0 = np.clip(rho0, EPS, 1.0 - EPS)
    om0 = (n_doc / K + gamma) * np.ones(K) if init_omega is None else np.asarray(init_omega, dtype=np.float64)
    om0 = np.maximum(om0, EPS)
    c0 = np.concatenate([-np.log(1.0 / rho0 - 1.0), np.log(om0)])

    def fun(c):
        rho = np.clip(1.0 / (1.0 + np.exp(-c[:K])), EPS, 1.0 - EPS)
        omega = np.exp(c[K:])
        f, g = _objective(np.concatenate([rho, omega]), t_vec, n_doc, gamma, kappa, K)
        return f, g * np.concatenate([rho * (1.0 - rho), omega])

    last = None
    for factr in (1e5, 1e7, 1e9, 1e10, 1e11):     # progressively weaker tolerances until one run converges
        try:
            with warnings.catch_warnings():
                warnings.filterwarnings("error", category=RuntimeWarning, message="overflow")
                c, f, info = _fmin_l_bfgs_b(fun, c0, factr)
            if info["warnflag"] > 1:
                raise ValueError("FAILURE: " + str(info["task"]))
            rho = np.clip(1.0 / (1.0 + np.exp(-c[:K])), EPS, 1.0 - EPS)
            return rho, np.exp(c[K:]), f
        except (ValueError, RuntimeWarning, FloatingPointError) as err:
            last = err
    if init_rho is not None:
        return find_optimum_rho_omega(sum_log_pi, start_alpha_log_pi, n_doc, gamma, alpha, kappa)
    raise ValueError(str(last))


# ---------------------------------------------------------------------------------------------------------------------
# HDP-HMM terms of the variational bound for a HARD assignment (GPI_HDP.py:2651-2750; bnpy's HDPHMMUtil restated).
def _c_dir(a):
    """Cumulant of a (row-wise) Dirichlet, summed over rows."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return gammaln(np.sum(a)) - np.sum(gammaln(a))
    return np.sum(gammaln(np.sum(a, axis=1))) - np.sum(gammaln(a))


def _l_top(rho, omega, alpha, start_alpha, kappa, gamma):
    """Top-level stick-breaking terms E[log p(u)] - E[log q(u)] plus the constants of the Dirichlet rows."""
    K = rho.size
    eta1, eta0 = rho * omega, (1.0 - rho) * omega
    dg_omega = digamma(omega)
    e_log_u, e_log_1mu = digamma(eta1) - dg_omega, digamma(eta0) - dg_omega
    c_beta = lambda a1, a0: np.sum(gammaln(a1 + a0)) - np.sum(gammaln(a1)) - np.sum(gammaln(a0))   # noqa: E731
    diff_c_beta = K * c_beta(1.0, gamma) - c_beta(eta1, eta0)
    t_alpha = K * K * np.log(alpha) + K * np.log(start_alpha)
    kv = np.arange(K, 0, -1, dtype=np.float64)
    if kappa > 0:
        coef_u, coef_1mu = K + 1.0 + eta1, K * kv + 1.9 + gamma - eta0      # (the reference's constants, GPI_HDP.py:2717-2718)
        t_beta = np.sum(rho2beta(rho, "K")) * (np.log(alpha + kappa) - np.log(kappa))
        t_kappa = K * (np.log(kappa) - np.log(alpha + kappa))
    else:
        coef_u, coef_1mu = (K + 1) + 1.0 - eta1, (K + 1) * kv + gamma - eta0
        t_beta = t_kappa = 0.0
    return t_alpha + t_kappa + t_beta + diff_c_beta + coef_u @ e_log_u + coef_1mu @ e_log_1mu


def elbo_linear_terms(rho, omega, alpha, start_alpha, kappa, gamma, trans_theta, start_theta, start_count, trans_count):
    """GPI_HDP.calcELBO_LinearTerms (GPI_HDP.py:2651-2680): L_top - cumulants of q(pi) + the slack terms
    (counts + prior - pseudo-counts) . E[log pi].  The row normaliser of the transition slack is digamma(sum theta)."""
    rho, omega = np.asarray(rho, dtype=np.float64), np.asarray(omega, dtype=np.float64)
    trans_theta, start_theta = np.asarray(trans_theta, dtype=np.float64), np.asarray(start_theta, dtype=np.float64)
    trans_count = np.array(trans_count, dtype=np.float64)
    K = trans_count.shape[0]
    ebeta = rho2beta(rho, "K" if start_theta.shape[0] == rho.size else "K+1")
    l_start = np.inner(start_count + start_alpha * ebeta - start_theta, digamma(start_theta) - digamma(np.sum(start_theta)))
    prior = alpha * np.tile(ebeta, (K, 1))
    prior[:, :K] += kappa * np.eye(K)
    l_trans = np.sum((trans_count + prior - trans_theta) * (digamma(trans_theta) - digamma(np.sum(trans_theta, axis=1))[:, None]))
    return _l_top(rho, omega, alpha, start_alpha, kappa, gamma) - _c_dir(trans_theta) - _c_dir(start_theta) + l_start + l_trans


def elbo_entropy(resp, resp_pair, eps=1e-30):
    """GPI_HDP.calcELBO_NonlinearTerms (GPI_HDP.py:2682-2700): H[q(z)] from the state and pair tables."""
    sigma = resp_pair / (resp_pair.sum(axis=2)[:, :, None] + eps) + eps
    h_table = -np.sum(resp_pair * np.log(sigma), axis=0)
    h_start = -np.sum(resp * np.log(resp + eps), axis=0)
    return float(h_table.sum() + h_start.sum())
