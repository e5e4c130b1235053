"""hdpgpc/hdpgpc/get_data.py, the one function every driver calls: prior estimators of the LDS noises."""
import numpy as np


def compute_estimators_LDS(samples, n_f=None):
    """get_data.py:295-322.  samples [N, T, D] (lead 0 is used): returns (std, std_dif, bound_std, bound_std_dif) -
    the mean per-point variance over the first n_f segments, the mean squared step between consecutive segments, both
    scaled by 0.02 when the data is not standardised, and the bounds the drivers pass on."""
    samples = np.asarray(samples, dtype=np.float64)
    if n_f is None:
        n_f = samples.shape[0] - 2
    a = samples[:n_f, :, 0].T                      # [T, n_f]
    b = samples[1:n_f + 1, :, 0].T
    dev = a - a.mean(axis=1, keepdims=True)
    std = float(np.mean(np.sum(dev * dev, axis=1) / n_f))
    step = b - a
    std_dif = float(np.mean(np.sum(step * step, axis=1) / n_f))
    if std > 1:
        std, std_dif = std * 0.02, std_dif * 0.02
    std_dif = float(min(max(std, std_dif), std * 1.5))
    print("Sigma estimated:", str(std))
    print("Gamma estimated:", str(std_dif))
    return std, std_dif, (std * 1e-5, std * 2.0), (std_dif * 1e-5, 1.0)
