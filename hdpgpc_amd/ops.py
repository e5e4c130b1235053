"""Tensor-level wrappers over the C-ABI.  torch tensors are device containers only: every function
takes fp64 CUDA (ROCm) tensors, enqueues HIP kernels on the current stream and returns tensors."""
import ctypes
import math

import numpy as np
import torch

from . import _ffi

LOG2PI = math.log(2.0 * math.pi)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def env_flag(name):
    import os
    return os.environ.get(name, "0") not in ("", "0")


def _stream():
    # the raw handle of torch's current stream (torch.cuda.current_stream() builds a Stream object: 9 us per call, and the
    # online step makes hundreds of calls per beat)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class _PinnedRing:
    """Small host arrays (index lists, descriptor tables, K x K transition logs) go to the device through a ring of pinned
    staging buffers with asynchronous copies: torch.as_tensor(array, device=...) is a BLOCKING copy that first waits for
    everything queued on the stream - the online step made ~40 of them per beat, each one a hidden synchronisation."""
    SLOTS, NBYTES = 64, 1 << 16

    def __init__(self):
        self.bufs = [torch.empty(self.NBYTES, dtype=torch.uint8).pin_memory() for _ in range(self.SLOTS)]
        self.views = [b.numpy() for b in self.bufs]
        self.events = [None] * self.SLOTS
        self.i = 0


_ring = None
_NP_OF = {torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8, torch.bool: np.bool_}


def to_dev(a, dtype, device):
    """A host array / list / CPU tensor as a device tensor of `dtype` without a stream synchronisation."""
    global _ring
    if torch.is_tensor(a):
        if a.is_cuda:
            return a.to(dtype=dtype).contiguous()
        a = a.numpy()
    arr = np.ascontiguousarray(a, dtype=_NP_OF[dtype])
    dev = torch.device(device)
    if dev.type != "cuda" or arr.nbytes == 0 or arr.nbytes > _PinnedRing.NBYTES:
        return torch.as_tensor(arr, dtype=dtype, device=device)
    if _ring is None:
        _ring = _PinnedRing()
    r = _ring
    i = r.i
    r.i = (i + 1) % r.SLOTS
    if r.events[i] is not None:
        r.events[i].synchronize()          # the copy that last used this slot (64 uploads ago) has long finished
    nb = arr.nbytes
    r.views[i][:nb] = arr.reshape(-1).view(np.uint8)
    out = torch.empty(arr.shape, dtype=dtype, device=device)
    out.view(-1).view(torch.uint8).copy_(r.bufs[i][:nb], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    r.events[i] = ev
    return out


def _dev64(t, name):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise TypeError(f"{name} must be a contiguous fp64 tensor on the GPU")
    return t


def raise_on_info(info, what):
    """Mirror torch.linalg.cholesky's error behaviour from a per-item LAPACK info tensor (host sync)."""
    host = info.reshape(-1).cpu()          # one copy (torch.nonzero on the device is three launches and a sync: 0.3 ms)
    bad = torch.nonzero(host)
    if bad.numel():
        i = int(bad[0, 0])
        raise torch.linalg.LinAlgError(
            f"{what}: (Batch element {i}): The factorization could not be completed because the input is not "
            f"positive-definite (the leading minor of order {int(host[i])} is not positive-definite).")


def gram_rbf(x, y, c, ell, noise=0.0):
    """a1: (ConstantKernel(c)*RBF(ell)+WhiteKernel(noise))(x, y); y=None is the one-argument call."""
    x = _dev64(x.reshape(-1), "x")
    yv = None if y is None else _dev64(y.reshape(-1), "y")
    ny = x.numel() if yv is None else yv.numel()
    K = torch.empty((x.numel(), ny), dtype=torch.float64, device=x.device)
    _ffi.check(_ffi.lib.hgp_gram_rbf_f64(_ptr(x), x.numel(), _ptr(yv), ny, c, ell, noise, _ptr(K), _stream()), "gram_rbf")
    return K


def potrf_batched(A, jitter_rel=1e-8, add_diag=0.0, want_inv=False, want_logdet=False):
    """a3: batched GPI_model._chol_spd.  A [b,T,T] (not modified).  Returns (L, info[, Linv][, logdet])."""
    A = _dev64(A, "A")
    if A.dim() == 2:
        A = A.unsqueeze(0)
    b, T, _ = A.shape
    L = A.clone()
    info = torch.zeros(b, dtype=torch.int32, device=A.device)
    Linv = torch.empty_like(L) if want_inv else None
    logdet = torch.empty(b, dtype=torch.float64, device=A.device) if want_logdet else None
    _ffi.check(_ffi.lib.hgp_potrf_batched_f64(_ptr(L), T, b, float(jitter_rel), float(add_diag), _ptr(Linv), _ptr(logdet),
                                              _ptr(info), _stream()), "potrf_batched")
    out = [L, info]
    if want_inv:
        out.append(Linv)
    if want_logdet:
        out.append(logdet)
    return tuple(out)


def chol_inverse(A, jitter_rel=0.0, add_diag=0.0, out=None, info=None, work=None):
    """Z = chol(0.5 (A + A^T) + shift I)^{-1} for a batch [b,T,T] (T <= 256; A untouched).  Returns (Z, info); `out` / `info`
    may be caller-allocated (capture-safe, no allocation).  work [b,T,T] (optional, T > 128): large batches factor once into
    it and take L^-1 from L (hgp_chol_inverse_ws_f64) instead of one factorisation per block column."""
    A = _dev64(A, "A")
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    b, T, _ = A3.shape
    Z = torch.empty_like(A3) if out is None else out
    if info is None:
        info = torch.zeros(b, dtype=torch.int32, device=A.device)
    if work is not None:
        if work.numel() < A3.numel():
            raise ValueError("chol_inverse: workspace smaller than the batch")
        _ffi.check(_ffi.lib.hgp_chol_inverse_ws_f64(_ptr(A3), T, b, float(jitter_rel), float(add_diag), _ptr(Z), _ptr(_dev64(work, "work")),
                                                    _ptr(info), _stream()), "chol_inverse_ws")
        return Z, info
    _ffi.check(_ffi.lib.hgp_chol_inverse_batched_f64(_ptr(A3), T, b, float(jitter_rel), float(add_diag), _ptr(Z), _ptr(info),
                                                     _stream()), "chol_inverse")
    return Z, info


def copy_list(items_dev, n_items, max_n):
    """n_items copies dst[0..n) = src[0..n) described by the device-resident int64 table items_dev [n_items, 3] = (src, dst, n)."""
    _ffi.check(_ffi.lib.hgp_copy_list_f64(_ptr(items_dev), int(n_items), int(max_n), _stream()), "copy_list")


MAX_CHUNK = 64  # segments per work item: one factorisation is reused for up to this many right-hand sides


def build_items(mat_of_group, add_of_group, group_sizes):
    """Split groups (segments sorted by group) into work items of at most MAX_CHUNK segments."""
    item_mat, item_add, item_off, item_cnt = [], [], [], []
    off = 0
    for m, a, n in zip(mat_of_group, add_of_group, group_sizes):
        done = 0
        while done < n:
            c = min(MAX_CHUNK, n - done)
            item_mat.append(m)
            item_add.append(a)
            item_off.append(off + done)
            item_cnt.append(c)
            done += c
        off += n
    return (np.asarray(item_mat, np.int32), np.asarray(item_add, np.float64), np.asarray(item_off, np.int32),
            np.asarray(item_cnt, np.int32))


def score_groups(Y, mean, Sigma, item_mat, item_add, item_off, item_cnt, seg_ids=None, jitter_rel=1e-8,
                 want_logdet=False, want_info=True, item_mean=None, strides=None):
    """a4+a6: quad[n] = (Y[n]-mean[s])^T cov_s^{-1} (Y[n]-mean[s]) for the segments of each work item.

    Y [N,T]; mean [S,T] or None; Sigma [S,T,T]; item_* host or device int/float arrays; seg_ids [sum cnt] or None.
    """
    Y = _dev64(Y, "Y")
    dev = Y.device
    N, T = Y.shape
    if strides is None:       # strides = (mean_stride, sigma_stride) in doubles: states that sit inside larger per-cluster records
        Sigma = _dev64(Sigma, "Sigma")
        if Sigma.dim() == 2:
            Sigma = Sigma.unsqueeze(0)
        if mean is not None:
            mean = _dev64(mean.reshape(-1, T), "mean")
    mstride, sstride = (T, T * T) if strides is None else (int(strides[0]), int(strides[1]))

    def up(a, dt):
        return to_dev(a, dt, dev)

    im, ia = up(item_mat, torch.int32), (None if item_add is None else up(item_add, torch.float64))
    io, ic = up(item_off, torch.int32), up(item_cnt, torch.int32)
    sid = None if seg_ids is None else up(seg_ids, torch.int32)
    imean = None if item_mean is None else up(item_mean, torch.int32)
    quad = torch.zeros(N, dtype=torch.float64, device=dev)
    logdet = torch.zeros(N, dtype=torch.float64, device=dev) if want_logdet else None
    info = torch.zeros(N, dtype=torch.int32, device=dev) if want_info else None
    _ffi.check(_ffi.lib.hgp_score_groups_f64(_ptr(Y), T, _ptr(mean), mstride, _ptr(Sigma), sstride, T, _ptr(im), _ptr(imean),
                                             _ptr(ia), _ptr(io), _ptr(ic), im.numel(), _ptr(sid), jitter_rel, _ptr(quad),
                                             _ptr(logdet), _ptr(info), _stream()), "score_groups")
    return quad, logdet, info


def score_each(Y, mean, Sigma, seg_mat, seg_mean=None, seg_add=None, jitter_rel=1e-8, want_logdet=False, want_info=True,
               symmetric=False, strides=None):
    """a6 for member segments: segment i against its own state.  Y [n,T]; mean [S,T]; Sigma [S,T,T]; seg_* [n].
    symmetric=True promises Sigma == Sigma^T exactly (upper triangle read only)."""
    Y = _dev64(Y, "Y")
    dev = Y.device
    n, T = Y.shape
    if strides is None:
        Sigma = _dev64(Sigma, "Sigma")
        if mean is not None:
            mean = _dev64(mean.reshape(-1, T), "mean")
    mstride, sstride = (T, T * T) if strides is None else (int(strides[0]), int(strides[1]))

    def up(a, dt):
        if a is None:
            return None
        return to_dev(a, dt, dev)

    sm, sme, sa = up(seg_mat, torch.int32), up(seg_mean, torch.int32), up(seg_add, torch.float64)
    if T > 128:   # cooperative kernels: one work item (one workgroup) per segment
        ar = torch.arange(n, dtype=torch.int32, device=dev)
        return score_groups(Y, mean, Sigma, sm, sa, ar, torch.ones(n, dtype=torch.int32, device=dev), jitter_rel=jitter_rel,
                            want_logdet=want_logdet, want_info=want_info, item_mean=sme, strides=strides)
    quad = torch.zeros(n, dtype=torch.float64, device=dev)
    logdet = torch.zeros(n, dtype=torch.float64, device=dev) if want_logdet else None
    info = torch.zeros(n, dtype=torch.int32, device=dev) if want_info else None
    _ffi.check(_ffi.lib.hgp_score_each_f64(_ptr(Y), T, _ptr(mean), mstride, _ptr(Sigma), sstride, T, _ptr(sm), _ptr(sme), _ptr(sa), n,
                                           jitter_rel, int(bool(symmetric)), _ptr(quad), _ptr(logdet), _ptr(info), _stream()),
               "score_each")
    return quad, logdet, info


class PairsPlan:
    """a2+a5 for an N x K batch: per-cluster operators (plan) + the per-pair kernels.

    acc_tol selects, per cluster and on the device, between the two evaluations of cov_f (hgp_pairs_plan_set_accuracy):
    clusters whose accuracy_bound() exceeds it (ill-conditioned K~, e.g. the drivers' ini_lengthscale = 3.0) are scored
    by the solve-based kernel (the reference's operation order, GPI.py:489-501); 0 = all clusters, < 0 = none."""

    def __init__(self, T, Ts_max, theta, device="cuda", acc_tol=1e-9):
        theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).reshape(-1, 3))
        self.T, self.Ts_max, self.K = int(T), int(Ts_max), theta.shape[0]
        self.theta = theta
        nbytes = _ffi.lib.hgp_pairs_plan_device_bytes(self.T, self.Ts_max, self.K)
        if nbytes == 0:
            raise ValueError("bad plan shape")
        self._buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._h = ctypes.c_void_p()
        self._score_out = False
        _ffi.check(_ffi.lib.hgp_pairs_plan_create(ctypes.byref(self._h), self.T, self.Ts_max, self.K,
                                                  theta.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                  _ptr(self._buf), nbytes), "pairs_plan_create")
        self.info = torch.zeros(self.K, dtype=torch.int32, device=device)
        self._routed = False
        self.set_accuracy(acc_tol)

    def set_accuracy(self, tol):
        """Threshold on accuracy_bound() above which a cluster takes the solve-based kernel (next update())."""
        self.acc_tol = float(tol)
        _ffi.check(_ffi.lib.hgp_pairs_plan_set_accuracy(self._h, self.acc_tol), "pairs_plan_set_accuracy")
        self._routed = False           # the per-cluster routing flags belong to the previous threshold until update() runs
        return self

    def update(self, x_basis, mean, Sigma):
        """x_basis [T]; mean [K,T] (= C f on the basis grid); Sigma [K,T,T]."""
        x_basis = _dev64(x_basis.reshape(-1), "x_basis")
        mean = _dev64(mean.reshape(self.K, self.T), "mean")
        Sigma = _dev64(Sigma.reshape(self.K, self.T, self.T), "Sigma")
        self._keep = (x_basis, mean, Sigma)
        _ffi.check(_ffi.lib.hgp_pairs_plan_update(self._h, _ptr(x_basis), _ptr(mean), _ptr(Sigma), _ptr(self.info),
                                                  _stream()), "pairs_plan_update")
        self._routed = True
        return self

    def scalars(self):
        """[K,8] device view: c, ell, noise, iso flag, mean diag Sigma, jitter, ||K~^-1||_inf, solve-based flag."""
        base = _ffi.lib.hgp_pairs_plan_scalars(self._h)
        off = base - self._buf.data_ptr()
        return self._buf[off:off + self.K * 64].view(torch.float64).view(self.K, 8)

    def accuracy_bound(self):
        """eps * (c ||K~^-1||_inf)^2 per cluster: growth of the rounding error of the explicit-operator evaluation
        relative to the reference's triangular solves (host sync).  ~1e-10 at the reference's length-scale.  The plan
        compares it with acc_tol on the device; nothing on the scoring path needs this host copy."""
        s = self.scalars().cpu().numpy()
        return np.finfo(np.float64).eps * (s[:, 0] * s[:, 6]) ** 2

    def solve_based(self):
        """Boolean [K] (host sync): clusters the last update() routed to the solve-based kernel."""
        return self.scalars()[:, 7].cpu().numpy() != 0.0

    def loglik(self, x, y, first_noise=None, want_logdet=True, want_info=True, sel=None, score=False):
        """x, y [N,Ts] -> (quad [N,K], logdet [N,K] or None, info [N,K] or None).

        sel [N] int32 (optional): segment n is scored against cluster sel[n] only; outputs (and first_noise) are [N].
        score=True: the first output is the reference's score -0.5 quad - 0.5 Ts log(2 pi) (GPI_model.py:285), written by the
        kernels themselves (hgp_pairs_plan_set_score_output) - no arithmetic launch behind the pair kernels.
        Every output element is written by exactly one pair kernel: the buffers are not cleared first."""
        if not self._routed:
            raise RuntimeError("PairsPlan: update() must run after set_accuracy() / before the first loglik() - the kernels skip "
                               "clusters by the routing flags update() writes, and the outputs are not cleared")
        x = _dev64(x, "x")
        y = _dev64(y, "y")
        N, Ts = x.shape
        dev = x.device
        shape = (N,) if sel is not None else (N, self.K)
        if first_noise is not None:
            first_noise = _dev64(first_noise.reshape(shape), "first_noise")
        if sel is not None:
            # a selection outside [0, K) would leave its output element unwritten: host arrays are checked here, device arrays
            # get outputs that start as NaN / info = 1 (two fill launches on the per-segment path only)
            if not torch.is_tensor(sel) or not sel.is_cuda:
                sel_h = np.asarray(sel)
                if sel_h.size and (sel_h.min() < 0 or sel_h.max() >= self.K):
                    raise ValueError(f"PairsPlan.loglik: sel must lie in [0, {self.K})")
                quad = torch.empty(shape, dtype=torch.float64, device=dev)
                info = torch.empty(shape, dtype=torch.int32, device=dev) if want_info else None
            else:
                quad = torch.full(shape, float("nan"), dtype=torch.float64, device=dev)
                info = torch.ones(shape, dtype=torch.int32, device=dev) if want_info else None
            sel = to_dev(sel, torch.int32, dev)
        else:
            quad = torch.empty(shape, dtype=torch.float64, device=dev)
            info = torch.empty(shape, dtype=torch.int32, device=dev) if want_info else None
        logdet = torch.empty(shape, dtype=torch.float64, device=dev) if want_logdet else None
        if bool(score) != self._score_out:
            _ffi.check(_ffi.lib.hgp_pairs_plan_set_score_output(self._h, int(bool(score))), "pairs_plan_set_score_output")
            self._score_out = bool(score)
        _ffi.check(_ffi.lib.hgp_loglik_pairs_f64(self._h, _ptr(x), _ptr(y), N, Ts, _ptr(first_noise), _ptr(sel), _ptr(quad),
                                                 _ptr(logdet), _ptr(info), _stream()), "loglik_pairs")
        return quad, logdet, info

    def score(self, x, y, first_noise=None, sel=None):
        """The reference's score: -0.5 quad - 0.5 Ts log(2 pi)  (GPI_model.py:285, no log-determinant)."""
        q, _, info = self.loglik(x, y, first_noise, want_logdet=False, sel=sel, score=True)
        return q, info

    def close(self):
        if self._h:
            _ffi.lib.hgp_pairs_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_PLANS = {}                        # (device, T, Ts, K, theta) -> PairsPlan, in least-recently-used order
PLAN_CACHE_BYTES = 1 << 30         # device bytes the cached plans may hold together (the newest plan always stays)


def plan_cache(T, Ts, K, theta, device):
    """A PairsPlan for K clusters that share one theta, from a process-wide LRU cache bounded by PLAN_CACHE_BYTES.  Plans are
    stateless between calls as far as callers are concerned: every use starts with update()."""
    key = (str(device), int(T), int(Ts), int(K), tuple(theta))
    plan = _PLANS.pop(key, None)
    if plan is None:
        plan = PairsPlan(T, Ts, np.repeat(np.asarray(theta, dtype=np.float64)[None], int(K), 0), device=device)
    _PLANS[key] = plan                                     # most recently used = last
    total = sum(p._buf.numel() for p in _PLANS.values())
    for k in list(_PLANS):
        if total <= PLAN_CACHE_BYTES or k == key:
            break
        old = _PLANS.pop(k)
        total -= old._buf.numel()
        old.close()
        old._buf = None
    return plan


def gemm_batched(A, B, transA=False, transB=False, alpha=1.0, add=None, beta=1.0, out=None):
    """C[b] = alpha op(A[b]) op(B[b]) (+ beta add[b]) on v_mfma_f64_16x16x4_f64; A [b,m,k] (or [m,k]), B [b,k,n] (or [k,n]).
    add: optional [m,n] (shared by the batch) or [b,m,n] addend fused into the epilogue; out: optional preallocated
    contiguous [b,m,n] (or [m,n]) result."""
    A = _dev64(A, "A")
    B = _dev64(B, "B")
    a3 = A if A.dim() == 3 else A.unsqueeze(0)
    b3 = B if B.dim() == 3 else B.unsqueeze(0)
    batch = max(a3.shape[0], b3.shape[0])
    M, Kd = (a3.shape[2], a3.shape[1]) if transA else (a3.shape[1], a3.shape[2])
    N = b3.shape[1] if transB else b3.shape[2]
    sA = 0 if a3.shape[0] == 1 else a3.shape[1] * a3.shape[2]
    sB = 0 if b3.shape[0] == 1 else b3.shape[1] * b3.shape[2]
    if out is None:
        C = torch.empty((batch, M, N), dtype=torch.float64, device=A.device)
    else:
        C = out if out.dim() == 3 else out.unsqueeze(0)
        if tuple(C.shape) != (batch, M, N) or not C.is_contiguous() or C.dtype != torch.float64:
            raise ValueError("gemm_batched: out must be a contiguous float64 [batch, M, N]")
    if add is None:
        _ffi.check(_ffi.lib.hgp_gemm_batched_f64(int(transA), int(transB), M, N, Kd, alpha, _ptr(a3), a3.shape[2], sA, _ptr(b3),
                                                 b3.shape[2], sB, 0.0, _ptr(C), N, M * N, batch, _stream()), "gemm_batched")
    else:
        D = _dev64(add, "add")
        d3 = D if D.dim() == 3 else D.unsqueeze(0)
        if tuple(d3.shape[1:]) != (M, N) or d3.shape[0] not in (1, batch):
            raise ValueError("gemm_batched: add must be [M, N] or [batch, M, N]")
        sD = 0 if d3.shape[0] == 1 else M * N
        _ffi.check(_ffi.lib.hgp_gemm_add_batched_f64(int(transA), int(transB), M, N, Kd, alpha, _ptr(a3), a3.shape[2], sA, _ptr(b3),
                                                     b3.shape[2], sB, beta, _ptr(d3), N, sD, _ptr(C), N, M * N, batch, _stream()),
                   "gemm_add_batched")
    if out is not None:
        return out
    return C if (A.dim() == 3 or B.dim() == 3) else C[0]


def add_diag_mean(R, S, factor, out=None):
    """out[b] = R[b] + factor * max(mean |diag S[b]|, eps) I  for [b,T,T] stacks (MNIW jitter, GPI_model.py:1312-1316)."""
    R, S = _dev64(R, "R"), _dev64(S, "S")
    b, T, _ = R.shape
    if out is None:
        out = torch.empty_like(R)
    _ffi.check(_ffi.lib.hgp_add_diag_mean_f64(_ptr(R), _ptr(S), T, b, float(factor), _ptr(out), _stream()), "add_diag_mean")
    return out


def rts_chain(J, P, AM, M, Cv):
    """Sequential part of the RTS smoother for all steps in one launch (in place on M [n,T] and Cv [n,T,T]); T <= 96."""
    n, T = M.shape[0], Cv.shape[1]
    _ffi.check(_ffi.lib.hgp_rts_chain_f64(_ptr(J), _ptr(P), _ptr(AM), _ptr(M), _ptr(Cv), n, T, _stream()), "rts_chain")


def lds_chain_scatter(f_post, c_post, f_sm_prev, P_sm_prev, stF, stFsm, stP, stPsm, pos):
    """8f-1 glue: append the new filtered state (rows pos + 1) and overwrite the re-smoothed previous one (rows pos)."""
    T = stP.shape[1]
    _ffi.check(_ffi.lib.hgp_lds_chain_scatter_f64(_ptr(f_post), _ptr(c_post), _ptr(f_sm_prev), _ptr(P_sm_prev), _ptr(stF),
                                                  _ptr(stFsm), _ptr(stP), _ptr(stPsm), _ptr(pos), T, _stream()), "lds_chain_scatter")


def lat_error(f_cur, f_prev, A, Gamma, covprev):
    """a8 batched: returns (-0.5 (mahal + trace) [b], info [b]); the caller adds -0.5 T log 2pi."""
    f_cur, f_prev = _dev64(f_cur, "f_cur"), _dev64(f_prev, "f_prev")
    A, Gamma, covprev = _dev64(A, "A"), _dev64(Gamma, "Gamma"), _dev64(covprev, "covprev")
    b, T = f_cur.shape
    out = torch.empty(b, dtype=torch.float64, device=A.device)
    info = torch.zeros(b, dtype=torch.int32, device=A.device)
    nws = _ffi.lib.hgp_matrix_lik_ws_bytes(T, b) if T > 128 else 0     # T <= 128: one fused kernel, no workspace
    ws = torch.empty(nws, dtype=torch.uint8, device=A.device) if nws else None
    _ffi.check(_ffi.lib.hgp_lat_error_f64(_ptr(f_cur), _ptr(f_prev), _ptr(A), _ptr(Gamma), _ptr(covprev), T, b, _ptr(out),
                                          _ptr(info), _ptr(ws), nws, _stream()), "lat_error")
    return out, info


def mniw_loglik(M, Sigma, m_mean, m_r_cov, scale, scale_is_diagonal=None):
    """a9 batched.  M, Sigma [b,T,T]; prior (m_mean, scale, optional m_r_cov) [T,T] shared or [b,T,T] per item.
    scale_is_diagonal: None = check it here (one device comparison, host sync); callers that know (the hot path's
    prior scale is sigma I) pass True / False."""
    M, Sigma = _dev64(M, "M"), _dev64(Sigma, "Sigma")
    b, T, _ = M.shape
    m_mean, scale = _dev64(m_mean, "m_mean"), _dev64(scale, "scale")
    stride = 0 if m_mean.dim() == 2 else T * T
    if scale_is_diagonal is None:
        scale_is_diagonal = bool(torch.equal(scale, torch.diag_embed(torch.diagonal(scale, dim1=-2, dim2=-1))))
    if m_r_cov is not None:
        m_r_cov = _dev64(m_r_cov, "m_r_cov")
    out = torch.empty(b, dtype=torch.float64, device=M.device)
    info = torch.zeros(b, dtype=torch.int32, device=M.device)
    nws = _ffi.lib.hgp_matrix_lik_ws_bytes(T, b) if T > 128 else 0     # T <= 128: one fused kernel, no workspace
    ws = torch.empty(nws, dtype=torch.uint8, device=M.device) if nws else None
    _ffi.check(_ffi.lib.hgp_mniw_loglik_f64(_ptr(M), _ptr(Sigma), _ptr(m_mean), _ptr(m_r_cov), _ptr(scale), int(scale_is_diagonal), stride, T, b,
                                            _ptr(out), _ptr(info), _ptr(ws), nws, _stream()), "mniw_loglik")
    return out, info


def warp_cov(x, rho, omega, diag_add, normalize=True):
    x = _dev64(x.reshape(-1), "x")
    K = torch.empty((x.numel(), x.numel()), dtype=torch.float64, device=x.device)
    _ffi.check(_ffi.lib.hgp_warp_cov_f64(_ptr(x), x.numel(), rho, omega, diag_add, int(bool(normalize)), _ptr(K), _stream()),
               "warp_cov")
    return K


def chol_rank1(L, v, alpha=None, beta=None):
    """config 5: chol(alpha L L^T + beta v v^T) by a rank-1 update (O(T^2)), batched; returns (L_new, info)."""
    L = _dev64(L, "L")
    L3 = (L if L.dim() == 3 else L.unsqueeze(0)).clone()
    b, T, _ = L3.shape
    v = _dev64(v.reshape(b, T), "v")
    al = None if alpha is None else torch.as_tensor(alpha, dtype=torch.float64, device=L.device).reshape(b).contiguous()
    be = None if beta is None else torch.as_tensor(beta, dtype=torch.float64, device=L.device).reshape(b).contiguous()
    info = torch.zeros(b, dtype=torch.int32, device=L.device)
    _ffi.check(_ffi.lib.hgp_chol_rank1_f64(_ptr(L3), _ptr(v), _ptr(al), _ptr(be), T, b, _ptr(info), _stream()), "chol_rank1")
    return (L3 if L.dim() == 3 else L3[0]), info


def lds_chain_gather(stA, stG, stC, stS, stPsm, stP, stF, stFsm, pos, out, Y=None, y_row0=0, y_out=None):
    """8f-1 glue: row pos[0] of the eight state stacks -> out[6 T T + 2 T] (A, G, C, S, Psm, P, F, Fsm); optionally the
    observation Y[pos[0] - y_row0] -> y_out."""
    T = stA.shape[1]
    _ffi.check(_ffi.lib.hgp_lds_chain_gather_f64(_ptr(stA), _ptr(stG), _ptr(stC), _ptr(stS), _ptr(stPsm), _ptr(stP), _ptr(stF),
                                                 _ptr(stFsm), _ptr(pos), T, _ptr(out), _ptr(Y), int(y_row0), _ptr(y_out),
                                                 _stream()), "lds_chain_gather")
    return out


def lds_chain_finish(part, ee, Snew, info1, info2, W, n0, Nf, bad_count, stA, stG, stC, stS, pos, annealing, sync, info0=None):
    """8f-1 glue: element-wise tail of the two MNIW updates + append of A, Gamma, C, Sigma + counters (see the header)."""
    T = stA.shape[1]
    _ffi.check(_ffi.lib.hgp_lds_chain_finish_f64(T, _ptr(part), _ptr(ee), _ptr(Snew), _ptr(info1), _ptr(info2), _ptr(info0), _ptr(W), _ptr(n0),
                                                 _ptr(Nf), _ptr(bad_count), _ptr(stA), _ptr(stG), _ptr(stC), _ptr(stS), _ptr(pos),
                                                 int(bool(annealing)), _ptr(sync), _stream()), "lds_chain_finish")


class GemmList:
    """A device-resident list of products  C = alpha op(A) op(B) + beta D (+ add_eye I), run as ONE launch (hgp_gemm_list_f64).
    Built once from tensors whose storage outlives the list; `add` takes tensors (2-D, or 1-D read as a column)."""

    def __init__(self, device):
        self.device = device
        self._items, self._keep, self.tiles, self._item_tiles = [], [], 0, []
        self._dev = self._map = None

    @staticmethod
    def _dims(t):
        if t.dim() == 1:
            if t.numel() > 1 and t.stride(0) != 1:
                raise ValueError("gemm list vectors must be contiguous (a strided 1-D view would be read as a column of stride 1)")
            return t.shape[0], 1, 1
        if t.stride(-1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
            raise ValueError("gemm list operands must have unit inner stride and non-overlapping rows")
        return t.shape[0], t.shape[1], t.stride(0)

    def add(self, A, B, out, D=None, transA=False, transB=False, alpha=1.0, beta=1.0, add_eye=0.0, out2=None):
        ar, ac, lda = self._dims(A)
        br, bc, ldb = self._dims(B)
        M, K = (ac, ar) if transA else (ar, ac)
        K2, N = (bc, br) if transB else (br, bc)
        orow, ocol, ldc = self._dims(out)
        if K != K2 or (orow, ocol) != (M, N) or max(M, N, K) > 256:
            raise ValueError(f"gemm list item {M}x{K} . {K2}x{N} -> {orow}x{ocol}")
        ldd = 0
        if D is not None:
            dr, dc, ldd = self._dims(D)
            if (dr, dc) != (M, N):
                raise ValueError("gemm list addend shape")
        if out2 is not None and self._dims(out2) != (orow, ocol, ldc):
            raise ValueError("gemm list second output must match the first")
        for t in (A, B, out, D, out2):
            if t is not None:
                _dev64_any(t)
                self._keep.append(t)
        it = _ffi.GemmItem(A.data_ptr(), B.data_ptr(), D.data_ptr() if D is not None else None, out.data_ptr(),
                           out2.data_ptr() if out2 is not None else None, M, N, K, lda, ldb, ldc, ldd, int(transA), int(transB),
                           float(alpha), float(beta), float(add_eye))
        self._items.append(it)
        self._item_tiles.append(((M + 15) // 16) * ((N + 15) // 16))
        self.tiles += self._item_tiles[-1]
        self._dev = None
        return out

    @classmethod
    def concat(cls, lists):
        """One list holding the items of several lists, in order (chain_batch: one launch per level for many chains)."""
        out = cls(lists[0].device)
        for l_ in lists:
            out._items += l_._items
            out._keep += l_._keep
            out._item_tiles += l_._item_tiles
            out.tiles += l_.tiles
        return out

    def finalize(self):
        import numpy as np
        arr = (_ffi.GemmItem * len(self._items))(*self._items)
        host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
        self._dev = host.to(self.device)
        self._map = None
        # per-tile map instead of the per-wave walk over the list (one dependent load per item: 7.8 vs 6.9 us for a 5-item level of one
        # chain under rocprofv3, tools/time_gemm_list.py); 16-bit item index, longer lists walk
        if 1 < len(self._items) <= 65535:
            m = np.concatenate([(i << 16) | np.arange(n, dtype=np.uint32) for i, n in enumerate(self._item_tiles)]).astype(np.uint32)
            self._map = torch.from_numpy(m.view(np.int32)).to(self.device)
            self._cum = np.concatenate([[0], np.cumsum(self._item_tiles)])
        return self

    def run(self):
        if self._dev is None:
            self.finalize()
        if self._map is not None:
            _ffi.check(_ffi.lib.hgp_gemm_list_mapped_f64(_ptr(self._dev), len(self._items), _ptr(self._map), self.tiles, _stream()),
                       "gemm_list_mapped")
            return
        _ffi.check(_ffi.lib.hgp_gemm_list_f64(_ptr(self._dev), len(self._items), self.tiles, _stream()), "gemm_list")

    def run_range(self, first, count):
        """One launch over the items [first, first + count) of the list (a long list holding the levels of many steps)."""
        if self._dev is None:
            self.finalize()
        if self._map is not None and first == 0:
            _ffi.check(_ffi.lib.hgp_gemm_list_mapped_f64(_ptr(self._dev), int(count), _ptr(self._map), int(self._cum[count]), _stream()),
                       "gemm_list_mapped")
            return
        base = ctypes.c_void_p(self._dev.data_ptr() + first * ctypes.sizeof(_ffi.GemmItem))
        _ffi.check(_ffi.lib.hgp_gemm_list_f64(base, int(count), sum(self._item_tiles[first:first + count]), _stream()), "gemm_list")


def _dev64_any(t):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float64):
        raise TypeError("gemm list operands must be fp64 tensors on the GPU")


def chol_inverse_rhs(A, Linv, rhs, rhs_out, info, rhs_on=None, rhs_trans=False, jitter_rel=0.0, add_diag=0.0):
    """Z = chol(0.5 (A + A^T) + shift I)^{-1} -> Linv and Y = Z op(rhs) -> rhs_out from ONE factorisation per matrix of the batch
    [b,T,T] (T <= 128), all outputs caller-allocated (capture-safe, no allocation)."""
    A = _dev64(A, "A")
    b, T, _ = A.shape
    _ffi.check(_ffi.lib.hgp_chol_inverse_rhs_batched_f64(_ptr(A), T, b, float(jitter_rel), float(add_diag), _ptr(Linv), _ptr(rhs),
                                                         _ptr(rhs_on), int(bool(rhs_trans)), _ptr(rhs_out), _ptr(info), _stream()),
               "chol_inverse_rhs")


def lds_chain_gather2(stA, stG, stC, stS, stPsm, stP, stF, stFsm, pos, out, Y, y_row0, y_out, W, Rp):
    """lds_chain_gather + the jittered right covariances Rp = W[1] + 1e-2 mean|diag W[2]| I of the two MNIW updates."""
    T = stA.shape[1]
    _ffi.check(_ffi.lib.hgp_lds_chain_gather2_f64(_ptr(stA), _ptr(stG), _ptr(stC), _ptr(stS), _ptr(stPsm), _ptr(stP), _ptr(stF),
                                                  _ptr(stFsm), _ptr(pos), T, _ptr(out), _ptr(Y), int(y_row0), _ptr(y_out), _ptr(W),
                                                  _ptr(Rp), _stream()), "lds_chain_gather2")


def lds_chain_finish2(f_post, c_post, f_sm_prev, P_sm_prev, y, part, Snew, info1, info2, W, n0, Nf, bad_count, stA, stG, stC, stS, stF,
                      stFsm, stP, stPsm, pos, annealing, sync):
    """lds_chain_scatter + lds_chain_finish in one launch ((y1 - y2)(y1 - y2)^T formed inside)."""
    T = stA.shape[1]
    _ffi.check(_ffi.lib.hgp_lds_chain_finish2_f64(T, _ptr(f_post), _ptr(c_post), _ptr(f_sm_prev), _ptr(P_sm_prev), _ptr(y), _ptr(part),
                                                  _ptr(Snew), _ptr(info1), _ptr(info2), _ptr(W), _ptr(n0), _ptr(Nf), _ptr(bad_count),
                                                  _ptr(stA), _ptr(stG), _ptr(stC), _ptr(stS), _ptr(stF), _ptr(stFsm), _ptr(stP),
                                                  _ptr(stPsm), _ptr(pos), int(bool(annealing)), _ptr(sync), _stream()),
               "lds_chain_finish2")


def trsv_lower_quad(G, y):
    """|| tril(G)^{-1} y ||^2 (a10 as written in the reference)."""
    G, y = _dev64(G, "G"), _dev64(y.reshape(-1), "y")
    out = torch.empty(1, dtype=torch.float64, device=G.device)
    _ffi.check(_ffi.lib.hgp_trsv_lower_quad_f64(_ptr(G), G.shape[1], _ptr(y), y.numel(), _ptr(out), _stream()), "trsv_lower_quad")
    return out[0]


def trsv_lower_solve(G, y):
    """alpha = tril(G)^{-T} tril(G)^{-1} y and || tril(G)^{-1} y ||^2  (the reference's cho_solve((K, True), y), GPI.py:1043)."""
    G, y = _dev64(G, "G"), _dev64(y.reshape(-1), "y")
    alpha = torch.empty_like(y)
    quad = torch.empty(1, dtype=torch.float64, device=G.device)
    _ffi.check(_ffi.lib.hgp_trsv_lower_solve_f64(_ptr(G), G.shape[1], _ptr(y), y.numel(), _ptr(alpha), _ptr(quad), _stream()),
               "trsv_lower_solve")
    return alpha, quad[0]


def lml_grad(x, alpha, Kinv, c, ell, noise):
    """a10 gradient w.r.t. (log c, log ell, log noise): 0.5 tr((alpha alpha^T - Kinv) dK/dtheta)  (GPI.py:1046-1051)."""
    x, alpha, Kinv = _dev64(x.reshape(-1), "x"), _dev64(alpha.reshape(-1), "alpha"), _dev64(Kinv, "Kinv")
    out = torch.empty(3, dtype=torch.float64, device=x.device)
    _ffi.check(_ffi.lib.hgp_lml_grad_f64(_ptr(x), _ptr(alpha), _ptr(Kinv), x.numel(), float(c), float(ell), float(noise), _ptr(out),
                                         _stream()), "lml_grad")
    return out



def hmm_messages(q, log_pi, log_trans, want_pair=True):
    """forward / backward messages of the switching variable and the log pair responsibilities
    (GPI_HDP.forward, backward, coupled_state_coef).  q [N,K] log-observations; returns (fmsg, marg, bmsg, log_resp_pair)."""
    q, log_pi, log_trans = _dev64(q, "q"), _dev64(log_pi.reshape(-1), "log_pi"), _dev64(log_trans, "log_trans")
    N, K = q.shape
    fmsg = torch.empty((N, K), dtype=torch.float64, device=q.device)
    marg = torch.empty(N, dtype=torch.float64, device=q.device)
    bmsg = torch.empty((N, K), dtype=torch.float64, device=q.device)
    pair = torch.empty((N, K, K), dtype=torch.float64, device=q.device) if want_pair else None
    _ffi.check(_ffi.lib.hgp_hmm_messages_f64(_ptr(q), _ptr(log_pi), _ptr(log_trans), N, K, _ptr(fmsg), _ptr(marg), _ptr(bmsg),
                                             _ptr(pair), _stream()), "hmm_messages")
    return fmsg, marg, bmsg, pair


def hmm_local_terms(Q, log_pi, log_trans, want_pair=True):
    """The local step of the switching variable for a batch of score matrices Q [B,N,K] sharing log_pi / log_trans
    (hgp_hmm_local_terms_f64): LogLik normalisation, messages and hard assignment per matrix.  Returns device tensors
    (labels [B,N] int64, pair_first [B,N] int64 or None, last_log [B,K])."""
    Q, log_pi, log_trans = _dev64(Q, "Q"), _dev64(log_pi.reshape(-1), "log_pi"), _dev64(log_trans, "log_trans")
    B, N, K = Q.shape
    dev = Q.device
    qn, fm, bm = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    marg = torch.empty((B, N), dtype=torch.float64, device=dev)
    labels = torch.empty((B, N), dtype=torch.int64, device=dev)
    pair = torch.empty((B, N), dtype=torch.int64, device=dev) if want_pair else None
    last = torch.empty((B, K), dtype=torch.float64, device=dev)
    _ffi.check(_ffi.lib.hgp_hmm_local_terms_f64(_ptr(Q), _ptr(log_pi), _ptr(log_trans), N, K, B, _ptr(qn), _ptr(fm), _ptr(marg), _ptr(bm),
                                                _ptr(labels), _ptr(pair), _ptr(last), _stream()), "hmm_local_terms")
    return labels, pair, last


def loglik_rows(q):
    """GPI_HDP.LogLik(axis=1) on the device: (q - rowmax, rowmax); unchanged input if any row maximum is infinite."""
    q = _dev64(q, "q")
    N, K = q.shape
    out = torch.empty_like(q)
    rowmax = torch.empty(N, dtype=torch.float64, device=q.device)
    _ffi.check(_ffi.lib.hgp_loglik_rows_f64(_ptr(q), N, K, _ptr(out), _ptr(rowmax), _stream()), "loglik_rows")
    return out, rowmax


def assign(fmsg, bmsg, want_resp=False):
    """GPI_HDP._safe_exp(LogLik(log(alpha * beta))): labels [N] int64 (first arg-max per row) and, optionally, the one-hot resp."""
    fmsg, bmsg = _dev64(fmsg, "fmsg"), _dev64(bmsg, "bmsg")
    N, K = fmsg.shape
    labels = torch.empty(N, dtype=torch.int64, device=fmsg.device)
    resp = torch.empty((N, K), dtype=torch.float64, device=fmsg.device) if want_resp else None
    _ffi.check(_ffi.lib.hgp_assign_f64(_ptr(fmsg), _ptr(bmsg), N, K, _ptr(labels), _ptr(resp), _stream()), "assign")
    return (labels, resp) if want_resp else labels


def warp_batch(x, Yt, Ym, n_ctrl, iters, noise, lam_s, lam_a, lr, weights=None, u0=None, want_trace=True):
    """8f-4: B monotone time-warps fitted by Adam in one launch.  x [T]; Yt [B,T,D]; Ym [T,D] (shared) or [B,T,D].
    Returns (u [B,n_ctrl], x_warp [B,T], y_warp [B,T,D], trace [B,iters,4] or None)."""
    x, Yt, Ym = _dev64(x.reshape(-1), "x"), _dev64(Yt, "Yt"), _dev64(Ym, "Ym")
    B, T, D = Yt.shape
    stride = 0 if Ym.dim() == 2 else T * D
    dev = x.device
    w = None if weights is None else _dev64(weights.reshape(B), "weights")
    u0 = None if u0 is None else _dev64(u0.reshape(n_ctrl), "u0")
    u = torch.empty((B, n_ctrl), dtype=torch.float64, device=dev)
    xw = torch.empty((B, T), dtype=torch.float64, device=dev)
    yw = torch.empty((B, T, D), dtype=torch.float64, device=dev)
    tr = torch.empty((B, iters, 4), dtype=torch.float64, device=dev) if want_trace else None
    _ffi.check(_ffi.lib.hgp_warp_batch_f64(_ptr(x), _ptr(Yt), _ptr(Ym), stride, T, B, D, int(n_ctrl), int(iters), float(noise),
                                           float(lam_s), float(lam_a), float(lr), _ptr(w), _ptr(u0), _ptr(u), _ptr(xw), _ptr(yw),
                                           _ptr(tr), _stream()), "warp_batch")
    return u, xw, yw, tr
