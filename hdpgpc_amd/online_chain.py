"""The online step's member updates as PERSISTENT chains (BASELINE configs[4]; GPI_HDP.py:1906-2208 with
GPI_model.py:325-375,705-716,966-1115,1300-1344).

Per beat the reference asks every cluster "what would you look like with this beat?": a deep copy, one Kalman update
(estimate_new), the same update again (include_weighted_sample), the two-step smoother (backwards_pair), the two MNIW updates
(bayesian_new_params), then the latent-transition scores of ALL the copy's members and the MNIW likelihood of its new
parameters - candidate after candidate, ~150 dependent launches each (round 3: 3 700 launches and 62-80 ms per beat at T = 256).
Nothing a candidate computes feeds another one.  Here every cluster model of a lead owns a slot of an ``OnlinePool``:

* its per-step lists live in growing stacks ([rows, T, T]; the model's lists are views, a12) with the device-side position,
  counters and MNIW distributions of GPI_model._chain_alloc - the cluster IS a chain that is never torn down;
* ``candidates(y)`` runs the member step of ALL clusters side by side with the level-fused lists of the offline chains
  (hgp_gemm_list_f64, one launch per dependency level over every cluster's items; the two inversions batched over
  [4 x clusters] / [2 x clusters] matrices) as a DRY run: the new rows land behind each chain's end, the re-smoothed previous
  state stays in the step's buffers, nothing of the cluster changes (finish flags 2 | 4, include/hdpgpc_hip.h).  One Kalman
  update serves estimate_new and the inclusion (the last filtered and smoothed states of an online chain coincide);
* of a candidate's latent-transition scores only three can differ from the cluster's cached ones (member 0 reads the LAST
  transition parameters, the previous member was re-smoothed, the new member): 3 x clusters a8 items and 2 x clusters a9 items
  are gathered by ONE copy-list launch and scored by one batched call each;
* ``commit(g, ...)`` is the same step for the cluster that absorbs the beat, for real and without the smoother (the reference's
  commit does not call backwards_pair, GPI_HDP.py:2186-2196): the previous smoothed mean takes the smoother's place in the
  MNIW update and the previous row is left alone (finish flag 4).
"""
import ctypes

import numpy as np
import torch

from . import _ffi, ops
from .GPI_model import LOG2PI, StackList, matrix_normal_inv_wishart

f64 = torch.float64
_STACKS = ("A", "G", "C", "S", "Psm", "P", "F", "Fsm")          # order of hgp_chain_gather_desc.st
_LISTS = {"A": "A", "G": "Gamma", "C": "C", "S": "Sigma", "Psm": "cov_f_sm", "P": "cov_f", "F": "f_star", "Fsm": "f_star_sm"}
_SH4 = ("X4", "RH4", "Z4", "Y4", "WK4")
_SH2 = ("S__", "S_", "Zs", "Y3", "WK2")


class _LevelLists:
    """The level lists of every slot, level-major in device memory with room for `cap` slots.  All slots have the same item
    shapes, so the per-tile maps of hgp_gemm_list_mapped_f64 depend on the slot index only and are laid down once for the whole
    capacity; adopting a cluster uploads that slot's items (one small copy per level) instead of rebuilding every list."""

    def __init__(self, cap, device):
        self.cap, self.device = cap, device
        self.per = self.tiles = self.dev = self.map = None
        self.keep = {}

    def set_slot(self, c, lv):
        isz = ctypes.sizeof(_ffi.GemmItem)
        if self.per is None:
            self.per = [len(l_._items) for l_ in lv]
            self.tiles = [list(l_._item_tiles) for l_ in lv]
            self.dev, self.map = [], []
            for per, tiles in zip(self.per, self.tiles):
                if self.cap * per > 65535:
                    raise ValueError("online pool: too many clusters for the 16-bit item index of the tile map")
                self.dev.append(torch.zeros(self.cap * per * isz, dtype=torch.uint8, device=self.device))
                one = np.concatenate([(i << 16) | np.arange(n, dtype=np.uint32) for i, n in enumerate(tiles)]).astype(np.uint32)
                allm = (one[None, :] + ((np.arange(self.cap, dtype=np.uint32) * per) << 16)[:, None]).reshape(-1)
                self.map.append(torch.from_numpy(allm.view(np.int32)).to(self.device))
        for l, l_ in enumerate(lv):
            assert len(l_._items) == self.per[l] and list(l_._item_tiles) == self.tiles[l]
            arr = (_ffi.GemmItem * self.per[l])(*l_._items)
            host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
            self.dev[l][c * self.per[l] * isz:(c + 1) * self.per[l] * isz].copy_(host)
        self.keep[c] = lv                      # the operands' storage must outlive the lists

    def run(self, l, lo, hi):
        per, tps = self.per[l], sum(self.tiles[l])
        isz = ctypes.sizeof(_ffi.GemmItem)
        if lo == 0:
            _ffi.check(_ffi.lib.hgp_gemm_list_mapped_f64(ops._ptr(self.dev[l]), per * hi, ops._ptr(self.map[l]), tps * hi, ops._stream()),
                       "gemm_list_mapped")
        else:
            base = ctypes.c_void_p(self.dev[l].data_ptr() + lo * per * isz)
            _ffi.check(_ffi.lib.hgp_gemm_list_f64(base, per * (hi - lo), tps * (hi - lo), ops._stream()), "gemm_list")


class _Slot:
    __slots__ = ("g", "ch", "rows", "N", "ini_noise", "def_diag", "bad0")


class OnlinePool:
    """All cluster models of one lead of an online GPI_HDP, as persistent chains (see the module docstring)."""

    def __init__(self, T, device, annealing, cap=32):
        self.T, self.device, self.annealing = int(T), device, bool(annealing)
        self.slots = []
        self.cap = 0
        self.ybuf = torch.zeros((1, T), dtype=f64, device=device)
        self._pending = None
        self._alloc(cap)

    # ------------------------------------------------------------------ storage
    def _alloc(self, cap):
        T, dev = self.T, self.device
        new = lambda *shape: torch.zeros(shape, dtype=f64, device=dev)            # noqa: E731
        self.cap = cap
        self.WS = 6 * T * T + 2 * T
        self.ws_all = new(cap, self.WS)
        self.shared = {k: new(cap * 4, T, T) for k in _SH4}
        self.shared.update({k: new(cap * 2, T, T) for k in _SH2})
        self.shared["i4"] = torch.zeros(cap * 4, dtype=torch.int32, device=dev)
        self.shared["i2"] = torch.zeros(cap * 2, dtype=torch.int32, device=dev)
        self.rhs_on = torch.tensor([1, 1, 0, 0] * cap, dtype=torch.int32, device=dev)
        self.bad_all = torch.zeros((cap, 2), dtype=torch.int32, device=dev)      # committed steps
        self.badc_all = torch.zeros((cap, 2), dtype=torch.int32, device=dev)     # candidate steps (cleared per beat)
        self.sync_all = torch.zeros(cap, dtype=torch.int32, device=dev)
        self.est_mean, self.mean_last = new(cap, T), new(cap, T)
        # the step buffers of every slot (GPI_model._chain_lists: 21 T x T matrices and 6 vectors) come out of ONE arena, zeroed once:
        # adopting a cluster used to cost 43 allocations and 22 MB of memsets at T = 256
        self.ARENA = 21 * T * T + 6 * T + 128          # (every piece starts on a 32-byte boundary)
        self.arena = new(cap, self.ARENA)
        self.ini_noise_all = new(cap)
        self.lists = _LevelLists(cap, dev)
        self.descs_dirty = True
        self._base = np.zeros((cap, 8), dtype=np.int64)        # byte addresses of the stacks / step buffers of every slot
        self._bufp = np.zeros((cap, 3), dtype=np.int64)
        # inputs of the batched a8 (three members per cluster) and a9 (two parameter pairs per cluster) calls
        self.LF_cur, self.LF_prev = new(cap * 3, T), new(cap * 3, T)
        self.LA, self.LG, self.LC = new(cap * 3, T, T), new(cap * 3, T, T), new(cap * 3, T, T)
        self.MN_M, self.MN_S, self.MN_mean, self.MN_scale = (new(cap * 2, T, T) for _ in range(4))

    def _grow(self):
        """Twice the cluster capacity: the per-cluster slices of the shared buffers move, so every slot's lists are rebuilt."""
        old = self.slots
        self.slots = []
        self._alloc(self.cap * 2)
        for sl in old:
            self._bind(sl)

    def _views(self, c):
        v = {k: self.shared[k][4 * c:4 * c + 4] for k in ("X4", "RH4", "Z4", "Y4", "i4")}
        v.update({k: self.shared[k][2 * c:2 * c + 2] for k in ("S__", "S_", "Zs", "Y3", "i2")})
        return v

    def _bind(self, sl):
        """Give the slot its index, its slices of the shared buffers and its level lists."""
        c = len(self.slots)
        sl.g._slot = c
        ch = sl.ch
        ch["ws"] = self.ws_all[c]
        ch["bad"], ch["sync"] = self.bad_all[c], self.sync_all[c:c + 1]
        ch["Y"], ch["y_row0"] = self.ybuf, -1
        arena, used = self.arena[c], [0]
        arena.zero_()

        def alloc(*shape):
            n = int(np.prod(shape))
            out = arena[used[0]:used[0] + n].view(*shape)
            used[0] += (n + 3) & ~3
            return out

        ch["alloc"] = alloc
        sl.g._chain_lists(ch, views=self._views(c))
        del ch["alloc"]
        T, tt = self.T, self.T * self.T
        Cw = ch["ws"][2 * tt:3 * tt].view(T, T)
        ch["lv"][6].add(Cw, ch["bufs"]["f_post"], self.est_mean[c])          # C_last f_post: the mean estimate_new scores against
        lvm = ops.GemmList(self.device)                                      # C_last f_last: the mean the beat is scored against
        lvm.add(Cw, ch["ws"][6 * tt:6 * tt + T], self.mean_last[c])
        self.lists.set_slot(c, ch["lv"] + [lvm])
        self._base[c] = [ch[k].data_ptr() for k in _STACKS]
        self._bufp[c] = [ch["bufs"][k].data_ptr() for k in ("f_post", "f_sm_prev", "P_sm_prev")]
        self.descs_dirty = True
        dd = sl.g
        self.ini_noise_all[c] = 1e-2 * torch.mean(torch.diagonal(dd.Sigma[0]))
        self.MN_mean[2 * c].copy_(dd.C_def), self.MN_mean[2 * c + 1].copy_(dd.A_def)
        self.MN_scale[2 * c].copy_(dd.Sigma_def), self.MN_scale[2 * c + 1].copy_(dd.Gamma_def)
        self.slots.append(sl)

    @staticmethod
    def supports(g):
        """The chain step covers: dynamic model, no estimation limit, at least one member, last filtered = last smoothed state
        (always true for a chain grown online), no tracked rank-1 factor."""
        if g.N < 1 or g.estimation_limit != np.inf or g._rank1_on() or not g._is_dynamic() or not g._dyn_prior():
            return False
        n = len(g.f_star)
        if not (n == len(g.f_star_sm) == len(g.cov_f) == len(g.cov_f_sm) == len(g.A) == len(g.Gamma) == len(g.C) == len(g.Sigma)):
            return False
        same = lambda a, b: a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape) or bool(torch.equal(a, b))   # noqa: E731
        return same(g.f_star[-1], g.f_star_sm[-1]) and same(g.cov_f[-1], g.cov_f_sm[-1])

    def adopt(self, g):
        """Move the model's per-step lists into a slot's stacks (the model keeps reading them through views)."""
        if len(self.slots) == self.cap:
            self._grow()
        T, dev = self.T, self.device
        L = len(g.f_star)
        sl = _Slot()
        sl.g, sl.N, sl.rows = g, L - 1, max(8, 2 * L)
        ch = {}
        for key in _STACKS:
            lst = getattr(g, _LISTS[key])
            shape = (T, 1) if key in ("F", "Fsm") else (T, T)
            buf = torch.zeros((sl.rows,) + shape, dtype=f64, device=dev)
            buf[:L] = (lst.stack() if isinstance(lst, StackList) else torch.stack(list(lst))).reshape((L,) + shape)
            ch[key] = buf
        ch["pos"] = torch.tensor([L - 1], dtype=torch.int64, device=dev)
        ch["Nf"] = torch.tensor([float(g.N)], dtype=f64, device=dev)
        ch["n0"] = torch.tensor([float(g.internal_params.n0)], dtype=f64, device=dev)
        eye = torch.eye(T, dtype=f64, device=dev)
        mi, mo = g.internal_params, g.observation_params
        ch["W"] = torch.stack((torch.stack((mi.m_mean, mo.m_mean)),
                               torch.stack((eye if mi.m_r_cov is None else mi.m_r_cov, eye if mo.m_r_cov is None else mo.m_r_cov)),
                               torch.stack((mi.scale, mo.scale)))).contiguous()
        sl.ch = ch
        sl.ini_noise = None                                   # log_sq_error's `first` inflation: on the device (self.ini_noise_all)
        if getattr(g, "_def_diag_key", None) == (id(g.Sigma_def), id(g.Gamma_def)):
            sl.def_diag = g._def_diag
        else:
            sl.def_diag = all(bool(torch.equal(s_, torch.diag(torch.diagonal(s_)))) for s_ in (g.Sigma_def, g.Gamma_def))
        sl.bad0 = 0
        self._bind(sl)
        self._rebind_lists(sl, float(mi.n0))
        return sl

    def _rebind_lists(self, sl, n0):
        """The model's lists and MNIW objects as views of the slot's stacks."""
        g, ch, L = sl.g, sl.ch, sl.N + 1
        for key in _STACKS:
            setattr(g, _LISTS[key], StackList(ch[key][:L]))
        W = ch["W"]
        g.internal_params = matrix_normal_inv_wishart(W[0, 0], W[1, 0], n0, W[2, 0])
        g.observation_params = matrix_normal_inv_wishart(W[0, 1], W[1, 1], n0, W[2, 1])
        g._stk = {k: v for k, v in g._stk.items() if k in ("_lat_all", "_lat_col")}

    def _more_rows(self, sl):
        ch = sl.ch
        rows = sl.rows * 2
        for key in _STACKS:
            buf = torch.zeros((rows,) + tuple(ch[key].shape[1:]), dtype=f64, device=self.device)
            buf[:sl.rows] = ch[key]
            ch[key] = buf
        sl.rows = rows
        lat = sl.g._stk.get("_lat_all")
        self._rebind_lists(sl, float(sl.g.internal_params.n0))
        if lat is not None:           # the key holds data pointers of the old stacks
            sl.g._stk["_lat_all"] = (self._lat_key(sl.g), lat[1])
        self._base[sl.g._slot] = [ch[k].data_ptr() for k in _STACKS]
        self.descs_dirty = True

    # ------------------------------------------------------------------ launch tables
    def _prepare(self):
        """Descriptor arrays of the gather / finish launches for the current slots (re-uploaded when a slot is added or its
        stacks move)."""
        p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
        T = self.T
        gd, fd_dry, fd_real = [], [], []
        for c, sl in enumerate(self.slots):
            ch, b = sl.ch, sl.ch["bufs"]
            g = _ffi.ChainGatherDesc()
            for i, k in enumerate(_STACKS):
                g.st[i] = ch[k].data_ptr()
            g.pos, g.out, g.Y, g.y_out, g.W, g.Rp = p(ch["pos"]), p(ch["ws"]), p(self.ybuf), p(b["y"]), p(ch["W"]), p(b["X4"][2:4])
            g.y_row0, g.T = -1, T
            gd.append(g)
            for flags, bad, lst in ((2 | 4, self.badc_all[c], fd_dry), (4, self.bad_all[c], fd_real)):
                f = _ffi.ChainFinishDesc()
                f.f_post, f.c_post, f.f_sm_prev, f.P_sm_prev, f.y = p(b["f_post"]), p(b["c_post"]), p(b["f_sm_prev"]), p(b["P_sm_prev"]), p(b["y"])
                f.part, f.Snew, f.info1, f.info2 = p(b["part"]), p(b["S__"]), p(ch["i4"]), p(ch["i2"])
                f.W, f.n0, f.Nf, f.bad_count = p(ch["W"]), p(ch["n0"]), p(ch["Nf"]), p(bad)
                f.stA, f.stG, f.stC, f.stS = p(ch["A"]), p(ch["G"]), p(ch["C"]), p(ch["S"])
                f.stF, f.stFsm, f.stP, f.stPsm = p(ch["F"]), p(ch["Fsm"]), p(ch["P"]), p(ch["Psm"])
                f.pos, f.sync, f.T, f.annealing = p(ch["pos"]), p(ch["sync"]), T, int(self.annealing) | flags
                lst.append(f)
        from .chain_batch import _descs
        self.gdev, self.fdev_dry, self.fdev_real = _descs(gd, self.device), _descs(fd_dry, self.device), _descs(fd_real, self.device)
        self.riding = self.slots[0].ch["riding"]
        self.descs_dirty = False

    def _gather(self, lo, hi):
        if self.descs_dirty:
            self._prepare()
        gsz = ctypes.sizeof(_ffi.ChainGatherDesc)
        _ffi.check(_ffi.lib.hgp_lds_chain_gather2_batched_f64(ctypes.c_void_p(self.gdev.data_ptr() + lo * gsz), hi - lo, self.T, ops._stream()),
                   "chain_gather2_batched")

    def _step(self, lo, hi, dry, gather=True):
        """The member step of slots [lo, hi): one launch per dependency level (GPI_model._chain_step2 for many chains)."""
        if self.descs_dirty:
            self._prepare()
        T, k, sh = self.T, hi - lo, self.shared
        fsz = ctypes.sizeof(_ffi.ChainFinishDesc)
        if gather:
            self._gather(lo, hi)
        run = lambda l: self.lists.run(l, lo, hi)        # noqa: E731
        s4, s2 = slice(4 * lo, 4 * hi), slice(2 * lo, 2 * hi)
        for l in range(4):
            run(l)
        if self.riding:
            ops.chol_inverse_rhs(sh["X4"][s4], sh["Z4"][s4], sh["RH4"][s4], sh["Y4"][s4], sh["i4"][s4], rhs_on=self.rhs_on[s4])
        else:
            ops.chol_inverse(sh["X4"][s4], out=sh["Z4"][s4], info=sh["i4"][s4], work=sh["WK4"][s4])
            run(10)
        for l in range(4, 8):
            run(l)
        if not dry:        # no smoother in the committed step: the previous smoothed mean stands where f_sm_prev would
            for sl in self.slots[lo:hi]:
                tt = T * T
                sl.ch["bufs"]["f_sm_prev"].copy_(sl.ch["ws"][6 * tt + T:6 * tt + 2 * T])
        run(8)
        if self.riding:
            ops.chol_inverse_rhs(sh["S__"][s2], sh["Zs"][s2], sh["S_"][s2], sh["Y3"][s2], sh["i2"][s2], rhs_trans=True, add_diag=1e-8)
        else:
            ops.chol_inverse(sh["S__"][s2], 0.0, 1e-8, out=sh["Zs"][s2], info=sh["i2"][s2], work=sh["WK2"][s2])
            run(11)
        run(9)
        fdev = self.fdev_dry if dry else self.fdev_real
        _ffi.check(_ffi.lib.hgp_lds_chain_finish2_batched_f64(ctypes.c_void_p(fdev.data_ptr() + lo * fsz), k, T, ops._stream()),
                   "chain_finish2_batched")

    # ------------------------------------------------------------------ the beat under the clusters' last states
    def begin_beat(self, y):
        """Gather every cluster's last state (row pos of its stacks) and score the beat y [T] under it: log_sq_error(x, y, i=-1)
        of GPI_HDP.py:1973 for all clusters in three launches.  Returns (scores, LAPACK infos) in SLOT order [M] (device); the
        gathered state stays valid for candidates() of the same beat."""
        M, T = len(self.slots), self.T
        tt = T * T
        for sl in self.slots:
            if sl.N + 2 > sl.rows:
                self._more_rows(sl)
        self.ybuf.copy_(y.reshape(1, T))
        self._gather(0, M)
        self.lists.run(len(self.lists.per) - 1, 0, M)          # mean_last = C_last f_last
        ar = np.arange(M, dtype=np.int32)
        quad, _, info = ops.score_each(self.ybuf.expand(M, T).contiguous(), self.mean_last, self.ws_all[0, 3 * tt:], ar, ar, None,
                                       strides=(T, self.WS))
        return -0.5 * quad - 0.5 * T * LOG2PI, info

    # ------------------------------------------------------------------ candidates
    def candidates(self, t_new, q_lat_cols, indexes, extra=None):
        """Every cluster with the beat of begin_beat(y) added (dry run).  q_lat_cols [T_all, M']: the clusters' current latent-transition
        columns in SLOT order; indexes[c] = member segment ids of slot c.  Returns
        (est [M] device: estimate_new's score, cols [T_all, M] device: the candidates' latent-transition columns,
         lds [M] host floats: return_LDS_param_likelihood of the candidates).
        extra (optional): the would-be NEW cluster (one member, GPI_HDP.py:1990-1996) - its single latent-transition score and its two
        parameter likelihoods ride the same batched a8 / a9 calls; two more return values then: (its a8 score - device scalar, its
        return_LDS_param_likelihood - host float)."""
        M, T = len(self.slots), self.T
        tt = T * T
        if extra is not None and (M >= self.cap or extra.N != 1 or not extra._dyn_prior()):
            extra = None                                        # no room behind the last slot (or not the plain case): the caller's job
        self.badc_all[:M].zero_()
        self._step(0, M, dry=True, gather=False)               # begin_beat(y) of this beat gathered the state
        # estimate_new: the beat against (C_last f_post, Sigma_last), `first` inflation for one-member clusters
        Y = self.ybuf.expand(M, T).contiguous()
        add = self.ini_noise_all[:M] * ops.to_dev(np.array([1.0 if sl.N == 1 else 0.0 for sl in self.slots]), f64, self.device)
        ar = np.arange(M, dtype=np.int32)
        quad, _, info = ops.score_each(Y, self.est_mean, self.ws_all[0, 3 * tt:], ar, ar, add, strides=(T, self.WS))
        est = -0.5 * quad - 0.5 * T * LOG2PI
        # a8 / a9 inputs by ONE copy-list launch.  Rows: N = members so far = index of the last row; N + 1 = the dry run's row
        N = np.array([sl.N for sl in self.slots], dtype=np.int64)
        one = N == 1
        base, bufp = self._base[:M], self._bufp[:M]
        iA, iG, iC, iS, iPsm, iP, iF, iFsm = range(8)
        row = lambda k, r, n: base[:, k] + r * (8 * n)                           # noqa: E731  (byte address of a stack row)
        f_post, f_smp, P_smp = bufp[:, 0], bufp[:, 1], bufp[:, 2]
        src, dst, cnt = [], [], []

        def put(s, dbuf, j, n):
            src.append(s), dst.append(dbuf.data_ptr() + (np.arange(M, dtype=np.int64) * (dbuf.shape[0] // self.cap) + j) * (8 * n))
            cnt.append(np.full(M, n, dtype=np.int64))

        # member 0: cur = prev = row 1, cov = row 1 (the re-smoothed one when it is also the previous member), par = the NEW row
        f1 = np.where(one, f_smp, row(iFsm, 1, T))
        put(f1, self.LF_cur, 0, T), put(f1, self.LF_prev, 0, T)
        put(row(iA, N + 1, tt), self.LA, 0, tt), put(row(iG, N + 1, tt), self.LG, 0, tt)
        put(np.where(one, P_smp, row(iPsm, 1, tt)), self.LC, 0, tt)
        # member N - 1 (N >= 2; a repeat of member 0's inputs otherwise, ignored): cur = re-smoothed row N, prev / cov = row N - 1, par = N
        Nm = np.maximum(N - 1, 1)
        put(f_smp, self.LF_cur, 1, T), put(row(iFsm, Nm, T), self.LF_prev, 1, T)
        put(row(iA, N, tt), self.LA, 1, tt), put(row(iG, N, tt), self.LG, 1, tt), put(row(iPsm, Nm, tt), self.LC, 1, tt)
        # the new member: cur = f_post, prev / cov = the re-smoothed row N, par = the NEW row
        put(f_post, self.LF_cur, 2, T), put(f_smp, self.LF_prev, 2, T)
        put(row(iA, N + 1, tt), self.LA, 2, tt), put(row(iG, N + 1, tt), self.LG, 2, tt), put(P_smp, self.LC, 2, tt)
        # a9: (C, Sigma) and (A, Gamma) of the NEW row against their priors
        put(row(iC, N + 1, tt), self.MN_M, 0, tt), put(row(iS, N + 1, tt), self.MN_S, 0, tt)
        put(row(iA, N + 1, tt), self.MN_M, 1, tt), put(row(iG, N + 1, tt), self.MN_S, 1, tt)
        nl, nm = 3 * M, 2 * M
        diag = all(sl.def_diag for sl in self.slots)
        if extra is not None:   # one a8 item (member 0 of a one-member cluster: GPI_model._lat_indices) and two a9 items behind the slots'
            e = extra
            fs, cs = e.f_star_sm[1].contiguous(), e.cov_f_sm[1].contiguous()
            Ae, Ge, Ce, Se = (m_.contiguous() for m_ in (e.A[-1], e.Gamma[-1], e.C[-1], e.Sigma[-1]))
            defs = [m_.contiguous() for m_ in (e.C_def, e.Sigma_def, e.A_def, e.Gamma_def)]
            self._extra_keep = (fs, cs, Ae, Ge, Ce, Se, defs)
            one_ = lambda t, buf, j, n: (src.append(np.array([t.data_ptr()], dtype=np.int64)),                       # noqa: E731
                                         dst.append(np.array([buf.data_ptr() + j * 8 * n], dtype=np.int64)), cnt.append(np.array([n], dtype=np.int64)))
            one_(fs, self.LF_cur, nl, T), one_(fs, self.LF_prev, nl, T), one_(Ae, self.LA, nl, tt), one_(Ge, self.LG, nl, tt), one_(cs, self.LC, nl, tt)
            one_(Ce, self.MN_M, nm, tt), one_(Se, self.MN_S, nm, tt), one_(Ae, self.MN_M, nm + 1, tt), one_(Ge, self.MN_S, nm + 1, tt)
            one_(defs[0], self.MN_mean, nm, tt), one_(defs[1], self.MN_scale, nm, tt)
            one_(defs[2], self.MN_mean, nm + 1, tt), one_(defs[3], self.MN_scale, nm + 1, tt)
            nl, nm = nl + 1, nm + 2
            if getattr(e, "_def_diag_key", None) == (id(e.Sigma_def), id(e.Gamma_def)):
                diag = diag and e._def_diag
            else:
                diag = diag and all(bool(torch.equal(s_, torch.diag(torch.diagonal(s_)))) for s_ in (e.Sigma_def, e.Gamma_def))
        table = np.stack([np.concatenate(src), np.concatenate(dst), np.concatenate(cnt)], axis=1)
        tdev = ops.to_dev(table, torch.int64, self.device)
        ops.copy_list(tdev, table.shape[0], tt)
        lat, info_l = ops.lat_error(self.LF_cur[:nl], self.LF_prev[:nl], self.LA[:nl], self.LG[:nl], self.LC[:nl])
        lat = lat - 0.5 * T * LOG2PI
        if diag:
            mn, info_m = ops.mniw_loglik(self.MN_M[:nm], self.MN_S[:nm], self.MN_mean[:nm], None, self.MN_scale[:nm], scale_is_diagonal=True)
        else:
            mn, info_m = ops.mniw_loglik(self.MN_M[:nm], self.MN_S[:nm], self.MN_mean[:nm], None, self.MN_scale[:nm], scale_is_diagonal=False)
        lds_dev = torch.sum(mn.view(nm // 2, 2), dim=1) / T * 100.0
        # one host round trip for everything the loop branches on
        flat = torch.cat([lds_dev, info.to(f64), info_l.to(f64), info_m.to(f64), self.badc_all[:M].reshape(-1).to(f64)]).cpu().numpy()
        nx = nm // 2
        lds = flat[:M]
        if flat[nx:nx + M + nl + nm].any():       # score [M] / a8 [nl] / a9 [nm] infos
            bad = int(np.nonzero(flat[nx:nx + M + nl + nm])[0][0])
            what = "log_sq_error" if bad < M else ("log_lat_error" if bad < M + nl else "log_likelihood_MNIW")
            raise torch.linalg.LinAlgError(f"{what}: the input is not positive-definite (online candidate step)")
        badc = flat[nx + M + nl + nm:].reshape(M, 2)
        if badc[:, 1].any():
            raise torch.linalg.LinAlgError("posterior / backwards_pair: the input is not positive-definite (online candidate step)")
        # the candidates' columns: the cluster's own column with (up to) three entries replaced
        cols = q_lat_cols.clone()
        rr, cc, vv = [], [], []
        for c, sl in enumerate(self.slots):
            idx = indexes[c]
            rr += [idx[0], t_new]
            cc += [c, c]
            vv += [3 * c, 3 * c + 2]
            if sl.N >= 2:
                rr.append(idx[sl.N - 1]), cc.append(c), vv.append(3 * c + 1)
        dev = self.device
        ix = ops.to_dev(np.array([rr, cc, vv]), torch.int64, dev)
        cols[ix[0], ix[1]] = lat[ix[2]]
        if extra is not None:
            return est, cols, lds, lat[3 * M], float(flat[M])
        return est, cols, lds

    # ------------------------------------------------------------------ commit
    @staticmethod
    def _lat_key(g, h_ini=1.0):
        return (h_ini, len(g.indexes), len(g.Gamma), g.f_star_sm[-1].data_ptr(), g.cov_f_sm[-1].data_ptr())

    def commit(self, g, index, x_train, y):
        """include_weighted_sample(h = 1) + bayesian_new_params(1) of the cluster that absorbs the beat (no smoother).  The launches
        are enqueued and the host-side bookkeeping is done here; the two status words the reference looks at (a failed filter
        step raises, a failed MNIW factorisation keeps the previous distributions, GPI_model.py:1068) are read by finish_commit(),
        so that the caller's host work (the HDP global step) overlaps the device work."""
        self.finish_commit()
        sl = self.slots[g._slot]
        T = self.T
        if sl.N + 2 > sl.rows:
            self._more_rows(sl)
        lat_old = g._stk.get("_lat_all")
        if lat_old is not None and lat_old[0] != self._lat_key(g):
            lat_old = None
        yv = g.cond_to_torch(y).reshape(-1, 1)
        self.ybuf.copy_(yv.reshape(1, T))
        c = g._slot
        self._step(c, c + 1, dry=False)
        sl.N += 1
        g.N += 1
        g.indexes.append(int(index))
        g.x_train.append(x_train)
        g.y_train.append(yv)
        self._rebind_lists(sl, float(g.internal_params.n0) + 1.0)
        info = None
        if lat_old is not None:
            # latent-transition scores: member 0 now reads the new last parameters, the new member is added; the rest is unchanged
            n = sl.N                                          # members now; rows 0..n
            ch = sl.ch
            fs, ps = ch["Fsm"], ch["Psm"]
            cur = torch.stack((fs[1], fs[n])).reshape(2, T)
            prv = torch.stack((fs[1], fs[n - 1])).reshape(2, T)
            A2, G2 = torch.stack((ch["A"][n], ch["A"][n])), torch.stack((ch["G"][n], ch["G"][n]))
            C2 = torch.stack((ps[1], ps[n - 1]))
            out, info = ops.lat_error(cur, prv, A2, G2, C2)
            out = out - 0.5 * T * LOG2PI
            new = torch.cat([out[0:1], lat_old[1][1:], out[1:2]])
            g._stk["_lat_all"] = (self._lat_key(g), new)
            col = g._stk.get("_lat_col")                       # the scattered column: two entries change
            if col is not None and col[0] is lat_old[1] and col[1].shape[0] > int(index):
                col[1][ops.to_dev([g.indexes[0], int(index)], torch.int64, self.device)] = out
                g._stk["_lat_col"] = (new, col[1])
        self._pending = (sl, info)

    def finish_commit(self):
        if self._pending is None:
            return
        (sl, info), self._pending = self._pending, None
        g, c = sl.g, sl.g._slot
        flat = torch.cat([self.bad_all[c].to(f64)] + ([] if info is None else [info.to(f64)])).tolist()     # one round trip
        if flat[1] != 0:
            raise torch.linalg.LinAlgError("posterior: the input is not positive-definite (online step)")
        if any(flat[2:]):
            raise torch.linalg.LinAlgError("log_lat_error: the input is not positive-definite (online step)")
        updated = int(flat[0]) == sl.bad0
        sl.bad0 = int(flat[0])
        if not updated:                                        # the MNIW update was skipped: the distributions kept their count
            if g.verbose:
                print("Alg error matrix ill conditioned.")
            g.internal_params.n0 -= 1.0
            g.observation_params.n0 -= 1.0


class CandidateView:
    """What the one-sample bound asks of a candidate cluster (offline_loop.full_LDS_elbo): its MNIW parameter likelihood."""

    def __init__(self, value):
        self._v = float(value)

    def lds_param_likelihood_value(self):
        return self._v
