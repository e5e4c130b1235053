"""Many independent member chains side by side ("the sequential LDS recursion does not shard over segments:
parallel over independent chains - clusters x leads x birth proposals").

One chain = GPI_model.full_pass_weighted of one model over its members (GPI_model.py:377-406): strictly sequential, ~14
dependent launches per member, each a few microseconds of work on a few compute units - the chip is >90 % idle.  The
variational loop, however, asks for many chains that do not depend on each other (the clusters a proposal changes, the leads,
the proposals of one exploration round, the classes of reload_model_from_labels).  ``run(jobs)`` advances all of them in
lock-step with the SAME number of launches per member step as one chain:

* every dependency level of the step is one ``hgp_gemm_list_f64`` launch over the concatenated item lists of all chains,
* the two inversions of the step are one ``hgp_chol_inverse_rhs_batched_f64`` launch each over [4 * chains] / [2 * chains]
  matrices, gather and finish are the descriptor-array launches ``hgp_lds_chain_*2_batched_f64`` (blockIdx.y = chain),
* chains are sorted by length; when the shortest live chain ends the launches simply shrink to a prefix (every buffer and list
  is laid out chain-major), each phase captured once as a hipGraph and replayed,
* the backward (RTS) recursions then run concurrently, one stream per chain.

Arithmetic per chain is exactly that of GPI_model._chain_step2: results are bit-identical to running the chains one after the
other.  Chains the graphed path does not cover (soft members, irregular grids, T > 128, fewer than 4 members) fall back to
GPI_model.full_pass_weighted one by one.
"""
import ctypes

import numpy as np
import torch

from . import _ffi, ops

f64 = torch.float64


class Job:
    """One chain: model `gp` absorbs the segments with resp > 0.99; `prev` = (q, q_lat) handed back when there are none."""

    def __init__(self, gp, x_trains, y_trains, resp, prev=(None, None)):
        self.gp, self.x, self.y, self.resp, self.prev = gp, x_trains, y_trains, torch.as_tensor(resp), prev
        self.active = torch.nonzero(self.resp > 0.99, as_tuple=False).reshape(-1).tolist()
        self.out = None


def _graphable(job):
    gp = job.gp
    a = job.active
    if len(a) < 4 or gp.x_basis.shape[0] > 128 or gp.estimation_limit != np.inf:
        return False
    if not bool(torch.any(gp.Gamma[-1] != 0)) or not bool(torch.all(job.resp[a] == 1.0)):
        return False
    X2 = job.x[..., 0] if job.x.ndim == 3 else job.x
    return bool(torch.equal(X2[a], gp.x_basis.reshape(1, -1).expand(len(a), -1)))


def _descs(structs, dev):
    arr = (type(structs[0]) * len(structs))(*structs)
    return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)


def run(jobs):
    """Execute the jobs; returns [(q, q_lat)] in job order (scores of every segment under the finished model, as
    full_pass_weighted returns them)."""
    for j in jobs:
        j.x, j.y = j.gp.cond_to_torch(j.x), j.gp.cond_to_torch(j.y)
    by_T = {}                                          # chains advance in lock-step only with chains of their own basis length
    for j in jobs:
        if len(j.active) and _graphable(j):
            by_T.setdefault(int(j.gp.x_basis.shape[0]), []).append(j)
    groups = [g for g in by_T.values() if len(g) >= 2]  # a lone chain has nothing to run beside
    fast = [j for g in groups for j in g]
    for j in jobs:
        if not len(j.active):
            j.out = j.prev
        elif not any(j is f for f in fast):
            j.out = j.gp.full_pass_weighted(j.x, j.y, j.resp, q=j.prev[0], q_lat=j.prev[1])
    for g in groups:
        _run_fast(g)
    return [j.out for j in jobs]


def _run_fast(jobs):
    dev = jobs[0].gp.device
    T = jobs[0].gp.x_basis.shape[0]
    assert all(j.gp.x_basis.shape[0] == T for j in jobs)
    # the first member of a fresh model takes the eager path (kernel fit, prior-predictive first step)
    for j in jobs:
        gp = j.gp
        head = 1 if gp.N == 0 else 0
        for index in j.active[:head]:
            gp.include_weighted_sample(index, j.x[index], j.x[index], j.y[index], 1.0)
            gp.backwards_pair(1.0)
            gp.bayesian_new_params(1.0)
        j.rest = j.active[head:]
        gp._check_pending()
    jobs = sorted(jobs, key=lambda j: -len(j.rest))    # longest first: the live set is always a prefix
    nc = len(jobs)
    new = lambda *shape: torch.zeros(shape, dtype=f64, device=dev)      # noqa: E731
    shared = {"X4": new(nc * 4, T, T), "RH4": new(nc * 4, T, T), "Z4": new(nc * 4, T, T), "Y4": new(nc * 4, T, T),
              "S__": new(nc * 2, T, T), "S_": new(nc * 2, T, T), "Zs": new(nc * 2, T, T), "Y3": new(nc * 2, T, T),
              "i4": torch.zeros(nc * 4, dtype=torch.int32, device=dev), "i2": torch.zeros(nc * 2, dtype=torch.int32, device=dev)}
    rhs_on = torch.tensor([1, 1, 0, 0] * nc, dtype=torch.int32, device=dev)
    chs, gd, fd = [], [], []
    for c, j in enumerate(jobs):
        gp = j.gp
        ch = gp._chain_alloc(len(j.rest))
        ch["Y"] = (j.y[j.rest][..., 0] if j.y.ndim == 3 else j.y[j.rest]).reshape(len(j.rest), -1).contiguous()
        ch["y_row0"] = int(ch["pos"][0])
        gp._chain_lists(ch, views={k: v[(4 if k in ("X4", "RH4", "Z4", "Y4", "i4") else 2) * c:(4 if k in ("X4", "RH4", "Z4", "Y4", "i4") else 2) * (c + 1)]
                                   for k, v in shared.items()})
        b = ch["bufs"]
        p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
        g = _ffi.ChainGatherDesc()
        for i, k in enumerate(("A", "G", "C", "S", "Psm", "P", "F", "Fsm")):
            g.st[i] = ch[k].data_ptr()
        g.pos, g.out, g.Y, g.y_out, g.W, g.Rp = p(ch["pos"]), p(ch["ws"]), p(ch["Y"]), p(b["y"]), p(ch["W"]), p(b["X4"][2:4])
        g.y_row0, g.T = ch["y_row0"], T
        f = _ffi.ChainFinishDesc()
        f.f_post, f.c_post, f.f_sm_prev, f.P_sm_prev, f.y = p(b["f_post"]), p(b["c_post"]), p(b["f_sm_prev"]), p(b["P_sm_prev"]), p(b["y"])
        f.part, f.Snew, f.info1, f.info2 = p(b["part"]), p(b["S__"]), p(ch["i4"]), p(ch["i2"])
        f.W, f.n0, f.Nf, f.bad_count = p(ch["W"]), p(ch["n0"]), p(ch["Nf"]), p(ch["bad"])
        f.stA, f.stG, f.stC, f.stS = p(ch["A"]), p(ch["G"]), p(ch["C"]), p(ch["S"])
        f.stF, f.stFsm, f.stP, f.stPsm = p(ch["F"]), p(ch["Fsm"]), p(ch["P"]), p(ch["Psm"])
        f.pos, f.sync, f.T, f.annealing = p(ch["pos"]), p(ch["sync"]), T, int(bool(gp.annealing))
        chs.append(ch), gd.append(g), fd.append(f)
    gdev, fdev = _descs(gd, dev), _descs(fd, dev)
    # level lists, chain-major: items [0, n_l * k) of level l belong to the first k chains
    n_lv = len(chs[0]["lv"])
    per = [len(chs[0]["lv"][l]._items) for l in range(n_lv)]
    merged = [ops.GemmList.concat([ch["lv"][l] for ch in chs]).finalize() for l in range(n_lv)]
    stream = ops._stream

    def step(k):                                           # one member of the first k chains
        _ffi.check(_ffi.lib.hgp_lds_chain_gather2_batched_f64(ctypes.c_void_p(gdev.data_ptr()), k, T, stream()), "chain_gather2_batched")
        for l in range(4):
            merged[l].run_range(0, per[l] * k)
        ops.chol_inverse_rhs(shared["X4"][:4 * k], shared["Z4"], shared["RH4"], shared["Y4"], shared["i4"], rhs_on=rhs_on)
        for l in range(4, 9):
            merged[l].run_range(0, per[l] * k)
        ops.chol_inverse_rhs(shared["S__"][:2 * k], shared["Zs"], shared["S_"], shared["Y3"], shared["i2"], rhs_trans=True, add_diag=1e-8)
        merged[9].run_range(0, per[9] * k)
        _ffi.check(_ffi.lib.hgp_lds_chain_finish2_batched_f64(ctypes.c_void_p(fdev.data_ptr()), k, T, stream()), "chain_finish2_batched")

    lengths = [len(j.rest) for j in jobs]
    done = 0
    for k in range(nc, 0, -1):                             # phase: the first k chains are alive for lengths[k-1] - done steps
        n_it = lengths[k - 1] - done
        if n_it > 0:
            _replay(lambda: step(k), n_it, unroll=8)
            done += n_it
    keep = (shared, gdev, fdev, merged, rhs_on)             # alive until the work is done
    side = [torch.cuda.Stream() for _ in jobs]
    main = torch.cuda.current_stream()
    for j, ch, s in zip(jobs, chs, side):                  # commit + backward recursion, one stream per chain
        gp = j.gp
        gp._chain_commit(ch, j.rest, j.x, j.y)
        bad = ch["bad"].tolist()
        if bad[1] != 0:
            raise torch.linalg.LinAlgError(f"posterior / backwards_pair: the input is not positive-definite (LDS step {bad[1]})")
        s.wait_stream(main)
        with torch.cuda.stream(s):
            gp._backwards_graphed()
    for s in side:
        main.wait_stream(s)
    for j in jobs:
        gp = j.gp
        gp._check_pending()
        gp._stk = {}
        j.out = (gp.compute_sq_err_all(j.x, j.y), gp.compute_q_lat_all(j.x))
    del keep


def _replay(fn, n_iter, unroll=8):
    """fn() n_iter times: once eagerly on a side stream (warm-up = first iteration), then `unroll` iterations captured as ONE
    hipGraph and replayed; the remainder eagerly."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    n_iter -= 1
    if n_iter <= 0:
        return
    unroll = max(1, min(int(unroll), n_iter))
    if n_iter >= 2 * unroll:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(unroll):
                fn()
        for _ in range(n_iter // unroll):
            graph.replay()
        n_iter = n_iter % unroll
        torch.cuda.current_stream().synchronize()          # the graph object dies with this frame
    for _ in range(n_iter):
        fn()
