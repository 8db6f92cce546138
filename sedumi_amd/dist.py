"""Multi-GPU layer of the normal-equations hot path: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in the CPU tests).  Two shardings, the ones SURVEY.md section 8(e) identifies:

* ``ColumnShardedAda`` -- the columns of ADA' are independent once the scaling data is replicated
  (getada3.c:311-351 walks constraint by constraint): every rank forms a contiguous column panel
  (`sdm_plan_getada_cols`) and the panels are exchanged with ONE all-gather (values + absd).  This is the shard of a
  problem whose ADA' is a single dense supernode (MAXCUT, control07): factor and solves then run replicated.
* ``SubtreeShardedSolver`` -- independent elimination-tree subtrees (connected components of the ADA' pattern, e.g.
  the 64 diagonal blocks of the block-diagonal config) are dealt to the ranks; ADA', factor and solves of a
  component never leave its rank -- no data-path collective -- and the only exchange is the all-gather of the
  solution segments y that the (host-side) IPM needs in full.

* ``SeparatorShardedSolver`` -- subtrees of ONE connected elimination tree: the supernodes at the top of the tree (the
  separators every subtree's update matrices meet in) are owned by rank 0, the subtrees below them are dealt to the ranks;
  every rank factors its subtrees, ONE reduce brings the subtree roots' fronts (their update matrices) to the owner, which
  extend-adds them into the separator fronts and factors those; the forward sweep passes the roots' update vectors the same
  way (one reduce), the backward sweep broadcasts the separators' solution.

All classes take an already initialised process group; nothing here launches processes.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.csgraph as csgraph

from . import problem
from .plan import Plan


def _torch():
    import torch
    import torch.distributed as dist
    return torch, dist


def balanced_ranges(weights, parts):
    """Contiguous ranges [b_r, b_{r+1}) with roughly equal weight sums (prefix-sum cuts)."""
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    cuts = [0]
    for r in range(1, parts):
        cuts.append(int(np.searchsorted(cum, cum[-1] * r / parts, side="left")))
    cuts.append(w.size)
    return np.maximum.accumulate(np.asarray(cuts))


class ColumnShardedAda:
    """ADA' formed as column panels, one per rank, all-gathered into every rank's resident plan."""

    def __init__(self, plan: Plan, group=None, device=None, col_weights=None):
        torch, dist = _torch()
        self.plan, self.group = plan, group
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:                                        # single process without a process group
            self.world, self.rank = 1, 0
        self.device = device if device is not None else torch.device("cpu")
        m = plan.m
        jc = np.asarray(plan.ADA_pattern.indptr, dtype=np.int64)
        w = np.diff(jc).astype(np.float64) if col_weights is None else np.asarray(col_weights, dtype=np.float64)
        self.cols = balanced_ranges(w, self.world)                     # column cuts
        self.voff = jc[self.cols]                                       # value offsets of the panels
        self.maxv = int(np.max(np.diff(self.voff))) if self.world else 0
        self.maxc = int(np.max(np.diff(self.cols))) if self.world else 0
        self.chunk = self.maxv + self.maxc                              # [values | absd] padded to a common length
        self.send = torch.zeros(self.chunk, dtype=torch.float64, device=self.device)
        self.recv = torch.zeros(self.chunk * self.world, dtype=torch.float64, device=self.device)
        assert m == self.cols[-1]

    def getada(self):
        torch, dist = _torch()
        r = self.rank
        j0, j1 = int(self.cols[r]), int(self.cols[r + 1])
        self.plan.getada_cols(j0, j1)
        nv = int(self.voff[r + 1] - self.voff[r])
        if nv:
            self.plan.copy("ada", self.send, int(self.voff[r]), nv, to_plan=False)
        if j1 > j0:
            self.plan.copy("absd", self.send[self.maxv:], j0, j1 - j0, to_plan=False)
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        for q in range(self.world):
            if q == r:
                continue
            base = q * self.chunk
            nvq, ncq = int(self.voff[q + 1] - self.voff[q]), int(self.cols[q + 1] - self.cols[q])
            if nvq:
                self.plan.copy("ada", self.recv[base:base + nvq], int(self.voff[q]), nvq, to_plan=True)
            if ncq:
                self.plan.copy("absd", self.recv[base + self.maxv:base + self.maxv + ncq], int(self.cols[q]), ncq, to_plan=True)


class BlockShardedAda:
    """ADA' = sum over PSD blocks of their contributions (getada3.c:305-359 / spscale.c:473-491 treat the blocks
    independently): the PSD blocks are dealt to the ranks (longest-processing-time first on n_k^2 x touching
    constraints), rank 0 also carries the LP / Lorentz rows; every rank forms its partial ADA' and absd on the COMMON
    pattern in its own resident plan, and ONE all-reduce (sum) over [values | absd] assembles ADA' on every rank --
    the form a replicated factorisation wants; a reduce to one owner is the same call with dist.reduce.  This is the
    shard for problems with at least as many PSD blocks as ranks; single-block problems use ColumnShardedAda."""

    def __init__(self, P, L, ADApattern, group=None, device_index=0, device=None):
        torch, dist = _torch()
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.device = device if device is not None else torch.device("cpu")
        s = np.asarray(P.K["s"], dtype=np.float64).ravel()
        At = sp.csc_matrix(P.At)
        _, _, _, _, _, psd = problem._cone_layout(P.K)
        # cost of a block: n_k^2 per constraint that touches it
        touch = np.zeros(s.size)
        rows = At.indices
        if s.size:
            inb = np.searchsorted(psd, rows, side="right") - 1
            ok = rows >= psd[0]
            cols = np.repeat(np.arange(At.shape[1]), np.diff(At.indptr))
            pairs = np.unique(np.stack((inb[ok], cols[ok])), axis=1)
            touch = np.bincount(pairs[0], minlength=s.size).astype(np.float64)
        cost = s ** 2 * np.maximum(touch, 1.0)
        load = np.zeros(self.world)
        owner = np.zeros(s.size, dtype=np.int64)
        for k in np.argsort(-cost, kind="stable"):
            r = int(np.argmin(load)); owner[k] = r; load[r] += cost[k]
        self.blocks_of = [np.flatnonzero(owner == r) for r in range(self.world)]
        self.P = P
        mine = self.blocks_of[self.rank]
        self.sub, self.rows = problem.block_subproblem(P, mine, keep_lq=(self.rank == 0))
        self.plan = Plan(device_index)
        self.plan.set_chol(L, ADApattern)
        self.plan.set_ada(self.sub.At, self.sub.Ablkjc, self.sub.K, problem.lorentz_pattern(self.sub))
        self.nv, self.m = self.plan.nnzADA, self.plan.m
        self.buf = torch.zeros(self.nv + self.m, dtype=torch.float64, device=self.device)
        # absd_j = (LP / Lorentz part of ADA'_jj) + sum |a_j .* z_j| for every constraint j with PSD nonzeros, 0 otherwise
        # (getada3.c:333-351).  A rank's kernel adds the first term only for the columns that have PSD nonzeros in ITS blocks;
        # for a column whose PSD nonzeros all sit in other ranks' blocks rank 0 -- which carries the LP / Lorentz rows, so its
        # partial ADA'_jj IS that first term there -- adds it by hand before the reduction.
        self.fix_cols = self.fix_diag = None
        if self.rank == 0 and self.world > 1 and s.size:
            haspsd = np.zeros(self.m, dtype=bool); haspsd[np.unique(pairs[1])] = True
            mine0 = np.isin(pairs[0], self.blocks_of[0])
            has0 = np.zeros(self.m, dtype=bool); has0[np.unique(pairs[1][mine0])] = True
            cols0 = np.flatnonzero(haspsd & ~has0)
            if cols0.size:
                pat = sp.csc_matrix(self.plan.ADA_pattern)
                diag = np.array([pat.indptr[j] + int(np.searchsorted(pat.indices[pat.indptr[j]:pat.indptr[j + 1]], j)) for j in cols0], dtype=np.int64)
                assert np.array_equal(pat.indices[diag], cols0)
                self.fix_cols = torch.as_tensor(self.nv + cols0, device=self.device)
                self.fix_diag = torch.as_tensor(diag, device=self.device)

    def upload_scaling(self, d, ud, qpr=None):
        """Scaling of the FULL problem: rank 0 keeps d.l / d.det (and the DAt.q values), every rank its blocks of udsqr."""
        if self.rank == 0:
            self.plan.upload("dl", d["l"]); self.plan.upload("ddet", d["det"])
            if qpr is not None and np.size(qpr):
                self.plan.upload("qpr", qpr)
        else:
            self.plan.upload("dl", np.ones(1))
        self.plan.upload("udsqr", problem.block_udsqr(self.P, self.blocks_of[self.rank], ud))

    def getada(self):
        """Partial ADA' + absd of this rank's blocks, then one all-reduce: the plan holds the full ADA' afterwards."""
        torch, dist = _torch()
        self.plan.getada()
        if self.world == 1:
            return
        self.plan.copy("ada", self.buf, 0, self.nv, to_plan=False)
        self.plan.copy("absd", self.buf[self.nv:], 0, self.m, to_plan=False)
        if self.fix_cols is not None:
            self.buf[self.fix_cols] += self.buf[self.fix_diag]
        dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
        self.plan.copy("ada", self.buf, 0, self.nv, to_plan=True)
        self.plan.copy("absd", self.buf[self.nv:], 0, self.m, to_plan=True)


# ----------------------------------------------------------------------------------------- subtree sharding
def components(P):
    """Connected components of the ADA' pattern of problem P = groups of constraints that share a cone variable
    (entry level for LP rows, block level for Lorentz / PSD blocks: getsymbada.m:41-60).  Returns (labels[m], B)
    with B the m x ngroups incidence matrix used for the pattern."""
    B = problem.coupling_incidence(P)
    ncomp, lab = csgraph.connected_components(sp.csr_matrix(B @ B.T), directed=False)
    return ncomp, lab, B


def split_problem(P, world):
    """Deal the connected components of ADA' to `world` ranks (longest-processing-time first on an m_c^3 + nnz cost)
    and build one sub-problem per rank: its constraints and exactly the cone blocks they touch, in SeDuMi's internal
    row order.  Returns [(subproblem, constraint_indices, rowmap)] per rank (subproblem None for an idle rank);
    rowmap = rows of P.At kept, in order."""
    ncomp, lab, _ = components(P)
    At = sp.csc_matrix(P.At)
    sizes = np.bincount(lab, minlength=ncomp)
    nnzc = np.bincount(lab, weights=np.diff(At.indptr), minlength=ncomp)
    cost = sizes.astype(np.float64) ** 3 / 3 + nnzc
    load = np.zeros(world)
    owner = np.zeros(ncomp, dtype=np.int64)
    for c in np.argsort(-cost, kind="stable"):
        r = int(np.argmin(load))
        owner[c] = r; load[r] += cost[c]
    out = []
    for r in range(world):
        cols = np.flatnonzero(owner[lab] == r)
        out.append(problem.subproblem(P, cols) + (cols,) if cols.size else (None, None, cols))
    return [(sub, cols, rows) for (sub, rows, cols) in out]


class SubtreeShardedSolver:
    """Every rank owns whole connected components of ADA': local ADA', factor and solves; y is all-gathered."""

    def __init__(self, P, group=None, device_index=0, device=None, pars=None):
        torch, dist = _torch()
        from . import mex
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:                                        # single process without a process group
            self.world, self.rank = 1, 0
        self.device = device if device is not None else torch.device("cpu")
        self.m = P.m
        parts = split_problem(P, self.world)
        self.cols_of = [c for (_, c, _) in parts]
        self.sub, self.cols, self.rows = parts[self.rank]
        self.pars = pars
        self.maxc = max((c.size for c in self.cols_of), default=0)
        self.send = torch.zeros(max(self.maxc, 1), dtype=torch.float64, device=self.device)
        self.recv = torch.zeros(max(self.maxc, 1) * self.world, dtype=torch.float64, device=self.device)
        self.plan = None
        if self.sub is not None:
            ADApat = problem.symb_ada(self.sub)
            L = mex.symbchol(ADApat)
            self.plan = Plan(device_index)
            self.plan.set_chol(L, ADApat)
            self.plan.set_ada(self.sub.At, self.sub.Ablkjc, self.sub.K, problem.lorentz_pattern(self.sub))
            self.L = L

    def upload_scaling(self, d, ud, P):
        """Scaling data of the FULL problem P (d.l, d.det, udsqr and, with Lorentz cones, d.q1 / d.q2): every rank
        keeps its own rows / blocks.  With Lorentz cones the values of DAt.q (getDAtm.m:39-44, the input of the
        getada2 term) are formed on the device from the rank's q1 / q2 right here."""
        if self.plan is None:
            return
        dl, ddet, uds = problem.sub_scaling(P, self.sub, self.rows, d, ud)
        self.plan.upload("dl", dl); self.plan.upload("ddet", ddet); self.plan.upload("udsqr", uds)
        if self.sub.K["q"].size:
            if "q1" not in d or "q2" not in d:
                raise ValueError("SubtreeShardedSolver.upload_scaling: the problem has Lorentz cones; d must carry "
                                 "q1 and q2 (getDAtm.m:39-44) so that DAt.q can be formed for the getada2 term")
            q1, q2 = problem.sub_scaling_q(P, self.sub, d)
            self.plan.upload("q1", q1); self.plan.upload("q2", q2)
            self.plan.getdatq()

    def factor(self):
        if self.plan is not None:
            self.plan.getada(); self.plan.blkchol(self.pars, True)

    def upload_rhs(self, rhs):
        """Right-hand side of the FULL problem: every rank keeps its own segment in HBM (for solve_resident)."""
        if self.plan is not None:
            self.plan.upload("rhs", np.asarray(rhs, dtype=np.float64)[self.cols])

    def solve_resident(self):
        """The solve of the resident right-hand side; the gathered solution segments stay in `self.recv` (device
        tensor when the group's backend is RCCL): no host round trip -- the form the IPM loop would use."""
        torch, dist = _torch()
        n = self.cols.size if self.cols is not None else 0
        if self.plan is not None:
            self.plan.ldlsolve()
            if self.world == 1 and self.device.type == "cpu":
                return None                      # one rank without a device collective: y stays in the plan
            self.plan.copy("y", self.send, 0, n, to_plan=False)
        if self.world == 1:
            self.recv[:self.send.numel()] = self.send
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)   # the only exchange: solution segments
        return self.recv

    def solve(self, rhs):
        """y = ADA' \\ rhs (full-length host vectors in, full-length host vector out)."""
        torch, dist = _torch()
        n = self.cols.size if self.cols is not None else 0
        if self.plan is not None:
            self.plan.upload("rhs", np.asarray(rhs, dtype=np.float64)[self.cols])
            self.plan.ldlsolve()
            self.plan.copy("y", self.send, 0, n, to_plan=False)
        if self.world == 1:
            self.recv[:self.send.numel()] = self.send
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)   # the only exchange: solution segments
        out = np.zeros(self.m)
        host = self.recv.cpu().numpy()
        for q, c in enumerate(self.cols_of):
            if c is not None and c.size:
                out[c] = host[q * self.send.numel():q * self.send.numel() + c.size]
        return out


# ----------------------------------------------------------------------------------------- separator sharding
def supernodal_etree(L):
    """parent[s] of every supernode of the symbolic factor L = {L (pattern), xsuper}: the supernode of the first row below the
    supernode's own columns -- what blkLDL relinks a finished supernode to (blkchol2.c:550-554: snode[lindx[xlindx[s] + n_s]]) --
    and the flops n m^2 - n^2 m + n^3/3 of each."""
    LL = sp.csc_matrix(L["L"])
    xs = np.asarray(L["xsuper"], dtype=np.int64).ravel() - 1
    nsuper = xs.size - 1
    snode = np.zeros(LL.shape[0], dtype=np.int64)
    for s in range(nsuper):
        snode[xs[s]:xs[s + 1]] = s
    parent = np.full(nsuper, -1, dtype=np.int64)
    cost = np.zeros(nsuper)
    for s in range(nsuper):
        f, n = xs[s], xs[s + 1] - xs[s]
        rows = LL.indices[LL.indptr[f]:LL.indptr[f + 1]]
        ms = rows.size
        if ms > n:
            parent[s] = snode[np.sort(rows)[n]]
        cost[s] = n * ms * ms - n * n * ms + n ** 3 / 3.0
    return parent, cost, xs


def proportional_split(parent, cost, world):
    """Top of the tree + subtrees: starting from the roots, the heaviest subtree is opened (its root joins the top, its children
    become subtrees of their own) until there are at least 2 x world subtrees or none heavier than total / (2 world) can be
    opened; the subtrees are then dealt to the ranks (longest first; rank 0 starts with the top's work).  Returns (top mask,
    owner rank per supernode: the top is rank 0's, roots list)."""
    nsuper = parent.size
    children = [[] for _ in range(nsuper)]
    for s in range(nsuper):
        if parent[s] >= 0:
            children[parent[s]].append(s)
    sub = cost.copy()
    for s in range(nsuper):                     # postordered: children before parents
        if parent[s] >= 0:
            sub[parent[s]] += sub[s]
    total = float(sub[parent < 0].sum())
    work = [int(s) for s in np.flatnonzero(parent < 0)]
    top = np.zeros(nsuper, dtype=bool)
    while world > 1:
        cand = [s for s in work if children[s]]
        if not cand:
            break
        big = max(cand, key=lambda t: sub[t])
        if len(work) >= 2 * world and sub[big] <= total / (2 * world):
            break
        work.remove(big); top[big] = True; work.extend(children[big])
    owner = np.zeros(nsuper, dtype=np.int64)
    load = np.zeros(world); load[0] = float(cost[top].sum())
    root_of = np.full(nsuper, -1, dtype=np.int64)
    for r0 in sorted(work, key=lambda t: -sub[t]):
        rk = int(np.argmin(load)); load[rk] += sub[r0]
        stack = [r0]
        while stack:
            t = stack.pop(); owner[t] = rk; root_of[t] = r0; stack.extend(children[t])
    return top, owner, sorted(work)


class SeparatorShardedSolver:
    """LDL' and solves of ONE symmetric matrix whose elimination tree is connected, sharded over the ranks by subtrees (SURVEY.md
    8e rows blkchol / fwblkslv / bwblkslv).  Every rank holds the symbolic factor and the full front arena (equal layout
    everywhere, so slices of it travel as they are) and the values of the whole matrix (what the ADA' layers above leave on every
    rank); it FACTORS only the supernodes it owns (sdm_plan_set_active_supernodes).  Exchanges, all on device tensors:
      factor   one reduce (sum; every slice has one writer) of the fronts of the subtree roots of ranks 1.. to rank 0, which owns the
               top of the tree (the top is NOT distributed: its levels run on rank 0 while the others wait for the broadcast);
      forward  one reduce of those roots' update vectors to rank 0;
      backward one broadcast of the top's solution;  then one all-reduce of the masked solution segments.
    """

    def __init__(self, X, group=None, device_index=0, device=None, L=None):
        torch, dist = _torch()
        from . import mex
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.device = device if device is not None else torch.device("cpu")
        X = sp.csc_matrix(X); X.sort_indices()
        self.m = X.shape[0]
        self.L = L if L is not None else mex.symbchol(X)          # ordmmd + symfct of the library (bit-exact with the reference): the same on every rank
        parent, cost, xs = supernodal_etree(self.L)
        self.nsuper = parent.size
        self.top, self.owner, self.roots = proportional_split(parent, cost, self.world)
        active = (self.owner == self.rank) & ~self.top
        if self.rank == 0:
            active |= self.top
        self.plan = Plan(device_index)
        self.plan.set_active_supernodes(active.astype(np.int32))
        self.plan.set_chol(self.L, X)
        lay = self.lay = self.plan.front_layout(self.nsuper)
        self.nlevels = lay["nlevels"]
        self.ltop = int(lay["level"][self.top].min()) if self.top.any() else self.nlevels
        # packed exchange buffers: the fronts / update vectors of the subtree roots that have to TRAVEL (those of ranks other than
        # rank 0, which owns the top: its own roots stay where they are -- and may not even be factored yet when the others arrive)
        self.xroots = [s for s in self.roots if self.owner[s] != 0]
        self.f_off = np.concatenate(([0], np.cumsum(lay["fsize"][self.xroots]))).astype(np.int64)
        self.w_off = np.concatenate(([0], np.cumsum(lay["ms"][self.xroots]))).astype(np.int64)
        self.fbuf = torch.zeros(int(self.f_off[-1]), dtype=torch.float64, device=self.device)
        self.wbuf = torch.zeros(int(self.w_off[-1]), dtype=torch.float64, device=self.device)
        self.xbuf = torch.zeros(self.m, dtype=torch.float64, device=self.device)
        perm = np.asarray(self.L["perm"], dtype=np.int64).ravel() - 1
        mine = np.zeros(self.m, dtype=bool)
        for s in np.flatnonzero(active):
            mine[perm[xs[s]:xs[s + 1]]] = True
        self.mask = torch.as_tensor(mine, device=self.device)           # entries of the solution this rank computes (the others of its y are never written)

    def _roots(self, mine):
        """(slot in the exchange buffers, supernode) of the travelling roots this rank owns (mine) / receives (not mine)."""
        return [(i, s) for i, s in enumerate(self.xroots) if (self.owner[s] == self.rank) == mine]

    def factor(self, values, pars=None, absd=None):
        """values: ADA' (the matrix) in the order of its pattern, complete on every rank."""
        torch, dist = _torch()
        pl, lay = self.plan, self.lay
        pl.upload("ada", values)
        if absd is not None:
            pl.upload("absd", absd)
        pl.blkchol_begin(pars, absd is not None)
        if self.world == 1 or not self.top.any():
            pl.blkchol_levels(0, self.nlevels); pl.blkchol_end()
            return
        if self.rank != 0:
            pl.blkchol_levels(0, self.nlevels)
        else:
            pl.blkchol_levels(0, self.ltop)
        self.fbuf.zero_()
        for i, s in self._roots(True):
            n = int(lay["fsize"][s])
            pl.copy("fronts", self.fbuf[int(self.f_off[i]):], int(lay["foff"][s]), n, to_plan=False)
        dist.reduce(self.fbuf, dst=0, op=dist.ReduceOp.SUM, group=self.group)
        if self.rank == 0:
            for i, s in self._roots(False):
                n = int(lay["fsize"][s])
                pl.copy("fronts", self.fbuf[int(self.f_off[i]):], int(lay["foff"][s]), n, to_plan=True)
            pl.blkchol_levels(self.ltop, self.nlevels)
        pl.blkchol_end()

    def solve(self, rhs):
        """x = (L D L')^{-1} rhs in the original order, complete on every rank (a device tensor of length m)."""
        torch, dist = _torch()
        pl, lay = self.plan, self.lay
        pl.upload("rhs", rhs)
        if self.world == 1 or not self.top.any():                      # a forest (or one rank): nothing meets at the top
            pl.solve_levels(3, 0, self.nlevels); pl.solve_levels(4, 0, self.nlevels)
            pl.copy("y", self.xbuf, 0, self.m, to_plan=False)
            if self.world > 1:
                self.xbuf.masked_fill_(~self.mask, 0.0)
                dist.all_reduce(self.xbuf, op=dist.ReduceOp.SUM, group=self.group)
            return self.xbuf
        # forward: every rank its subtrees; the roots' update vectors to the owner of the top; the top
        pl.solve_levels(3, 0, self.nlevels if self.rank != 0 else self.ltop)
        self.wbuf.zero_()
        for i, s in self._roots(True):
            pl.copy("wvec", self.wbuf[int(self.w_off[i]):], int(lay["woff"][s]), int(lay["ms"][s]), to_plan=False)
        dist.reduce(self.wbuf, dst=0, op=dist.ReduceOp.SUM, group=self.group)
        if self.rank == 0:
            for i, s in self._roots(False):
                pl.copy("wvec", self.wbuf[int(self.w_off[i]):], int(lay["woff"][s]), int(lay["ms"][s]), to_plan=True)
            pl.solve_levels(3, self.ltop, self.nlevels)
            pl.solve_levels(4, self.ltop, self.nlevels)               # backward through the top
            pl.copy("xfin", self.xbuf, 0, self.m, to_plan=False)
        dist.broadcast(self.xbuf, src=0, group=self.group)
        if self.rank != 0:
            pl.copy("xfin", self.xbuf, 0, self.m, to_plan=True)
            pl.solve_levels(4, 0, self.nlevels)
        else:
            pl.solve_levels(4, 0, self.ltop)
        pl.copy("y", self.xbuf, 0, self.m, to_plan=False)
        self.xbuf.masked_fill_(~self.mask, 0.0)
        dist.all_reduce(self.xbuf, op=dist.ReduceOp.SUM, group=self.group)
        return self.xbuf


# ----------------------------------------------------------------------------------------- one dense front over the ranks
class BlockCyclicFactor:
    """LDL' of ONE dense front (a single-supernode factor: MAXCUT, control07, every dense ADA') over the ranks, 1-D block-column-cyclic
    (SURVEY.md 8e row blkchol: "block-cyclic dense LDL' with panel broadcasts").  Every rank holds the same plan on the
    launch-per-panel path and the same ADA' values; tile column c (64 columns) belongs to rank (c // blk) % world:

      for every panel q:   all ranks launch panel q -- its owner factors it (cholonBlk, blkchol2.c:96-167, and the rows below), every
                           rank applies the trailing updates that are due to ITS tile columns (precorrect, blkchol2.c:346-420);
                           ONE broadcast from the owner: the panel's columns of the front followed by its record (d, lb, pivot
                           decisions, progress counters, the transposed diagonal block) -- what the relinking of a finished supernode
                           to its parent (blkchol2.c:550-554) becomes when the supernode itself is spread over ranks.

    Afterwards every rank holds the complete factor, d and the pivot lists (the solves are replicated: they do not shard),
    bit for bit what the single plan computes: per tile the same operations in the same order (tests/test_distributed.py).
    """

    def __init__(self, L, ADApat, group=None, device_index=0, device=None, blk=1, plan=None):
        """plan: an existing plan of this factor to take over (its set_chol must have followed set_one_launch_fronts(False): bench.py hands in
        the plan that also forms ADA'); otherwise one is made."""
        torch, dist = _torch()
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.device = device if device is not None else torch.device("cpu")
        xs = np.asarray(L["xsuper"]).ravel()
        if xs.size != 2:
            raise ValueError("BlockCyclicFactor: the symbolic factor must be ONE supernode (a dense front)")
        self.m = int(sp.csc_matrix(ADApat).shape[0])
        self.blk = int(blk)
        if plan is None:
            plan = Plan(device_index)
            plan.set_one_launch_fronts(False)                        # the launch-per-panel path: the panels are exchanged between its launches
            plan.set_chol(L, ADApat)
        self.plan = plan
        self.plan.set_column_owner(self.world, self.rank, self.blk)
        self.npanel = (self.m + 63) // 64
        self.nrec = 4 * 64 + 2 + 64 * 64
        _, n0 = self.plan.panel_slice(0)
        self.buf = torch.zeros(n0 + self.nrec, dtype=torch.float64, device=self.device)

    def owner(self, panel):
        return (panel // self.blk) % self.world

    def factor(self, values, pars=None, absd=None):
        """values: ADA' in the order of its pattern, the same on every rank (what the ADA' layers above leave there)."""
        torch, dist = _torch()
        self.plan.upload("ada", values)
        if absd is not None:
            self.plan.upload("absd", absd)
        self.factor_resident(pars, absd is not None)

    def factor_resident(self, pars=None, use_absd=True):
        """The same on the ADA' / absd the plan already holds (complete and equal on every rank: e.g. after ColumnShardedAda.getada)."""
        torch, dist = _torch()
        pl = self.plan
        pl.blkchol_begin(pars, use_absd)
        for q in range(self.npanel):
            pl.blkchol_panels(0, 1, q, q + 1)
            if self.world == 1:
                continue
            src = self.owner(q)
            off, n = pl.panel_slice(q)
            if self.rank == src:
                pl.panel_record(q, unpack=False)
                pl.copy("fronts", self.buf, off, n, to_plan=False)
                pl.copy("panelrec", self.buf[n:], 0, self.nrec, to_plan=False)
            dist.broadcast(self.buf[:n + self.nrec], src=src, group=self.group)
            if self.rank != src:
                pl.copy("fronts", self.buf, off, n, to_plan=True)
                pl.copy("panelrec", self.buf[n:], 0, self.nrec, to_plan=True)
                pl.panel_record(q, unpack=True)
        pl.blkchol_end()
