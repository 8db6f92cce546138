"""Synthetic cone problems directly in SeDuMi's internal (post-pretransfo) form.

The hot path consumes ``At`` (N x m CSC, rows = [x0 | LP | Lorentz trace |
Lorentz norm-bound | real PSD blocks, lower triangle folded]) plus the K fields
of pretransfo.m:531-542.  These builders produce that form for the workloads
named in BASELINE.json / SURVEY.md section 8(d) without needing the reference's
example files (which do not travel to the GPU box):

  control_like   -- control07-shaped SDP: m=666, K.s=[70 35], dense ADA'
  maxcut         -- MAXCUT relaxation, one dense PSD block, A_i = e_i e_i'
  blockdiag_sdp  -- nblk PSD blocks, each constraint touches one block
  random_sdp     -- small mixed LP + Lorentz + PSD problems for parity tests
  lp_dense_cols  -- sparse LP with a few dense columns (dpr1fact path)

Host utilities only (numpy/scipy); nothing here is on the timed path.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def make_K(lpN, q, s, hs=()):
    """K struct (dict of float arrays, MATLAB conventions) as left by pretransfo.m:486-542.
    lpN includes the artificial x0 row.  s = real symmetric PSD blocks, hs = Hermitian PSD blocks (stored behind
    the real ones as [Re; Im], 2 n^2 rows each; K.s lists both, K.rsdpN = number of real blocks)."""
    q = np.asarray(q, dtype=np.float64).ravel()
    sr = np.asarray(s, dtype=np.float64).ravel()
    sh = np.asarray(hs, dtype=np.float64).ravel()
    s = np.concatenate((sr, sh))
    blkstart = np.cumsum(np.concatenate(([lpN + 1, q.size], q - 1, sr ** 2, 2 * sh ** 2))).astype(np.float64)
    mb = blkstart[np.cumsum([0, 1, q.size])]
    return {
        "f": 0.0, "l": float(lpN), "q": q.reshape(1, -1), "r": np.zeros((0, 1)), "s": s.reshape(1, -1),
        "rsdpN": float(sr.size), "N": float(blkstart[-1] - 1), "blkstart": blkstart.reshape(1, -1),
        "rLen": float(sr.sum()), "hLen": float(sh.sum()), "qMaxn": float(q.max() if q.size else 0),
        "rMaxn": float(sr.max() if sr.size else 0), "hMaxn": float(sh.max() if sh.size else 0), "mainblks": mb.reshape(1, -1),
        "qblkstart": blkstart[1:2 + q.size].reshape(1, -1), "sblkstart": blkstart[1 + q.size:].reshape(1, -1),
        "lq": float(mb[-1] - 1),
    }


def partitA(At, K):
    """Ablkjc (m x 3 doubles, 0-based offsets) -- what partitA.c:137-146 returns for K.mainblks."""
    At = sp.csc_matrix(At)
    m = At.shape[1]
    mb = K["mainblks"].ravel().astype(np.int64) - 1
    out = np.zeros((m, 3))
    for j in range(m):
        b, e = At.indptr[j], At.indptr[j + 1]
        out[j, :] = b + np.searchsorted(At.indices[b:e], mb)
    return out


class Problem:
    def __init__(self, At, K, name):
        At = sp.csc_matrix(At, dtype=np.float64)
        At.sum_duplicates(); At.sort_indices()
        self.At, self.K, self.name = At, K, name
        self.m = At.shape[1]
        self.Ablkjc = partitA(At, K)


def _psd_rows(K):
    """0-based first row of every PSD block (real blocks n^2 rows, Hermitian blocks 2 n^2) and the block orders."""
    start = K["sblkstart"].ravel().astype(np.int64) - 1
    return start, K["s"].ravel().astype(np.int64)


def control_like(seed=0, m=666, n1=70, n2=35):
    """control07-shaped (examples/control07.mat after pretransfo): per constraint ~160 folded nonzeros in
    the 70-block (36 with a 70-entry diagonal, 35 with ~630, the rest ~140) and one entry of the 35-block
    for the first n2(n2+1)/2 constraints."""
    rng = np.random.default_rng(seed)
    K = make_K(1, [], [n1, n2])
    start, ns = _psd_rows(K)
    N = int(K["N"])
    tri1 = [(r, c) for c in range(n1) for r in range(c, n1)]
    tri2 = [(r, c) for c in range(n2) for r in range(c, n2)]
    rows, cols, vals = [], [], []
    for j in range(m):
        if j < 36:
            pos = [(i, i) for i in range(n1)]
        else:
            cnt = 631 if j >= m - 35 else int(rng.integers(100, 200))
            idx = rng.choice(len(tri1), size=min(cnt, len(tri1)), replace=False)
            pos = [tri1[i] for i in idx]
        for (r, c) in pos:
            rows.append(start[0] + r + c * n1); cols.append(j); vals.append(rng.standard_normal() * (1.0 if r == c else 2.0))
        if j < len(tri2):
            r, c = tri2[j]
            rows.append(start[1] + r + c * n2); cols.append(j); vals.append(1.0 if r == c else 2.0)
    At = sp.csc_matrix((vals, (rows, cols)), shape=(N, m))
    return Problem(At, K, f"control_like(m={m},s=[{n1},{n2}])")


def maxcut(n, seed=2):
    """MAXCUT SDP relaxation: K.s=n, A_i = e_i e_i' (SURVEY.md 8d config 4)."""
    K = make_K(1, [], [n])
    start, _ = _psd_rows(K)
    rows = start[0] + np.arange(n) * (n + 1)
    At = sp.csc_matrix((np.ones(n), (rows, np.arange(n))), shape=(int(K["N"]), n))
    return Problem(At, K, f"maxcut(n={n})")


def blockdiag_sdp(nblk=64, n=200, mper=150, nnz=20, seed=4):
    """nblk PSD blocks of order n; constraint j touches block j//mper only with `nnz` random lower-triangle
    entries plus a diagonal shift."""
    rng = np.random.default_rng(seed)
    K = make_K(1, [], [n] * nblk)
    start, _ = _psd_rows(K)
    m = nblk * mper
    rows, cols, vals = [], [], []
    for j in range(m):
        k = j // mper
        r = rng.integers(0, n, size=nnz); c = rng.integers(0, n, size=nnz)
        lo, hi = np.minimum(r, c), np.maximum(r, c)
        rows.extend(start[k] + hi + lo * n); cols.extend([j] * nnz)
        vals.extend(rng.standard_normal(nnz) * np.where(hi == lo, 1.0, 2.0))
        dsel = rng.integers(0, n, size=3)
        rows.extend(start[k] + dsel * (n + 1)); cols.extend([j] * 3); vals.extend([1.0, 1.0, 1.0])
    At = sp.csc_matrix((vals, (rows, cols)), shape=(int(K["N"]), m))
    return Problem(At, K, f"blockdiag_sdp({nblk}x{n},m={m})")


def random_sdp(m=30, lp=6, q=(3, 4), s=(5, 7, 4), dens=0.3, seed=0, block_local=False, hs=()):
    """Small mixed LP + Lorentz + PSD problem.  block_local=True makes every constraint touch a single
    cone block (sparse ADA' pattern: exercises ordering / multi-supernode factor)."""
    rng = np.random.default_rng(seed)
    K = make_K(lp + 1, q, s, hs)
    N = int(K["N"])
    bs = K["blkstart"].ravel().astype(np.int64) - 1
    nq = len(q)
    start, ns = _psd_rows(K)
    nreal = len(s)
    rows, cols, vals = [], [], []
    nblocks = 1 + nq + len(s) + len(hs)
    # block_local: keep the number of constraints per block below the block's dimension so that ADA' stays
    # nonsingular (otherwise the pivots of the dependent constraints are pure rounding noise)
    caps = [int(0.7 * lp)] + [int(0.7 * qk) for qk in q] + [int(0.7 * n * (n + 1) / 2) for n in ns]
    picks = []
    if block_local:
        left = list(caps)
        for j in range(m):
            avail = [b for b in range(nblocks) if left[b] > 0]
            if not avail:
                raise ValueError("random_sdp(block_local): m exceeds the capacity of the cone blocks")
            b = int(avail[rng.integers(0, len(avail))])
            left[b] -= 1
            picks.append(b)
    for j in range(m):
        pick = picks[j] if block_local else -1
        if lp and (pick in (-1, 0)):
            for r in range(1, lp + 1):
                if rng.random() < dens:
                    rows.append(r); cols.append(j); vals.append(rng.standard_normal())
        for k in range(nq):
            if pick not in (-1, 1 + k):
                continue
            if rng.random() < 0.8:
                rows.append(lp + 1 + k); cols.append(j); vals.append(rng.standard_normal())
            for r in range(bs[1 + k], bs[2 + k]):
                if rng.random() < dens:
                    rows.append(r); cols.append(j); vals.append(rng.standard_normal())
        for k, n in enumerate(ns):
            if pick not in (-1, 1 + nq + k):
                continue
            for c in range(n):
                for r in range(c, n):
                    if rng.random() < dens:
                        rows.append(start[k] + r + c * n); cols.append(j)
                        vals.append(rng.standard_normal() * (1.0 if r == c else 2.0))
                    if k >= nreal and r > c and rng.random() < dens:      # imaginary plane: strictly lower, folded (x2)
                        rows.append(start[k] + n * n + r + c * n); cols.append(j)
                        vals.append(rng.standard_normal() * 2.0)
        if not any(cc == j for cc in cols[-1:]):      # never leave a constraint empty
            rows.append(1 if lp else start[0]); cols.append(j); vals.append(1.0)
    At = sp.csc_matrix((vals, (rows, cols)), shape=(N, m))
    return Problem(At, K, f"random_sdp(m={m},seed={seed})")


def lp_dense_cols(m=200, n=2000, dens=0.01, ndense=4, seed=1):
    """Sparse LP (m constraints, n variables) with `ndense` fully dense variable rows."""
    rng = np.random.default_rng(seed)
    K = make_K(n + 1, [], [])
    A = sp.random(n, m, density=dens, random_state=rng, format="lil", data_rvs=rng.standard_normal)
    for i in range(ndense):
        A[i, :] = rng.standard_normal(m)
    for j in range(m):
        A[ndense + (j % (n - ndense)), j] = 1.0 + rng.random()
    At = sp.vstack([sp.csc_matrix((1, m)), sp.csc_matrix(A)], format="csc")
    return Problem(At, K, f"lp_dense_cols(m={m},n={n})")


# ------------------------------------------------------------------ host helpers
def dense_symbolic(m):
    """symbchol.m:75-77: the dense shortcut (perm = 1:m, one supernode, L.L = tril(ones))."""
    return {"perm": np.arange(1, m + 1, dtype=np.float64).reshape(-1, 1),
            "L": sp.csc_matrix(np.tril(np.ones((m, m)))), "xsuper": np.array([[1.0], [m + 1.0]]),
            "tmpsiz": np.array([[0.0]])}


def dense_pattern(m):
    return sp.csc_matrix(np.ones((m, m)))


def lorentz_pattern(P):
    """Pattern of DAt.q (len(K.q) x m): trace entry or any norm-bound entry of the block present
    (sedumi.m:370-375: findblks + spones(extractA))."""
    K, At = P.K, sp.csc_matrix(P.At)
    nq = K["q"].size
    m = P.m
    if nq == 0:
        return sp.csc_matrix((0, m))
    lpN = int(K["l"])
    qb = K["qblkstart"].ravel().astype(np.int64) - 1
    rows, cols = [], []
    for j in range(m):
        r = At.indices[At.indptr[j]:At.indptr[j + 1]]
        tr = r[(r >= lpN) & (r < lpN + nq)] - lpN
        nb = r[(r >= qb[0]) & (r < qb[-1])]
        blk = np.searchsorted(qb, nb, side="right") - 1
        for k in np.unique(np.concatenate((tr, blk))):
            rows.append(int(k)); cols.append(j)
    Q = sp.csc_matrix((np.ones(len(rows)), (rows, cols)), shape=(nq, m))
    Q.sort_indices()
    return Q


def spd_scaling(K, seed=0, cond=1e2, identity=False):
    """Scaling inputs of one IPM iteration: d.l, d.det (positive), udsqr = [vec(D_k)] with D_k SPD."""
    rng = np.random.default_rng(seed)
    lpN, nq = int(K["l"]), K["q"].size
    e = np.log10(cond) / 2
    if identity:
        dl, ddet = np.ones(lpN), np.ones(nq)
    else:
        dl, ddet = 10.0 ** rng.uniform(-e, e, lpN), 10.0 ** rng.uniform(-e / 2, e / 2, nq)
    ud = []
    nreal = int(np.asarray(K.get("rsdpN", K["s"].size)).ravel()[0])
    for k, n in enumerate(K["s"].ravel().astype(int)):
        herm = k >= nreal
        if identity:
            D = np.eye(n, dtype=complex if herm else float)
        else:
            R = rng.standard_normal((n, n)) / np.sqrt(n)
            if herm:
                R = R + 1j * rng.standard_normal((n, n)) / np.sqrt(n)
            D = np.eye(n) + 0.1 * (R + R.conj().T)
            w = np.linalg.eigvalsh(D).min()
            if w < 0.2:
                D += (0.2 - w) * np.eye(n)
        if herm:                                           # A.8: Hermitian blocks as [vec(Re); vec(Im)]
            ud.append(np.concatenate((D.real.ravel(order="F"), D.imag.ravel(order="F"))))
        else:
            ud.append(D.ravel(order="F"))
    return {"l": dl, "det": ddet}, (np.concatenate(ud) if ud else np.zeros(0))


# ------------------------------------------------------------------ pattern of ADA' and independent sub-problems
def _cone_layout(K):
    """Row ranges of the internal variable order (A.1): lpN, Lorentz block sizes, first Lorentz trace row, first
    norm-bound row of every Lorentz block, (first row, length) of every PSD block -- all 0-based."""
    lpN = int(np.asarray(K["l"]).ravel()[0])
    q = np.asarray(K["q"], dtype=np.float64).ravel().astype(np.int64)
    s = np.asarray(K["s"], dtype=np.float64).ravel().astype(np.int64)
    nreal = int(np.asarray(K.get("rsdpN", s.size)).ravel()[0])
    bs = np.asarray(K["blkstart"], dtype=np.float64).ravel().astype(np.int64) - 1
    qnorm = bs[1:1 + q.size + 1]                             # starts of the norm-bound parts (+ end)
    psd = bs[1 + q.size:]                                     # starts of the PSD blocks (+ end)
    return lpN, q, s, nreal, qnorm, psd


def coupling_incidence(P):
    """m x G 0/1 matrix: constraint j is incident to group g when it has a nonzero in it.  Groups = every LP and
    Lorentz row on its own (entry-level coupling), every Lorentz block, every PSD block (block-level coupling) --
    the three terms of getsymbada.m:41-60."""
    At = sp.csc_matrix(P.At)
    N, m = At.shape
    lpN, q, s, nreal, qnorm, psd = _cone_layout(P.K)
    psd0 = int(psd[0]) if s.size else N
    rows = At.indices
    cols = np.repeat(np.arange(m), np.diff(At.indptr))
    grp = rows.astype(np.int64).copy()
    ispsd = rows >= psd0
    if s.size:
        grp[ispsd] = psd0 + q.size + (np.searchsorted(psd, rows[ispsd], side="right") - 1)
    B = sp.csr_matrix((np.ones(grp.size), (cols, grp)), shape=(m, psd0 + q.size + s.size))
    if q.size:
        Q = sp.csr_matrix(lorentz_pattern(P).T)              # m x nq
        Qc = sp.csr_matrix((np.ones(Q.nnz), (Q.nonzero()[0], psd0 + Q.nonzero()[1])), shape=B.shape)
        B = B + Qc
    B.data[:] = 1.0
    return sp.csr_matrix(B)


def symb_ada(P):
    """Nonzero pattern of ADA' as SeDuMi builds it (getsymbada.m:41-60), values all 1; dense when over 90 % full."""
    B = coupling_incidence(P)
    m = B.shape[0]
    S = sp.csc_matrix(B @ B.T)
    if m == 0 or S.nnz > 0.9 * m * m:
        return sp.csc_matrix(np.ones((m, m)))
    S = sp.csc_matrix((np.ones(S.nnz), S.indices, S.indptr), shape=S.shape)
    S.sort_indices()
    return S


def subproblem(P, cols):
    """The sub-problem made of the constraints `cols` and exactly the cone blocks they touch (row 0, the artificial
    x0 variable, is always kept).  Returns (Problem, kept rows of P.At in order)."""
    At = sp.csc_matrix(P.At)[:, np.asarray(cols, dtype=np.int64)]
    lpN, q, s, nreal, qnorm, psd = _cone_layout(P.K)
    touched = np.zeros(At.shape[0], dtype=bool)
    touched[np.unique(At.indices)] = True
    lp_rows = [0] + [r for r in range(1, lpN) if touched[r]]
    qkeep = [k for k in range(q.size) if touched[lpN + k] or touched[qnorm[k]:qnorm[k + 1]].any()]
    skeep = [k for k in range(s.size) if touched[psd[k]:psd[k + 1]].any()]
    rows = list(lp_rows) + [lpN + k for k in qkeep]
    for k in qkeep:
        rows.extend(range(int(qnorm[k]), int(qnorm[k + 1])))
    for k in skeep:
        rows.extend(range(int(psd[k]), int(psd[k + 1])))
    rows = np.asarray(rows, dtype=np.int64)
    K = make_K(len(lp_rows), q[qkeep], [s[k] for k in skeep if k < nreal], [s[k] for k in skeep if k >= nreal])
    sub = Problem(sp.csc_matrix(At.tocsr()[rows, :]), K, f"{P.name}[{len(cols)} constraints]")
    sub.kept = {"lp": np.asarray(lp_rows), "q": np.asarray(qkeep, dtype=np.int64), "s": np.asarray(skeep, dtype=np.int64)}
    return sub, rows


def sub_scaling(P, sub, rows, d, ud):
    """Scaling data (d.l, d.det, udsqr) of the full problem restricted to the cone blocks kept by `subproblem`."""
    lpN, q, s, nreal, qnorm, psd = _cone_layout(P.K)
    dl = np.asarray(d["l"], dtype=np.float64).ravel()[sub.kept["lp"]]
    ddet = np.asarray(d["det"], dtype=np.float64).ravel()[sub.kept["q"]] if q.size else np.zeros(0)
    lens = np.where(np.arange(s.size) < nreal, s ** 2, 2 * s ** 2)
    off = np.concatenate(([0], np.cumsum(lens)))
    ud = np.asarray(ud, dtype=np.float64).ravel()
    uds = np.concatenate([ud[off[k]:off[k + 1]] for k in sub.kept["s"]]) if sub.kept["s"].size else np.zeros(0)
    return dl, ddet, uds


def sub_scaling_q(P, sub, d):
    """d.q1 (one per Lorentz cone) and d.q2 (norm-bound parts, concatenated) of the full problem restricted to the
    Lorentz cones kept by `subproblem` -- the inputs of getDAtm.m:39-44 for the sub-problem's DAt.q."""
    lpN, q, s, nreal, qnorm, psd = _cone_layout(P.K)
    keep = sub.kept["q"]
    q1 = np.asarray(d["q1"], dtype=np.float64).ravel()[keep]
    q2f = np.asarray(d["q2"], dtype=np.float64).ravel()
    base = int(qnorm[0]) if q.size else 0
    q2 = np.concatenate([q2f[int(qnorm[k]) - base:int(qnorm[k + 1]) - base] for k in keep]) if keep.size else np.zeros(0)
    return q1, q2


def block_subproblem(P, blocks, keep_lq):
    """All m constraints, but only the PSD blocks listed in `blocks` (indices into K.s, ascending) and -- when keep_lq --
    the LP / Lorentz rows; otherwise a single artificial x0 row.  The partial ADA' of such sub-problems on the common
    pattern add up to ADA' (spscale.c:473-491 loops the blocks independently).  Returns (Problem, kept rows of P.At)."""
    At = sp.csc_matrix(P.At)
    lpN, q, s, nreal, qnorm, psd = _cone_layout(P.K)
    blocks = [int(b) for b in blocks]
    psd0 = int(psd[0]) if s.size else At.shape[0]
    rows = list(range(psd0)) if keep_lq else [0]
    for k in blocks:
        rows.extend(range(int(psd[k]), int(psd[k + 1])))
    rows = np.asarray(rows, dtype=np.int64)
    K = make_K(lpN if keep_lq else 1, q if keep_lq else [], [s[k] for k in blocks if k < nreal], [s[k] for k in blocks if k >= nreal])
    sub = Problem(sp.csc_matrix(At.tocsr()[rows, :]), K, f"{P.name}[blocks {blocks}{'+lq' if keep_lq else ''}]")
    sub.blocks = np.asarray(blocks, dtype=np.int64)
    return sub, rows


def block_udsqr(P, blocks, ud):
    """The slices of udsqr (concatenated D_k, Hermitian blocks [Re; Im]) that belong to the PSD blocks `blocks`."""
    lpN, q, s, nreal, qnorm, psd = _cone_layout(P.K)
    lens = np.where(np.arange(s.size) < nreal, s ** 2, 2 * s ** 2)
    off = np.concatenate(([0], np.cumsum(lens)))
    ud = np.asarray(ud, dtype=np.float64).ravel()
    return np.concatenate([ud[off[k]:off[k + 1]] for k in blocks]) if len(blocks) else np.zeros(0)
