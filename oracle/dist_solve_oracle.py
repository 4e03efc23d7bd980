"""TEST INFRASTRUCTURE (CPU checker, numpy): the algebra of the distributed linear solve of the HIP library
(openimucameracalibrator_amd/csrc/kernels_bcr.hip, "Distributed block cyclic reduction") restated with dense
linear algebra -- what every rank computes from the rows it owns, what travels, how the top system is put together, how the step
comes back.  The reference has no counterpart (its solve is one SPARSE_NORMAL_CHOLESKY inside ceres::Solve,
spline_trajectory_estimator.impl.h:257-272): this file pins the DECOMPOSITION the product uses, independent of its kernels:

  the damped system  [[B, E], [E^T, C]] x = rhs  with B block tridiagonal in blocks of `bs` columns (the band), E the arrow rows,
  C the corner; rank k owns the blocks [b_k, b_k+1) and the ROWS of those blocks only (its diagonal blocks, its couplings to the
  right, its arrow rows, its right-hand side); rank 0 also holds the corner.

  forward(rank):   eliminate the rank's interior blocks (all but its first, the separator) -> a message
                   [D_sep, F_sep, coupling (sep, ghost), D_ghost, F_ghost, corner part, rhs parts] -- the Schur complement of the
                   interior onto (separator, next rank's separator = ghost, arrow), split by who owns what
  top system:      D_top[k] = D_sep(k) + D_ghost(k - 1), couplings, arrow rows, corner = sum of the parts: solved by every rank
  backward(rank):  the interior from the separator's, the ghost's and the arrow's solution

Only tests/ imports this module; the product never does."""
import numpy as np


def random_system(nblk, bs, a, seed=0, last=None):
    """A random symmetric positive definite band + arrow system: `nblk` blocks of `bs` columns (the last one `last` columns),
    block tridiagonal band, `a` arrow columns.  Returns (M dense, rhs, block boundaries)."""
    rng = np.random.default_rng(seed)
    last = bs if last is None else last
    edges = np.concatenate([np.arange(nblk) * bs, [(nblk - 1) * bs + last]])
    pb = int(edges[-1])
    n = pb + a
    J = np.zeros((4 * n, n))
    # rows that touch two neighbouring blocks and the arrow: the products J^T J have exactly the band + arrow pattern
    for r in range(4 * n):
        k = rng.integers(0, nblk)
        cols = np.arange(edges[k], edges[min(k + 2, nblk)])
        pick = rng.choice(cols, size=min(len(cols), 6), replace=False)
        J[r, pick] = rng.normal(size=len(pick))
        J[r, pb:] = rng.normal(size=a) * 0.3
    M = J.T @ J + 1e-3 * np.eye(n)
    rhs = rng.normal(size=n)
    return M, rhs, edges


def owned_rows(M, rhs, edges, a, b0, b1):
    """What rank with the blocks [b0, b1) holds: the rows of its blocks (band part up to the next block, arrow entries, rhs)."""
    pb = int(edges[-1])
    lo, hi = int(edges[b0]), int(edges[b1])
    nxt = int(edges[min(b1 + 1, len(edges) - 1)])
    return dict(lo=lo, hi=hi, nxt=nxt, band=M[lo:hi, lo:nxt].copy(), arrow=M[lo:hi, pb:].copy(), rhs=rhs[lo:hi].copy(),
                sep=int(edges[b0 + 1]) - lo, ghost=nxt - hi)


def forward(own, a, corner=None, rhs_arrow=None):
    """The rank's message: its interior eliminated.  `corner` / `rhs_arrow`: the arrow corner and the arrow part of the right-hand
    side, held by rank 0 only (the others pass None and contribute their Schur updates alone)."""
    ns, ng = own["sep"], own["ghost"]
    n_own = own["hi"] - own["lo"]
    A = own["band"]                                   # rows: own blocks; columns: own blocks + ghost block
    sep = slice(0, ns); inter = slice(ns, n_own); gh = slice(n_own, n_own + ng)
    Aii = A[inter, inter]
    # boundary = [separator | ghost | arrow]; couplings of the interior to it (the band is symmetric: A[sep, inter] = A[inter, sep]^T)
    Bi = np.concatenate([A[sep, inter].T, A[inter, gh], own["arrow"][inter]], axis=1)      # interior x boundary
    ri = own["rhs"][inter]
    if n_own > ns:
        X = np.linalg.solve(Aii, np.concatenate([Bi, ri[:, None]], axis=1))
        S = Bi.T @ X[:, :-1]; sr = Bi.T @ X[:, -1]
    else:
        S = np.zeros((ns + ng + a, ns + ng + a)); sr = np.zeros(ns + ng + a)
    o_s, o_g, o_a = slice(0, ns), slice(ns, ns + ng), slice(ns + ng, ns + ng + a)
    msg = dict(
        D_sep=A[sep, sep] - S[o_s, o_s], F_sep=own["arrow"][sep] - S[o_s, o_a], r_sep=own["rhs"][sep] - sr[o_s],
        S=A[sep, gh] - S[o_s, o_g],                      # coupling (separator, ghost)
        D_ghost=-S[o_g, o_g], F_ghost=-S[o_g, o_a], r_ghost=-sr[o_g],
        C=(corner if corner is not None else 0.0) - S[o_a, o_a], r_arrow=(rhs_arrow if rhs_arrow is not None else 0.0) - sr[o_a],
    )
    keep = dict(Aii=Aii, Bi=Bi, ri=ri, ns=ns, ng=ng)
    return msg, keep


def top_solve(msgs, a):
    """The separators' system from all ranks' messages, solved: (x of every separator, x of the arrow)."""
    N = len(msgs)
    sizes = [m["D_sep"].shape[0] for m in msgs]
    off = np.concatenate([[0], np.cumsum(sizes)])
    nt = int(off[-1])
    T = np.zeros((nt + a, nt + a)); r = np.zeros(nt + a)
    for k, m in enumerate(msgs):
        s = slice(off[k], off[k + 1])
        T[s, s] += m["D_sep"]; T[s, nt:] += m["F_sep"]; T[nt:, s] += m["F_sep"].T; r[s] += m["r_sep"]
        T[nt:, nt:] += m["C"]; r[nt:] += m["r_arrow"]
        if k + 1 < N:
            g = slice(off[k + 1], off[k + 2])
            T[s, g] += m["S"]; T[g, s] += m["S"].T
            T[g, g] += m["D_ghost"]; T[g, nt:] += m["F_ghost"]; T[nt:, g] += m["F_ghost"].T; r[g] += m["r_ghost"]
    x = np.linalg.solve(T, r)
    return [x[off[k]:off[k + 1]] for k in range(N)], x[nt:]


def backward(keep, x_sep, x_ghost, x_arrow):
    """The rank's own solution: separator + interior."""
    if keep["Aii"].shape[0] == 0:
        return x_sep.copy()
    xb = np.concatenate([x_sep, x_ghost if keep["ng"] else np.zeros(0), x_arrow])
    xi = np.linalg.solve(keep["Aii"], keep["ri"] - keep["Bi"] @ xb)
    return np.concatenate([x_sep, xi])


def solve_distributed(M, rhs, edges, a, b0s):
    """All ranks in one process: returns the solution of M x = rhs through the decomposition (b0s: first block of every rank + the block count)."""
    pb = int(edges[-1])
    N = len(b0s) - 1
    msgs, keeps = [], []
    for k in range(N):
        own = owned_rows(M, rhs, edges, a, b0s[k], b0s[k + 1])
        m, kp = forward(own, a, corner=M[pb:, pb:] if k == 0 else None, rhs_arrow=rhs[pb:] if k == 0 else None)
        msgs.append(m); keeps.append(kp)
    xs, xa = top_solve(msgs, a)
    parts = [backward(keeps[k], xs[k], xs[k + 1] if k + 1 < N else np.zeros(0), xa) for k in range(N)]
    return np.concatenate(parts + [xa])
