"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's PARSDMM hot path.

Reference: slimgroup/SetIntersectionProjection.jl v0.2.5 (pure Julia).  Every function
below cites the reference ``file:line`` it follows (paths relative to the reference
root).  This is OUR code: it restates the algorithm, it is not a copy of any source.

How this oracle is pinned (the reference ships NO golden vectors and Julia is not
installed here or on the GPU box, so the reference itself cannot be run):
  * the reference's own known-answer / property tests are re-run against this file in
    ``tests/test_oracle_pins.py`` (prox_l2s 7/4, cardinality closed forms, D_x/D_z ==
    diff/h and TV block order, CDS SpMV == CSC SpMV, CDS_scaled_add exact, cg with exact
    initial guess, update_y_l == 4-line formula, feasible input returned untouched,
    single identity-operator set == direct projector, converged result feasible);
  * what stays UNPINNED: the summation order inside OpenBLAS dot/nrm2/asum and whether its
    axpy fuses the multiply-add (hardware dependent).  Conventions chosen here, and
    mirrored by the HIP engine: (a) element-wise updates follow the reference's
    ``Blas_active=false`` formulas (plain IEEE mul/add, no FMA), (b) every reduction
    (dot, norm, asum) is accumulated in float64 and rounded once to TF.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------------------
# boundary types  (src/SetIntersectionProjection.jl:95-149)
# --------------------------------------------------------------------------------------


@dataclass
class compgrid:
    """User-side grid struct: only .d (spacings) and .n (sizes) are read
    (test/runtests.jl:18-21, src/get_TD_operator.jl:21-28)."""
    d: Tuple
    n: Tuple


@dataclass
class PARSDMM_options:
    """src/SetIntersectionProjection.jl:110-128 (same defaults)."""
    x_min_solver: str = "CG_normal"
    maxit: int = 200
    evol_rel_tol: float = 1e-3
    feas_tol: float = 5e-2
    obj_tol: float = 1e-3
    rho_ini: Sequence[float] = (10.0,)
    rho_update_frequency: int = 2
    gamma_ini: float = 1.0
    adjust_rho: bool = True
    adjust_gamma: bool = True
    adjust_feasibility_rho: bool = True
    Blas_active: bool = True
    feasibility_only: bool = False
    FL: Any = np.float32
    parallel: bool = False
    zero_ini_guess: bool = True
    Minkowski: bool = False


@dataclass
class set_definitions:
    """src/SetIntersectionProjection.jl:142-149."""
    set_type: str
    TD_OP: str
    min: Any
    max: Any
    app_mode: Tuple[str, str]
    custom_TD_OP: Tuple[Any, bool] = ((), False)


@dataclass
class set_properties:
    """src/SetIntersectionProjection.jl:132-140."""
    ncvx: List[bool] = field(default_factory=list)
    AtA_diag: List[bool] = field(default_factory=list)
    dense: List[bool] = field(default_factory=list)
    TD_n: List[Tuple] = field(default_factory=list)
    tag: List[Tuple[str, str, str, str]] = field(default_factory=list)
    banded: List[bool] = field(default_factory=list)
    AtA_offsets: List[np.ndarray] = field(default_factory=list)


@dataclass
class log_type_PARSDMM:
    """src/SetIntersectionProjection.jl:95-108.  Arrays are float64-backed but hold
    TF-rounded values, like the reference's Array{Real}."""
    set_feasibility: np.ndarray
    r_dual: np.ndarray
    r_pri: np.ndarray
    r_dual_total: np.ndarray
    r_pri_total: np.ndarray
    obj: np.ndarray
    evol_x: np.ndarray
    rho: np.ndarray
    gamma: np.ndarray
    cg_it: np.ndarray
    cg_relres: np.ndarray
    timing: Any = None


def convert_options(options: PARSDMM_options, TF) -> None:
    """src/convert_options!.jl:6-15."""
    options.evol_rel_tol = TF(options.evol_rel_tol)
    options.feas_tol = TF(options.feas_tol)
    options.obj_tol = TF(options.obj_tol)
    options.rho_ini = [TF(r) for r in options.rho_ini]
    options.gamma_ini = TF(options.gamma_ini)


# --------------------------------------------------------------------------------------
# canonical reductions (float64 accumulate, one rounding to TF) -- see module docstring
# --------------------------------------------------------------------------------------


def _sumsq64(x) -> float:
    x64 = np.asarray(x, dtype=np.float64)
    return float(np.dot(x64, x64))


def nrm2(x, TF):
    return TF(math.sqrt(_sumsq64(x)))


def dot(x, y, TF):
    return TF(float(np.dot(np.asarray(x, np.float64), np.asarray(y, np.float64))))


def asum(x, TF):
    return TF(float(np.sum(np.abs(np.asarray(x, np.float64)))))


def _nanmax(v) -> float:
    """Julia's maximum(): NaN-propagating."""
    v = np.asarray(v, dtype=np.float64)
    if np.isnan(v).any():
        return float("nan")
    return float(v.max())


# --------------------------------------------------------------------------------------
# ordered sparse kernels (Julia SparseArrays semantics: fixed, index-ascending sum order)
# --------------------------------------------------------------------------------------


def _ordered_seg_sum(ptr: np.ndarray, vals: np.ndarray, TF) -> np.ndarray:
    """out[c] = ((0 + vals[ptr[c]]) + vals[ptr[c]+1]) + ... accumulated in TF."""
    n = len(ptr) - 1
    out = np.zeros(n, dtype=TF)
    if n == 0:
        return out
    cnt = np.diff(ptr)
    for j in range(int(cnt.max()) if len(cnt) else 0):
        sel = np.nonzero(cnt > j)[0]
        out[sel] = out[sel] + vals[ptr[sel] + j]
    return out


def csc_mul(A: sp.csc_matrix, x: np.ndarray) -> np.ndarray:
    """s = A*x as Julia's mul!(s, A, x) does for SparseMatrixCSC (src/update_y_l.jl:43):
    every row accumulates its products in ascending column order."""
    TF = x.dtype.type
    R = sp.csr_matrix(A)
    R.sort_indices()
    prods = (R.data.astype(TF) * x[R.indices]).astype(TF)
    return _ordered_seg_sum(R.indptr, prods, TF)


def csc_mul_adj(A: sp.csc_matrix, v: np.ndarray) -> np.ndarray:
    """A'*v as mul!(tmp, A', v) (src/rhs_compose.jl:28, src/update_y_l.jl:84): every
    column accumulates its products in ascending row order."""
    TF = v.dtype.type
    C = sp.csc_matrix(A)
    C.sort_indices()
    prods = (C.data.astype(TF) * v[C.indices]).astype(TF)
    return _ordered_seg_sum(C.indptr, prods, TF)


def ata_ordered(A: sp.csc_matrix, TF) -> sp.csc_matrix:
    """AtA = A'*A (src/PARSDMM_precompute_distribute.jl:47) with Julia's Gustavson
    accumulation order: entry (i,j) sums A[k,i]*A[k,j] over ascending k."""
    R = sp.csr_matrix(A)
    R.sort_indices()
    n = R.shape[1]
    cnt = np.diff(R.indptr)
    rows = np.repeat(np.arange(R.shape[0]), cnt)
    # all ordered pairs (a, b) of entries that share a row k
    ii, jj, kk, vv = [], [], [], []
    maxc = int(cnt.max()) if len(cnt) else 0
    pos = np.arange(len(R.indices)) - R.indptr[rows]
    for a in range(maxc):
        for b in range(maxc):
            sel = np.nonzero((cnt > a) & (cnt > b))[0]
            ea = R.indptr[sel] + a
            eb = R.indptr[sel] + b
            ii.append(R.indices[ea])
            jj.append(R.indices[eb])
            kk.append(sel)
            vv.append((R.data[ea].astype(TF) * R.data[eb].astype(TF)).astype(TF))
    del pos
    ii = np.concatenate(ii); jj = np.concatenate(jj)
    kk = np.concatenate(kk); vv = np.concatenate(vv)
    order = np.lexsort((kk, ii, jj))           # by column j, then row i, then k ascending
    ii, jj, vv = ii[order], jj[order], vv[order]
    key = jj.astype(np.int64) * n + ii
    starts = np.nonzero(np.diff(np.concatenate(([-1], key))))[0]
    ptr = np.concatenate((starts, [len(key)]))
    vals = _ordered_seg_sum(ptr, vv, TF)
    return sp.csc_matrix((vals, (ii[starts], jj[starts])), shape=(n, n))


# --------------------------------------------------------------------------------------
# operators  (src/get_discrete_Grad.jl, src/get_TD_operator.jl)
# --------------------------------------------------------------------------------------


def _fwd_diff(n: int, h, TF) -> sp.csc_matrix:
    """(n-1) x n forward difference, entries fl(-1/h), fl(+1/h)
    (src/get_discrete_Grad.jl:22-23,58-60)."""
    neg = (np.ones(n - 1, TF) * TF(-1)) / TF(h)
    pos = (np.ones(n - 1, TF) * TF(1)) / TF(h)
    return sp.diags([neg, pos], [0, 1], shape=(n - 1, n), dtype=TF, format="csc")


def get_discrete_Grad(n: Sequence[int], h: Sequence, TD_type: str, TF) -> sp.csc_matrix:
    """src/get_discrete_Grad.jl:16-37 (2-D) and :51-76 (3-D).  Grid is column-major,
    dim 1 ("x") fastest; 2-D: dim 2 is "z"; 3-D: dims are x, y, z."""
    I = lambda k: sp.identity(k, dtype=TF, format="csc")
    if len(n) == 2:
        n1, n2 = n
        Dx, Dz = _fwd_diff(n1, h[0], TF), _fwd_diff(n2, h[1], TF)
        D2z = sp.kron(Dz, I(n1), format="csc")
        D2x = sp.kron(I(n2), Dx, format="csc")
        if TD_type == "D_z":
            return D2z
        if TD_type == "D_x":
            return D2x
        if TD_type in ("TV", "D2D"):
            return sp.vstack([D2z, D2x], format="csc")       # z block first (:31-33)
    else:
        n1, n2, n3 = n
        Dx, Dy, Dz = (_fwd_diff(n1, h[0], TF), _fwd_diff(n2, h[1], TF),
                      _fwd_diff(n3, h[2], TF))
        D3z = sp.kron(Dz, sp.kron(I(n2), I(n1), format="csc"), format="csc")
        D3y = sp.kron(I(n3), sp.kron(Dy, I(n1), format="csc"), format="csc")
        D3x = sp.kron(I(n3), sp.kron(I(n2), Dx, format="csc"), format="csc")
        if TD_type == "D_z":
            return D3z
        if TD_type == "D_y":
            return D3y
        if TD_type == "D_x":
            return D3x
        if TD_type in ("TV", "D3D"):
            return sp.vstack([D3z, D3y, D3x], format="csc")  # z, y, x (:69-72)
    raise ValueError("unknown derivative operator " + TD_type)


def get_TD_operator(comp_grid, TD_type: str, TF):
    """src/get_TD_operator.jl:12-95 -- banded operators only (identity, D_x, D_y, D_z,
    TV); the JOLI transforms are outside the hot-path scope."""
    n = tuple(int(v) for v in comp_grid.n)
    if len(n) == 3 and n[2] == 1:
        n = n[:2]
    h = tuple(TF(v) for v in comp_grid.d[:len(n)])
    N = int(np.prod(n))
    if TD_type == "identity":
        return sp.identity(N, dtype=TF, format="csc"), True, False, n, True
    if TD_type in ("DFT", "DCT"):   # the transform is folded into the projector, TD_OP becomes I (setup_constraints.jl:76-80)
        return sp.identity(N, dtype=TF, format="csc"), True, True, n, True
    if TD_type == "D_xz" and len(n) == 2:      # src/get_TD_operator.jl:66-70: D_z on the (n1-1, n2) grid of D_x's output
        Dx = get_discrete_Grad(n, h, "D_x", TF)
        Dz = get_discrete_Grad((n[0] - 1, n[1]), h, "D_z", TF)
        A = sp.csc_matrix(Dz @ Dx, dtype=TF)
        A.sort_indices()
        return A, False, False, (n[0] - 1, n[1] - 1), True
    A = get_discrete_Grad(n, h, TD_type, TF)
    if len(n) == 2:
        n1, n2 = n
        TD_n = {"TV": ((n1 - 1) + n1, n2 + (n2 - 1)), "D2D": ((n1 - 1) + n1, n2 + (n2 - 1)),
                "D_z": (n1, n2 - 1), "D_x": (n1 - 1, n2)}[TD_type]
    else:
        n1, n2, n3 = n
        TD_n = {"TV": (3 * n1 - 1, 3 * n2 - 1, 3 * n3 - 1), "D3D": (3 * n1 - 1, 3 * n2 - 1, 3 * n3 - 1),
                "D_z": (n1, n2, n3 - 1), "D_y": (n1, n2 - 1, n3), "D_x": (n1 - 1, n2, n3)}[TD_type]
    return A, False, False, TD_n, True


# --------------------------------------------------------------------------------------
# CDS (compressed diagonal storage)  (src/mat2CDS.jl, src/CDS_MVp.jl, src/CDS_scaled_add!.jl)
# --------------------------------------------------------------------------------------


def mat2CDS(A: sp.spmatrix, TF=None):
    """src/mat2CDS.jl:7-32.  R[r,b] = A[r, r+off_b]; offsets ascending; for off>0 the
    band's tail is zero-padded, for off<0 its head (:22-29).  R is column-major (N x d)."""
    C = sp.coo_matrix(A)
    TF = TF or A.dtype.type
    N = A.shape[0]
    offs = np.unique(C.col.astype(np.int64) - C.row.astype(np.int64))
    R = np.zeros((N, len(offs)), dtype=TF, order="F")
    Acsr = sp.csr_matrix(A)
    for b, off in enumerate(offs):
        dA = Acsr.diagonal(int(off)).astype(TF)
        if off >= 0:
            R[:len(dA), b] = dA
        else:
            R[N - len(dA):, b] = dA
    return R, offs.astype(np.int64)


def CDS_MVp(R: np.ndarray, offset: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """y += A*x, band after band in the order given (src/CDS_MVp.jl:9-28; the threaded
    twin CDS_MVp_MT.jl:17-23 keeps the same per-row order)."""
    N = R.shape[0]
    for i, d in enumerate(int(o) for o in offset):
        r0, r1 = max(0, -d), min(N, N - d)
        if r1 > r0:
            y[r0:r1] = y[r0:r1] + R[r0:r1, i] * x[r0 + d:r1 + d]
    return y


def Ax_CDS(x, Q, Q_offsets):
    """Ax_CDS_MT: fill!(Ax_out,0) then CDS_MVp_MT (src/argmin_x.jl:72-78)."""
    return CDS_MVp(Q, Q_offsets, x, np.zeros(len(x), dtype=x.dtype))


def CDS_scaled_add(A, B, A_offsets, B_offsets, alpha) -> None:
    """A[:,col(B_off_k)] += alpha*B[:,k]; error if A lacks the diagonal
    (src/CDS_scaled_add!.jl:8-26)."""
    TF = A.dtype.type
    for k, off in enumerate(B_offsets):
        cols = np.nonzero(np.asarray(A_offsets) == off)[0]
        if len(cols) == 0:
            raise ValueError("attempted to update a diagonal in A in CDS storage that does not exist. "
                             "A and B need to have the same nonzero diagonals")
        for c in cols:
            A[:, c] = A[:, c] + TF(alpha) * B[:, k]


def Q_update(Q, AtA, set_Prop, rho, ind_updated, log, i, Q_offsets) -> np.ndarray:
    """CDS branch of src/Q_update!.jl:45-48 (i is the 0-based log row)."""
    TF = Q.dtype.type
    for ii in ind_updated:
        CDS_scaled_add(Q, AtA[ii], Q_offsets, set_Prop.AtA_offsets[ii], TF(rho[ii]) - TF(log.rho[i, ii]))
    return Q


# --------------------------------------------------------------------------------------
# prox maps and projectors  (src/prox_*.jl, src/projectors/*.jl)
# --------------------------------------------------------------------------------------


def prox_l2s(x, rho, m):
    """x = (x*rho + m) / (rho + 1.0): numerator in TF, division in Float64 because of the
    1.0 literal, rounded to TF on store (src/prox_l2s!.jl:3-6)."""
    TF = x.dtype.type
    num = (x * TF(rho) + m).astype(np.float64)
    x[:] = (num / (np.float64(TF(rho)) + 1.0)).astype(TF)
    return x


def prox_l1(x, rho):
    """src/prox_l1!.jl:8-10 (threshold 1/rho)."""
    TF = x.dtype.type
    x[:] = np.sign(x) * np.maximum(TF(0), np.abs(x) - (TF(1) / TF(rho)))
    return x


def project_bounds(x, LB, UB):
    """src/projectors/project_bounds!.jl:3-25 (scalar or per-element bounds)."""
    TF = x.dtype.type
    if np.ndim(LB) == 0:
        x[:] = np.maximum(TF(LB), np.minimum(x, TF(UB)))
    else:
        x[:] = np.maximum(np.asarray(LB, TF), np.minimum(x, np.asarray(UB, TF)))
    return x


def _accumulate_pairwise(c, v, s, i1, n, TF):
    """Julia Base._accumulate_pairwise! (what cumsum! uses for floats): blocks of <128
    sequentially, halves recursively."""
    if n < 128:
        s_seq = np.cumsum(v[i1:i1 + n], dtype=TF)
        c[i1:i1 + n] = s + s_seq
        return s_seq[-1]
    n2 = n >> 1
    s_ = _accumulate_pairwise(c, v, s, i1, n2, TF)
    s_ = TF(s_ + _accumulate_pairwise(c, v, TF(s + s_), i1 + n2, n - n2, TF))
    return s_


def julia_cumsum(v):
    TF = v.dtype.type
    c = np.empty_like(v)
    if len(v) == 0:
        return c
    c[0] = v[0]
    if len(v) > 1:
        _accumulate_pairwise(c, v, TF(v[0]), 1, len(v) - 1, TF)
    return c


def l1ball_theta_duchi(absv, b):
    """Threshold of src/projectors/project_l1_Duchi!.jl:33-46: descending sort, cumsum in
    TF, serial scan, rho=max(1,rho), theta=max(0,(sv[rho]-b)/rho)."""
    TF = absv.dtype.type
    lv = len(absv)
    u = np.sort(absv)[::-1].astype(TF)
    sv = julia_cumsum(u)
    kk = np.arange(1, lv + 1).astype(TF)
    cond = (u > ((sv - TF(b)) / kk)) & (np.arange(1, lv + 1) < lv)
    stop = np.nonzero(~cond)[0]
    rho = int(stop[0]) if len(stop) else lv
    rho = max(1, rho)
    return max(TF(0), TF((sv[rho - 1] - TF(b)) / TF(rho)))


def project_l1_Duchi(v, b):
    """src/projectors/project_l1_Duchi!.jl:21-52 (real input)."""
    TF = v.dtype.type
    if TF(b) <= TF(0):
        raise ValueError("Radius of L1 ball is negative")
    if asum(v, TF) <= TF(b):
        return v
    theta = l1ball_theta_duchi(np.abs(v), b)
    v[:] = np.sign(v) * np.maximum(np.abs(v) - theta, TF(0))
    return v


def project_l2(x, sigma):
    """src/projectors/project_l2!.jl:3-16."""
    TF = x.dtype.type
    nl2 = nrm2(x, TF)
    if nl2 <= TF(sigma):
        return x
    x *= TF(sigma) / nl2
    return x


def project_annulus(x, sigma_min, sigma_max):
    """src/projectors/project_annulus!.jl:3-21 (zero vector -> constant fill :16-17)."""
    TF = x.dtype.type
    nl2 = nrm2(x, TF)
    if TF(sigma_min) <= nl2 <= TF(sigma_max):
        return x
    if nl2 > TF(sigma_max):
        x *= TF(sigma_max) / nl2
    elif nl2 > 0:
        x *= TF(sigma_min) / nl2
    else:
        x[:] = (np.ones(len(x), TF) * (np.float64(TF(sigma_min)) / math.sqrt(len(x)))).astype(TF)   # sqrt(Int) is Float64
    return x


def project_cardinality(x, k: int):
    """Vector mode of src/projectors/project_cardinality!.jl:3-21: stable sortperm by abs,
    descending; everything after the k-th is zeroed."""
    TF = x.dtype.type
    order = np.argsort(-np.abs(x), kind="stable")
    x[order[int(k):]] = TF(0)
    return x


def _slices(X, mode):
    """Views of the slices of a tensor in the reference's order: (slice, x) -> X[i,:,:], y -> X[:,i,:], z -> X[:,:,i]."""
    if mode[0] != "slice":
        raise ValueError("mode[1] for rank / nuclear norm projections can only be: slice")
    ax = {"x": 0, "y": 1, "z": 2}[mode[1]]
    return [np.take(X, i, axis=ax) for i in range(X.shape[ax])], ax


def _per_slice(x, n, mode, fun):
    TF = x.dtype.type
    X = x.reshape(n, order="F").copy(order="F")
    if X.ndim == 2:
        X[...] = fun(X.astype(np.float64)).astype(TF)
    else:
        sl, ax = _slices(X, mode)
        for i, S in enumerate(sl):
            idx = [slice(None)] * 3
            idx[ax] = i
            X[tuple(idx)] = fun(S.astype(np.float64)).astype(TF)
    x[:] = X.reshape(-1, order="F")
    return x


def project_rank(x, r: int, n, mode=("matrix", "")):
    """src/projectors/project_rank!.jl:3-48: matrix, or every x / y / z slice of a tensor."""
    def trunc(S):
        U, s, Vt = np.linalg.svd(S, full_matrices=False)
        return (U[:, :r] * s[:r]) @ Vt[:r, :]
    return _per_slice(x, n, mode, trunc)


def project_nuclear(x, sigma, n, mode=("matrix", "")):
    """src/projectors/project_nuclear!.jl:3-62: singular values projected onto the l1 ball of radius sigma."""
    TF = x.dtype.type

    def shrink(S):
        U, s, Vt = np.linalg.svd(S, full_matrices=False)
        if s.sum() <= float(sigma):
            return S          # inside the ball: U*S*Vt is S up to rounding (project_nuclear!.jl:19-23); returned untouched
        s = project_l1_Duchi(s.astype(TF), TF(sigma)).astype(np.float64)
        return (U * s) @ Vt
    return _per_slice(x, n, mode, shrink)


def project_bounds_mode(x, LB, UB, n, mode):
    """src/projectors/project_bounds!.jl:38-88: bounds per fiber; x is reshaped to n (= TD_n).
    Note the order min(max(x, LB), UB) (the vector method clips with UB first)."""
    X = x.reshape(n, order="F")
    if mode[0] == "slice":
        raise ValueError("bound constraints per slice of a tensor currently not implemented, yet...")
    ax = {"x": 0, "y": 1, "z": X.ndim - 1}[mode[1]]
    if X.ndim == 2 and mode[1] == "y":
        raise ValueError("2-D models: fiber modes are x and z")
    shp = [1] * X.ndim
    shp[ax] = X.shape[ax]
    L, U = np.asarray(LB).reshape(shp), np.asarray(UB).reshape(shp)
    x[:] = np.minimum(np.maximum(X, L), U).reshape(-1, order="F")
    return x


def _card_cols(M, k):
    """every column of M (segment-major copy): zero all but the k largest magnitudes, ties to the earlier index."""
    L = M.shape[0]
    if k >= L:
        return M
    order = np.argsort(-np.abs(M), axis=0, kind="stable")     # sortperm(by=abs, rev=true) is stable
    drop = order[k:, :]
    np.put_along_axis(M, drop, 0, axis=0)
    return M


def project_cardinality_mode(x, k: int, n, mode):
    """src/projectors/project_cardinality!.jl:23-146: cardinality per fiber (2-D: x, z; 3-D: x, y, z) or per slice."""
    X = x.reshape(n, order="F").copy(order="F")
    if X.ndim == 2:
        if mode[0] != "fiber" or mode[1] not in ("x", "z"):
            raise ValueError("for 2D models, the mode of application for project_cardinality! needs to be (fiber,x) or (fiber,z)")
        if mode[1] == "x":
            X = _card_cols(X, k)
        else:
            X = _card_cols(np.ascontiguousarray(X.T), k).T
    else:
        n1, n2, n3 = X.shape
        ax = {"x": 0, "y": 1, "z": 2}[mode[1]]
        if mode[0] == "fiber":
            Y = np.moveaxis(X, ax, 0).reshape(X.shape[ax], -1).copy()
            Y = _card_cols(Y, k)
            rest = [d for i, d in enumerate(X.shape) if i != ax]
            X = np.moveaxis(Y.reshape([X.shape[ax]] + rest), 0, ax)
        else:   # slice: permutedims [2,3,1] / [1,3,2] / identity, column-major reshape -> remaining dims, lower one fastest
            perm = {0: (1, 2, 0), 1: (0, 2, 1), 2: (0, 1, 2)}[ax]
            Y = np.transpose(X, perm)
            shp = Y.shape
            Y = _card_cols(Y.reshape(shp[0] * shp[1], shp[2], order="F").copy(), k)
            X = np.transpose(Y.reshape(shp, order="F"), np.argsort(perm))
    x[:] = np.asarray(X).reshape(-1, order="F")
    return x


def project_histogram_relaxed(x, LB, UB):
    """src/projectors/project_histogram_relaxed.jl:9-27: j-th smallest entry clipped to [LB[j], UB[j]] (UB first)."""
    idx = np.argsort(x, kind="stable")
    v = x[idx]
    v = np.maximum(LB, np.minimum(v, UB))
    x[idx] = v
    return x


def project_subspace(x, A, orth: bool, n=None, mode=("matrix", "")):
    """src/projectors/project_subspace!.jl:10-125 (float64 arithmetic, rounded once)."""
    TF = x.dtype.type
    A = np.asarray(A, np.float64)

    def P(M):
        t = A.T @ M
        if not orth:
            t = np.linalg.solve(A.T @ A, t)
        return A @ t
    if mode[0] in ("matrix", "tensor"):
        x[:] = P(x.astype(np.float64)).astype(TF)
        return x
    X = x.reshape(n, order="F").astype(np.float64)
    if X.ndim == 2:
        if mode[0] == "slice":
            raise ValueError("mode[1] for project_subspace! must be: fiber")
        if mode[1] == "x":
            X = P(X)
        elif mode[1] == "z":
            X = P(X.T).T
        else:
            raise ValueError("mode[2] for project_subspace! with 2D array input must be: x, or z")
    else:
        if mode[0] != "slice":
            raise ValueError("for 3D models, the mode of application for project_subspace! needs to be (slice,x) or (slice,y) or (slice,z)")
        ax = {"x": 0, "y": 1, "z": 2}[mode[1]]
        perm = {0: (1, 2, 0), 1: (0, 2, 1), 2: (0, 1, 2)}[ax]
        Y = np.transpose(X, perm)
        shp = Y.shape
        Y = P(Y.reshape(shp[0] * shp[1], shp[2], order="F"))
        X = np.transpose(Y.reshape(shp, order="F"), np.argsort(perm))
    x[:] = X.reshape(-1, order="F").astype(TF)
    return x


def project_l1_dft(x, b, n):
    """x -> Re(F' project_l1_Duchi!(F x, b)) with F the unitary DFT (src/get_projector.jl:29-35 with
    A = joDFT, src/projectors/project_l1_Duchi!.jl:29-32,49 for complex input).  joDFT's normalisation
    is not pinned by any reference test; unitary is assumed (the operator declares AtA_diag = true)."""
    TF = x.dtype.type
    Z = np.fft.fftn(x.reshape(n, order="F").astype(np.float64), norm="ortho")
    a = np.abs(Z)
    if not (a.sum() > float(b)):
        return x      # inside the ball: F'F = I, return x untouched instead of an FFT round trip (pure rounding noise)
    theta = float(l1ball_theta_duchi(a.reshape(-1, order="F"), float(b)))       # the same scan, incl. its lv-1 cap (:42)
    Z = np.where(a > 0, Z / np.maximum(a, 1e-300), 0) * np.maximum(a - theta, 0)
    x[:] = np.real(np.fft.ifftn(Z, norm="ortho")).reshape(-1, order="F").astype(TF)
    return x


def project_bounds_dft(x, UB, n):
    """x -> Re(F' (UB .* F x)), F the unitary DFT: project_bounds! on a complex vector with binary bounds
    (src/projectors/project_bounds!.jl:27-36) under src/get_projector.jl:8-9 with TD_OP = "DFT" (joDFT normalisation
    unpinned, unitary assumed as for project_l1_dft)."""
    TF = x.dtype.type
    Z = np.fft.fftn(x.reshape(n, order="F").astype(np.float64), norm="ortho")
    Z = Z * np.asarray(UB, np.float64).reshape(n, order="F")
    x[:] = np.real(np.fft.ifftn(Z, norm="ortho")).reshape(-1, order="F").astype(TF)
    return x


def get_projector(constraint: set_definitions, TF, comp_grid=None, TD_n=None) -> Callable:
    """src/get_projector.jl:3-103 for the banded operators (and the DFT-folded l1 ball)."""
    st = constraint.set_type
    n = tuple(int(v) for v in comp_grid.n) if comp_grid is not None else None
    if n is not None and len(n) == 3 and n[2] == 1:
        n = n[:2]
    tdn = tuple(int(v) for v in TD_n) if TD_n is not None else n
    mode = tuple(constraint.app_mode)
    whole = mode[0] in ("matrix", "tensor")
    if constraint.TD_OP == "DCT":
        # x -> C' P(C x), C the orthonormal DCT-II along every dimension (joDCT normalisation unpinned, orthonormal assumed)
        import scipy.fft as sfft
        if st == "l1":
            inner = lambda c: project_l1_Duchi(c, constraint.max)
        elif st == "cardinality":
            inner = lambda c: project_cardinality(c, int(constraint.max))
        elif st == "bounds":
            inner = lambda c: project_bounds(c, constraint.min, constraint.max)
        elif st == "l2":
            inner = lambda c: project_l2(c, constraint.max)
        elif st == "annulus":
            inner = lambda c: project_annulus(c, constraint.min, constraint.max)
        else:
            raise NotImplementedError(f"{st} behind the DCT")

        def through_dct(x):
            TFx = x.dtype.type
            c = sfft.dctn(x.reshape(n, order="F").astype(np.float64), norm="ortho").reshape(-1, order="F").astype(TFx)
            if st == "l1" and asum(c, TFx) <= TFx(constraint.max):
                return x                                       # inside the ball: C'C = I, x returned untouched
            c = inner(c)
            x[:] = sfft.idctn(c.astype(np.float64).reshape(n, order="F"), norm="ortho").reshape(-1, order="F").astype(TFx)
            return x
        return through_dct
    if constraint.TD_OP == "DFT" and st == "l1":
        return lambda x: project_l1_dft(x, constraint.max, n)
    if constraint.TD_OP == "DFT" and st == "bounds":
        return lambda x: project_bounds_dft(x, constraint.max, n)
    if constraint.TD_OP == "DFT" and st in ("l2", "annulus"):
        def through_dft(x):                                  # literal x -> Re(F' P(F x)) with the unitary DFT
            TFx = x.dtype.type
            Z = np.fft.fftn(x.reshape(n, order="F").astype(np.float64), norm="ortho").reshape(-1, order="F")
            nz = float(np.sqrt((np.abs(Z) ** 2).sum()))
            if st == "l2":
                Z = Z if nz <= float(constraint.max) else Z * (float(constraint.max) / nz)
            else:
                lo, hi = float(constraint.min), float(constraint.max)
                if nz > hi:
                    Z = Z * (hi / nz)
                elif nz < lo and nz > 0:
                    Z = Z * (lo / nz)
                elif nz < lo:
                    Z = np.full_like(Z, lo / np.sqrt(len(Z)))
            x[:] = np.real(np.fft.ifftn(Z.reshape(n, order="F"), norm="ortho")).reshape(-1, order="F").astype(TFx)
            return x
        return through_dft
    if st == "rank":
        return lambda x: project_rank(x, int(constraint.max), tdn, mode)
    if st == "nuclear":
        return lambda x: project_nuclear(x, constraint.max, tdn, mode)
    if st == "subspace":
        A, orth = constraint.custom_TD_OP
        return lambda x: project_subspace(x, A, bool(orth), n, mode)
    if st == "histogram":
        return lambda x: project_histogram_relaxed(x, constraint.min, constraint.max)
    if st == "bounds":
        if whole:
            return lambda x: project_bounds(x, constraint.min, constraint.max)
        return lambda x: project_bounds_mode(x, constraint.min, constraint.max, tdn, mode)
    if st == "cardinality":
        if whole:
            return lambda x: project_cardinality(x, int(constraint.max))
        return lambda x: project_cardinality_mode(x, int(constraint.max), tdn, mode)
    if not whole:
        raise NotImplementedError(f"{st} with app_mode {mode}")
    if st == "prox_l1":
        return lambda x: prox_l1(x, constraint.max)
    if st == "l1":
        return lambda x: project_l1_Duchi(x, constraint.max)
    if st == "l2":
        return lambda x: project_l2(x, constraint.max)
    if st == "annulus":
        return lambda x: project_annulus(x, constraint.min, constraint.max)
    raise NotImplementedError(st)


# --------------------------------------------------------------------------------------
# one-off setup  (src/setup_constraints.jl, src/PARSDMM_precompute_distribute.jl)
# --------------------------------------------------------------------------------------


def setup_constraints(constraint: List[set_definitions], comp_grid, TF):
    """src/setup_constraints.jl:17-102 (banded operators)."""
    P_sub, TD_OP = [], []
    sp_ = set_properties()
    for c in constraint:
        if np.ndim(c.min) == 0:
            if isinstance(c.min, (float, np.floating)):
                c.min, c.max = TF(c.min), TF(c.max)                     # :32-38
        else:
            c.min, c.max = np.asarray(c.min, TF), np.asarray(c.max, TF)   # :39-42
        if c.set_type in ("nuclear", "rank") and c.app_mode[0] in ("matrix", "tensor") and len(comp_grid.n) == 3 \
                and comp_grid.n[2] > 1:
            raise ValueError("requested rank or nuclear norm constraints on a tensor, use mode=(slice,x) e.t.c. to "
                             "define constraints per slice")                                  # :60-62
        if c.set_type in ("l1", "l2") and c.app_mode[0] in ("slice", "fiber"):
            raise ValueError("l1 and l2 constraints only available for matrix or tensor mode, currently")
        A, AtA_diag, dense, TD_n, banded = get_TD_operator(comp_grid, c.TD_OP, TF)
        cust = c.custom_TD_OP[0] if c.set_type != "subspace" else ()
        if not (isinstance(cust, (tuple, list)) and len(cust) == 0):      # :70-72  A = constraint[i].custom_TD_OP[1]
            A = sp.csc_matrix(cust, dtype=TF)
            A.sort_indices()
            AtA_diag, dense = False, False
        P_sub.append(get_projector(c, TF, comp_grid, TD_n))
        TD_OP.append(A)
        sp_.AtA_diag.append(AtA_diag); sp_.dense.append(dense); sp_.TD_n.append(TD_n)
        sp_.banded.append(banded); sp_.AtA_offsets.append(None)
        sp_.tag.append((c.set_type, c.TD_OP, c.app_mode[0], c.app_mode[1]))
        if c.set_type in ("rank", "cardinality"):                        # :89-97
            ncvx = True
        elif c.set_type in ("bounds", "histogram") and c.TD_OP != "identity" and TF(np.max(c.min)) > TF(0):
            ncvx = True
        else:
            ncvx = False
        sp_.ncvx.append(ncvx)
    return P_sub, TD_OP, sp_


def PARSDMM_precompute_distribute(TD_OP, set_Prop, comp_grid, options):
    """src/PARSDMM_precompute_distribute.jl:6-77 (serial, all-banded => CDS)."""
    TF = options.FL
    n = tuple(int(v) for v in comp_grid.n)
    N = int(np.prod(n))
    if not options.feasibility_only:                                     # :17-26
        TD_OP.append(sp.identity(N, dtype=TF, format="csc"))
        set_Prop.TD_n.append(n); set_Prop.AtA_offsets.append(np.array([0], np.int64))
        set_Prop.banded.append(True); set_Prop.AtA_diag.append(True)
        set_Prop.ncvx.append(False); set_Prop.dense.append(False)
        set_Prop.tag.append(("distance squared", "identity", "matrix", ""))
    p = len(TD_OP)
    AtA = []
    for i in range(p):                                                   # :44-48
        if set_Prop.AtA_diag[i]:
            M = sp.identity(N, dtype=TF, format="csc")
        else:
            M = ata_ordered(TD_OP[i], TF)
        R, off = mat2CDS(M, TF)                                          # :52-59
        AtA.append(R)
        set_Prop.AtA_offsets[i] = off
    y = [np.zeros(TD_OP[i].shape[0], TF) for i in range(p)]              # :62-67
    l = [np.zeros(TD_OP[i].shape[0], TF) for i in range(p)]
    return TD_OP, AtA, l, y


# --------------------------------------------------------------------------------------
# iteration-body steps
# --------------------------------------------------------------------------------------


def PARSDMM_precompute_distribute_Minkowski(TD_OP_c1, TD_OP_c2, TD_OP_sum, prop_c1, prop_c2, prop_sum, comp_grid, options):
    """src/PARSDMM_precompute_distribute_Minkowski.jl:6-173 (banded operators) -> (TD_OP, set_Prop, AtA, l, y)."""
    import copy
    TF = np.dtype(options.FL).type
    N = int(np.prod([int(v) for v in comp_grid.n]))
    Z = sp.csc_matrix((N, N), dtype=TF)
    I = sp.identity(N, dtype=TF, format="csc")
    AtA = []
    for ops, prop, where in ((TD_OP_c1, prop_c1, 1), (TD_OP_c2, prop_c2, 2), (TD_OP_sum, prop_sum, 3)):
        for i, A in enumerate(ops):
            B = I if (prop.dense[i] and prop.AtA_diag[i]) else ata_ordered(A, TF)          # :36-47
            blocks = {1: [[B, Z], [Z, Z]], 2: [[Z, Z], [Z, B]], 3: [[B, B], [B, B]]}[where]
            AtA.append(sp.bmat(blocks, format="csc", dtype=TF))
    TD_OP = ([sp.hstack([A, sp.csc_matrix(A.shape, dtype=TF)], format="csc") for A in TD_OP_c1] +          # :91-103
             [sp.hstack([sp.csc_matrix(A.shape, dtype=TF), A], format="csc") for A in TD_OP_c2] +
             [sp.hstack([A, A], format="csc") for A in TD_OP_sum])
    prop = copy.deepcopy(prop_c1)
    for other in (prop_c2, prop_sum):
        for f in ("AtA_diag", "AtA_offsets", "TD_n", "banded", "dense", "ncvx", "tag"):
            getattr(prop, f).extend(copy.deepcopy(getattr(other, f)))
    if not options.feasibility_only:                                                       # :106-116
        TD_OP.append(sp.hstack([I, I], format="csc"))
        prop.TD_n.append(tuple(int(v) for v in comp_grid.n)); prop.AtA_offsets.append(np.array([0], np.int64))
        prop.banded.append(True); prop.AtA_diag.append(False); prop.dense.append(False)
        prop.ncvx.append(False); prop.tag.append(("distance squared", "identity", "matrix", ""))
        AtA.append(sp.bmat([[I, I], [I, I]], format="csc", dtype=TF))
    s_ = len(TD_OP)
    for i in range(s_):                                                                    # :139-146
        AtA[i], prop.AtA_offsets[i] = mat2CDS(AtA[i], TF)
    y = [np.zeros(TD_OP[i].shape[0], TF) for i in range(s_)]
    l = [np.zeros(TD_OP[i].shape[0], TF) for i in range(s_)]
    return TD_OP, prop, AtA, l, y


def rhs_compose(l, y, rho, TD_OP, p, N, only=None):
    """rhs = sum_i A_i'(rho_i y_i + l_i), sets added in order into a zero-filled rhs
    (src/rhs_compose.jl:24-36).  `only`: restrict to a subset of sets (the partial sum one
    worker contributes in the reference's parallel mode, rhs_compose.jl:17-20)."""
    TF = y[0].dtype.type
    rhs = np.zeros(N, TF)
    for ii in (range(p) if only is None else only):
        rhs = rhs + csc_mul_adj(TD_OP[ii], TF(rho[ii]) * y[ii] + l[ii])
    return rhs


def cg(Afun, b, tol, maxIter, x):
    """src/cg.jl:44-128 with M = identity (z aliases r).  Returns (x, flag, relres, iter).
    Reductions: one float64-accumulated ||r||^2 serves norm(r), dot(z,r) and the next
    dot(r,z) (identical vectors in the reference)."""
    TF = b.dtype.type
    n = len(b)
    nr0 = nrm2(b, TF)
    if nr0 == 0:                                                         # :51
        return np.zeros(n, TF), -9, TF(0), 0
    r = b - Afun(x)                                                      # :56
    p = r.copy()
    ss = _sumsq64(r)                      # float64 sum; norm(r) = TF(sqrt(ss)), dot(r,r) = TF(ss)
    rr = TF(ss)
    if TF(TF(math.sqrt(ss)) / nr0) <= tol:                               # :77-80
        return x, 0, TF(0), 1
    flag, lastIter, res_last = -1, 0, TF(0)
    for it in range(1, maxIter + 1):                                     # :86
        lastIter = it
        Ap = Afun(p)
        gamma = rr                                                       # dot(r,z)      :90
        alpha = TF(gamma / dot(p, Ap, TF))                               # :92
        if np.isposinf(alpha) or alpha < 0:                              # :95-97
            flag = -2
            res_last = TF(0)   # resvec[lastIter] was never written
            break
        x += alpha * p                                                   # :99
        r -= alpha * Ap                                                  # :101
        ss = _sumsq64(r)
        rr = TF(ss)
        res_last = TF(TF(math.sqrt(ss)) / nr0)                           # :104
        if res_last <= tol:                                              # :108-110
            flag = 0
            break
        beta = TF(rr / gamma)                                            # :114
        p = r + beta * p                                                 # :118
    return x, flag, res_last, lastIter


def argmin_x(Q, rhs, x, x_solve_tol_ref, i, Q_offsets):
    """CDS branch of src/argmin_x.jl:23-39 (i is the 1-based PARSDMM iteration)."""
    TF = x.dtype.type
    Af = lambda v: Ax_CDS(v, Q, Q_offsets)
    eps = np.finfo(TF).eps
    with np.errstate(all="ignore"):
        relres0 = float(np.float64(0.1) * np.float64(nrm2(Af(x) - rhs, TF)) / np.float64(nrm2(rhs, TF)))
    cand = _julia_max(relres0, float(TF(10) * eps))
    if i < 3:
        tol = TF(cand)
    else:
        tol = TF(_julia_min(cand, float(x_solve_tol_ref)))
    x, flag, relres, it = cg(Af, rhs, tol, 1000, x)
    return x, it, relres, tol


def _julia_max(a: float, b: float) -> float:
    return float("nan") if (math.isnan(a) or math.isnan(b)) else max(a, b)


def _julia_min(a: float, b: float) -> float:
    return float("nan") if (math.isnan(a) or math.isnan(b)) else min(a, b)


def update_y_l(x, p, i, y, y_old, l, l_old, rho, gamma, prox, TD_OP, log, P_sub, counter,
               x_hat, r_pri, s, feasibility_only=False, only=None):
    """src/update_y_l.jl:6-109, ``Blas_active=false`` formulas (:64-78).  i is the 1-based
    iteration; counter the 1-based feasibility row."""
    TF = x.dtype.type
    eps = np.finfo(TF).eps
    for ii in (range(p) if only is None else only):
        rho1 = TF(1) / TF(rho[ii])                                       # :33-34
        g = TF(gamma[ii]); r_ = TF(rho[ii])
        y_old[ii][:] = y[ii]; l_old[ii][:] = l[ii]                       # :39-40
        s[ii][:] = csc_mul(TD_OP[ii], x)                                 # :43
        if g == 1:                                                       # :65-70
            y[ii][:] = s[ii] - l[ii] * rho1
            y[ii] = prox[ii](y[ii])
            r_pri[ii][:] = -s[ii] + y[ii]
            l[ii][:] = l[ii] + r_ * r_pri[ii]
        else:                                                            # :71-77
            x_hat[ii][:] = g * s[ii] + (TF(1) - g) * y[ii]
            y[ii][:] = x_hat[ii] - l[ii] * rho1
            y[ii] = prox[ii](y[ii])
            r_pri[ii][:] = -s[ii] + y[ii]
            l[ii][:] = l[ii] + r_ * (-x_hat[ii] + y[ii])
        log.r_pri[i - 1, ii] = nrm2(r_pri[ii], TF)                       # :81
        x_hat[ii][:] = y[ii] - y_old[ii]                                 # :82
        log.r_dual[i - 1, ii] = r_ * nrm2(csc_mul_adj(TD_OP[ii], x_hat[ii]), TF)   # :84
        if i % 10 == 0 and ((not feasibility_only and ii < p - 1) or feasibility_only):   # :90-99
            x_hat[ii][:] = s[ii]
            P_sub[ii](x_hat[ii])
            log.set_feasibility[counter - 1, ii] = TF(nrm2(x_hat[ii] - s[ii], TF)
                                                      / TF(nrm2(s[ii], TF) + TF(100) * eps))
    if i % 10 == 0:                                                      # :103-105
        counter += 1
    return counter


def bb_scalars(TF, d_dHh_dlh, n_d_H_hat, n_d_l_hat, n_d_l, n_d_G_hat, d_dGh_dl,
               rho, gamma, adjust_rho, adjust_gamma):
    """Scalar Barzilai-Borwein rule of src/adapt_rho_gamma.jl:55-126 in TF arithmetic."""
    safeguard = TF(1e-10) if TF == np.float64 else TF(1e-6)              # :31-35
    eps_corr = TF(0.3)                                                   # :37
    sqrt = lambda v: TF(np.sqrt(TF(v)))
    alpha_reliable = beta_reliable = False
    alpha_corr = beta_corr = TF(0)
    if (n_d_H_hat * n_d_l_hat) > safeguard and (n_d_H_hat * n_d_H_hat) > safeguard and d_dHh_dlh > safeguard:
        alpha_reliable = True
        alpha_corr = d_dHh_dlh / (n_d_H_hat * n_d_l_hat)
    if (n_d_G_hat * n_d_l) > safeguard and (n_d_G_hat * n_d_G_hat) > safeguard and d_dGh_dl > safeguard:
        beta_reliable = True
        beta_corr = d_dGh_dl / (n_d_G_hat * n_d_l)
    alpha_comp = beta_comp = False
    alpha_hat = beta_hat = TF(0)
    if alpha_reliable and alpha_corr > eps_corr:                         # :67-77
        alpha_comp = True
        mg = d_dHh_dlh / (n_d_H_hat * n_d_H_hat)
        sd = (n_d_l_hat * n_d_l_hat) / d_dHh_dlh
        alpha_hat = mg if (TF(2) * mg) > sd else sd - mg / TF(2)
    if beta_reliable and beta_corr > eps_corr:                           # :79-89
        beta_comp = True
        mg = d_dGh_dl / (n_d_G_hat * n_d_G_hat)
        sd = (n_d_l * n_d_l) / d_dGh_dl
        beta_hat = mg if (TF(2) * mg) > sd else sd - mg / TF(2)
    rho, gamma = TF(rho), TF(gamma)
    if adjust_rho:                                                       # :92-115
        if alpha_comp and beta_comp:
            rho = sqrt(alpha_hat * beta_hat)
        elif alpha_comp:
            rho = alpha_hat
        elif beta_comp:
            rho = beta_hat
    if adjust_gamma:                                                     # :102-125
        if alpha_comp and beta_comp:
            gamma = TF(1) + ((TF(2) * sqrt(alpha_hat * beta_hat)) / (alpha_hat + beta_hat))
        elif alpha_comp:
            gamma = TF(1.9)
        elif beta_comp:
            gamma = TF(1.1)
        else:
            gamma = TF(1.5)
    return TF(rho), TF(gamma)


def adapt_rho_gamma(gamma, rho, adjust_gamma, adjust_rho, y, y_old, s, s_0, l, l_hat_0, l_0,
                    l_old, y_0, p, l_hat, only=None):
    """src/adapt_rho_gamma.jl:8-132.  Mutates rho, gamma, l_hat in place."""
    TF = y[0].dtype.type
    for ii in (range(p) if only is None else only):
        r_ = TF(rho[ii])
        l_hat[ii][:] = l_old[ii] + r_ * (-s[ii] + y_old[ii])             # :41
        d_l_hat = l_hat[ii] - l_hat_0[ii]                                # :42
        d_H_hat = s[ii] - s_0[ii]                                        # :43
        d_l = l[ii] - l_0[ii]                                            # :49
        d_G_hat = -(y[ii] - y_0[ii])                                     # :51
        rho[ii], gamma[ii] = bb_scalars(
            TF, dot(d_H_hat, d_l_hat, TF), nrm2(d_H_hat, TF), nrm2(d_l_hat, TF), nrm2(d_l, TF),
            nrm2(d_G_hat, TF), dot(d_G_hat, d_l, TF), rho[ii], gamma[ii], adjust_rho, adjust_gamma)
    return rho, gamma


def stop_PARSDMM(log, i, evol_rel_tol, feas_tol, obj_tol, adjust_rho, adjust_gamma,
                 adjust_feasibility_rho, ind_ref, counter, TF):
    """src/stop_PARSDMM.jl:7-54.  i, counter, ind_ref are 1-based like the reference."""
    stop = False
    f32 = lambda a: np.asarray(a, dtype=np.float64).astype(TF)
    with np.errstate(all="ignore"):
        if i > 6 and _nanmax(log.set_feasibility[counter - 2, :]) < feas_tol:        # :23
            a = f32(log.obj[i - 6:i]); b = f32(log.obj[i - 7:i - 1])
            if _nanmax(np.abs((a - b) / b)) < obj_tol:
                stop = True
        if i > 5 and _nanmax(log.evol_x[i - 6:i]) < evol_rel_tol:                     # :29
            stop = True
        if i > 20 and adjust_rho:                                                     # :35-46
            lo = max(i - 50, 1)
            if log.r_pri_total[i - 1] > _nanmax(log.r_pri_total[lo - 1:i - 1]):
                adjust_rho = adjust_feasibility_rho = adjust_gamma = False
                ind_ref = i
        if (not adjust_rho) and i > (ind_ref + 25):                                   # :49-52
            lo = max(ind_ref, max(i - 50, 1))
            if log.r_pri_total[i - 1] > _nanmax(log.r_pri_total[lo - 1:i - 1]):
                stop = True
    return stop, adjust_rho, adjust_gamma, adjust_feasibility_rho, ind_ref


# --------------------------------------------------------------------------------------
# the solver  (src/PARSDMM_initialize.jl, src/PARSDMM.jl)
# --------------------------------------------------------------------------------------


def assemble_Q(AtA, AtA_offsets, rho, TF):
    """src/PARSDMM_initialize.jl:216-230: Q_offsets = first-seen order over sets of a
    zero-padded offset table (so 0 is always present); Q[:,c] += rho_i*AtA_i[:,j]."""
    seen: List[int] = []
    for i in range(len(AtA)):
        for o in list(AtA_offsets[i]) + [0]:
            if int(o) not in seen:
                seen.append(int(o))
    Q_offsets = np.array(seen, np.int64)
    N = AtA[0].shape[0]
    Q = np.zeros((N, len(Q_offsets)), dtype=TF, order="F")
    for i in range(len(AtA)):
        for j, o in enumerate(AtA_offsets[i]):
            c = int(np.nonzero(Q_offsets == o)[0][0])
            Q[:, c] = Q[:, c] + TF(rho[i]) * AtA[i][:, j]
    return Q, Q_offsets


def PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, x=None, l=None, y=None,
            trace: Optional[list] = None, replay=None):
    """src/PARSDMM.jl:25-258 serial CDS path.  Returns (x, log_PARSDMM, l, y).
    replay = (rho_rows, gamma_rows) -- NOT part of the reference, a device of the parity tests: the rho / gamma history of
    ANOTHER run (its log.rho, log.gamma) is forced on this one -- after the rules of iteration i have run, rho and gamma
    take the values that run logged for iteration i + 1.  A threshold flip of the Barzilai-Borwein rule (the correlation of
    rounding noise falling on either side of 0.3) can then no longer separate the two runs, so everything ELSE -- operators,
    the x-step, every projector, the multiplier updates -- is compared over the whole solve at the tight tolerance."""
    TF = m.dtype.type
    convert_options(options, TF)                                                     # :43
    o = options
    maxit = int(o.maxit)
    rho_update_frequency = int(o.rho_update_frequency)
    adjust_rho, adjust_gamma, adjust_feasibility_rho = o.adjust_rho, o.adjust_gamma, o.adjust_feasibility_rho
    feasibility_only = o.feasibility_only
    mink = bool(getattr(o, "Minkowski", False))
    N = len(m)
    if x is None:
        x = np.zeros(N, TF)
    if mink:                                                                         # PARSDMM_initialize.jl:31-37
        if o.zero_ini_guess:
            N = 2 * len(x)
        else:
            assert len(x) == 2 * len(m)
            N = len(x)
    m_ext = np.concatenate([m, np.zeros(len(m), TF)]) if mink else m                 # :85-87
    # ---- PARSDMM_initialize (src/PARSDMM_initialize.jl:30-313) ----
    p = len(TD_OP)
    pp = p if feasibility_only else p - 1
    ind_ref = maxit
    rho = np.empty(p, TF)
    rho[:] = o.rho_ini[0] if len(o.rho_ini) == 1 else np.asarray(o.rho_ini, TF)      # :58-63
    prox = list(P_sub)
    if not feasibility_only:
        m_orig = m.copy()
        prox.append(lambda inp: prox_l2s(inp, rho[p - 1], m_orig))                   # :64-71
    eps = np.finfo(TF).eps
    feasibility_initial = np.zeros(len(P_sub), TF)
    for ii in range(len(P_sub)):                                                     # :97-99
        Am = csc_mul(TD_OP[ii], m_ext)
        feasibility_initial[ii] = TF(nrm2(P_sub[ii](Am.copy()) - Am, TF) / TF(nrm2(Am, TF) + TF(100) * eps))
    stop = bool(_nanmax(feasibility_initial) < o.feas_tol)                           # :101-104
    gamma_ini = TF(o.gamma_ini)
    for ii in range(pp):                                                             # :107-114
        if set_Prop.ncvx[ii]:
            rho_update_frequency, adjust_gamma, gamma_ini = 3, False, TF(0.75)
    M = [TD_OP[i].shape[0] for i in range(p)]
    if l is None or len(l) == 0:
        l = [np.zeros(M[i], TF) for i in range(p)]
    if y is None or len(y) == 0:
        y = [np.zeros(M[i], TF) for i in range(p)]
    gamma = np.full(p, gamma_ini, TF)
    z = lambda: [np.zeros(M[i], TF) for i in range(p)]
    y_0, y_old, l_0, l_old, l_hat_0, l_hat, x_hat, s_0, s, r_pri = (z(), z(), z(), z(), z(), z(), z(), z(), z(), z())
    x_old = np.zeros(N, TF)
    Q, Q_offsets = assemble_Q(AtA, set_Prop.AtA_offsets, rho, TF)                    # :216-230
    nfe = maxit                                                                      # :233 (maxit rows)
    log = log_type_PARSDMM(np.zeros((nfe, pp)), np.zeros((maxit, p)), np.zeros((maxit, p)),
                           np.zeros(maxit), np.zeros(maxit), np.zeros(maxit), np.zeros(maxit),
                           np.zeros((maxit, p)), np.zeros((maxit, p)), np.zeros(maxit, np.int64),
                           np.zeros(maxit))
    log.set_feasibility[0, :] = feasibility_initial                                  # :236
    if o.zero_ini_guess:                                                             # :304-313
        for v in l: v[:] = 0
        for v in y: v[:] = 0
        x[:] = 0
    # ---- back in PARSDMM ----
    if stop:                                                                         # PARSDMM.jl:63-82
        x[:len(m)] = m
        if mink and len(x) == len(m):
            x = np.concatenate([x, np.zeros(len(x), TF)])
        _truncate_log(log, 1, 1)
        return x, log, l, y
    if mink and len(x) == len(m):                                                    # :84-89
        x = np.concatenate([x, np.zeros(len(m), TF)])
    counter = 2                                                                      # :91
    x_solve_tol_ref = TF(1.0)                                                        # :93
    for i in range(1, maxit + 1):                                                    # :97
        rhs = rhs_compose(l, y, rho, TD_OP, p, N)                                    # :101
        x_old[:] = x                                                                 # :106
        x, it, relres, x_solve_tol_ref = argmin_x(Q, rhs, x, x_solve_tol_ref, i, Q_offsets)   # :107
        log.cg_it[i - 1] = it; log.cg_relres[i - 1] = relres
        counter = update_y_l(x, p, i, y, y_old, l, l_old, rho, gamma, prox, TD_OP, log, P_sub,
                             counter, x_hat, r_pri, s, feasibility_only)             # :133
        log.r_dual_total[i - 1] = _seq_sum(log.r_dual[i - 1, :], TF)                 # :134
        log.r_pri_total[i - 1] = _seq_sum(log.r_pri[i - 1, :], TF)                   # :138
        nd = nrm2((csc_mul(TD_OP[-1], x) if mink else x) - m, TF)                    # :139-143
        log.obj[i - 1] = TF(0.5) * TF(nd * nd)
        with np.errstate(all="ignore"):
            log.evol_x[i - 1] = TF(nrm2(x_old - x, TF) / nrm2(x, TF))                # :145
        log.rho[i - 1, :] = rho; log.gamma[i - 1, :] = gamma                         # :146-147
        if trace is not None:
            trace.append(dict(x=x.copy(), y=[v.copy() for v in y], l=[v.copy() for v in l],
                              rhs=rhs.copy(), tol=float(x_solve_tol_ref)))
        (stop, adjust_rho, adjust_gamma, adjust_feasibility_rho, ind_ref) = stop_PARSDMM(
            log, i, o.evol_rel_tol, o.feas_tol, o.obj_tol, adjust_rho, adjust_gamma,
            adjust_feasibility_rho, ind_ref, counter, TF)                            # :153
        if stop:
            _truncate_log(log, i, counter)                                           # :154-158
            return x, log, l, y
        if i == 1:                                                                   # :164-180
            for ii in range(p):
                l_hat[ii][:] = l_old[ii] + TF(rho[ii]) * (-s[ii] + y_old[ii])
                l_hat_0[ii][:] = l_hat[ii]; y_0[ii][:] = y[ii]; s_0[ii][:] = s[ii]; l_0[ii][:] = l[ii]
        if (adjust_rho or adjust_gamma) and i % rho_update_frequency == 0:           # :182
            adapt_rho_gamma(gamma, rho, adjust_gamma, adjust_rho, y, y_old, s, s_0, l, l_hat_0, l_0,
                            l_old, y_0, p, l_hat)
            if i > 1:                                                                # :192-206
                for ii in range(p):
                    l_hat_0[ii][:] = l_hat[ii]; y_0[ii][:] = y[ii]; s_0[ii][:] = s[ii]; l_0[ii][:] = l[ii]
        if adjust_feasibility_rho and i % 10 == 0:                                   # :213-223
            row = log.set_feasibility[counter - 2, :]
            if i > 10:
                k = _julia_argmax(row)
                rho[k] = TF(2.0) * rho[k]
        rho = np.maximum(np.minimum(rho, TF(1e4)), TF(1e-2))                         # :226 (new vector)
        if replay is not None and i < len(replay[0]):                                # (test device, see the docstring)
            rho = np.asarray(replay[0][i], np.float64).astype(TF)
            gamma[:] = np.asarray(replay[1][i], np.float64).astype(TF)
        ind_updated = [int(k) for k in np.nonzero(rho.astype(np.float64) != log.rho[i - 1, :])[0]]   # :230
        Q = Q_update(Q, AtA, set_Prop, rho, ind_updated, log, i - 1, Q_offsets)      # :243
        if i == maxit:                                                               # :249-252
            _truncate_log(log, i, counter)
    return x, log, l, y


def _julia_argmax(row) -> int:
    """findmax: first maximal element; a NaN wins (Julia isless ordering)."""
    row = np.asarray(row, np.float64)
    nan = np.nonzero(np.isnan(row))[0]
    if len(nan):
        return int(nan[0])
    return int(np.argmax(row))


def _seq_sum(row, TF):
    """sum() of a short Array{Real} row holding TF values: sequential TF additions."""
    acc = TF(row[0])
    for v in row[1:]:
        acc = TF(acc + TF(v))
    return acc


def _truncate_log(log, i, counter):
    """output_check_PARSDMM, src/PARSDMM.jl:261-278 (set_feasibility keeps `counter` rows)."""
    log.obj = log.obj[:i]; log.evol_x = log.evol_x[:i]
    log.r_pri_total = log.r_pri_total[:i]; log.r_dual_total = log.r_dual_total[:i]
    log.r_pri = log.r_pri[:i, :]; log.r_dual = log.r_dual[:i, :]
    log.cg_it = log.cg_it[:i]; log.cg_relres = log.cg_relres[:i]
    log.set_feasibility = log.set_feasibility[:counter, :]
    log.gamma = log.gamma[:i, :]; log.rho = log.rho[:i, :]


# --------------------------------------------------------------------------------------
# multilevel wrapper  (src/PARSDMM_multi_level.jl, src/interpolate_y_l.jl,
#                      src/setup_multi_level_PARSDMM.jl, src/constraint2coarse.jl)
# PARITY UNPINNED: the reference's multilevel test is disabled (test/runtests.jl:47) and the tie
# rounding of Interpolations.BSpline(Constant()) (Interpolations.jl 0.13, not vendored) is not covered by
# any reference test; half-way positions are taken to round up (floor(x + 1/2)), see _nn_index.
# --------------------------------------------------------------------------------------


def _nn_index(nc: int, nf: int) -> np.ndarray:
    """0-based source index of every fine index k: the grid point nearest to 1 + k (nc-1)/(nf-1), half-way
    positions going UP -- Interpolations.jl 0.13 (Project.toml:23; the package itself is not vendored) rounds
    Constant() positions with floor(x + 1/2) inside the axis (its `roundbounds`).  Exact integer arithmetic."""
    k = np.arange(nf, dtype=np.int64)
    if nf <= 1 or nc <= 1:
        return np.zeros(nf, np.int64)
    num, den = k * (nc - 1), nf - 1
    return (2 * num + den) // (2 * den)


def resample_nn(a, nc, nf):
    """itp = interpolate(reshape(a, nc), BSpline(Constant())); itp(range(1,stop=nc_d,length=nf_d)...)
    (src/PARSDMM_multi_level.jl:41-45,61-65)."""
    A = np.asarray(a).reshape(tuple(nc), order="F")
    for ax, (c, f) in enumerate(zip(nc, nf)):
        A = np.take(A, _nn_index(int(c), int(f)), axis=ax)
    return np.ascontiguousarray(A.reshape(-1, order="F"))


def _julia_round(v):
    return int(np.rint(v))


def constraint2coarse(constraint, comp_grid, cf):
    """src/constraint2coarse.jl:8-104 (mutates and returns the list, like the reference)."""
    n = tuple(int(v) for v in comp_grid.n)
    dim3 = len(n) == 3 and n[2] > 1
    for c in constraint:
        if c.set_type == "rank":
            c.max = min(c.max, min(n))
        if c.set_type == "cardinality":
            c.max = min(c.max, int(np.prod(n)))
        if c.set_type == "l1":
            c.max = c.max / (cf ** 3 if dim3 else cf ** 2)
        if c.set_type == "l2":
            c.max = c.max / (math.sqrt(cf ** 3) if dim3 else cf)
        if c.set_type == "nuclear" and not dim3:
            c.max = c.max / 2.7
    return constraint


def setup_multi_level_PARSDMM(m, n_levels, cf, comp_grid, constraint, options, mod=None):
    """src/setup_multi_level_PARSDMM.jl:7-137.  `mod` supplies setup_constraints /
    PARSDMM_precompute_distribute / compgrid (this module by default)."""
    import copy
    import sys
    mod = mod or sys.modules[__name__]
    TF = m.dtype.type
    P_sub, TD_OP, prop = mod.setup_constraints(copy.deepcopy(constraint), comp_grid, TF)
    TD_OP, AtA, l, y = mod.PARSDMM_precompute_distribute(TD_OP, prop, comp_grid, options)
    TD_OP_levels, AtA_levels, P_sub_levels, prop_levels, grid_levels = [TD_OP], [AtA], [P_sub], [prop], [comp_grid]
    constraint_level = copy.deepcopy(constraint)
    n0 = tuple(int(v) for v in comp_grid.n)
    for i in range(2, n_levels + 1):
        n = tuple(_julia_round(v / cf ** (i - 1)) for v in n0)                     # :66
        d = tuple((a / b) * dd for a, b, dd in zip(n0, n, comp_grid.d))            # :82
        g = mod.compgrid(d, n)
        constraint_level = constraint2coarse(constraint_level, g, cf)              # :87 (cumulative)
        P, A, pr = mod.setup_constraints(copy.deepcopy(constraint_level), g, TF)
        A, AtA_l, _, _ = mod.PARSDMM_precompute_distribute(A, pr, g, options)
        TD_OP_levels.append(A); AtA_levels.append(AtA_l); P_sub_levels.append(P); prop_levels.append(pr); grid_levels.append(g)
    return TD_OP_levels, AtA_levels, P_sub_levels, prop_levels, grid_levels, constraint_level


def interpolate_y_l(l, y, set_Prop_levels, comp_grid_levels, dim3, i, resample=resample_nn):
    """src/interpolate_y_l.jl:7-97, i = 0-based index of the FINER level.  The TV branch splits the
    multipliers with the block sizes of [D_x; D_y; D_z] although the storage order is [D_z; D_y; D_x]
    (:21-30 vs get_discrete_Grad.jl:72) -- replicated as written."""
    nc = tuple(int(v) for v in comp_grid_levels[i + 1].n)
    nf = tuple(int(v) for v in comp_grid_levels[i].n)
    for j in range(len(l)):
        tag = set_Prop_levels[i].tag[j][1]
        if tag in ("TV", "D2D", "D3D"):
            if dim3:
                shapes_c = [(nc[0] - 1, nc[1], nc[2]), (nc[0], nc[1] - 1, nc[2]), (nc[0], nc[1], nc[2] - 1)]
                shapes_f = [(nf[0] - 1, nf[1], nf[2]), (nf[0], nf[1] - 1, nf[2]), (nf[0], nf[1], nf[2] - 1)]
            else:
                shapes_c = [(nc[0] - 1, nc[1]), (nc[0], nc[1] - 1)]
                shapes_f = [(nf[0] - 1, nf[1]), (nf[0], nf[1] - 1)]
            ends = np.cumsum([int(np.prod(s)) for s in shapes_c])
            starts = np.concatenate(([0], ends[:-1]))
            l[j] = np.concatenate([resample(l[j][a:b], sc, sf) for a, b, sc, sf in zip(starts, ends, shapes_c, shapes_f)])
            y[j] = np.concatenate([resample(y[j][a:b], sc, sf) for a, b, sc, sf in zip(starts, ends, shapes_c, shapes_f)])
        else:
            tdn_f = tuple(int(v) for v in set_Prop_levels[i].TD_n[j])
            tdn_c = tuple(int(v) for v in set_Prop_levels[i + 1].TD_n[j])
            s = tuple(a - b for a, b in zip(nf, tdn_f))                                # :78
            fine = tuple(a - b for a, b in zip(nf, s))
            l[j] = resample(l[j], tdn_c, fine)
            y[j] = resample(y[j], tdn_c, fine)
    return l, y


def _carry_rho(options, log):
    """options.rho_ini = log.rho[end, :] (PARSDMM_multi_level.jl:57,83).  DEVIATION: when a level returned through
    the feasible-input exit (PARSDMM.jl:63-82) its one log row of rho is all zero and the reference would start the
    next level with rho = 0 (Q = 0, 1/rho = Inf, NaN iterates); the previous rho_ini is kept instead."""
    last = [float(v) for v in np.atleast_2d(log.rho)[-1, :]]
    if all(v > 0 for v in last):
        options.rho_ini = last


def PARSDMM_multi_level(m, TD_OP_levels, AtA_levels, P_sub_levels, set_Prop_levels, comp_grid_levels, options,
                        solver=None, resample=resample_nn):
    """src/PARSDMM_multi_level.jl:8-89."""
    solver = solver or PARSDMM
    n_levels = len(TD_OP_levels)
    n0 = tuple(int(v) for v in comp_grid_levels[0].n)
    dim3 = len(n0) == 3 and n0[2] > 1
    rho_orig = list(options.rho_ini)
    m_levels = [m] + [resample(m, n0, tuple(int(v) for v in comp_grid_levels[i].n)) for i in range(1, n_levels)]   # :40-48
    i = n_levels - 1
    options.zero_ini_guess = True                                                    # :53
    x, log, l, y = solver(m_levels[i], AtA_levels[i], TD_OP_levels[i], set_Prop_levels[i], P_sub_levels[i],
                          comp_grid_levels[i], options)
    _carry_rho(options, log)              # :57
    for i in range(n_levels - 2, -1, -1):
        nc = tuple(int(v) for v in comp_grid_levels[i + 1].n)
        nf = tuple(int(v) for v in comp_grid_levels[i].n)
        x = resample(x, nc, nf)                                                      # :61-67
        l, y = interpolate_y_l(list(l), list(y), set_Prop_levels, comp_grid_levels, dim3, i, resample)   # :74
        options.zero_ini_guess = False                                               # :81
        x, log, l, y = solver(m_levels[i], AtA_levels[i], TD_OP_levels[i], set_Prop_levels[i], P_sub_levels[i],
                              comp_grid_levels[i], options, x, l, y)
        _carry_rho(options, log)          # :83
    options.rho_ini = rho_orig                                                       # :87
    return x, log, l, y
