"""Host-side mirror of the reference's interface for the PARSDMM path, over the C ABI of
libsipx.so (include/sipx.h).  Same names, argument meaning and error behaviour as

    setup_constraints(constraint, comp_grid, TF)              src/setup_constraints.jl:17-102
    PARSDMM_precompute_distribute(TD_OP, set_Prop, grid, opt) src/PARSDMM_precompute_distribute.jl:6-77
    PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options[, x, l, y]) -> (x, log, l, y)
                                                              src/PARSDMM.jl:25-35,257

so that the parity tests read like the reference's own.  What differs, by design: TD_OP[i] is
a matrix-free operator DESCRIPTOR (TDOperator) instead of a SparseMatrixCSC and P_sub[i] is a
projector DESCRIPTOR (Projector) instead of an opaque closure -- both still behave like the
originals (``A @ x``, ``A.T @ v``, ``P(v)`` run the device kernels).  Nothing here computes on
the CPU; without libsipx.so or without a GPU every call raises SipxError.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

LIB_PATH = os.environ.get("SIPX_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsipx.so")

# every entry point declared in include/sipx.h
EXPORTED_SYMBOLS = [
    "sipx_last_error", "sipx_create", "sipx_destroy", "sipx_add_set", "sipx_set_rows", "sipx_num_terms",
    "sipx_finalize", "sipx_rhs_compose", "sipx_argmin_x", "sipx_update_y_l", "sipx_log_scalars",
    "sipx_adapt_rho_gamma", "sipx_q_update", "sipx_download", "sipx_parsdmm", "sipx_parsdmm_begin",
    "sipx_parsdmm_steps", "sipx_cds_spmv",
    "sipx_apply_op", "sipx_apply_op_adj", "sipx_project", "sipx_get_Q", "sipx_time_spmv", "sipx_kernel_stats",
    "sipx_debug_proj", "sipx_resample_nn", "sipx_set_q_mode", "sipx_apply_Q",
    "sipx_stream",
    "sipx_dev_rhs", "sipx_dev_x", "sipx_set_owned", "sipx_get_rhs", "sipx_prox_l2s",
    "sipx_rccl_unique_id", "sipx_set_comm_rccl", "sipx_set_comm", "sipx_slab", "sipx_warm_start_from", "sipx_set_decomp",
    "sipx_kernel_stats_json", "sipx_comm_info", "sipx_device_bytes", "sipx_reset",
]

SIPX_F32, SIPX_F64 = 0, 1
OPS = {"identity": 0, "D_x": 1, "D_y": 2, "D_z": 3, "TV": 4, "D2D": 4, "D3D": 4, "custom": 5}
PROJ = {"bounds": 0, "bounds_vec": 1, "l1": 2, "l2": 3, "annulus": 4, "cardinality": 5, "prox_l1": 6, "l1_dft": 7, "rank": 8,
        "nuclear": 9, "histogram": 10, "subspace": 11, "bounds_dft": 12}
MODES = {"matrix": 0, "tensor": 0, "fiber": 1, "slice": 2}
TRANSFORMS = {"DCT": 1}
SPECIAL_OPERATORS = ("DFT", "DCT", "wavelet", "curvelet")     # src/setup_constraints.jl:54
YL_FEAS, YL_BB, YL_FIRST = 1, 2, 4
Q_MODES = {"cds": 0, "stencil": 1}


class SipxError(RuntimeError):
    pass


class _SetDesc(C.Structure):
    _fields_ = [("op", C.c_int32), ("proj", C.c_int32), ("pmin", C.c_double), ("pmax", C.c_double),
                ("lb", C.c_void_p), ("ub", C.c_void_p), ("ncvx", C.c_int32), ("reserved", C.c_int32),
                ("mode", C.c_int32), ("dir", C.c_int32), ("basis", C.c_void_p), ("basis_rows", C.c_int64),
                ("basis_cols", C.c_int32), ("basis_orth", C.c_int32), ("component", C.c_int32),
                ("csc_colptr", C.c_void_p), ("csc_rowval", C.c_void_p), ("csc_nzval", C.c_void_p), ("csc_rows", C.c_int64),
                ("transform", C.c_int32), ("pad_", C.c_int32)]


class _Options(C.Structure):
    _fields_ = [("maxit", C.c_int32), ("evol_rel_tol", C.c_double), ("feas_tol", C.c_double),
                ("obj_tol", C.c_double), ("rho_update_frequency", C.c_int32), ("adjust_rho", C.c_int32),
                ("adjust_gamma", C.c_int32), ("adjust_feasibility_rho", C.c_int32)]


class _Log(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("set_feasibility", "r_dual", "r_pri", "r_dual_total", "r_pri_total",
                                           "obj", "evol_x", "rho", "gamma", "cg_it", "cg_relres")] + \
               [("timing_ms", C.c_double * 7), ("n_iter", C.c_int32), ("n_feas_rows", C.c_int32),
                ("stopped_feasible", C.c_int32)]


_lib = None
_default_device = 0


def set_default_device(device: int):
    """GPU used by contexts that are created implicitly (operator products, stand-alone projector calls)."""
    global _default_device
    _default_device = int(device)


def lib():
    """The loaded libsipx.so.  Raises when it has not been built: the product never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SipxError(f"{LIB_PATH} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`; "
                            "the engine has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.sipx_last_error.restype = C.c_char_p
        for name in ("sipx_stream", "sipx_dev_rhs", "sipx_dev_x"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.sipx_kernel_stats_json.restype = C.c_char_p
        L.sipx_kernel_stats_json.argtypes = [C.c_void_p, C.c_int]
        L.sipx_destroy.restype = None
        L.sipx_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise SipxError(lib().sipx_last_error().decode())


def _dtype_code(TF):
    TF = np.dtype(TF).type
    if TF == np.float32:
        return SIPX_F32
    if TF == np.float64:
        return SIPX_F64
    raise SipxError("FL must be Float32 or Float64")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------------------
# boundary types (src/SetIntersectionProjection.jl:95-149)
# --------------------------------------------------------------------------------------------------
@dataclass
class compgrid:
    d: Tuple
    n: Tuple


@dataclass
class PARSDMM_options:
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
    Q_mode: str = "cds"     # engine extension (not a reference field): "cds" = the reference's banded Q, "stencil" = sipx_set_q_mode(SIPX_Q_STENCIL)


def default_PARSDMM_options(options: PARSDMM_options, TF) -> PARSDMM_options:
    """src/default_PARSDMM_options.jl:6-34."""
    d = PARSDMM_options(FL=TF)
    for k, v in d.__dict__.items():
        setattr(options, k, v)
    return options


@dataclass
class set_definitions:
    set_type: str
    TD_OP: str
    min: Any
    max: Any
    app_mode: Tuple[str, str]
    custom_TD_OP: Tuple[Any, bool] = ((), False)


@dataclass
class set_properties:
    ncvx: List[bool] = field(default_factory=list)
    AtA_diag: List[bool] = field(default_factory=list)
    dense: List[bool] = field(default_factory=list)
    TD_n: List[Tuple] = field(default_factory=list)
    tag: List[Tuple[str, str, str, str]] = field(default_factory=list)
    banded: List[bool] = field(default_factory=list)
    AtA_offsets: List[Any] = field(default_factory=list)


@dataclass
class log_type_PARSDMM:
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


TIMING_SECTIONS = ("initialization", "form rhs for linear system", "argmin x", "argmin y and l update",
                   "stopping conditions check", "adjust rho and gamma", "Q-update")   # src/PARSDMM.jl @timeit names


def _grid(comp_grid):
    n = tuple(int(v) for v in comp_grid.n)
    if len(n) == 3 and n[2] == 1:
        n = n[:2]
    return n, tuple(float(v) for v in comp_grid.d[:len(n)])


# --------------------------------------------------------------------------------------------------
# descriptors standing in for TD_OP[i] and P_sub[i]
# --------------------------------------------------------------------------------------------------
class TDOperator:
    """Matrix-free stand-in for the SparseMatrixCSC a difference/identity TD_OP is in the reference."""

    def __init__(self, kind: str, comp_grid, TF, adjoint=False, component=0):
        if kind not in OPS:
            raise SipxError("provided an unknown transform domain operator. check function "
                            "get_TD_operator(comp_grid,TD_type,TF) for options")
        self.kind, self.comp_grid, self.TF, self.adjoint = kind, comp_grid, np.dtype(TF).type, adjoint
        n, _ = _grid(comp_grid)
        self.n = n
        N = int(np.prod(n))
        if kind == "identity":
            rows = N
        else:
            dirs = {"D_x": [0], "D_y": [1], "D_z": [len(n) - 1], "TV": list(range(len(n))), "D2D": list(range(len(n))),
                    "D3D": list(range(len(n)))}[kind]
            rows = sum(N // n[a] * (n[a] - 1) for a in dirs)
        # Minkowski block rows [A 0] (1), [0 A] (2), [A A] (3) act on [u; v]  (PARSDMM_precompute_distribute_Minkowski.jl:91-103)
        self.component = int(component)
        cols = 2 * N if self.component else N
        self.shape = (cols, rows) if adjoint else (rows, cols)

    @property
    def T(self):
        return TDOperator(self.kind, self.comp_grid, self.TF, not self.adjoint, self.component)

    def __matmul__(self, v):
        v = np.ascontiguousarray(v, dtype=self.TF)
        if v.shape != (self.shape[1],):
            raise SipxError("dimension mismatch")
        if self.component:
            plain = TDOperator(self.kind, self.comp_grid, self.TF, self.adjoint)
            N = int(np.prod(self.n))
            if self.adjoint:
                t = plain @ v
                z = np.zeros(N, self.TF)
                return np.concatenate([t if self.component in (1, 3) else z, t if self.component in (2, 3) else z])
            u, w = v[:N], v[N:]
            return plain @ (u if self.component == 1 else (w if self.component == 2 else (u + w).astype(self.TF)))
        out = np.empty(self.shape[0], self.TF)
        ctx = Context(self.comp_grid, self.TF)
        try:
            f = lib().sipx_apply_op_adj if self.adjoint else lib().sipx_apply_op
            _chk(f(ctx.h, OPS[self.kind], _ptr(v), _ptr(out)))
        finally:
            ctx.close()
        return out

    __mul__ = __matmul__


class CustomOperator:
    """A caller-supplied sparse TD_OP (constraint.custom_TD_OP[1], src/setup_constraints.jl:70-72): kept as a scipy CSC
    matrix on the host, shipped to the engine as SIPX_OP_CSC.  Behaves like the matrix for `@` and `.T` (host arithmetic,
    used by setup code only)."""
    kind = "custom"
    component = 0

    def __init__(self, A, comp_grid, TF):
        import scipy.sparse as sp
        self.TF = np.dtype(TF).type
        self.comp_grid = comp_grid
        self.n, _ = _grid(comp_grid)
        A = sp.csc_matrix(A, dtype=self.TF)
        A.sort_indices()
        if A.shape[1] != int(np.prod(self.n)):
            raise SipxError("custom TD_OP: the number of columns must equal the number of grid points")
        self.A = A
        self.shape = A.shape

    @property
    def T(self):
        return self.A.T.tocsc()

    def __matmul__(self, v):
        return (self.A @ np.asarray(v, self.TF)).astype(self.TF)

    def ata_cds(self):
        """mat2CDS(TD_OP' * TD_OP) (PARSDMM_precompute_distribute.jl:52-59, mat2CDS.jl:7-32) in the working precision."""
        G = (self.A.T @ self.A).tocsc().astype(self.TF)
        G.sort_indices()
        N = G.shape[0]
        coo = G.tocoo()
        offs = np.unique(coo.col.astype(np.int64) - coo.row.astype(np.int64))
        R = np.zeros((N, len(offs)), self.TF, order="F")
        col = {int(o): b for b, o in enumerate(offs)}
        for r, c, v in zip(coo.row, coo.col, coo.data):
            R[r, col[int(c) - int(r)]] = v
        return R, offs.astype(np.int64)


class Projector:
    """Descriptor of P_sub[i] (src/get_projector.jl:3-103); calling it projects in place on the device."""

    def __init__(self, constraint: set_definitions, comp_grid, TF):
        self.TF = np.dtype(TF).type
        self.comp_grid = comp_grid
        st = constraint.set_type
        n, _ = _grid(comp_grid)
        am = tuple(constraint.app_mode)
        if am[0] not in MODES:
            raise SipxError(f"unknown application mode {am!r}")
        self.mode, self.dir = MODES[am[0]], 0
        if self.mode:
            dirs = {"x": 0, "z": len(n) - 1} if len(n) == 2 else {"x": 0, "y": 1, "z": 2}
            if am[1] not in dirs:
                raise SipxError(f"application mode {am!r}: direction must be one of {sorted(dirs)} on this grid")
            self.dir = dirs[am[1]]
        self.lb = self.ub = self.basis = None
        self.basis_orth = False
        self.pmin = self.pmax = 0.0
        self.transform = 0
        if constraint.TD_OP == "DCT":
            # x -> C' P(C x) with the orthonormal DCT-II (get_projector.jl, special_operator_list branches); the l2 ball and
            # the annulus commute with an orthogonal transform and are applied to x itself
            if st in ("l2", "annulus"):
                self.kind, self.pmax = st, float(constraint.max)
                self.pmin = float(constraint.min) if st == "annulus" else 0.0
            elif st in ("l1", "cardinality"):
                self.kind, self.pmax, self.transform = st, float(constraint.max), TRANSFORMS["DCT"]
            elif st == "bounds" and np.ndim(constraint.min) == 0:
                self.kind, self.pmin, self.pmax, self.transform = "bounds", float(constraint.min), float(constraint.max), TRANSFORMS["DCT"]
            elif st == "bounds":
                self.kind, self.transform = "bounds_vec", TRANSFORMS["DCT"]
                self.lb = np.ascontiguousarray(constraint.min, self.TF)
                self.ub = np.ascontiguousarray(constraint.max, self.TF)
            else:
                raise SipxError(f"set type {st!r} behind the DCT is not built")
            if self.mode:
                raise SipxError("sets behind the DCT apply to the whole array (matrix / tensor mode)")
        elif constraint.TD_OP in SPECIAL_OPERATORS:
            if constraint.TD_OP == "DFT" and st == "l1":
                self.kind, self.pmax = "l1_dft", float(constraint.max)
            elif constraint.TD_OP == "DFT" and st == "bounds" and np.ndim(constraint.min) == 1:
                lb = np.asarray(constraint.min)
                ub = np.ascontiguousarray(constraint.max, self.TF)
                if np.any(lb != 0) or len(np.unique(ub)) != 2:        # project_bounds!.jl:31-32 (@assert)
                    raise SipxError("bounds in the DFT domain: LB must be all zeros and UB a two-valued mask")
                self.kind, self.ub = "bounds_dft", ub
            elif constraint.TD_OP == "DFT" and st in ("l2", "annulus"):
                # ||F x||_2 = ||x||_2 for the unitary DFT and the projection is a rescaling (or, for the zero vector of an
                # annulus, a constant fill whose DC-only spectrum transforms back to a constant): x -> Re(F' P(F x)) is
                # P applied to x itself, without the FFT round trip of get_projector.jl:39,47
                self.kind, self.pmax = st, float(constraint.max)
                self.pmin = float(constraint.min) if st == "annulus" else 0.0
            else:
                raise SipxError("of the orthogonal-transform sets only the l1 ball, masking bounds, the l2 ball and the annulus "
                                "in the DFT domain are built")
        elif st == "rank":
            self.kind, self.pmax = "rank", float(int(constraint.max))
        elif st == "nuclear":
            self.kind, self.pmax = "nuclear", float(constraint.max)
        elif st == "bounds":
            if np.ndim(constraint.min) == 0:
                self.kind, self.pmin, self.pmax = "bounds", float(constraint.min), float(constraint.max)
            else:
                self.kind = "bounds_vec"
                self.lb = np.ascontiguousarray(constraint.min, self.TF)
                self.ub = np.ascontiguousarray(constraint.max, self.TF)
        elif st == "histogram":
            self.kind = "histogram"
            self.lb = np.ascontiguousarray(constraint.min, self.TF)
            self.ub = np.ascontiguousarray(constraint.max, self.TF)
        elif st == "subspace":
            A, orth = constraint.custom_TD_OP
            self.kind = "subspace"
            self.basis = np.asfortranarray(A, dtype=self.TF)
            if self.basis.ndim != 2:
                raise SipxError("subspace constraints need a matrix A in custom_TD_OP[0]")
            self.basis_orth = bool(orth)
        elif st in ("l1", "l2", "prox_l1"):
            self.kind, self.pmax = st, float(constraint.max)
        elif st == "annulus":
            self.kind, self.pmin, self.pmax = st, float(constraint.min), float(constraint.max)
        elif st == "cardinality":
            self.kind, self.pmax = st, float(int(constraint.max))
        else:
            raise SipxError(f"set type {st!r} is not part of this engine")

    def check_rows(self, op: "TDOperator"):
        """Host-side shape checks of the vectors the descriptor points at (the engine reads them unchecked)."""
        n = list(op.n)
        if op.kind == "custom" and (self.mode or self.transform or self.kind in ("l1_dft", "bounds_dft", "rank", "nuclear",
                                                                                     "histogram", "subspace")):
            raise SipxError("custom sparse operators take the whole-array projectors only")
        if op.kind in ("D_x", "D_y", "D_z"):
            n[{"D_x": 0, "D_y": 1, "D_z": len(n) - 1}[op.kind]] -= 1
        rows = op.shape[0]
        if self.kind == "bounds_vec":
            if self.mode == MODES["slice"]:
                raise SipxError("bound constraints per slice of a tensor currently not implemented, yet...")
            want = n[self.dir] if self.mode == MODES["fiber"] else rows
            if self.lb.shape != (want,) or self.ub.shape != (want,):
                raise SipxError(f"bounds vectors need {want} entries for this operator / application mode")
        if self.kind == "bounds_dft" and self.ub.shape != (rows,):
            raise SipxError(f"the DFT-domain mask needs {rows} entries")
        if self.kind == "histogram" and (self.lb.shape != (rows,) or self.ub.shape != (rows,)):
            raise SipxError(f"histogram bounds need {rows} sorted entries")

    def desc(self, op: str, ncvx: bool) -> _SetDesc:
        d = _SetDesc()
        d.op, d.proj = OPS[op], PROJ[self.kind]
        d.pmin, d.pmax = self.pmin, self.pmax
        d.lb, d.ub = _ptr(self.lb), _ptr(self.ub)
        d.ncvx, d.reserved = int(bool(ncvx)), 0
        d.mode, d.dir = self.mode, self.dir
        d.transform = self.transform
        if self.basis is not None:
            d.basis, d.basis_rows, d.basis_cols = _ptr(self.basis), self.basis.shape[0], self.basis.shape[1]
            d.basis_orth = int(self.basis_orth)
        return d

    def __call__(self, v):
        if v.dtype.type != self.TF or not v.flags.c_contiguous:
            raise SipxError("projector input must be a contiguous vector of the working precision")
        grid_kind = self.mode != 0 or self.transform != 0 or self.kind in ("l1_dft", "bounds_dft", "rank", "nuclear", "histogram", "subspace")
        ctx = Context(self.comp_grid if grid_kind else compgrid((1.0, 1.0), (max(len(v), 1), 1)), self.TF)
        try:
            if grid_kind:
                self.check_rows(TDOperator("identity", self.comp_grid, self.TF))
            elif self.kind == "bounds_vec" and (self.lb.shape != v.shape or self.ub.shape != v.shape):
                raise SipxError("bounds vectors must match the projected vector")
            d = self.desc("identity", False)
            _chk(lib().sipx_project(ctx.h, C.byref(d), _ptr(v), C.c_int64(len(v))))
        finally:
            ctx.close()
        return v


# --------------------------------------------------------------------------------------------------
# one-off setup
# --------------------------------------------------------------------------------------------------
def get_TD_operator(comp_grid, TD_type: str, TF):
    """src/get_TD_operator.jl:12-95 for the banded operators."""
    n, _ = _grid(comp_grid)
    if TD_type in ("DFT", "DCT"):   # src/get_TD_operator.jl:45-51,80-86; setup_constraints.jl:76-80 swaps in the identity
        return TDOperator("identity", comp_grid, TF), True, True, n, False
    if TD_type == "D_xz":      # src/get_TD_operator.jl:66-70 (2-D only): TD_OP = D_z * D_x, shipped as a sparse matrix
        if len(n) != 2:
            raise SipxError("provided an unknown transform domain operator. check function "
                            "get_TD_operator(comp_grid,TD_type,TF) for options")
        import scipy.sparse as sp
        TFt = np.dtype(TF).type
        _, h = _grid(comp_grid)

        def fwd(k, hk):       # (k-1) x k forward difference with entries -+fl(1/h)   (get_discrete_Grad.jl:22-23)
            ih = TFt(1) / TFt(hk)
            return sp.diags([np.full(k - 1, -ih, TFt), np.full(k - 1, ih, TFt)], [0, 1], shape=(k - 1, k), dtype=TFt, format="csc")
        n1, n2 = n
        Dx = sp.kron(sp.identity(n2, dtype=TFt, format="csc"), fwd(n1, h[0]), format="csc")            # on (n1, n2)
        Dz = sp.kron(fwd(n2, h[1]), sp.identity(n1 - 1, dtype=TFt, format="csc"), format="csc")        # on (n1-1, n2)
        return CustomOperator((Dz @ Dx).astype(TFt), comp_grid, TF), False, False, (n1 - 1, n2 - 1), True
    A = TDOperator(TD_type, comp_grid, TF)
    if TD_type == "identity":
        return A, True, False, n, True
    if len(n) == 2:
        n1, n2 = n
        TD_n = {"TV": ((n1 - 1) + n1, n2 + (n2 - 1)), "D2D": ((n1 - 1) + n1, n2 + (n2 - 1)), "D_z": (n1, n2 - 1),
                "D_x": (n1 - 1, n2)}[TD_type]
    else:
        n1, n2, n3 = n
        TD_n = {"TV": (3 * n1 - 1, 3 * n2 - 1, 3 * n3 - 1), "D3D": (3 * n1 - 1, 3 * n2 - 1, 3 * n3 - 1),
                "D_z": (n1, n2, n3 - 1), "D_y": (n1, n2 - 1, n3), "D_x": (n1 - 1, n2, n3)}[TD_type]
    return A, False, False, TD_n, True


def setup_constraints(constraint: List[set_definitions], comp_grid, TF):
    """src/setup_constraints.jl:17-102 -> (P_sub, TD_OP, set_Prop)."""
    TF = np.dtype(TF).type
    P_sub, TD_OP, prop = [], [], set_properties()
    for c in constraint:
        if np.ndim(c.min) == 0:
            if isinstance(c.min, (float, np.floating)):
                c.min, c.max = TF(c.min), TF(c.max)
        else:
            c.min, c.max = np.asarray(c.min, TF), np.asarray(c.max, TF)
        if c.set_type in ("nuclear", "rank") and c.app_mode[0] in ("matrix", "tensor") and len(comp_grid.n) == 3:
            raise SipxError("requested rank or nuclear norm constraints on a tensor, use mode=(slice,x) e.t.c. to "
                            "define constraints per slice")
        if c.set_type in ("l1", "l2") and c.app_mode[0] in ("slice", "fiber"):
            raise SipxError("l1 and l2 constraints only available for matrix or tensor mode, currently")
        A, AtA_diag, dense, TD_n, banded = get_TD_operator(comp_grid, c.TD_OP, TF)
        cust = c.custom_TD_OP[0] if c.set_type != "subspace" else ()
        if not (isinstance(cust, (tuple, list)) and len(cust) == 0):          # setup_constraints.jl:70-72
            A = CustomOperator(cust, comp_grid, TF)
            AtA_diag, dense = False, False
        banded = True       # the engine keeps Q in CDS: a DFT set contributes the identity band like any orthogonal op
        P_sub.append(Projector(c, comp_grid, TF))
        TD_OP.append(A)
        prop.AtA_diag.append(AtA_diag); prop.dense.append(dense); prop.TD_n.append(TD_n)
        prop.banded.append(banded); prop.AtA_offsets.append(None)
        prop.tag.append((c.set_type, c.TD_OP, c.app_mode[0], c.app_mode[1]))
        if c.set_type in ("rank", "cardinality"):
            ncvx = True
        elif c.set_type in ("bounds", "histogram") and c.TD_OP != "identity" and TF(np.max(c.min)) > TF(0):
            ncvx = True
        else:
            ncvx = False
        prop.ncvx.append(ncvx)
    return P_sub, TD_OP, prop


def PARSDMM_precompute_distribute(TD_OP, set_Prop, comp_grid, options):
    """src/PARSDMM_precompute_distribute.jl:6-77.  AtA[i] = None means "generate the CDS bands of
    A_i'A_i on the device from the descriptor" (bit-identical to mat2CDS(TD_OP'*TD_OP))."""
    TF = np.dtype(options.FL).type
    n, _ = _grid(comp_grid)
    if not options.feasibility_only:
        TD_OP.append(TDOperator("identity", comp_grid, TF))
        set_Prop.TD_n.append(n); set_Prop.AtA_offsets.append(np.array([0], np.int64))
        set_Prop.banded.append(True); set_Prop.AtA_diag.append(True)
        set_Prop.ncvx.append(False); set_Prop.dense.append(False)
        set_Prop.tag.append(("distance squared", "identity", "matrix", ""))
    p = len(TD_OP)
    AtA = [None] * p
    st = {1: 1, 2: n[0], 3: n[0] * n[1] if len(n) > 2 else None}
    for i in range(p):
        kind = TD_OP[i].kind
        if kind == "custom":
            AtA[i], set_Prop.AtA_offsets[i] = TD_OP[i].ata_cds()
            continue
        dirs = {"identity": [], "D_x": [0], "D_y": [1], "D_z": [len(n) - 1]}.get(kind, list(range(len(n))))
        strides = [int(np.prod(n[:a])) for a in dirs]
        set_Prop.AtA_offsets[i] = np.array(sorted({0} | {s for s in strides} | {-s for s in strides}), np.int64)
    del st
    y = [np.zeros(TD_OP[i].shape[0], TF) for i in range(p)]
    l = [np.zeros(TD_OP[i].shape[0], TF) for i in range(p)]
    return TD_OP, AtA, l, y


def PARSDMM_precompute_distribute_Minkowski(TD_OP_c1, TD_OP_c2, TD_OP_sum, set_Prop_c1, set_Prop_c2, set_Prop_sum,
                                           comp_grid, options):
    """src/PARSDMM_precompute_distribute_Minkowski.jl:6-173 -> (TD_OP, set_Prop, AtA, l, y): the operators become the
    block rows [A 0], [0 A], [A A] acting on x = [u; v], the distance term [I I] is appended; AtA[i] = None (the 2N x 2N
    CDS bands are generated on the device)."""
    import copy
    TF = np.dtype(options.FL).type
    n, _ = _grid(comp_grid)
    groups = ((TD_OP_c1, 1), (TD_OP_c2, 2), (TD_OP_sum, 3))
    TD_OP = [TDOperator(A.kind, comp_grid, TF, component=c) for ops, c in groups for A in ops]
    prop = copy.deepcopy(set_Prop_c1)
    for other in (set_Prop_c2, set_Prop_sum):
        for f in ("AtA_diag", "AtA_offsets", "TD_n", "banded", "dense", "ncvx", "tag"):
            getattr(prop, f).extend(copy.deepcopy(getattr(other, f)))
    if not options.feasibility_only:
        TD_OP.append(TDOperator("identity", comp_grid, TF, component=3))
        prop.TD_n.append(n); prop.AtA_offsets.append(np.array([0], np.int64))
        prop.banded.append(True); prop.AtA_diag.append(False); prop.dense.append(False)
        prop.ncvx.append(False); prop.tag.append(("distance squared", "identity", "matrix", ""))
    s = len(TD_OP)
    N = int(np.prod(n))
    for i in range(s):
        kind, comp = TD_OP[i].kind, TD_OP[i].component
        dirs = {"identity": [], "D_x": [0], "D_y": [1], "D_z": [len(n) - 1]}.get(kind, list(range(len(n))))
        base = sorted({0} | {int(np.prod(n[:a])) for a in dirs} | {-int(np.prod(n[:a])) for a in dirs})
        off = sorted({o + sh for o in base for sh in (-N, 0, N)}) if comp == 3 else base
        prop.AtA_offsets[i] = np.array(off, np.int64)
    y = [np.zeros(TD_OP[i].shape[0], TF) for i in range(s)]
    l = [np.zeros(TD_OP[i].shape[0], TF) for i in range(s)]
    return TD_OP, prop, [None] * s, l, y


# --------------------------------------------------------------------------------------------------
# engine handle (phase-level API, include/sipx.h section A)
# --------------------------------------------------------------------------------------------------
class Context:
    def __init__(self, comp_grid, TF, device: Optional[int] = None):
        device = _default_device if device is None else device
        self.TF = np.dtype(TF).type
        n, h = _grid(comp_grid)
        self.n, self.N = n, int(np.prod(n))
        self.Nx = self.N                         # unknowns: 2N once a Minkowski component is added
        self.h = C.c_void_p()
        na = (C.c_int64 * len(n))(*n)
        ha = (C.c_double * len(n))(*h)
        _chk(lib().sipx_create(C.byref(self.h), _dtype_code(self.TF), len(n), na, ha, device))
        self.rows: List[int] = []
        self.p = self.pp = 0
        self._keep = []

    def close(self):
        if self.h:
            lib().sipx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_set(self, op: TDOperator, proj: Projector, ncvx=False, AtA=None, AtA_offsets=None) -> int:
        proj.check_rows(op)
        d = proj.desc(op.kind, ncvx)
        d.component = int(getattr(op, "component", 0))
        if d.component:
            self.Nx = 2 * self.N
        if op.kind == "custom":
            if AtA is None:
                raise SipxError("a custom sparse operator needs its AtA (PARSDMM_precompute_distribute computes it)")
            cp = np.ascontiguousarray(op.A.indptr, np.int64); ri = np.ascontiguousarray(op.A.indices, np.int64)
            nz = np.ascontiguousarray(op.A.data, self.TF)
            self._keep += [cp, ri, nz]
            d.csc_colptr, d.csc_rowval, d.csc_nzval = cp.ctypes.data, ri.ctypes.data, nz.ctypes.data
            d.csc_rows = int(op.A.shape[0])
        self._keep.append(proj)
        if AtA is not None:
            R = np.asfortranarray(AtA, dtype=self.TF)
            off = np.ascontiguousarray(AtA_offsets, np.int64)
            rc = lib().sipx_add_set(self.h, C.byref(d), _ptr(R), off.ctypes.data_as(C.c_void_p), int(R.shape[1]))
        else:
            rc = lib().sipx_add_set(self.h, C.byref(d), None, None, 0)
        if rc < 0:
            raise SipxError(lib().sipx_last_error().decode())
        r = C.c_int64()
        _chk(lib().sipx_set_rows(self.h, rc, C.byref(r)))
        self.rows.append(int(r.value))
        return rc

    def set_q_mode(self, mode: str):
        if mode not in Q_MODES:
            raise SipxError(f"unknown Q mode {mode!r} (cds | stencil)")
        _chk(lib().sipx_set_q_mode(self.h, Q_MODES[mode]))

    def apply_Q(self, x):
        x = np.ascontiguousarray(x, self.TF)
        if x.shape != (self.Nx,):
            raise SipxError("length of x does not match the number of unknowns")
        y = np.empty_like(x)
        _chk(lib().sipx_apply_Q(self.h, _ptr(x), _ptr(y)))
        return y

    def set_decomp(self, mode: str):
        """"sets": the reference's split by constraint set; "slab": every rank works on its z-slab of every set (sipx.h)."""
        if mode not in ("sets", "slab", "slab_full"):
            raise SipxError(f"unknown decomposition {mode!r} (sets | slab | slab_full)")
        # slab: a rank's arrays hold its planes only (sparse arrays); slab_full: whole arrays on every rank (levels of a multilevel solve)
        _chk(lib().sipx_set_decomp(self.h, {"sets": 0, "slab": 1, "slab_full": 2}[mode]))

    def set_owned(self, owned: Sequence[int]):
        a = np.zeros(len(self.rows) + 1, np.int32)       # constraint sets + the distance term
        a[:len(owned)] = np.asarray(owned, np.int32)[:len(a)]
        _chk(lib().sipx_set_owned(self.h, a.ctypes.data_as(C.c_void_p)))

    def finalize(self, m, rho_ini, gamma_ini, feasibility_only=False, zero_ini_guess=True, x0=None, l0=None, y0=None):
        m = np.ascontiguousarray(m, self.TF)
        if m.shape != (self.N,):
            raise SipxError("length of m does not match the grid")
        rho = np.ascontiguousarray(rho_ini, np.float64)
        npp = len(self.rows)
        feas = np.zeros(max(npp, 1))
        keep = []

        rows_all = list(self.rows) + ([] if feasibility_only else [self.N])

        def arr_list(lst):
            if lst is None or len(lst) == 0:
                return None
            if len(lst) != len(rows_all):
                raise SipxError("warm start: l and y need one vector per term (sets plus the distance term)")
            ptrs = (C.c_void_p * len(lst))()
            for i, a in enumerate(lst):
                a = np.ascontiguousarray(a, self.TF)
                if a.shape != (rows_all[i],):
                    raise SipxError(f"warm start: vector {i} has {a.shape} entries, operator {i} has {rows_all[i]} rows")
                keep.append(a)
                ptrs[i] = a.ctypes.data
            return ptrs
        x0a = None if x0 is None else np.ascontiguousarray(x0, self.TF)
        if x0a is not None and x0a.shape != (self.Nx,):
            raise SipxError("length of x does not match the grid")
        _chk(lib().sipx_finalize(self.h, _ptr(m), rho.ctypes.data_as(C.c_void_p), len(rho), C.c_double(gamma_ini),
                                 int(feasibility_only), int(zero_ini_guess), _ptr(x0a), arr_list(l0), arr_list(y0),
                                 feas.ctypes.data_as(C.c_void_p)))
        p, pp = C.c_int(), C.c_int()
        _chk(lib().sipx_num_terms(self.h, C.byref(p), C.byref(pp)))
        self.p, self.pp = p.value, pp.value
        if self.p > npp:
            self.rows.append(self.N)
        return feas[:npp]

    def reset(self, m, rho_ini, gamma_ini, zero_ini_guess=True, x0=None, l0=None, y0=None):
        """The same sets on the same grid, once more (sipx_reset): a new model m -- and, optionally, a new warm start and rho_ini --
        on this finalised context, without any allocation.  Returns the initial feasibilities.  A solve that follows gives the
        bits a newly built context gives."""
        m = np.ascontiguousarray(m, self.TF)
        if m.shape != (self.N,):
            raise SipxError("length of m does not match the grid")
        rho = np.ascontiguousarray(rho_ini, np.float64)
        npp = self.pp
        feas = np.zeros(max(npp, 1))
        keep = []

        def arr_list(lst):
            if lst is None or len(lst) == 0:
                return None
            if len(lst) != len(self.rows):
                raise SipxError("warm start: l and y need one vector per term (sets plus the distance term)")
            ptrs = (C.c_void_p * len(lst))()
            for i, a in enumerate(lst):
                a = np.ascontiguousarray(a, self.TF)
                if a.shape != (self.rows[i],):
                    raise SipxError(f"warm start: vector {i} has {a.shape} entries, operator {i} has {self.rows[i]} rows")
                keep.append(a)
                ptrs[i] = a.ctypes.data
            return ptrs
        x0a = None if x0 is None else np.ascontiguousarray(x0, self.TF)
        if x0a is not None and x0a.shape != (self.Nx,):
            raise SipxError("length of x does not match the grid")
        _chk(lib().sipx_reset(self.h, _ptr(m), rho.ctypes.data_as(C.c_void_p), len(rho), C.c_double(gamma_ini),
                              int(zero_ini_guess), _ptr(x0a), arr_list(l0), arr_list(y0), feas.ctypes.data_as(C.c_void_p)))
        self.feasibility_initial = feas[:npp]
        return self.feasibility_initial

    # ---- phases ----
    def rhs_compose(self, rho):
        rho = np.ascontiguousarray(rho, np.float64)
        _chk(lib().sipx_rhs_compose(self.h, rho.ctypes.data_as(C.c_void_p)))

    def argmin_x(self, it, tol_ref):
        t, ci, rr, fl = C.c_double(tol_ref), C.c_int64(), C.c_double(), C.c_int()
        _chk(lib().sipx_argmin_x(self.h, int(it), C.byref(t), C.byref(ci), C.byref(rr), C.byref(fl)))
        return t.value, ci.value, rr.value, fl.value

    def update_y_l(self, it, flags, rho, gamma):
        rho = np.ascontiguousarray(rho, np.float64); gamma = np.ascontiguousarray(gamma, np.float64)
        rp, rd, fe = np.zeros(self.p), np.zeros(self.p), np.zeros(max(self.pp, 1))
        _chk(lib().sipx_update_y_l(self.h, int(it), int(flags), rho.ctypes.data_as(C.c_void_p),
                                   gamma.ctypes.data_as(C.c_void_p), rp.ctypes.data_as(C.c_void_p),
                                   rd.ctypes.data_as(C.c_void_p), fe.ctypes.data_as(C.c_void_p)))
        return rp, rd, fe[:self.pp]

    def log_scalars(self):
        a, b = C.c_double(), C.c_double()
        _chk(lib().sipx_log_scalars(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def adapt_rho_gamma(self, adjust_rho, adjust_gamma, rho, gamma):
        rho = np.array(rho, np.float64); gamma = np.array(gamma, np.float64)
        _chk(lib().sipx_adapt_rho_gamma(self.h, int(adjust_rho), int(adjust_gamma), rho.ctypes.data_as(C.c_void_p),
                                        gamma.ctypes.data_as(C.c_void_p)))
        return rho, gamma

    def q_update(self, rho_new, rho_old):
        a = np.ascontiguousarray(rho_new, np.float64); b = np.ascontiguousarray(rho_old, np.float64)
        _chk(lib().sipx_q_update(self.h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))

    def download(self, want_ly=True, x_out=None):
        # the reference overwrites the x argument in place (cg.jl:95, PARSDMM.jl:64): a caller's array of the right kind is the
        # destination itself -- its pages exist already, a fresh array pays a page fault per 4 KiB of the copy
        if (x_out is not None and isinstance(x_out, np.ndarray) and x_out.dtype == self.TF and x_out.shape == (self.Nx,)
                and x_out.flags.c_contiguous and x_out.flags.writeable):
            x = x_out
        else:
            x = np.empty(self.Nx, self.TF)
        l = [np.zeros(r, self.TF) for r in self.rows] if want_ly else None
        y = [np.zeros(r, self.TF) for r in self.rows] if want_ly else None

        def ptrs(lst):
            if lst is None:
                return None
            a = (C.c_void_p * len(lst))()
            for i, v in enumerate(lst):
                a[i] = v.ctypes.data
            return a
        _chk(lib().sipx_download(self.h, _ptr(x), ptrs(l), ptrs(y)))
        return x, l, y

    def warm_start_from(self, coarse: "Context"):
        """x, l, y of this (finer) level from a solved coarser context, resampled on the device (sipx_warm_start_from)."""
        _chk(lib().sipx_warm_start_from(self.h, coarse.h))

    def slab(self):
        """(row0, row1, chunk): this rank's rows of the x-step and the elements per rank of the padded exchange buffers."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(lib().sipx_slab(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def get_rhs(self):
        """rhs of the last rhs_compose (src/rhs_compose.jl:24-36), copied to the host."""
        rhs = np.empty(self.Nx, self.TF)
        _chk(lib().sipx_get_rhs(self.h, _ptr(rhs)))
        return rhs

    def get_Q(self):
        d = C.c_int()
        offs = np.zeros(32, np.int64)
        _chk(lib().sipx_get_Q(self.h, None, offs.ctypes.data_as(C.c_void_p), C.byref(d)))
        Q = np.empty((self.Nx, d.value), self.TF, order="F")
        _chk(lib().sipx_get_Q(self.h, _ptr(Q), offs.ctypes.data_as(C.c_void_p), C.byref(d)))
        return Q, offs[:d.value].copy()

    def time_spmv(self, reps=20):
        ms = C.c_double()
        _chk(lib().sipx_time_spmv(self.h, int(reps), C.byref(ms)))
        return ms.value

    def debug_proj(self, set_index: int, which: int = 0):
        out = np.zeros(16)
        _chk(lib().sipx_debug_proj(self.h, int(set_index), int(which), out.ctypes.data_as(C.c_void_p)))
        keys = ("need", "theta", "theta_prev", "hw", "spec_lo", "spec_hi", "lo", "hi", "asum", "vmax", "gathered",
                "overflow", "spec_ok", "michelot_its", "refine", "lean")
        d = dict(zip(keys, out))
        d["sampled"] = float(int(d["lean"]) >> 1)      # the probes of the last search were centred by the sampled estimate
        d["lean"] = float(int(d["lean"]) & 1)
        return d

    def kernel_stats(self, enable):
        """(launches, total ms) of the CG product since the last call; then (re)starts the collection: 0 / False = off,
        1 / True = that kernel only, 2 = every kernel (sipx.h, sipx_kernel_stats)."""
        n, ms = C.c_int64(), C.c_double()
        _chk(lib().sipx_kernel_stats(self.h, int(enable), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def kernel_stats_all(self, enable):
        """Per-kernel statistics gathered since the last call, as a dict (sipx_kernel_stats_json); `enable` as above.
        enable = -1: the engine's counters only (`slab_searches`, `rank_route`) -- nothing is synchronised, a running
        collection and its samples stay as they are."""
        import json
        txt = lib().sipx_kernel_stats_json(self.h, int(enable))
        if txt is None:
            raise SipxError(lib().sipx_last_error().decode())
        return json.loads(txt.decode())

    def device_bytes(self):
        """{"context": bytes this context allocated on its GPU, "device_used", "device_total": the runtime's figures} (sipx_device_bytes)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(lib().sipx_device_bytes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"context": a.value, "device_used": b.value, "device_total": c.value}

    def comm_info(self):
        """What the attached communicator reports: {"nranks", "rank", "version", "decomposition"} (sipx_comm_info)."""
        n, r, d = C.c_int(), C.c_int(), C.c_int()
        ver = C.create_string_buffer(64)
        _chk(lib().sipx_comm_info(self.h, C.byref(n), C.byref(r), ver, 64, C.byref(d)))
        return {"nranks": n.value, "rank": r.value, "version": ver.value.decode(), "decomposition": "slab" if d.value == 1 else "sets"}

    def _log_struct(self, options: PARSDMM_options):
        maxit, p, pp = int(options.maxit), self.p, self.pp
        o = _Options(maxit, float(options.evol_rel_tol), float(options.feas_tol), float(options.obj_tol),
                     int(options.rho_update_frequency), int(options.adjust_rho), int(options.adjust_gamma),
                     int(options.adjust_feasibility_rho))
        arrs = dict(set_feasibility=np.zeros((maxit, max(pp, 1))), r_dual=np.zeros((maxit, p)),
                    r_pri=np.zeros((maxit, p)), r_dual_total=np.zeros(maxit), r_pri_total=np.zeros(maxit),
                    obj=np.zeros(maxit), evol_x=np.zeros(maxit), rho=np.zeros((maxit, p)),
                    gamma=np.zeros((maxit, p)), cg_it=np.zeros(maxit, np.int64), cg_relres=np.zeros(maxit))
        lg = _Log()
        for k, a in arrs.items():
            setattr(lg, k, a.ctypes.data)
        return o, lg, arrs

    def _log_result(self, lg, arrs):
        it, nf, pp = lg.n_iter, lg.n_feas_rows, self.pp
        return log_type_PARSDMM(arrs["set_feasibility"][:nf, :pp], arrs["r_dual"][:it], arrs["r_pri"][:it],
                                arrs["r_dual_total"][:it], arrs["r_pri_total"][:it], arrs["obj"][:it],
                                arrs["evol_x"][:it], arrs["rho"][:it], arrs["gamma"][:it], arrs["cg_it"][:it],
                                arrs["cg_relres"][:it],
                                dict(zip(TIMING_SECTIONS, [t * 1e-3 for t in lg.timing_ms])))

    def parsdmm(self, options: PARSDMM_options):
        o, lg, arrs = self._log_struct(options)
        _chk(lib().sipx_parsdmm(self.h, C.byref(o), C.byref(lg)))
        return self._log_result(lg, arrs), bool(lg.stopped_feasible)

    # the same native solve advanced in pieces (sipx_parsdmm_begin / sipx_parsdmm_steps)
    def parsdmm_begin(self, options: PARSDMM_options):
        self._run = self._log_struct(options)            # keeps the log arrays alive
        o, lg, arrs = self._run
        _chk(lib().sipx_parsdmm_begin(self.h, C.byref(o), C.byref(lg)))

    def parsdmm_steps(self, nsteps: int) -> bool:
        done = C.c_int()
        _chk(lib().sipx_parsdmm_steps(self.h, int(nsteps), C.byref(done)))
        return bool(done.value)

    def parsdmm_log(self):
        o, lg, arrs = self._run
        return self._log_result(lg, arrs)


def build_context(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, x=None, l=None, y=None, device=None,
                  owned=None, attach=None) -> Context:
    """Everything PARSDMM_initialize allocates (src/PARSDMM_initialize.jl:117-230), on the device.  `attach(ctx)` is called
    before sipx_finalize: where a rank of a sharded solve is given its communicator (sharded.attach_comm)."""
    TF = np.dtype(m.dtype).type
    if not (np.isrealobj(m) and (x is None or np.isrealobj(x))):
        raise SipxError("input for PARSDMM is not real")                 # src/PARSDMM.jl:50-52
    pp = len(P_sub)
    p = len(TD_OP)
    if (not options.feasibility_only and p != pp + 1) or (options.feasibility_only and p != pp):
        raise SipxError("TD_OP must hold one operator per set plus the identity of the distance term "
                        "(output of PARSDMM_precompute_distribute)")
    ctx = Context(comp_grid, TF, device)
    try:
        for i in range(pp):
            A = AtA[i] if AtA is not None else None
            ctx.add_set(TD_OP[i], P_sub[i], set_Prop.ncvx[i], A, set_Prop.AtA_offsets[i] if A is not None else None)
        if owned is not None:
            ctx.set_owned(owned)
        if attach is not None:
            attach(ctx)
        ctx.set_q_mode(getattr(options, "Q_mode", "cds"))
        rho_ini = [float(TF(r)) for r in options.rho_ini]                # convert_options!.jl:6-15
        zero = bool(options.zero_ini_guess)                               # x, l, y are zero-filled then (PARSDMM_initialize.jl:304-313)
        feas0 = ctx.finalize(m, rho_ini, float(TF(options.gamma_ini)), options.feasibility_only,
                             zero, None if zero else x, None if zero else l, None if zero else y)
        ctx.feasibility_initial = feas0
    except Exception:
        ctx.close()
        raise
    return ctx


# Contexts kept between calls of PARSDMM (round 5).  The reference's callers wrap PARSDMM as a projector and call it again and again
# with the same sets on the same grid (examples/constrained_freq_FWI_simple.jl:468, examples/Constraint_examples_2D.jl:222-223,
# examples/Dykstra_parallel_vs_PARSDMM.jl:134); every such call of the reference allocates its 16 work vectors per set again
# (src/PARSDMM_initialize.jl:129-184).  Here a call looks its context up by (device, precision, grid, Q mode, set descriptors): a
# hit costs sipx_reset -- zero-fills, the upload of m, Q assembled again, the initial feasibility -- instead of sipx_create ...
# sipx_finalize.  Only descriptors without array arguments are cached (scalar bounds, radii, ranks ...: what their bytes are is what
# they are); a list with bound vectors, a subspace basis, a caller-supplied operator or explicit A'A bands builds a context per call
# as before.  SIPX_CONTEXT_CACHE=0 switches the cache off, SIPX_CONTEXT_CACHE=k keeps k contexts (default 2: each holds the whole
# device state of its problem); clear_context_cache() frees them.
_ctx_cache: "dict" = {}


_extra_caches: "list" = []           # (multilevel.py keeps the level contexts of its last call: cleared together with these)


def clear_context_cache():
    for ctx in list(_ctx_cache.values()):
        ctx.close()
    _ctx_cache.clear()
    for f in _extra_caches:
        f()


def _cache_limit() -> int:
    e = os.environ.get("SIPX_CONTEXT_CACHE")
    if e is None or e == "":
        return 2
    try:
        return max(0, int(e))
    except ValueError:
        return 2


def _context_key(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, device):
    """None when this problem must not be cached (array arguments inside a descriptor, explicit A'A bands, sharded set ownership)."""
    n, h = _grid(comp_grid)
    sig = []
    for i, P in enumerate(P_sub):
        if not isinstance(P, Projector) or P.lb is not None or P.ub is not None or P.basis is not None:
            return None
        op = TD_OP[i]
        if not isinstance(op, TDOperator) or op.kind == "custom" or (AtA is not None and AtA[i] is not None):
            return None
        sig.append((op.kind, int(getattr(op, "component", 0)), P.kind, P.pmin, P.pmax, P.mode, P.dir, P.transform, bool(set_Prop.ncvx[i])))
    dev = _default_device if device is None else device
    # (the engine reads its A/B switches when a context is built: a context built under other switches is another context)
    env = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("SIPX_")))
    return (dev, np.dtype(m.dtype).str, tuple(n), tuple(h), bool(options.feasibility_only), getattr(options, "Q_mode", "cds"), tuple(sig), env)


def PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, x=None, l=None, y=None, device=None, outputs="all"):
    """Drop-in for src/PARSDMM.jl:25-258 (serial path): returns (x, log_PARSDMM, l, y).
    outputs="x": l and y are not copied back (returned as None) -- a caller that uses PARSDMM as a projector reads x only
    (examples/constrained_freq_FWI_simple.jl:468 keeps `[1]` of the result)."""
    import time
    t0 = time.perf_counter()
    if outputs not in ("all", "x"):
        raise SipxError("outputs must be 'all' or 'x'")
    if not (np.isrealobj(m) and (x is None or np.isrealobj(x))):
        raise SipxError("input for PARSDMM is not real")                 # src/PARSDMM.jl:50-52
    limit = _cache_limit()
    key = _context_key(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, device) if limit > 0 else None
    ctx = _ctx_cache.pop(key, None) if key is not None else None
    reused = ctx is not None
    try:
        if reused:
            TF = np.dtype(m.dtype).type
            zero = bool(options.zero_ini_guess)
            rho_ini = [float(TF(r)) for r in options.rho_ini]
            ctx.reset(m, rho_ini, float(TF(options.gamma_ini)), zero, None if zero else x, None if zero else l, None if zero else y)
        else:
            ctx = build_context(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, x, l, y, device)
        t_init = time.perf_counter() - t0
        log, _ = ctx.parsdmm(options)
        log.timing[TIMING_SECTIONS[0]] = t_init          # "initialization": PARSDMM_initialize (src/PARSDMM.jl:40)
        xo, lo, yo = ctx.download(want_ly=(outputs == "all"), x_out=x)
    except Exception:
        if ctx is not None:
            ctx.close()
        raise
    if key is not None:
        while len(_ctx_cache) >= limit:                 # the oldest entry goes (dicts keep insertion order)
            _ctx_cache.pop(next(iter(_ctx_cache))).close()
        _ctx_cache[key] = ctx
    else:
        ctx.close()
    log.context_reused = reused
    if x is not None and xo is not x and len(x) == len(xo):
        x[:] = xo                         # the reference overwrites the x argument in place
        xo = x
    return xo, log, lo, yo


# --------------------------------------------------------------------------------------------------
# kernel-level helpers (parity tests / bench)
# --------------------------------------------------------------------------------------------------
def cds_spmv(R, offsets, x, device=None):
    """y = A x for a CDS matrix (fill! + CDS_MVp_MT, src/argmin_x.jl:72-78)."""
    device = _default_device if device is None else device
    TF = x.dtype.type
    R = np.asfortranarray(R, dtype=TF)
    off = np.ascontiguousarray(offsets, np.int64)
    x = np.ascontiguousarray(x)
    y = np.empty_like(x)
    _chk(lib().sipx_cds_spmv(_dtype_code(TF), C.c_int64(R.shape[0]), int(R.shape[1]), _ptr(R),
                             off.ctypes.data_as(C.c_void_p), _ptr(x), _ptr(y), device))
    return y


def prox_l2s(x, rho, m, device=None):
    """x = (x*rho + m) / (rho + 1.0) in place (src/prox_l2s!.jl:3-6), on the device."""
    device = _default_device if device is None else device
    TF = x.dtype.type
    if not x.flags.c_contiguous or m.dtype != x.dtype or m.shape != x.shape:
        raise SipxError("prox_l2s: x and m must be contiguous vectors of one precision and length")
    m = np.ascontiguousarray(m)
    _chk(lib().sipx_prox_l2s(_dtype_code(TF), C.c_int64(x.size), _ptr(x), C.c_double(float(rho)), _ptr(m), device))
    return x


def resample_nn(a, nc, nf, device=None):
    """Nearest-neighbour grid transfer (Interpolations.BSpline(Constant()) on range(1, stop=nc, length=nf)), on the device."""
    device = _default_device if device is None else device
    a = np.ascontiguousarray(a)
    TF = a.dtype.type
    nc, nf = [int(v) for v in nc], [int(v) for v in nf]
    if a.size != int(np.prod(nc)):
        raise SipxError("resample: array size does not match its shape")
    out = np.empty(int(np.prod(nf)), TF)
    _chk(lib().sipx_resample_nn(_dtype_code(TF), len(nc), (C.c_int64 * len(nc))(*nc), (C.c_int64 * len(nf))(*nf),
                                _ptr(a), _ptr(out), device))
    return out


def CDS_MVp(N, ndiags, R, offset, x, y):
    """Reference signature (src/CDS_MVp.jl:9-28): y += A x."""
    y += cds_spmv(R, offset, x)
    return y
