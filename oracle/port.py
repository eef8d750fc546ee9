"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of the C/OpenMP port (oracle/parsdmm_port.c).

Used by bench.py's cpu_baseline leg (kind "port") and by tests/test_port.py, which
cross-checks the port against the numpy oracle.  Never imported by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = {"identity": 0, "D_x": 1, "D_y": 2, "D_z": 3, "TV": 4}
PROJ = {"bounds": 0, "l1": 2}


def _lib(TF):
    name = "libparsdmm_port_f32.so" if np.dtype(TF) == np.float32 else "libparsdmm_port_f64.so"
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    return C.CDLL(path)


def run(n, h, sets, m, maxit, evol_rel_tol=1e-3, feas_tol=5e-2, obj_tol=1e-3, rho_ini=10.0, gamma_ini=1.0, freq=2,
        adjust_rho=True, adjust_gamma=True, adjust_feasibility_rho=True, nthreads=0):
    """sets: list of (set_type, TD_OP, min, max).  Returns dict(x, obj, r_pri_total, cg_it, rho, gamma,
    set_feasibility, evol_x, loop_seconds)."""
    TF = m.dtype.type
    lib = _lib(TF)
    # never the OpenMP default: a GPU box shows every hardware thread of its host behind a CPU quota of 16, and a team that large
    # spinning at every barrier of a small problem does not finish
    nthreads = int(nthreads) or host_threads()
    pp, p = len(sets), len(sets) + 1
    nd = len(n)
    na = (C.c_int64 * 3)(*(list(n) + [1] * (3 - nd)))
    ha = (C.c_double * 3)(*(list(map(float, h)) + [1.0] * (3 - nd)))
    ops = (C.c_int * pp)(*[OPS[s[1]] for s in sets])
    prj = (C.c_int * pp)(*[PROJ[s[0]] for s in sets])
    pmin = (C.c_double * pp)(*[float(TF(s[2])) for s in sets])
    pmax = (C.c_double * pp)(*[float(TF(s[3])) for s in sets])
    N = int(np.prod(n))
    x = np.zeros(N, TF)
    obj, rpt, evol = np.zeros(maxit), np.zeros(maxit), np.zeros(maxit)
    cg = np.zeros(maxit, np.int64)
    rho, gam, feas = np.zeros((maxit, p)), np.zeros((maxit, p)), np.zeros((maxit, max(pp, 1)))
    nit, nf, secs = C.c_int(), C.c_int(), C.c_double()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    m = np.ascontiguousarray(m)
    rc = lib.parsdmm_port(nd, na, ha, pp, ops, prj, pmin, pmax, P(m), int(maxit), C.c_double(evol_rel_tol),
                          C.c_double(feas_tol), C.c_double(obj_tol), C.c_double(rho_ini), C.c_double(gamma_ini),
                          int(freq), int(adjust_rho), int(adjust_gamma), int(adjust_feasibility_rho), int(nthreads),
                          P(x), P(obj), P(rpt), P(cg), P(rho), P(gam), P(feas), P(evol), C.byref(nit), C.byref(nf),
                          C.byref(secs))
    assert rc == 0
    it = nit.value
    return dict(x=x, obj=obj[:it], r_pri_total=rpt[:it], cg_it=cg[:it], rho=rho[:it], gamma=gam[:it],
                set_feasibility=feas[:nf.value, :pp], evol_x=evol[:it], loop_seconds=secs.value, n_iter=it)


def host_threads():
    """Threads for the CPU baseline: the affinity mask, capped by the cgroup CPU quota and by 16 (the
    CPU share of a one-GPU box); SIPX_CPU_THREADS overrides."""
    if os.environ.get("SIPX_CPU_THREADS"):
        return int(os.environ["SIPX_CPU_THREADS"])
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def time_baseline(name, n, h, kinds, budget_s=20.0):
    """cpu_baseline leg of bench.py: the same workload (grid, sets, synthetic model, zero tolerances)
    for as many iterations as fit the budget; reports iterations/sec of the iteration loop."""
    import bench
    TF = np.float32
    m = bench.synthetic_model(n, TF, 20240601 + 3)
    nthreads = host_threads()
    lib = _lib(TF)
    # radii sigma = 0.5 ||A m||_1 from plain numpy differences (setup, not timed)
    M3 = m.reshape(n, order="F").astype(np.float64)
    sets = []
    for k in kinds:
        if k == "bounds":
            sets.append(("bounds", "identity", 1600.0, 3900.0))
        else:
            op = k[3:]
            axes = {"D_x": [0], "D_y": [1], "D_z": [len(n) - 1], "TV": list(range(len(n)))}[op]
            s = sum(np.abs(np.diff(M3, axis=a)).sum() / h[a] for a in axes)
            sets.append(("l1", op, 0.0, 0.5 * s))
    # one bounded run (operator / AtA setup is not timed): about 10-30 s of iteration-loop work
    iters = int(min(25, max(3, round(budget_s * 2.1e7 / int(np.prod(n))))))     # 256^3: the 25 iterations the GPU run covers (5 warm-up + 20 timed), ~10 s
    r = run(n, h, sets, m, iters, 0.0, 0.0, 0.0, nthreads=nthreads)
    return {"value": r["n_iter"] / r["loop_seconds"], "unit": "it/s", "cores": nthreads, "kind": "port",
            "sample": f"same workload ({name}: {'x'.join(map(str, n))} f32, same sets and model), first {r['n_iter']} "
                      f"PARSDMM iterations ({int(r['cg_it'].sum())} CG its) in {r['loop_seconds']:.1f} s; "
                      "C/OpenMP port keeping the reference's structure (per-diagonal CDS SpMV, un-fused CG, "
                      "explicit sparse TD_OP products, sort-based l1 projection)"}
