"""Generates the committed golden fixtures (run once, here; the reference tree is not on the GPU box).

Input : a 128x128 crop of the reference's only data file, examples/Data/compass_velocity.mat
        (read as DATA with scipy.io.loadmat; cf. examples/projection_intersection_2D.jl:38-45),
        stored as c1_compass_128_m.npy (Float32, 64 KiB).
Output: what the CPU oracle (oracle/parsdmm_oracle.py) returns for BASELINE config 1
        -- 2-D 128x128 Float32, {bounds [1600, 3900] on I, l1-ball on TV with sigma = 0.5||TV m||_1},
        default options with maxit=500 (examples/projection_intersection_2D.jl:17-24) --
        stored as c1_compass_128_x.npy and c1_compass_128_log.json.
The reference itself cannot be run (no Julia toolchain), so these are ORACLE outputs pinned by the
reference's known-answer tests (tests/test_oracle_pins.py), not outputs of the Julia package."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import parsdmm_oracle as O  # noqa: E402


def c1_problem(m, mod, TF=np.float32):
    n, h = (128, 128), (25.0, 6.0)
    g = mod.compgrid(h, n)
    TV = O.get_TD_operator(O.compgrid(h, n), "TV", TF)[0]
    c = [mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
         mod.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(TV @ m).sum()), ("matrix", ""))]
    opt = mod.PARSDMM_options(FL=TF, maxit=500)
    P, A, prop = mod.setup_constraints(c, g, TF)
    A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
    return g, opt, P, A, prop, AtA


if __name__ == "__main__":
    import scipy.io as sio
    D = sio.loadmat("/root/reference/examples/Data/compass_velocity.mat")["Data"]
    m2 = D[100:228, 700:828].T                      # (x, z) like permutedims(m,[2,1]) in the example
    m = np.ascontiguousarray(m2.reshape(-1, order="F").astype(np.float32))
    np.save(os.path.join(HERE, "c1_compass_128_m.npy"), m)
    g, opt, P, A, prop, AtA = c1_problem(m, O)
    x, log, l, y = O.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    np.save(os.path.join(HERE, "c1_compass_128_x.npy"), x)
    json.dump({k: np.asarray(getattr(log, k), np.float64).tolist() for k in
               ("obj", "evol_x", "r_pri_total", "r_dual_total", "cg_it", "cg_relres", "rho", "gamma", "set_feasibility")},
              open(os.path.join(HERE, "c1_compass_128_log.json"), "w"))
    print("iterations", len(log.obj), "m range", m.min(), m.max(), "x range", x.min(), x.max())
