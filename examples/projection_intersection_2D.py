"""The reference's examples/projection_intersection_2D.jl on the MI355X engine: set up constraints, project a 2-D velocity
model onto their intersection, look at the log.  Line for line the same calls as the Julia script (:17-24, :55-81), with the
package loaded as `sipx`; the model is the 128x128 crop of the compass velocity model kept under tests/golden/.

    python examples/projection_intersection_2D.py            (needs the built library and a GPU)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

sipx = load_package()

# PARSDMM options (projection_intersection_2D.jl:17-24)
options = sipx.PARSDMM_options()
options.FL = np.float32
options.adjust_gamma = True
options.adjust_rho = True
options.adjust_feasibility_rho = True
options.Blas_active = True
options.maxit = 500
TF = options.FL

# model to project and computational grid (:38-46): 25 m and 6 m between grid points
m = np.load(os.path.join(ROOT, "tests", "golden", "c1_compass_128_m.npy"))
comp_grid = sipx.compgrid((TF(25.0), TF(6.0)), (128, 128))

# constraints (:55-74): bounds on the velocity and on its vertical derivative
constraint = [
    sipx.set_definitions("bounds", "identity", 1480.0, 4500.0, ("matrix", "")),
    sipx.set_definitions("bounds", "D_z", 0.0, 1e6, ("matrix", "")),
]

options.parallel = False
P_sub, TD_OP, set_Prop = sipx.setup_constraints(constraint, comp_grid, options.FL)
TD_OP, AtA, l, y = sipx.PARSDMM_precompute_distribute(TD_OP, set_Prop, comp_grid, options)

print("\nPARSDMM serial (bounds and bounds on D_z):")
for _ in range(3):                                  # the reference times three calls (:79-81)
    t0 = time.perf_counter()
    x1, log_PARSDMM, _, _ = sipx.PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options)
    print("  %.3f s, %d iterations" % (time.perf_counter() - t0, len(log_PARSDMM.obj)))

Dz = sipx.get_TD_operator(comp_grid, "D_z", TF)[0]
print("  min(x) = %.1f, max(x) = %.1f, min(D_z x) = %.3e   (model: min(D_z m) = %.3e)" %
      (x1.min(), x1.max(), (Dz @ x1).min(), (Dz @ m).min()))
print("  final objective 1/2||m - x||^2 = %.4e, set feasibility %s" %
      (log_PARSDMM.obj[-1], np.array2string(log_PARSDMM.set_feasibility[-2], precision=2)))
print("  timing (s):", {k: round(v, 4) for k, v in log_PARSDMM.timing.items()})
