import sys, numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package
sipx = load_package()
from oracle import parsdmm_oracle as O
TF = np.float64
rng = np.random.default_rng(31)
M = 8
v = (rng.uniform(2.0, 3.0, M) * rng.choice([-1.0, 1.0], M)).astype(TF)
b = 0.8 * float(np.abs(v).sum())
want = O.project_l1_Duchi(v.copy(), TF(b))
c = sipx.set_definitions("l1", "identity", 0.0, b, ("matrix", ""))
got = sipx.host.Projector(c, sipx.compgrid((1.0, 1.0), (M, 1)), TF)(v.copy())
print("v", v); print("want", want); print("got ", got)
print("theta std", (np.abs(v).sum() - b) / M, "cap", (np.abs(v).sum() - np.abs(v).min() - b) / (M - 1))
