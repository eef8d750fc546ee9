"""Golden fixtures of BASELINE config 1 (2-D 128x128 Float32, {bounds, l1 on TV}; real-data crop of
the reference's examples/Data/compass_velocity.mat).  CPU: the oracle still reproduces the committed
vectors.  GPU: the HIP engine matches them at the reference's own Float32 tolerance."""
import json
import os

import numpy as np
import pytest

from oracle import parsdmm_oracle as O
from tests.golden.make_golden import c1_problem

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    m = np.load(os.path.join(G, "c1_compass_128_m.npy"))
    x = np.load(os.path.join(G, "c1_compass_128_x.npy"))
    log = json.load(open(os.path.join(G, "c1_compass_128_log.json")))
    return m, x, log


def test_oracle_reproduces_golden():
    m, xg, lg = _load()
    g, opt, P, A, prop, AtA = c1_problem(m, O)
    x, log, l, y = O.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    assert np.array_equal(x, xg)
    assert np.array_equal(np.asarray(log.obj), np.asarray(lg["obj"]))
    assert np.array_equal(np.asarray(log.cg_it), np.asarray(lg["cg_it"]))
    # the projection is feasible (test/test_PARSDMM.jl:86-89)
    assert xg.min() >= 1600 - 1.5 * 0.05 * 1600 and xg.max() <= 3900 * (1 + 1.5 * 0.05)


@pytest.mark.gpu
def test_engine_matches_golden(sipx):
    m, xg, lg = _load()
    g, opt, P, A, prop, AtA = c1_problem(m, sipx)
    x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    K = min(6, len(log.obj), len(lg["obj"]))
    assert np.array_equal(log.cg_it[:K], np.asarray(lg["cg_it"])[:K])
    assert np.allclose(log.obj[:K], np.asarray(lg["obj"])[:K], rtol=5e-4)
    assert np.allclose(log.rho[:K], np.asarray(lg["rho"])[:K], rtol=5e-4)
    err = np.linalg.norm(x.astype(np.float64) - xg) / np.linalg.norm(xg)
    assert err < 5e-4, err                                   # test/test_PARSDMM_parallel.jl:72
    for i, (lo, hi) in enumerate([(1600.0, 3900.0)]):
        assert x.min() >= lo * (1 - 0.075) and x.max() <= hi * (1 + 0.075)
