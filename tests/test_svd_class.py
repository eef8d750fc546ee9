"""What a Float32 LAPACK SVD -- the arithmetic of the reference's rank projector, src/projectors/project_rank!.jl:26-45: svd() in TF --
leaves on a slice of BASELINE config 4's model, in the measure the engine's slice-rank projector accepts its Ritz pairs by
(csrc/ext_proj.hip, k_sub_residual): the backward error of a computed triplet on the slice itself.  CPU only (numpy's svd on a
Float32 array is LAPACK's sgesdd, the routine behind Julia's svd)."""
import numpy as np

ENGINE_EPS_BW = 2.0 ** -23          # ExtImpl::RankKnobs::eps_bw


def test_float32_svd_backward_error_class():
    """For theta = v'Gv (G = X'X) the residual rho = G v - theta v is orthogonal to v and (sigma, u = X v / sigma, v) is an EXACT
    singular triplet of X + E with E = -u rho' / sigma, ||E||_2 = ||rho|| / sigma: the backward error of the computed right vector.
    On a 512 x 512 slice (a constant plus white noise: one singular value 200 times the flat rest) the Float32 SVD's top 32 triplets
    carry ||E|| / ||X||_2 between 1e-8 (the dominant one) and several 1e-7; the engine accepts a pair at 2^-23 = 1.2e-7, i.e. at the median of
    what the reference's own arithmetic delivers, and the rank-32 projection built on such vectors is as far from the exact one as
    Float32 storage of the result is (1e-7 of the slice's norm)."""
    rng = np.random.default_rng(1)
    n, r = 512, 32
    X = (2700.0 + 150.0 * rng.standard_normal((n, n))).astype(np.float32)
    U, s, Vt = np.linalg.svd(X, full_matrices=False)                 # Float32: sgesdd
    Xd = X.astype(np.float64)
    U64, s64, Vt64 = np.linalg.svd(Xd, full_matrices=False)
    G = Xd.T @ Xd
    V = Vt.T.astype(np.float64)[:, :r]
    V /= np.linalg.norm(V, axis=0)
    th = np.einsum("ij,ij->j", V, G @ V)
    rho = np.linalg.norm(G @ V - V * th, axis=0)
    bw = rho / (s64[0] * np.sqrt(th))                               # ||E||_2 / ||X||_2 per triplet
    assert bw[0] < 1e-7                                              # the dominant triplet
    assert 5e-8 < np.median(bw[1:]) < 1e-6 and bw[1:].max() < 5e-6, (np.median(bw[1:]), bw[1:].max())
    # the engine's level is not tighter than the median and not looser than a few times the largest of the reference's arithmetic
    assert ENGINE_EPS_BW <= 1.2 * np.median(bw[1:]) and ENGINE_EPS_BW < bw[1:].max(), (np.median(bw[1:]), bw[1:].max())
    P32 = (U[:, :r] * s[:r]) @ Vt[:r]
    P64 = (U64[:, :r] * s64[:r]) @ Vt64[:r]
    assert np.linalg.norm(P32 - P64) < 1e-6 * np.linalg.norm(P64)
