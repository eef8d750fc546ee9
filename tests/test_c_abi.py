"""The C ABI driven from a plain C host program (tests/c_abi/smoke.c): include/sipx.h must be valid C99, and the library must
be usable without Python or torch in the process -- what the reference-side ccall binding relies on."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "setintersectionprojection.jl_amd")


def _build(tmp_path, src="smoke.c"):
    exe = str(tmp_path / ("sipx_" + src[:-2]))
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", src), "-o", exe, "-L", PKG, "-lsipx", "-lm",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_c99_and_the_library_links_from_c(tmp_path):
    if not os.path.exists(os.path.join(PKG, "libsipx.so")):
        pytest.skip("libsipx.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    _build(tmp_path)
    _build(tmp_path, "phases.c")


def test_c_host_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not os.path.exists(os.path.join(PKG, "libsipx.so")):
        pytest.skip("libsipx.so not built")
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_host_program_runs_the_solver(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "error path:" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_phase_level_sequence_from_c_equals_the_whole_solve(tmp_path, dtype):
    """tests/c_abi/phases.c: rhs_compose -> argmin_x -> update_y_l -> log_scalars -> adapt_rho_gamma -> q_update with the
    scalar rules on the host (what a Julia shim keeping PARSDMM.jl's loop ccalls) == sipx_parsdmm, bit for bit."""
    r = subprocess.run([_build(tmp_path, "phases.c"), dtype], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit for bit" in r.stdout and "OK" in r.stdout
