"""bench.py with more than one rank, rehearsed on CPU (the HIP engine cannot run here): `--dry-comm` replays the collective
skeleton of the PARSDMM iteration of both decompositions over gloo, so that the launcher (plain `--gpus N` and
torch.distributed.run), the rendezvous, the ONE-JSON-line stdout contract, the both-decompositions flow, the comm probe and
the watchdog have run with world > 1 before the first real multi-GPU run does."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _one_line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    assert len(lines[0]) < 4096, len(lines[0])        # what the driver keeps of stdout is a few KB: the line must fit whole
    return json.loads(lines[0])


def _check(d, world):
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["dry_comm"] is True and d["invalid_as_measurement"] is True and "dry-comm" in d["config"]["workload"]
    assert set(d["decompositions"]) == {"slab", "sets"} and d["decomposition"] == "slab"
    assert d["value"] == d["decompositions"]["slab"]["value"]                    # the headline names one decomposition, always
    assert d["faster_decomposition"] in ("slab", "sets")
    for v in d["decompositions"].values():
        assert v["value"] > 0
    assert d["comm"]["rccl_nranks"] == world
    probe = d["comm_probe_us"]
    assert len([k for k, v in probe.items() if isinstance(v, float)]) >= 8, probe


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 3])
def test_bench_spawns_its_ranks_dry(world):
    env = dict(os.environ, SIPX_DRY_COMM_CPU="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--dry-comm", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(_one_line(r.stdout), world)


@pytest.mark.timeout(240)
def test_bench_under_torch_distributed_run_dry():
    """The driver's launch line for N > 1."""
    env = dict(os.environ, SIPX_DRY_COMM_CPU="1")
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--dry-comm", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(_one_line(r.stdout), 2)


@pytest.mark.timeout(120)
def test_watchdog_names_the_stalled_rank():
    env = dict(os.environ, SIPX_DRY_COMM_CPU="1", SIPX_BENCH_TEST_HANG="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-comm", "--steps", "3", "--warmup", "1", "--deadline", "6"],
                       capture_output=True, text=True, timeout=100, env=env)
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                               # no half-written bench line
    assert "bench watchdog" in r.stderr and "rank 1: " in r.stderr and "this rank stalls here" in r.stderr


def test_iteration_skeletons_match_the_design():
    """DESIGN 5's collective counts per PARSDMM iteration of the headline list at 2 CG iterations."""
    sys.path.insert(0, ROOT)
    import bench
    for world in (2, 8):
        slab = bench.iteration_skeleton("slab", 2, world)
        sets = bench.iteration_skeleton("sets", 2, world)
        assert sum(1 for op, a in sets if a == "N") == 2 and sum(1 for op, a in slab if a == "N") == 0
        assert len(slab) == bench.SLAB_SMALL_COLLECTIVES_AT_2_CG


def test_headline_fits_what_the_driver_keeps():
    """Round 3's bench line grew to 24 KB and the driver, which keeps a few KB of stdout, could not parse it.  The headline
    built from that very record (profiles/r03_bench_default.json) and from the fattest N > 1 records must stay under 4 KB and
    keep the contract's keys plus the figures a review reads first; a record fatter still sheds its optional objects."""
    sys.path.insert(0, ROOT)
    import bench
    for name in ("r03_bench_default.json", "r03_bench_2ranks_share_one_gpu_rehearsal.json", "r03_bench_4ranks_share_one_gpu_rehearsal.json"):
        out = json.load(open(os.path.join(ROOT, "profiles", name)))
        line = bench.headline(out, "bench_detail.json")
        assert len(line) < 3072, (name, len(line))
        h = json.loads(line)
        for k in CONTRACT + ("dominant_kernel", "iteration_roofline", "c3_512", "c4_512", "c5", "comm"):
            assert k in h, (name, k)
        assert {"kernel", "bound", "peak", "achieved", "unit", "frac", "traffic", "launches", "avg_launch_ms"} <= set(h["roofline"])
        assert abs(h["value"] - out["value"]) <= 1e-4 * out["value"] and h["detail"] == "bench_detail.json"
        if out["n_gpus"] == 1:
            assert {"value", "unit", "cores", "kind", "sample"} <= set(h["cpu_baseline"])
        else:
            assert set(h["decompositions"]) == {"slab", "sets"} and len(h["comm_probe_us"]) >= 8
        fat = dict(out, comm_probe_us={f"probe_{k}_us (padding)": float(k) for k in range(400)})
        line = bench.headline(fat, None)
        assert len(line) < bench.LINE_LIMIT and json.loads(line)["value"] > 0 and "roofline" in json.loads(line)
