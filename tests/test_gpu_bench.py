"""bench.py end to end on the GPU: the one-line JSON contract at N=1, and the
multi-rank path (partition by global sample number, barrier, one reduce per
step, max-over-ranks timing) rehearsed with two ranks that share the one GPU of
the test box over gloo.  With RCCL the same code runs one rank per GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SMALL = ["--size", "160", "120", "--spp", "8", "--steps", "2", "--warmup", "1"]


def run(cmd, env=None):
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def check_contract(out, n_gpus):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == n_gpus and out["steps"] == 2 and out["warmup"] == 1
    assert out["unit"] == "Msamples/s" and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["value"] > 0 and out["ms_per_step"] > 0
    # value = whole-job samples / max-over-ranks time
    samples = 160 * 120 * 8 * n_gpus
    assert abs(out["value"] - samples / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * out["value"]
    r = out["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel_ms"] > 0


def test_bench_single_gpu_contract():
    out = run([sys.executable, "bench.py", "--cpu-spp", "1"] + SMALL)
    check_contract(out, 1)
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0
    assert out["other_numeric"]["numeric"] == "fast"
    assert 4.0 < out["segments_per_sample"] < 6.0


def test_bench_two_ranks_share_one_gpu_over_gloo():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2"] + SMALL,
              env={"PT_BENCH_BACKEND": "gloo"})
    check_contract(out, 2)
    assert "cpu_baseline" not in out
    assert "gloo" in out["config"]["parallelism"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started plainly, the way the round-end scaling run starts it:
    the script spawns its two ranks itself (gloo here: both share the test box's one GPU)."""
    out = run([sys.executable, "bench.py", "--gpus", "2"] + SMALL, env={"PT_BENCH_BACKEND": "gloo"})
    check_contract(out, 2)
    assert "gloo" in out["config"]["parallelism"]


def test_bench_makes_an_rccl_process_group_at_one_rank():
    """PT_BENCH_FORCE_PG=1: the "nccl" (RCCL) process group is initialised at world size 1 and
    the per-step reduce and the barriers run through it - the code the multi-GPU launch uses,
    executed once on the one-GPU box."""
    out = run([sys.executable, "bench.py", "--cpu-spp", "0"] + SMALL, env={"PT_BENCH_FORCE_PG": "1"})
    check_contract(out, 1)
    assert out["config"]["process_group"] == "nccl, world 1"


def test_bench_rank_failure_exits_non_zero():
    """A rank that raises must end with a non-zero code at once, so that the launcher tears the
    other ranks down: here a scene name that does not exist."""
    p = subprocess.run([sys.executable, "bench.py", "--scene", "no_such_scene", "--cpu-spp", "0"] + SMALL,
                       cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "Traceback" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_two_ranks_with_row_bands_over_gloo():
    """--partition rows: every rank renders all the step's samples of its own band of rows."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
               "--partition", "rows"] + SMALL, env={"PT_BENCH_BACKEND": "gloo"})
    check_contract(out, 2)
    assert "row bands" in out["config"]["parallelism"]
