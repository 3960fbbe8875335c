"""bench.py's multi-rank path, as far as one GPU allows (round 4): the RCCL branch with ONE rank under torch.distributed.run, and
`python bench.py --gpus 2` starting its ranks itself (both ranks on cuda:0, collectives over gloo: BENCH_SAME_DEVICE / BENCH_BACKEND
are the rehearsal switches of bench.py, never set by the driver).  The driver's 1-GPU invocation itself is covered too: its line
must carry `roofline`, `cpu_baseline`-less when asked, and the placement report."""
import json
import math
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SMALL = ["--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--envs", "8", "--particles", "200000", "--mesh", "128",
         "--steady-steps", "20", "--probe-passes", "3"]


def _line(cmd, extra_env=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]                # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def test_rccl_branch_with_one_rank():
    """torch.distributed.run --nproc-per-node 1: process group "nccl" (= RCCL) on cuda:0, the all-gather of returns and the
    max-over-ranks all-reduce really run through it."""
    out = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                 "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL)
    assert out["collective_backend"] == "nccl" and out["n_gpus"] == 1
    assert len(out["returns_all_gather_ms"]) == 1 and math.isfinite(out["returns_all_gather_ms"][0])
    assert out["returns_all_gather_ms"][0] >= 0.0 and len(out["per_rank_ms_per_step"]) == 1
    assert out["value"] > 0 and out["bad_positions"] == 0 and out["energy_drift"] < 1e-3
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["achieved"] > 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver runs its 1-GPU leg, with N = 1): the
    script starts the two ranks as child processes and relays rank 0's line; whole-job value = both ranks' environments."""
    out = _line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                {"BENCH_SAME_DEVICE": "1", "BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["collective_backend"] == "gloo"
    assert len(out["per_rank_ms_per_step"]) == 2 and len(out["returns_all_gather_ms"]) == 2
    assert all(math.isfinite(t) for t in out["returns_all_gather_ms"])
    per_rank = 8 * 200000 * 4
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 4 / (2 * per_rank) - 1) < 1e-9      # value = all ranks' particle-steps / time
    assert out["config"]["sharding"].startswith("2 x 8 envs")


def test_single_process_line_and_orders():
    """The driver's own form (no launcher, one GPU), in both orders of the regions: the default times the K steps last, behind
    cold_start / steady_state / the event pass; --order cold first."""
    warm = _line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL)
    cold = _line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--order", "cold"] + SMALL)
    assert warm["collective_backend"] is None and warm["order"][-1] == "timed" and cold["order"][1] == "timed"
    assert warm["cold_start"]["steps"] == 4 and warm["steady_state"]["steps"] == 20 and cold["cold_start"] is None
    for out in (warm, cold):
        assert out["placement"]["outcome"] in ("none", "found", "patience", "timeout", "memory")
        assert out["config"]["refresh"] == "every_step" and out["roofline"]["frac"] > 0
        assert "survey_112B_equivalent" in out and "algorithmic_frac_of_step" not in out
