"""bench.py as the driver runs it: the JSON contract of the default line, and the multi-rank launch path
(`python bench.py --gpus 2` spawns two ranks itself; on a one-GPU box the ranks share the device, which
exercises the launch / sharding / collective / gather plumbing, not scaling)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_default_line_contract():
    d = _run(["--steps", "5", "--warmup", "2", "--no-extras"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["unit"] == "points/s" and d["value"] > 1e5
    assert "workload" in d["config"]
    # the timed forms' accuracy rides on the line: the committed 48-set sweep's worst for the forms that ran
    assert d["accuracy"]["within_bar"] is True and d["accuracy"]["sweep_worst_of_the_timed_forms"] <= d["accuracy"]["bar"] == 1e-5


@pytest.mark.parametrize("exchange", ["all_to_all", "all_gather"])
def test_grid_workload_two_ranks_with_a_collective(exchange):
    """config-4 shape on a small grid: 2 ranks, 2 images, 33^3 points each, sharded + one exchange + MC.  all_to_all
    (default): every rank receives the full grid of the ONE image it meshes; all_gather: both grids"""
    d = _run(["--gpus", "2", "--workload", "grid", "--grid-res", "32", "--grid-images", "2", "--steps", "2",
              "--warmup", "1", "--exchange", exchange])
    assert d["n_gpus"] == 2
    got = (d["all_gather"] or {}).get("bytes_received_per_rank", 0)
    want_images = 1 if exchange == "all_to_all" else 2
    assert d["all_gather"]["exchange"] == exchange and want_images * 33 ** 3 * 4 <= got < (want_images + 0.5) * 33 ** 3 * 4
    assert d["config"]["images"] == 2 and d["value"] > 0
    assert d["mesh"]["images_meshed_on_rank0"] == 1


def test_query_workload_two_ranks_are_replicas():
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras"])
    assert d["n_gpus"] == 2 and d["config"]["points_per_step_per_gpu"] == 2048
