"""world_size-2 gloo tests of the sharded dense-grid path (CPU).  (-m "not gpu")

The per-shard compute is injected: here it is a deterministic function of the flat grid
index (and, in one test, the oracle's MLP on a tiny grid), so that what is tested is the
partition, the padding, the all_gather and the reassembly into ``.dist`` order."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from disn_amd import parallel as par


def test_shard_range_properties():
    for total in (1, 7, 125, 274625, 16974593):
        for world in (1, 2, 3, 8):
            edges = [par.shard_range(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
                assert a1 == b0 and a1 >= a0
            sizes = par.shard_sizes(total, world)
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        par.shard_range(10, 2, 2)


def _worker(rank, world, port, total, n_images, q, exchange="all_gather"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def query_fn(b, k0, k1):
            calls.append((b, k0, k1))
            k = torch.arange(k0, k1, dtype=torch.float64)
            return (torch.sin(k * 0.37 + b) * 3.0).to(torch.float32)

        out = par.sharded_grid(query_fn, n_images, total, torch.device("cpu"), exchange=exchange)
        if exchange == "all_to_all":
            q.put((rank, out[0].numpy(), calls, out[1]))
        else:
            q.put((rank, out.numpy(), calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("total,n_images", [(125, 1), (4913, 3)])
def test_sharded_grid_gloo_world2(total, n_images):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    k = np.arange(total, dtype=np.float64)
    expect = np.stack([(np.sin(k * 0.37 + b) * 3.0).astype(np.float32) for b in range(n_images)])
    for rank, out, calls in results:
        assert out.shape == (n_images, total)
        assert np.array_equal(out, expect), "rank %d: sharded result differs from the single-rank order" % rank
        k0, k1 = par.shard_range(total, world, rank)
        assert calls == [(b, k0, k1) for b in range(n_images)]   # every rank touched only its slice


@pytest.mark.timeout(300)
@pytest.mark.parametrize("total,n_images", [(125, 1), (4913, 3), (1000, 4)])
def test_sharded_grid_all_to_all_gloo_world2(total, n_images):
    """exchange="all_to_all": rank r ends with the FULL grids of the images it owns (b % world == r) and nothing else;
    an odd image count gives the ranks different receive sizes"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + total + 333) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n_images, q, "all_to_all")) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    k = np.arange(total, dtype=np.float64)
    seen = []
    for rank, out, calls, own in results:
        assert own == par.owned_images(n_images, world, rank) and out.shape == (len(own), total)
        for i, b in enumerate(own):
            assert np.array_equal(out[i], (np.sin(k * 0.37 + b) * 3.0).astype(np.float32)), (rank, b)
        k0, k1 = par.shard_range(total, world, rank)
        assert calls == [(b, k0, k1) for b in range(n_images)]   # every rank still evaluates its slice of EVERY image
        seen += own
    assert sorted(seen) == list(range(n_images))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange", ["all_gather", "all_to_all"])
def test_sharded_grid_gloo_world8_config4_shape(exchange):
    """BASELINE config 4's launch shape on CPU: 8 ranks, 8 images, a grid whose point count leaves the same remainder
    mod 8 as 257^3 (65^3 = 274 625 = 8 * 34 328 + 1: one rank's slice is a point longer) -- both exchanges"""
    world, n_images, total = 8, 8, 65 ** 3
    assert total % world == (257 ** 3) % world == 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 777 + (11 if exchange == "all_to_all" else 0)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n_images, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    k = np.arange(total, dtype=np.float64)
    expect = np.stack([(np.sin(k * 0.37 + b) * 3.0).astype(np.float32) for b in range(n_images)])
    sizes = par.shard_sizes(total, world)
    assert sum(sizes) == total and sorted(set(sizes)) == [34328, 34329]
    seen = []
    for res in results:
        rank, out, calls = res[0], res[1], res[2]
        k0, k1 = par.shard_range(total, world, rank)
        assert calls == [(b, k0, k1) for b in range(n_images)]
        if exchange == "all_to_all":
            own = res[3]
            assert own == par.owned_images(n_images, world, rank) == [rank]   # one image per GPU: the rank that meshes it
            assert np.array_equal(out, expect[own])
            seen += own
        else:
            assert np.array_equal(out, expect), "rank %d" % rank
    if exchange == "all_to_all":
        assert sorted(seen) == list(range(n_images))


def test_single_process_equals_unsharded():
    total = 343

    def query_fn(b, k0, k1):
        return torch.arange(k0, k1, dtype=torch.float32) + 1000 * b

    out = par.sharded_grid(query_fn, 2, total, torch.device("cpu"))
    assert torch.equal(out[1], torch.arange(total, dtype=torch.float32) + 1000)
    out2, own = par.sharded_grid(query_fn, 2, total, torch.device("cpu"), exchange="all_to_all")
    assert own == [0, 1] and torch.equal(out2, out)
    with pytest.raises(ValueError):
        par.sharded_grid(query_fn, 2, total, torch.device("cpu"), exchange="ring")


# ---------------------------------------------------------------- data-parallel training ----------
def test_shard_batch():
    assert [par.shard_batch(64, 8, r) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        par.shard_batch(20, 8, 0)


def _ddp_worker(rank, world, port, q):
    """Two ranks, each with the gradient of its half of a quadratic loss: after the bucketed exchange
    plus the 1/world scale inside Adam, both ranks hold the same parameters as a single process that
    saw the whole batch (oracle/train_oracle.adam_step is the optimizer reference)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import train_oracle as T
        n, head = 1000, 300
        rng = np.random.default_rng(5)
        w0 = rng.standard_normal(n)
        targets = rng.standard_normal((world, n))            # rank r fits targets[r]
        w, m, v = w0.copy(), np.zeros(n), np.zeros(n)
        red = par.GradientReducer(head, use_side_stream=False)
        assert red.world == world
        for t in range(1, 4):
            g = torch.from_numpy(w - targets[rank])            # d/dw 0.5*|w - target_r|^2
            red.start_head(g)
            red.finish(g)
            w, m, v = T.adam_step(w, g.numpy() / world, m, v, t, 1e-2)
        q.put((rank, w))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradient_reducer_gloo_world2_matches_single_process():
    from oracle import train_oracle as T
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 77) % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 1000
    rng = np.random.default_rng(5)
    w = rng.standard_normal(n)
    targets = rng.standard_normal((world, n))
    m, v = np.zeros(n), np.zeros(n)
    for t in range(1, 4):
        g = (w[None] - targets).mean(0)                        # gradient of the mean over the global batch
        w, m, v = T.adam_step(w, g, m, v, t, 1e-2)
    assert np.array_equal(results[0], results[1])              # replicas stay bit-identical
    assert np.allclose(results[0], w, rtol=0, atol=1e-12)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("workload", ["grid", "train", "query"])
def test_bench_launch_path_dry_run_world_2(workload):
    """the argument path the driver would use for the scaling runs -- `python bench.py --gpus N
    --workload W` self-launching `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    ...`, RANK / WORLD_SIZE from the environment, rendezvous, barrier, max-over-ranks time, ONE contract line from rank
    0 -- exercised on the CPU over gloo with the compute left out (bench.py --dry-run); for the grid workload the ranks'
    flat-index slices must tile the 257^3 grid"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", workload, "--dry-run",
                        "--dist-backend", "gloo", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       timeout=500, cwd=root)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in line
    assert line["dry_run"] is True and line["value"] is None and line["n_gpus"] == 2 and line["steps"] == 3
    assert "workload" in line["config"] and "model" not in line["config"]
    if workload == "grid":
        assert line["grid_points"] == 257 ** 3 and line["slice_rank0"] == [0, (257 ** 3 + 1) // 2]
