#!/usr/bin/env python
"""Benchmark of the DISN SDF-query hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): SDF point queries / s on a 137x137 image with a 2048-point batch.
One STEP = one pass of the whole hot path over one batch of synthetic input, nothing cached:
resize 137->224, VGG-16, five tap up-samples, projection, gather, both point MLPs and their sum
for 2048 random query points (BASELINE config 2: "1xMI355X: VGG-16 encode + 2048 random query
points, img_feat_twostream, fp32, random-init weights").  Inputs are resident in HBM before the
timed region.  With N>1 every rank runs the same per-GPU workload on its own image / points
(replicas: the 2048-point step has no exchange step), value = N*2048*K / max-over-ranks time.

Besides the contract line's fields the JSON carries
  roofline      -- the dominant kernel family (implicit-GEMM 3x3 conv, fp32 MFMA): algorithmic
                   FLOP of the 13 conv launches of one step / their summed duration measured with
                   events on the launch stream, against the 157.3 TFLOP/s fp32-MFMA peak;
  roofline_gather / roofline_mlp -- the same for the gather (HBM-bound, 29 440 B/point) and the
                   point MLP (fp32 MFMA, 3.67 MFLOP/point with the global block folded);
  query_only    -- encoder amortised (the >=1e7 pts/s target of north_star applies here);
  grid256       -- wall-clock of a full 257^3 dense-grid evaluation on this GPU (config 3, no MC);
  cpu_baseline  -- the oracle (numpy/torch-CPU restatement of the reference's algorithm as
                   written) timed on this box's host cores on the same workload (rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
GATHER_BYTES_PER_PT = 29440        # 4 taps x 1472 ch x 4 B read + 1472 x 4 B write
MLP_FLOP_PER_PT = 2 * 1835904      # 3.67 MFLOP/pt, global 1024x512 block folded per image
VGG_LAYERS = [(3, 64, 224), (64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56),
              (256, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 28), (512, 512, 14), (512, 512, 14),
              (512, 512, 14)]
N_POINTS = 2048


def ev_time_ms(fn, reps, torch):
    """average duration of fn() in ms, events recorded on the stream the kernels run on"""
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


def train_feed(torch, dev, B, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    feed = {"imgs": torch.rand((B, 137, 137, 3), device=dev, generator=g),
            "sample_pc": torch.rand((B, N_POINTS, 3), device=dev, generator=g) - 0.5,
            "trans_mat": torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                                        [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]] * B,
                                      dtype=torch.float32, device=dev),
            "sdf": 0.05 * torch.randn((B, N_POINTS, 1), device=dev, generator=g)}
    feed["sample_pc_rot"] = feed["sample_pc"].clone()
    return feed


def train_bench(args, torch, dist, dev, world, rank, launched):
    """timed region = K full optimizer steps (forward, get_loss, backward of every variable, gradient
    exchange, Adam) with the batch resident in HBM"""
    from disn_amd.train_sdf import Trainer
    from disn_amd.weights import WeightStore
    B = args.train_batch
    bf = args.train_dtype == "bf16"
    tr = Trainer(WeightStore.random_init(0, mode="he"), dev, batch_size=B * world, precision=args.train_dtype)
    feed = train_feed(torch, dev, B, 1000 + rank)
    for _ in range(args.warmup):
        tr.step(feed)
    torch.cuda.synchronize()
    if launched:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, losses, _ = tr.step(feed)
    torch.cuda.synchronize()
    if launched:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if launched:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(losses["overall_loss"])
    assert np.isfinite(loss)
    tr.close()
    return {"metric": "training samples/sec (train_sdf.py step: VGG-16 + two-stream SDF net, 2048 points/sample, "
                      "all variables trained, Adam)",
            "value": world * B * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 5 shape, %s: data-parallel training step, %d samples x %d "
                                   "points per GPU, random-init (he) weights" % (
                                       "bf16 multiply / fp32 accumulate, master weights and optimizer" if bf
                                       else "fp32 accuracy (the reference's precision; precision=%s)" % args.train_dtype,
                                       B, N_POINTS),
                       "global_batch": B * world, "points_per_sample": N_POINTS,
                       "parallelism": "dp%d (one sum all-reduce of the flat gradient buffer in two buckets, the "
                                      "fc+MLP bucket under the conv backward)" % world},
            "final_loss": loss}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / grid / cpu legs")
    ap.add_argument("--cpu-runs", type=int, default=5)
    ap.add_argument("--workload", choices=("query", "train"), default="query",
                    help="query: BASELINE.json metric (default); train: config-5 training step")
    ap.add_argument("--train-batch", type=int, default=8, help="images per GPU per training step")
    ap.add_argument("--train-dtype", choices=("f32", "f32_mfma", "bf16"), default="f32",
                    help="--workload train: f32 = the reference's precision (forward / data-gradient GEMMs as "
                         "a three-term bf16 split on the bf16 MFMA pipes, same error as f32_mfma = everything "
                         "on the f32-input MFMA); bf16 = mixed precision (bf16 multiply, fp32 accumulate / "
                         "master weights / optimizer)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # torch.distributed.run / torchrun
    if launched:       # also with one rank: same RCCL init / barrier / all-reduce sequence as N ranks
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; reporting n_gpus=%d" % (args.gpus, world, world), file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())

    from disn_amd import ops
    from disn_amd.engine import SdfEngine
    from disn_amd.weights import WeightStore

    if args.workload == "train":
        # BASELINE config 5 (not the north-star metric): data-parallel training step, B images x 2048
        # points per GPU, fp32, TF Adam; gradients exchanged by RCCL under the convolution backward
        line = train_bench(args, torch, dist, dev, world, rank, launched)
        if rank == 0:
            print(json.dumps(line))
        if launched:
            dist.barrier()
            dist.destroy_process_group()
        return

    store = WeightStore.random_init(0, mode="xavier")          # "random-init weights" (create_sdf.py:184-192)
    eng = SdfEngine(store, dev)
    rng = np.random.default_rng(1000 + rank)
    img = torch.from_numpy(rng.random((1, 137, 137, 3), dtype=np.float32)).to(dev)
    pts = torch.from_numpy((rng.random((1, N_POINTS, 3), dtype=np.float32) * 2 - 1).astype(np.float32)).to(dev)
    tm = torch.tensor([[[-68.453156, 5.5086656, -0.37556022], [-17.138561, -84.685486, -0.250198],
                        [-47.284092, -3.6569588, 0.2493176], [101.133705, 101.34268, 1.4305686]]],
                      dtype=torch.float32, device=dev)       # demo/demo.py:272-276

    def step():
        # rows A..H, every step, through the single overlapped entry (disn_encode_query)
        return eng.encode_query(img, pts, tm)[1]

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if launched:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if launched:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if launched:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(out).all())
    value = world * N_POINTS * args.steps / dt
    line = {
        "metric": "SDF point queries/sec (137x137 img, 2048-pt batch, VGG-16 encode + two-stream query per batch)",
        "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 2: VGG-16 encode + 2048 random query points, img_feat_twostream, "
                               "fp32, random-init (xavier) weights, nothing cached between steps",
                   "images_per_step_per_gpu": 1, "points_per_step_per_gpu": N_POINTS,
                   "parallelism": "replicas x%d (no data-path collective)" % world},
    }

    if rank == 0:
        print("[bench] main line: %.4g points/s, %.4f ms/step" % (value, line["ms_per_step"]), file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_extras:   # roofline / cpu legs: N=1 only (bench contract)
        # ---- roofline of the dominant kernel family: the 13 implicit-GEMM conv launches ----------
        # Each layer is timed through the kernel the step runs for it at B = 1: the three-term bf16
        # split ("x3": fp32-accurate, bf16 MFMA pipes) wherever api.hip selects it, else the
        # f32-input MFMA kernel (conv1_1, K = 27).
        layers, tot_ms, tot_flop = [], 0.0, 0.0
        x3_off = os.environ.get("DISN_X3", "1") == "0"
        for cin, cout, hw in VGG_LAYERS:
            x = torch.rand((1, hw, hw, cin), device=dev)
            wraw = torch.randn((9 * cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
            b = torch.zeros(cout, device=dev)
            use_x3 = cin != 3 and not x3_off
            if use_x3:
                w3 = ops.pack_kn_x3(wraw)
                wsb = torch.empty(max(ops.lib().disn_conv3x3_x3_workspace_bytes(1, hw, hw, cin, cout), 256),
                                  dtype=torch.uint8, device=dev)
                o = torch.empty((1, hw, hw, cout), device=dev)
                ms = ev_time_ms(lambda: ops.conv3x3_x3(x, w3, b, cout, True, wsb, o), 20, torch)
            else:
                w = ops.pack_kn(wraw)
                ms = ev_time_ms(lambda: ops.conv3x3(x, w, b, cout, True), 20, torch)
            fl = 2.0 * hw * hw * cout * 9 * cin
            layers.append({"cin": cin, "cout": cout, "hw": hw, "ms": round(ms, 5), "tflops": round(fl / ms / 1e9, 2),
                           "kernel": "gemm_bf16_mfma<*,*,CONV3,3>" if use_x3 else "gemm_f32_mfma<*,*,CONV3*>"})
            tot_ms += ms
            tot_flop += fl
        ach = tot_flop / tot_ms / 1e9
        # HBM-side bytes of the same 13 launches from the PMC passes of the last profiled round
        # (profiles/pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 fetch
        # correction, write counter calibrated on the gather's known output bytes); None if absent
        traffic, pmc = None, {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = pmc["conv_family_per_step"]["hbm_bytes"]
        except Exception:
            pass
        # fp32-equivalent ceiling of the three-term method: 6 bf16 MFMAs (2.5 PFLOP/s dense) per product block
        peak_x3 = 2500.0 / 6.0
        line["roofline"] = {"kernel": "13 conv launches of one VGG-16 forward, B=1: gemm_bf16_mfma<64,64,CONV3,3> "
                                      "(three-term bf16 split, fp32-accurate) for 12 layers, gemm_f32_mfma for conv1_1",
                            "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                            "peak_note": "peak = dense f32-input MFMA (157.3), the arithmetic type of the result; "
                                         "the three-term kernels issue bf16 MFMAs whose fp32-equivalent ceiling is "
                                         "2500/6 = 417 TFLOP/s",
                            "frac_of_three_term_ceiling": ach / peak_x3, "traffic": traffic,
                            "traffic_note": "memory-side bytes per step of the 13 conv launches + their split-K "
                                            "reduces (FETCH_SIZE x2 + calibrated WRITE_SIZE; L2 misses served by MALL "
                                            "count), PMC passes tools/gpu_pmc_traffic.sh -> profiles/pmc_traffic.json; "
                                            "algorithmic ~180 MB (three-plane bf16 weights 88 + inputs 36 + outputs 54)",
                            "flop_per_step": tot_flop, "ms_per_step": tot_ms, "layers": layers}
        # ---- gather (HBM bound) -----------------------------------------------------------------
        enc = eng.encode(img)
        g = {}
        for n in (N_POINTS, 262144):
            p = torch.rand((1, n, 3), device=dev) * 2 - 1
            xy = ops.project(p, tm)
            feat = torch.empty((1, n, 1472), device=dev)
            ms = ev_time_ms(lambda: ops.gather(enc.featmap, xy, feat), 20, torch)
            gbs = n * GATHER_BYTES_PER_PT / ms / 1e6
            g["n%d" % n] = {"ms": ms, "achieved": gbs, "frac": gbs / PEAK_HBM_GBS,
                            "traffic": (pmc.get("gather_n%d" % n) or {}).get("hbm_bytes"),
                            "algorithmic_bytes": n * GATHER_BYTES_PER_PT}
        line["roofline_gather"] = {"kernel": "gather_kernel", "bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "bytes_per_point": GATHER_BYTES_PER_PT, **g}
        # ---- point MLP + query-only (encoder amortised) ------------------------------------------
        q = {}
        for n in (N_POINTS, 65536, 262144):
            p = torch.rand((1, n, 3), device=dev) * 2 - 1
            from disn_amd.engine import FOLD_MIN_POINTS
            folded = n >= FOLD_MIN_POINTS       # engine default: local fold2/conv1 folded into the feature map
            eng.query(enc, p, tm)               # (builds the folded map once; it is per-image state)
            ms = ev_time_ms(lambda: eng.query(enc, p, tm), 10, torch)
            flop = MLP_FLOP_PER_PT - (2 * 1472 * 512 if folded else 0)
            q["n%d" % n] = {"ms": ms, "points_per_s": n / ms * 1e3, "folded_local_stream": folded,
                            "executed_flop_per_point": flop,
                            "mlp_tflops_lower_bound": n * flop / ms / 1e9}
        line["query_only"] = q
        best = max(v["mlp_tflops_lower_bound"] for v in q.values())
        line["roofline_mlp"] = {"kernel": "gemm_bf16_mfma<128,128,DENSE,3> x8 (+gather, embed, final) per chunk",
                                "bound": "mfma", "achieved": best, "peak": PEAK_FP32_MFMA_TFLOPS,
                                "unit": "TFLOP/s", "frac": best / PEAK_FP32_MFMA_TFLOPS,
                                "flop_per_point": MLP_FLOP_PER_PT - 2 * 1472 * 512,
                                "note": "EXECUTED flops: from 32768 points per image on the 1472 feature rows "
                                        "of the local fold2/conv1 are pre-multiplied into the feature map once per "
                                        "image (disn_fold_local, 28 GFLOP), which removes 1.51 of the 3.67 "
                                        "MFLOP/point as written; the whole chunk (gather, embed, final) is in "
                                        "the time, so this is a lower bound on the GEMM rate"}
        # ---- config 3: full 257^3 grid on one GPU (no marching cubes yet) -----------------------
        from disn_amd import create_sdf as cs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc3 = eng.encode(img)
        full = cs.dense_grid_sdf(eng, enc3, 0, tm, [-1, -1, -1, 1, 1, 1], 256)
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
        line["grid256"] = {"points": 257 ** 3, "seconds": t1, "points_per_s": 257 ** 3 / t1,
                           "includes": "encode + all chunks + /10, single GPU, no marching cubes"}
        # config 3 end to end: + marching cubes on the device (+ the .obj the reference writes)
        from disn_amd import isosurface as iso
        iso.marching_cubes(full, [-1, -1, -1, 1, 1, 1], 256, 0.0)         # warm-up (workspace, code)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc3 = eng.encode(img)
        full = cs.dense_grid_sdf(eng, enc3, 0, tm, [-1, -1, -1, 1, 1, 1], 256)
        verts, faces = iso.marching_cubes(full, [-1, -1, -1, 1, 1, 1], 256, 0.0)
        torch.cuda.synchronize()
        t2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        iso.write_obj("/tmp/disn_bench_mesh.obj", verts, faces)
        t3 = time.perf_counter() - t0
        line["grid256_mesh"] = {"seconds": t2, "vertices": int(verts.shape[0]), "triangles": int(faces.shape[0]),
                                "obj_write_seconds": t3,
                                "includes": "encode + 257^3 SDF + marching cubes on one GPU (random-init weights: "
                                            "the iso-surface of an untrained net); .obj write timed separately"}
        del full
        # ---- training step (BASELINE config 5 shape, one GPU's share: 8 samples x 2048 points) -----
        try:
            from disn_amd.train_sdf import Trainer
            tr = Trainer(WeightStore.random_init(0, mode="he"), dev, batch_size=8)
            tfeed = train_feed(torch, dev, 8, 1)
            for _ in range(2):
                tr.step(tfeed)
            ms_fb = ev_time_ms(lambda: tr.forward_backward(tfeed), 5, torch)
            ms_step = ev_time_ms(lambda: tr.step(tfeed), 5, torch)
            # 3x the forward MACs of the convolutions and the two point MLPs (data + weight gradients)
            conv_flop = 8 * sum(2.0 * hw * hw * cout * 9 * cin for cin, cout, hw in VGG_LAYERS)
            mlp_flop = 8 * N_POINTS * MLP_FLOP_PER_PT
            line["train_step"] = {"samples": 8, "points_per_sample": N_POINTS, "ms_forward_backward": ms_fb,
                                  "ms_step": ms_step, "samples_per_s": 8 / ms_step * 1e3, "dtype": "f32",
                                  "mfma_tflops": 3 * (conv_flop + mlp_flop) / ms_fb / 1e9,
                                  "note": "forward + get_loss + gradient of all 56 variables + TF Adam; "
                                          "python bench.py --workload train [--gpus N] times it as the main line"}
            tr.close()
            del tr
            torch.cuda.empty_cache()
            tb = Trainer(WeightStore.random_init(0, mode="he"), dev, batch_size=8, precision="bf16")
            for _ in range(2):
                tb.step(tfeed)
            ms_b = ev_time_ms(lambda: tb.step(tfeed), 5, torch)
            line["train_step"]["mixed_precision"] = {
                "ms_step": ms_b, "samples_per_s": 8 / ms_b * 1e3, "dtype": "bf16",
                "note": "bf16 multiply (v_mfma_f32_32x32x16_bf16) with fp32 accumulate in the conv / MLP "
                        "forward, data-gradient and 128-tile weight-gradient GEMMs; fp32 activations, "
                        "master weights, gradients and Adam; --workload train --train-dtype bf16"}
            tb.close()
            del tb, tfeed
            torch.cuda.empty_cache()
        except Exception as e:  # the north-star line must survive a failure of this leg
            line["train_step"] = {"error": repr(e)}
        # ---- CPU baseline: the oracle on the same workload, host cores -----------------------------
        from oracle import disn_oracle as O
        Wn = store.arrays
        feed = {"imgs": img.cpu().numpy(), "sample_pc": pts.cpu().numpy(), "sample_pc_rot": pts.cpu().numpy(),
                "trans_mat": tm.cpu().numpy()}
        O.get_model(feed, Wn)
        ts = []
        for _ in range(max(1, args.cpu_runs)):
            t0 = time.perf_counter()
            O.get_model(feed, Wn)
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        cpu_name = ""
        try:
            cpu_name = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        # one-thread figure: torch intra-op threads = 1; numpy's BLAS pool is limited through
        # threadpoolctl when available
        nthreads = torch.get_num_threads()
        one = None
        try:
            torch.set_num_threads(1)
            try:
                from threadpoolctl import threadpool_limits
                limiter = threadpool_limits(limits=1)
            except Exception:
                limiter = None
            t0 = time.perf_counter()
            O.get_model(feed, Wn)
            one = time.perf_counter() - t0
            if limiter is not None:
                limiter.unregister() if hasattr(limiter, "unregister") else limiter.restore_original_limits()
        except Exception:
            pass
        finally:
            torch.set_num_threads(nthreads)
        line["cpu_baseline"] = {"value": N_POINTS / med, "unit": "points/s", "cores": nthreads,
                                "kind": "port", "seconds_per_step": med,
                                "value_1thread": (N_POINTS / one) if one else None,
                                "sample": "%d full steps (encode + 2048 points) of the numpy/torch-CPU oracle after "
                                          "1 warm-up, median; nproc=%d; %s" % (len(ts), os.cpu_count(), cpu_name)}
    if rank == 0:
        print(json.dumps(line))
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
