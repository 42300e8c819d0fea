#!/usr/bin/env python3
"""bench.py -- clouds/s of EPC-Net 256-d global-descriptor extraction on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" = one pass of the hot path (epc_net_forward: kNN graph -> ProxyConv blocks -> conv5 -> G_VLAD -> 256-d
descriptor) over one batch of synthetic clouds per GPU.  Workload at any N = BASELINE.json configs[1]: EPC-Net
inference, batch 64 x 4096 x 3 fp32 per GPU, NetVLAD K=64, 256-D output; inputs are resident in HBM before the timed
region.  Descriptor extraction shards over GPUs with no data-path collective (clouds are independent in inference,
SURVEY.md 8e) -> weak scaling; value = clouds all ranks processed / max-over-ranks time.

Extra objects on the JSON line:
  roofline     dominant kernel = conv5 + L2 + soft-assignment (conv5_kernel<256,VLAD>), bound = MFMA.  It runs on the
               half-precision MFMA in split arithmetic (one fp16 value per activation; weights as fp16 hi + MX-fp6 lo, the
               lo product on the K=64 scaled f8f6f4 MFMA; f32 accumulate; descriptor error 1.5e-6 against the f32 oracle), so it is priced against the dense
               bf16/fp16 peak (2.5 PFLOP/s): achieved = ALGORITHMIC FLOPs per launch (2.684 GFLOP per cloud,
               DESIGN.md) / its average duration, measured with HIP events recorded by the library on the launch
               stream inside the timed region (on every --profile-every-th step); the matrix pipe executes 1.2x that
               in bf16-rate units (frac is capped at 0.83).
  overlapped   a second timed region after the first: the same K steps with --in-flight (2) of them in flight on the
               engine's HIP streams (InferenceEngine.submit).  Reported beside `value`, never as `value`: kernels that
               share the chip take longer individually, so the roofline figures belong to the one-stream region.
  pipeline_hbm HBM bytes one step moves (PMC counters of the committed profile x launches per step) over the step time,
               against 8 TB/s -- the north_star's "fraction of the HBM roofline"; secondary, the path is not HBM-bound.
  cpu_baseline the CPU oracle (numpy restatement of the reference's dense (N,N)-mask formulation, batch = 1 cloud per
               call as evaluate.py:86-90 does) timed on this box's host cores on a bounded sample.  Rank 0, N = 1 only.
               It is NOT TensorFlow (not installable here) -- kind "port".  `index_form`: the same oracle with neighbour
               lists + gathers (this repo's formulation) instead of the dense mask, so that the algorithmic gain can be
               told from the hardware gain.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PARAMS = {"CLUSTER_SIZE": 64, "FEATURE_OUTPUT_DIM": 256, "KNN": 20, "INPUT_DIM": 3, "GROUPS": 4}
OUTER = "query_triplets"
N_POINTS = 4096
# algorithmic FLOPs per cloud of the dominant kernel: conv5 256->1024 + assignment 1024->64
CONV5_ASSIGN_FLOPS = 2.0 * N_POINTS * 256 * 1024 + 2.0 * N_POINTS * 1024 * 64
F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 (matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16 MFMA (the 5 PF headline includes 2:1 sparsity)
# Matrix-pipe work of the dominant kernel in units of one bf16/fp16 MFMA product per algorithmic product: conv5 = fp16
# hi product (1) + MX-fp6 lo product on the K=64 scaled MFMA, which runs at four times the bf16 rate (0.25); assignment = 1
# (single-fp16 cluster weights): (1.25 x 2147.5 + 536.9) / 2684.4 = 1.2, so `frac` (algorithmic / half-precision peak) is
# capped at 0.83.  (EPC-Net-L's conv5 feeds a max-pool and stays on the 3-product split-bf16 form.)
SPLIT_PRODUCTS = {"epc-net": 1.2, "epc-net-l": 3}
FLOPS_PER_CLOUD = {"epc-net": 3.747e9, "epc-net-l": 1.355e9}
# arithmetic of the dominant kernel (not a precision claim: results are f32-accurate, tests/test_gpu_parity.py)
DTYPE = {"epc-net": "f16+f6", "epc-net-l": "bf16x3"}


def pkg(name=""):
    return importlib.import_module("epc-net_amd" + ("." + name if name else ""))


def build_store(arch, device, seed):
    V = pkg("variables")
    st = V.reset_default_store(device=device, seed=seed)
    M = pkg("models." + arch)
    with V.variable_scope(OUTER):
        M.declare_variables(PARAMS, N_POINTS)
    st.randomize_statistics(seed)
    return st


def cpu_baseline(arch, store, budget_s, max_clouds):
    """Time the oracle on host cores.  The ONLY place bench.py touches oracle/ (never the measured GPU path)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import epcnet_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    w = {k[len(OUTER) + 1:]: v.detach().cpu().numpy() for k, v in store.vars.items()}
    pcs = O.synthetic_clouds(max_clouds + 1, N_POINTS, 1234)
    O.forward(pcs[:1, None], w, arch=arch)  # warm-up (BLAS threads, page faults)
    done, t0 = 0, time.perf_counter()
    while done < max_clouds and (time.perf_counter() - t0) < budget_s:
        O.forward(pcs[done + 1:done + 2, None], w, arch=arch)  # batch = 1 cloud per call (evaluate.py:86-90)
        done += 1
    dt = time.perf_counter() - t0
    # the same oracle in the kNN-index formulation this repo's kernels use (neighbour lists + gathers instead of the dense
    # (N,N) mask and mask @ x): separates the algorithmic gain from the hardware gain.  Bounded: ~6 s.
    done2, t1 = 0, time.perf_counter()
    while done2 < 8 and (time.perf_counter() - t1) < 6.0:
        O.forward(pcs[done2 + 1:done2 + 2, None], w, arch=arch, formulation="lists")
        done2 += 1
    dt2 = time.perf_counter() - t1
    return {"value": round(done / dt, 4), "unit": "clouds/s", "cores": int(threads), "kind": "port",
            "index_form": {"value": round(done2 / dt2, 4), "unit": "clouds/s",
                           "sample": "%d clouds, same oracle with neighbour lists + gathers instead of the dense mask, %.1f s"
                                     % (done2, dt2)},
            "sample": "%d clouds x %d pts, 1 cloud per call, numpy/OpenBLAS restatement of the reference's dense "
                      "(N,N)-mask graph (oracle/epcnet_oracle.py), %.1f s; host has %d logical CPUs"
                      % (done, N_POINTS, dt, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=400.0, help="extra untimed steps after the warm-up (0 = none)")
    ap.add_argument("--batch", type=int, default=64, help="clouds per step per GPU (configs[1]: 64)")
    ap.add_argument("--arch", default="epc-net", choices=["epc-net", "epc-net-l"])
    ap.add_argument("--precision", default="f32", choices=["f32", "fast"],
                    help="arithmetic of the EPC-Net path (include/epcnet.h EPC_PRECISION_*)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="steps kept in flight on the engine's HIP streams (InferenceEngine.submit); 1 = one stream")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="record the stage-boundary HIP events on every n-th timed step (the events cost ~5 %% of a step)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--cpu-clouds", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d (WORLD_SIZE=%d)"
                         % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    E = pkg("engine")
    store = build_store(args.arch, device, seed=0)            # same weights on every rank
    lanes = max(1, args.in_flight)
    eng = E.InferenceEngine(args.arch, PARAMS, store, outer=OUTER, micro_batch=args.batch, in_flight=lanes,
                            precision=args.precision)
    g = torch.Generator(device="cpu")
    g.manual_seed(100 + rank)                                  # every rank extracts different clouds
    xyz = (torch.rand((args.batch, N_POINTS, 3), generator=g) * 2.0 - 1.0).to(device)
    outs = [torch.empty((args.batch, 256), dtype=torch.float32, device=device) for _ in range(lanes)]
    out = outs[0]

    every = max(1, args.profile_every)
    profiles = {k: E.StageProfile() for k in range(0, args.steps, every)}
    scratch = E.StageProfile()
    for k in range(max(args.warmup, 0)):
        eng.forward(xyz, out=out, profile=scratch if k % every == 0 else None)   # the entry point the timed steps use
    if args.warmup > 0:
        torch.cuda.synchronize()
        scratch.elapsed_ms()                                   # first event query happens outside the timed region
        # Settle: a process that starts right after another GPU process has exited (pytest, then this) can see one
        # 70-80 ms dispatch stall while the driver tears the old process down -- observed as a single slow kNN launch.
        # Keep issuing UNTIMED steps until the device has been busy for --settle-ms since the first launch.
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            eng.forward(xyz, out=out)
            torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- the timed region: K steps on ONE stream; stage-boundary HIP events on every `every`-th step -------------
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        eng.forward(xyz, out=out, profile=profiles.get(k))
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    norms = out.norm(dim=1)
    if not bool(torch.isfinite(out).all()) or float((norms - 1).abs().max()) > 1e-3:
        raise SystemExit("descriptors are not unit-norm / finite: refusing to report a number")

    # ---- second, separately reported region: the same K steps with `lanes` of them in flight ----------------------
    # InferenceEngine.submit deals successive steps over the engine's own HIP streams (own workspace and output buffer
    # each): step k+1's kNN (VALU-bound) then runs beside step k's conv5 / aggregate (MFMA- / HBM-bound).  Every step
    # still does the whole path.  Kernels that share the chip take longer individually, so per-kernel roofline figures
    # come from the one-stream region above and this region only reports its step rate.
    overlapped = None
    if lanes > 1:
        for k in range(4 * lanes):
            eng.submit(xyz, out=outs[k % lanes])
        fence()
        t1 = time.perf_counter()
        for k in range(args.steps):
            eng.submit(xyz, out=outs[k % lanes])
        fence()
        el2 = max_over_ranks(time.perf_counter() - t1)
        if not all(torch.equal(out, o) for o in outs[1:]):
            raise SystemExit("lanes disagree with the one-stream descriptors: refusing to report a number")
        overlapped = {"steps_in_flight_per_gpu": lanes, "value": round(world * args.batch * args.steps / el2, 2),
                      "unit": "clouds/s", "ms_per_step": round(el2 / args.steps * 1e3, 4),
                      "how": "InferenceEngine.submit: successive steps on %d HIP streams, same kernels, same results" % lanes}

    stage = {}
    for pr in profiles.values():
        for k, v in pr.elapsed_ms().items():
            stage[k] = stage.get(k, 0.0) + v / len(profiles)
    conv5_ms = stage["conv5"]
    conv5_flops = (CONV5_ASSIGN_FLOPS if args.arch == "epc-net" else 2.0 * N_POINTS * 128 * 1024) * args.batch
    achieved = conv5_flops / (conv5_ms * 1e-3) / 1e12
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this same command
    # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; profiles/): measured at batch 64 only.
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_current.json")))
        if args.arch == "epc-net" and args.batch == 64:
            traffic = next(v["hbm_bytes_per_launch_corrected"] for k, v in pm["kernels"].items()
                           if k.startswith("void conv5_kernel<256, 0"))
    except Exception:
        traffic = None
    # HBM bytes of ONE step = the PMC bytes per launch of the pipeline's kernels (profiles/pmc_hbm_current.json, collected
    # by scripts/collect_profiles.sh on this configuration) x their launches per step: the figure behind "fraction of the
    # HBM roofline" (north_star); the path is MFMA / VALU-bound, so it is a secondary number.
    hbm_step = None
    try:
        per_step = {"morton_sort_kernel": 1, "void knn_topk_culled_kernel": 1, "proxyconv_block_f16_kernel": 4,
                    "void conv5_kernel<256, 0": 1, "vlad_aggregate_kernel": 1, "void vlad_fold_kernel": 1,
                    "hidden_gemm_kernel": 1, "head_finish_kernel": 1}
        if args.arch == "epc-net" and args.batch == 64:
            hbm_step = sum(v["hbm_bytes_per_launch_corrected"] * m for k, v in pm["kernels"].items()
                           for pre, m in per_step.items() if k.startswith(pre))
    except Exception:
        hbm_step = None
    clouds = world * args.batch * args.steps
    value = clouds / elapsed

    if rank == 0:
        line = {
            "metric": "point-clouds/sec (4096 pts) descriptor extraction",
            "value": round(value, 2), "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[args.arch], "data": "synthetic",
            "config": {"workload": "%s inference, batch %dx%dx3 fp32 per GPU, NetVLAD K=64, 256-D output "
                                   "(BASELINE.json configs[1])" % (args.arch, args.batch, N_POINTS),
                       "clouds_per_step_per_gpu": args.batch, "num_points": N_POINTS,
                       "weights": "seeded random init of the architecture (no checkpoint payloads exist)",
                       "parallelism": "independent clouds sharded over %d GPU(s), no data-path collective" % world,
                       "stage_events_on_every_nth_step": every},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "executed_tflops": round(achieved * SPLIT_PRODUCTS[args.arch], 3),
                         "executed_frac": round(achieved * SPLIT_PRODUCTS[args.arch] / BF16_MFMA_PEAK_TFLOPS, 4),
                         "vs_f32_mfma_peak": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                         "kernel": "conv5_kernel (conv5 + L2 + soft-assignment), split half-precision MFMA (%s), "
                                   "f32 accumulate" % DTYPE[args.arch],
                         "avg_launch_ms": round(conv5_ms, 4),
                         "algorithmic_flops_per_launch": conv5_flops},
            "stage_ms": {k: round(v, 4) for k, v in stage.items()},
            "pipeline_tflops": round(value / world * FLOPS_PER_CLOUD[args.arch] / 1e12, 3),
        }
        if hbm_step:
            gbps = hbm_step / (elapsed / args.steps) / 1e9
            line["pipeline_hbm"] = {"bytes_per_step": int(hbm_step), "achieved_GBps": round(gbps, 1), "peak_GBps": 8000.0,
                                    "frac": round(gbps / 8000.0, 4),
                                    "how": "PMC bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/) x launches per step / ms_per_step"}
        if overlapped is not None:
            line["overlapped"] = overlapped
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.arch, store, args.cpu_budget_s, args.cpu_clouds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
