#!/usr/bin/env python3
"""bench.py -- clouds/s of EPC-Net 256-d global-descriptor extraction on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1 from a plain shell: the script launches itself (one process per GPU, torch.distributed.run, RCCL) BEFORE touching
  the GPU and relays rank 0's JSON line; under an existing torchrun (WORLD_SIZE set) it is the rank process.

A "step" = one pass of the hot path (epc_net_forward: sort -> kNN graph -> ProxyConv blocks -> conv5 -> G_VLAD -> 256-d
descriptor) over one batch of synthetic clouds per GPU.  Workload at any N = BASELINE.json configs[1]: EPC-Net inference,
batch 64 x 4096 x 3 fp32 per GPU, NetVLAD K=64, 256-D output; inputs are resident in HBM before the timed region.
Descriptor extraction shards over GPUs with no data-path collective (clouds are independent in inference)
-> weak scaling; value = clouds all ranks processed / max-over-ranks time.

--steps K: the timed region is K steps between fences (barrier + synchronize, max over ranks); it is repeated --regions
times (default 7) and `value` / `ms_per_step` are the MEDIAN region's, `regions` holds min / max.

`value` / `dtype` / `roofline` belong to EPC_PRECISION_F32 (include/epcnet.h): the f32-equivalent arithmetic (conv layers:
both MFMA operands scaled by a per-row / per-column power of two and split into fp16 hi + lo, three products, f32 accumulate
= 2^-21 per product; f32 tensors in HBM) -- the arithmetic class of the reference's float32 graph, the one that meets the
1e-4 bar on the whole adversarial parity set (tests/test_gpu_adversarial.py: closer to the float64 result than numpy's
float32 on the ill-conditioned cases).  The same timed region is then repeated in EPC_PRECISION_FAST (one fp16 value per
activation, weights fp16 hi + MX-fp6 lo; right-or-refuses outside fp16's range) and reported in the `fast` object with its
own roofline -- never as `value`.

Objects on the JSON line:
  roofline     dominant kernel = conv5 + L2 norm + soft-assignment (conv5_kernel<256,VLAD>), bound = MFMA, priced against
               the dense bf16 peak (2.5 PFLOP/s): achieved = ALGORITHMIC FLOPs per launch (2.684 GFLOP per cloud)
               / its average duration from HIP events the library records on the launch stream inside the timed region
               (every --profile-every-th step).  executed_* = the matrix-pipe work in bf16-rate units (3 products per
               algorithmic product in f32-equivalent arithmetic; 1.2 in the fast one); mfma_util = the MFMA-busy fraction from
               the committed rocprofv3 PMC pass of this command (profiles/pmc_compute_current.json), null when absent.
  fast         the EPC_PRECISION_FAST region: value, ms_per_step, stage_ms, roofline, overlapped (two steps in flight).
  configs      bounded legs for the other BASELINE.json configs, each timed like the main region (barrier + synchronize on
               both sides, max over ranks):
               train_step      configs[2]: one quadruplet step (1 + 2 + 14 + 1 = 18 clouds x 4096, the reference's tuple:
                               configs/epc-net.yaml:28-34; train.py:238-277, 484-495), a FRESH tuple every step (the loss stays
                               non-zero), HIP-graph replay at every N (data-parallel over tuples at N > 1: three graphs around
                               the two RCCL all-reduces), f32-accurate arithmetic, with its own `roofline` (MFMA and, from the
                               committed counters, HBM); `eager` = the same step without graphs (side number);
               train_step_bf16 the same step in the arithmetic configs[2] names: bf16-stored (rows, 1024) activations and
                               gradients, one bf16 value per GEMM operand, f32 accumulate / statistics / master weights;
               train_step_22, train_step_bf16_22   both at BASELINE.json's literal size, 1 + 2 + 18 + 1 = 22 clouds;
               epc_net_l_b256  configs[3]: EPC-Net-L inference at batch 256 per GPU, with its own roofline;
               retrieval       configs[4] composed end to end (retrieval.evaluate_sharded = evaluate.py:293-332): 23 runs x
                               (400 + 120) synthetic clouds EXTRACTED sharded over the ranks, ONE RCCL all-gather of the
                               descriptors, every rank ranks its share of the queries against each run's database
                               (epc_pairwise_topk, k = 25) and books them on its device, ONE RCCL all-reduce of the per-pair
                               integer counters (no host loop, no serial tail on rank 0: `serial_ms`).
                               The process group is RCCL at N = 1 as well (a world of one rank: rccl_exercised true).
  pipeline_hbm HBM bytes one step moves (PMC counters of the committed profile x launches per step) over the step time,
               against 8 TB/s -- the north_star's "fraction of the HBM roofline"; secondary, the path is not HBM-bound.
  cpu_baseline the CPU oracle (numpy restatement of the reference's dense (N,N)-mask formulation, batch = 1 cloud per
               call as evaluate.py:86-90 does) timed on this box's host cores on a bounded sample.  Rank 0, N = 1 only.
               It is NOT TensorFlow (not installable here) -- kind "port".  `index_form`: the same oracle with neighbour
               lists + gathers (this repo's formulation) instead of the dense mask.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PARAMS = {"CLUSTER_SIZE": 64, "FEATURE_OUTPUT_DIM": 256, "KNN": 20, "INPUT_DIM": 3, "GROUPS": 4}
OUTER = "query_triplets"
N_POINTS = 4096
# algorithmic FLOPs per cloud of the dominant kernel: conv5 256->1024 + assignment 1024->64
CONV5_ASSIGN_FLOPS = 2.0 * N_POINTS * 256 * 1024 + 2.0 * N_POINTS * 1024 * 64
CONV5_L_FLOPS = 2.0 * N_POINTS * 128 * 1024
F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 (matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16 MFMA (the 5 PF headline includes 2:1 sparsity)
# Matrix-pipe work of the dominant kernel in units of one bf16/fp16 MFMA product per algorithmic product:
#   f32-equivalent: conv5 and the assignment both run hi*hi + hi*lo + lo*hi                                  -> 3
#   fast: conv5 = fp16 hi product (1) + MX-fp6 lo product on the K=64 scaled MFMA at four times the rate (0.25),
#         assignment = 1 (single-fp16 cluster weights): (1.25 x 2147.5 + 536.9) / 2684.4                     -> 1.2
SPLIT_PRODUCTS = {("epc-net", "f32"): 3.0, ("epc-net", "fast"): 1.2, ("epc-net-l", "f32"): 3.0}
FLOPS_PER_CLOUD = {"epc-net": 3.747e9, "epc-net-l": 1.355e9}
# arithmetic of the dominant kernel (not a precision claim: tests/test_gpu_parity.py, tests/test_gpu_adversarial.py)
DTYPE = {("epc-net", "f32"): "f16x3", ("epc-net", "fast"): "f16+f6", ("epc-net-l", "f32"): "f16x3"}
CONV5_KERNEL = {("epc-net", "f32"): "void conv5_vlad_f32_kernel<256>", ("epc-net", "fast"): "void conv5_kernel<256, 0, true, true>",
                ("epc-net-l", "f32"): "void conv5_max_f32_kernel<128>"}


def pkg(name=""):
    return importlib.import_module("epc-net_amd" + ("." + name if name else ""))


# Floor of every inference stage IN ITS PRESENT GEOMETRY ("Floor models": the arithmetic behind each line), EPC-Net,
# EPC_PRECISION_F32, 64 clouds x 4096 points.  Shader clocks from the profiled launches (GRBM_GUI_ACTIVE / 8 XCDs / duration): the
# conv5 kernel -- the one that keeps the matrix pipe busy -- runs at 2.04 GHz, every other stage at 2.42 GHz.  A stage at its floor moves
# only with a different algorithm / fewer instructions / fewer bytes, not with tuning.
MFMA_GHZ, VALU_GHZ = 2.04, 2.42


def stage_floors_epc_net_f32_b64():
    cus, simds = 256, 1024
    # kNN: 75.06 M vector instructions per launch (SQ_INSTS_VALU, profiles/*_pmc_compute.json) at one wave-instruction per SIMD and
    # four cycles: VALU issue is the only resource the kernel loads (0.84 of its cycles at this batch, 0.98 at EPC-Net-L's 256 clouds)
    knn = 75.06e6 / simds * 4 / (VALU_GHZ * 1e9)
    # ProxyConv block: 20 gathered 256-byte rows per point = 1.34 GB per launch.  Through the vector L1 at 64 B / clk / CU + the skeleton
    # (weight-pack staging, MFMA chain, epilogue: 0.020 ms) that would be 0.054 ms -- the figure of rounds 4-5, which assumes every row
    # L1-hot.  The round-2 ablations (docs/HISTORY_r01_r02.md:654-670: all rows L1-hot 0.071 of 0.100 ms; the gather in LDS 0.081-0.095;
    # index prefetch: no change) leave the L2 -> CU path of the ~70 % of rows a 32-KB L1 cannot hold as the bound, and no in-kernel lever
    # moves it at this row width.  The floor is therefore the kernel's own best instance -- block 4, whose gathered rows are the most
    # L2-local: 0.068 ms (0.054 flattered the step's distance to its floor).
    block = 0.068e-3
    # conv5 + assignment: 4 rounds of one 8-wave workgroup per CU; per workgroup a 256-KB prologue at 9 B / clk / CU (one CU's miss
    # queue at HBM latency), then 32 chunks of: 120 MFMAs x 16 cycles per wave, two waves per SIMD (3840 cycles of matrix pipe), the
    # two waves' 1900-cycle epilogues of which the half that fits under the MFMA issue slots (the vector pipe is blocked 8 of 16 cycles
    # per 16x16x32 MFMA) is hidden: 3840 + (3800 - 1920) cycles per chunk
    conv5 = 4 * (256 * 1024 / 9.0 + 32 * (3840 + (3800 - 1920))) / (MFMA_GHZ * 1e9)
    # aggregate: the 3-byte feature map + assignment fragments, 905 MB per launch, at the 6.3 TB/s a streaming read reaches
    aggregate = 905e6 / 6.3e12
    return {"sort": 0.012, "knn": round(knn * 1e3, 4), "conv1": 0.0, "block1": round(block * 1e3, 4), "block2": round(block * 1e3, 4),
            "block3": round(block * 1e3, 4), "block4": round(block * 1e3, 4), "conv5": round(conv5 * 1e3, 4),
            "aggregate": round(aggregate * 1e3, 4), "head": 0.020}


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


def self_launch(args, argv):
    """--gpus N > 1 from a plain shell: start one rank process per GPU and relay rank 0's line.  Runs BEFORE this process
    makes any HIP call (torch.cuda.device_count() does not initialise the runtime on this image) and starts the ranks as
    CHILDREN -- a process that has initialised the GPU must never be replaced by exec."""
    import torch
    have = torch.cuda.device_count()
    n = min(args.gpus, have) if have > 0 else 0
    if args.same_device and have > 0:
        n = args.gpus                                          # every rank on cuda:0 (dry run of the N > 1 code on a 1-GPU box)
    if n < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    child_argv = [a for a in argv]
    if n != args.gpus:   # degrade cleanly: run on what the box has and say so
        out, skip = [], False
        for a in child_argv:
            if skip:
                skip = False
                continue
            if a == "--gpus":
                skip = True
                continue
            if a.startswith("--gpus="):
                continue
            out.append(a)
        child_argv = out + ["--gpus", str(n), "--requested-gpus", str(args.gpus)]
    if n == 1:
        cmd = [sys.executable, os.path.abspath(__file__)] + child_argv
    else:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child_argv
    env = dict(os.environ, EPC_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


class Harness:
    """Timing discipline shared by every leg: barrier + synchronize on both sides, max over ranks."""

    def __init__(self, device, dist, world, rank):
        self.device, self.dist, self.world, self.rank = device, dist, world, rank

    def fence(self):
        import torch
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        import torch
        if self.dist is None:
            return seconds
        # (a host tensor under gloo -- the dry-run backend --, a device tensor under RCCL)
        on_host = self.dist.get_backend() == "gloo"
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if on_host else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok):
        """True on every rank iff `ok` on every rank (a leg that failed on one rank is skipped / reported by all of them)."""
        import torch
        if self.dist is None:
            return bool(ok)
        on_host = self.dist.get_backend() == "gloo"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if on_host else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def timed(self, fn, steps):
        """K calls of fn(k) between fences; returns max-over-ranks seconds."""
        self.fence()
        t0 = time.perf_counter()
        for k in range(steps):
            fn(k)
        self.fence()
        return self.max_over_ranks(time.perf_counter() - t0)

    def timed_regions(self, fn, steps, regions):
        """`regions` fenced regions of EXACTLY `steps` calls each (fn(region, k)); returns the sorted list of max-over-ranks
        seconds.  The reported figure is the MEDIAN region: a --steps 20 region of this path is ~25 ms, short enough for one
        DVFS excursion or one host hiccup to move it by several per cent, and the chip's clock differs box to box."""
        out = []
        for r in range(regions):
            out.append(self.timed(lambda k: fn(r, k), steps))
        return sorted(out)


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def lib_sha256():
    """SHA-256 of the libepcnet_hip.so this process loaded (stamped into profiles/pmc_*_current.json at collection time)."""
    import hashlib
    path = pkg("lib").LIB_PATH
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


_PMC_WHY = {}


def pmc_summary(name):
    """A committed rocprofv3 PMC summary -- only when it was collected on THIS build of the library (scripts/
    summarise_profiles.py stamps the .so's SHA-256 into it); otherwise None and the reason in _PMC_WHY[name]."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        _PMC_WHY[name] = "profiles/%s is absent" % name
        return None
    stamp = doc.get("lib_sha256")
    if not stamp:
        _PMC_WHY[name] = "profiles/%s carries no lib_sha256 stamp" % name
        return None
    if stamp != lib_sha256():
        _PMC_WHY[name] = ("profiles/%s was collected on another build of libepcnet_hip.so (stamp %s..., loaded %s...): re-run "
                          "scripts/collect_profiles.sh" % (name, stamp[:12], lib_sha256()[:12]))
        return None
    return doc


def kernel_entry(summary, prefix):
    if not summary:
        return None
    for k, v in summary.get("kernels", {}).items():
        if k.startswith(prefix):
            return v
    return None


def extraction_leg(H, E, store, arch, precision, batch, steps, warmup, settle_ms, every, lanes, regions=1):
    """The timed region of one (arch, precision): K steps on ONE stream (stage-boundary HIP events on every `every`-th
    step), then -- separately reported -- the same K steps with `lanes` of them in flight."""
    import torch
    device = H.device
    eng = E.InferenceEngine(arch, PARAMS, store, outer=OUTER, micro_batch=batch, in_flight=max(1, lanes), precision=precision)
    g = torch.Generator(device="cpu")
    g.manual_seed(100 + H.rank)                                  # every rank extracts different clouds
    xyz = (torch.rand((batch, N_POINTS, 3), generator=g) * 2.0 - 1.0).to(device)
    outs = [torch.empty((batch, 256), dtype=torch.float32, device=device) for _ in range(max(1, lanes))]
    out = outs[0]
    profiles = {(r, k): E.StageProfile() for r in range(regions) for k in range(0, steps, every)}
    scratch = E.StageProfile()
    for k in range(max(warmup, 0)):
        eng.forward(xyz, out=out, profile=scratch if k % every == 0 else None, check=False)   # the entry point the timed steps use
    if warmup > 0:
        torch.cuda.synchronize()
        scratch.elapsed_ms()                                   # first event query happens outside the timed region
        # Settle: a process that starts right after another GPU process has exited (pytest, then this) can see one
        # 70-80 ms dispatch stall while the driver tears the old process down -- observed as a single slow kNN launch.
        # Keep issuing UNTIMED steps until the device has been busy for --settle-ms since the first launch.
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
            eng.forward(xyz, out=out, check=False)
            torch.cuda.synchronize()
    region_s = H.timed_regions(lambda r, k: eng.forward(xyz, out=out, profile=profiles.get((r, k)), check=False), steps, regions)
    elapsed = median(region_s)
    norms = out.norm(dim=1)
    if not bool(torch.isfinite(out).all()) or float((norms - 1).abs().max()) > 1e-3:
        raise SystemExit("descriptors are not unit-norm / finite: refusing to report a number")

    # second, separately reported region: the same K steps with `lanes` of them in flight (InferenceEngine.submit deals
    # successive steps over the engine's own HIP streams, own workspace and output buffer each: step k+1's kNN (VALU-bound)
    # runs beside step k's conv5 / aggregate).  Kernels that share the chip take longer individually, so per-kernel
    # roofline figures come from the one-stream region and this region only reports its step rate.
    overlapped = None
    if lanes > 1:
        for k in range(4 * lanes):
            eng.submit(xyz, out=outs[k % lanes])
        el2 = median(H.timed_regions(lambda r, k: eng.submit(xyz, out=outs[k % lanes]), steps, regions))
        if not all(torch.equal(out, o) for o in outs[1:]):
            raise SystemExit("lanes disagree with the one-stream descriptors: refusing to report a number")
        overlapped = {"steps_in_flight_per_gpu": lanes, "value": round(H.world * batch * steps / el2, 2),
                      "unit": "clouds/s", "ms_per_step": round(el2 / steps * 1e3, 4),
                      "how": "InferenceEngine.submit: successive steps on %d HIP streams, same kernels, same results" % lanes}
    stage = {}
    for pr in profiles.values():
        for k, v in pr.elapsed_ms().items():
            stage[k] = stage.get(k, 0.0) + v / len(profiles)
    conv5_ms = stage["conv5"]
    conv5_flops = (CONV5_ASSIGN_FLOPS if arch == "epc-net" else CONV5_L_FLOPS) * batch
    achieved = conv5_flops / (conv5_ms * 1e-3) / 1e12
    key = (arch, precision)
    # HBM bytes / MFMA-busy fraction per launch of the dominant kernel from the committed rocprofv3 PMC summaries of this
    # same command (profiles/; collected by scripts/collect_profiles.sh at batch 64 / 256)
    std_batch = batch == (64 if arch == "epc-net" else 256)
    hbm = kernel_entry(pmc_summary("pmc_hbm_current.json"), CONV5_KERNEL[key]) if std_batch else None
    comp = kernel_entry(pmc_summary("pmc_compute_current.json"), CONV5_KERNEL[key]) if std_batch else None
    split = SPLIT_PRODUCTS[key]
    roofline = {"bound": "mfma", "achieved": round(achieved, 3), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                "traffic": hbm["hbm_bytes_per_launch_corrected"] if hbm else None,
                "executed_tflops": round(achieved * split, 3),
                "executed_frac": round(achieved * split / BF16_MFMA_PEAK_TFLOPS, 4),
                "mfma_util": comp.get("mfma_util") if comp else None,
                "vs_f32_mfma_peak": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                "kernel": "%s (conv5 + %s), %s MFMA, f32 accumulate"
                          % (CONV5_KERNEL[key][5:], "L2 + soft-assignment" if arch == "epc-net" else "global max-pool", DTYPE[key]),
                "avg_launch_ms": round(conv5_ms, 4), "algorithmic_flops_per_launch": conv5_flops}
    if hbm is None or comp is None:
        roofline["counters_unavailable"] = ("batch differs from the profiled configuration" if not std_batch else
                                            "; ".join(sorted(set(_PMC_WHY.values()))))
    res = {"value": round(H.world * batch * steps / elapsed, 2), "unit": "clouds/s", "dtype": DTYPE[key],
           "ms_per_step": round(elapsed / steps * 1e3, 4),
           "regions": {"count": regions, "steps_each": steps, "statistic": "median",
                       "ms_per_step_min": round(region_s[0] / steps * 1e3, 4), "ms_per_step_max": round(region_s[-1] / steps * 1e3, 4),
                       "value_min": round(H.world * batch * steps / region_s[-1], 2),
                       "value_max": round(H.world * batch * steps / region_s[0], 2)},
           "roofline": roofline,
           "stage_ms": {k: round(v, 4) for k, v in stage.items()},
           "pipeline_tflops": round(batch * steps / elapsed * FLOPS_PER_CLOUD[arch] / 1e12, 3)}
    if arch == "epc-net" and precision == "f32" and batch == 64:
        fl = stage_floors_epc_net_f32_b64()
        res["stage_floor_ms"] = fl
        res["stage_floor"] = {"sum_ms": round(sum(fl.values()), 4), "step_over_floor": round(elapsed / steps * 1e3 / sum(fl.values()), 3),
                              "model": "per-stage floors of the present kernels' geometry (conv5 at %.2f GHz, the other stages at %.2f): kNN = vector instructions / issue rate; "
                                       "block = the L2 -> CU miss path of the gathered rows a 32-KB L1 cannot hold (round-2 ablations); conv5 = 4 rounds x (256-KB prologue at 9 B/clk/CU + "
                                       "32 chunks x (MFMA issue + the epilogue share the MFMA issue slots do not hide)); aggregate = feat "
                                       "bytes / 6.3 TB/s (DESIGN.md 4, Floor models)" % (MFMA_GHZ, VALU_GHZ)}
    if overlapped is not None:
        res["overlapped"] = overlapped
    return res, elapsed


def train_counters(precision, n_clouds):
    """Counter summary of the training step from the committed rocprofv3 PMC passes (profiles/pmc_train_current.json, collected by
    scripts/collect_train_profiles.sh on the eager step of the same size) -- only while its lib_sha256 stamp is the loaded library's."""
    doc = pmc_summary("pmc_train_current.json")
    if not doc:
        return None, _PMC_WHY.get("pmc_train_current.json")
    leg = doc.get("legs", {}).get("%s_%d" % (precision, n_clouds))
    if not leg:
        return None, "profiles/pmc_train_current.json has no %s step at %d clouds" % (precision, n_clouds)
    return leg, None


def train_step_leg(H, steps, warmup, precision="bf16x6", eager_steps=0, n_neg=14):
    """configs[2]: the quadruplet step at full size, a fresh tuple every step.  HIP-graph replay at EVERY world size (one graph
    at N = 1; at N > 1 three graphs around the two RCCL all-reduces -- the head's gradients travel under the backbone's backward,
    training.TrainStep._graphed_step).  ``precision``:
    "bf16x6" = the f32-accurate split arithmetic (the reference computes in float32), "bf16" = one bf16 value per GEMM operand,
    the arithmetic BASELINE.json configs[2] names.  ``eager_steps`` > 0: a side number without graphs (launch-bound)."""
    import torch
    TR = pkg("training")
    store = build_store("epc-net", H.device, 0)                # every rank starts from the same weights
    params = dict(PARAMS, ARCH="epc-net", BATCH_NUM_QUERIES=1, DECAY_STEP=200000, BASE_LEARNING_RATE=5e-5, MARGIN_1=0.5,
                  MARGIN_2=0.2, TRAIN_PRECISION=precision)
    ts = TR.TrainStep(params, store, outer=OUTER)
    g = torch.Generator(device="cpu")
    g.manual_seed(7000 + H.rank)                               # data-parallel over tuples: every rank its own tuples
    n_tuples = 8
    ncl = 1 + 2 + n_neg + 1                                    # 18: the reference's tuple (configs/epc-net.yaml:28-34); 22: BASELINE.json's "18 neg"
    tuples = [(torch.rand((ncl, N_POINTS, 3), generator=g) * 2.0 - 1.0).to(H.device) for _ in range(n_tuples)]
    losses = []

    def run(use_graph):
        def one(k):
            t = tuples[k % n_tuples]
            loss, _, _ = ts.step(t[None, 0:1], t[None, 1:3], t[None, 3:3 + n_neg], t[None, 3 + n_neg:], epoch=0, graph=use_graph)
            losses.append(loss)
        return one

    for k in range(warmup):
        run(True)(k)
    del losses[:]
    elapsed = H.timed(run(True), steps)
    loss_vals = [float(x) for x in losses]
    ops = pkg("ops")
    ops.chain_persist_check(H.device)                          # an abandoned grid barrier of the persistent forward chain fails the leg loudly
    persistent = bool(ops.CHAIN_PERSIST_FWD and pkg("lib").lib().epc_chain_persist_ok(ncl * N_POINTS))
    flops = 3.0 * FLOPS_PER_CLOUD["epc-net"] * ncl
    tflops = flops * steps / elapsed / 1e12
    ms = elapsed / steps * 1e3
    ctr, ctr_why = train_counters(precision, ncl)
    # The step's kernels are single passes over its tensors ("Training step"): it is priced against HBM as well --
    # counter bytes per step (FETCH_SIZE x 2 + WRITE_SIZE summed over the step's launches) over the step time -- next to the
    # algorithmic bytes of the tensors each kernel has to move once (profiles/pmc_train_current.json: model_bytes_per_step).
    roof = {"bound": "hbm" if precision == "bf16" else "mfma", "achieved": round(tflops, 3), "peak": BF16_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tflops / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "vs_f32_mfma_peak": round(tflops / F32_MFMA_PEAK_TFLOPS, 4),
            "kernel": "whole step (forward + backward + Adam + moving averages), all launches of the replayed graph"}
    if ctr:
        gbps = ctr["hbm_bytes_per_step"] / (ms * 1e-3) / 1e9
        roof.update({"traffic": int(ctr["hbm_bytes_per_step"]), "hbm_achieved_GBps": round(gbps, 1), "hbm_peak_GBps": 8000.0,
                     "hbm_frac": round(gbps / 8000.0, 4), "model_bytes_per_step": ctr.get("model_bytes_per_step"),
                     "traffic_over_model": round(ctr["hbm_bytes_per_step"] / ctr["model_bytes_per_step"], 3) if ctr.get("model_bytes_per_step") else None,
                     "largest_kernels": ctr.get("largest_kernels")})
    else:
        roof["counters_unavailable"] = ctr_why
    out = {"workload": "EPC-Net quadruplet training step, 1 + 2 + %d + 1 = %d clouds x 4096 pts per GPU (BASELINE.json configs[2]; "
                       "train.py:238-277, 484-495), fresh tuple every step, HIP-graph replay%s"
                       % (n_neg, ncl, "" if H.world == 1 else " (three graphs around the two all-reduces of the data-parallel step: "
                          "the head's gradients travel under the backbone's backward)"),
           "value": round(H.world * steps / elapsed, 2), "unit": "tuples/s (%d clouds each)" % ncl, "steps": steps,
           "clouds_per_tuple": ncl, "ms_per_step": round(ms, 4),
           "dtype": ("bf16x6 / f16x3 forward, bf16x3 backward GEMMs (f32-accurate), f32 tensors" if precision == "bf16x6" else
                     "bf16: the (rows, 1024) activations and gradients of conv5 .. VLAD stored as bf16, one bf16 value per GEMM "
                     "operand, f32 accumulate / statistics / master weights (the 64-channel backbone's tensors are f32)"),
           "backbone_forward": ("one persistent launch: a workgroup per CU keeps its rows, a grid-wide barrier per BatchNorm (csrc/train_chain_persist.hip)"
                                if persistent else "13 fused launches (csrc/train_chain.hip)"),
           "tflops": round(tflops, 2), "algorithmic_flops_per_step": flops,
           # the whole step against the dense 16-bit matrix peak (96 % of its FLOPs are dense contractions)
           "roofline": roof,
           "loss_first": round(loss_vals[0], 5), "loss_last": round(loss_vals[-1], 5),
           "loss_mean": round(sum(loss_vals) / len(loss_vals), 5)}
    if eager_steps > 0:
        for k in range(3):
            run(False)(k)
        e2 = H.timed(run(False), eager_steps)
        out["eager"] = {"ms_per_step": round(e2 / eager_steps * 1e3, 4), "steps": eager_steps,
                        "note": "the same step launched from Python without HIP graphs (host launch-bound): a side number"}
    return out


def retrieval_leg(H, E, steps, warmup, rccl):
    """configs[4] composed end to end (evaluate.py:293-332): EXTRACTION of 23 runs x (400 database + 120 query) synthetic clouds
    sharded over the ranks -> ONE RCCL all-gather of the descriptors -> every rank ranks its share of the queries against each
    database run (epc_pairwise_topk, k = 25) and books them on its device -> ONE RCCL all-reduce of the integer counters.
    retrieval.evaluate_sharded is the product entry point; a step = one whole evaluation."""
    import numpy as np
    import torch
    D, R = pkg("distributed"), pkg("retrieval")
    runs, n_db, n_q = 23, 400, 120
    rank, world = H.rank, H.world
    store = build_store("epc-net", H.device, 0)
    eng = E.InferenceEngine("epc-net", PARAMS, store, outer=OUTER, micro_batch=64, in_flight=1, precision="f32")
    # the same clouds on every rank (same device generator seed, same hardware); a rank only EXTRACTS its shard.
    # query i of run n = a jittered copy of database cloud (7 i) % 400 of run (n + 1) % 23: its true neighbour there.
    g = torch.Generator(device=H.device)
    g.manual_seed(4242)
    dbs = [torch.rand((n_db, N_POINTS, 3), generator=g, device=H.device) * 2.0 - 1.0 for _ in range(runs)]
    qs = []
    for n in range(runs):
        src = dbs[(n + 1) % runs][(7 * torch.arange(n_q, device=H.device)) % n_db]
        qs.append(src + 0.002 * torch.randn((n_q, N_POINTS, 3), generator=g, device=H.device))
    rng = np.random.RandomState(5)
    extra = {(m, n): [list(rng.choice(n_db, size=rng.randint(0, 3), replace=False)) for _ in range(n_q)]
             for m in range(runs) for n in range(runs) if m != n}

    def truth(m, n):
        t = extra[(m, n)]
        if m == (n + 1) % runs:
            return [[(7 * i) % n_db] + [v for v in t[i] if v != (7 * i) % n_db] for i in range(n_q)]
        return t

    extract = lambda chunk: eng.forward(chunk, check=False)
    prev = D.force_collective(bool(rccl))                     # a world of one still issues its RCCL collectives
    tms, result = [], {}
    import time as _time
    t_pack = _time.perf_counter()
    packed = R.pack_truth(truth, [n_db] * runs, [n_q] * runs).to(H.device)     # once per dataset (like loading the pickles)
    t_pack = _time.perf_counter() - t_pack
    try:
        def one(_k, vectors=False):
            tm = {}
            result["res"] = R.evaluate_sharded(extract, dbs, qs, packed, device=H.device, batch_size=64, timings=tm,
                                               return_vectors=vectors)
            tms.append(tm)
        for i in range(warmup):
            one(i)
        del tms[:]
        elapsed = H.timed(one, steps)
        timed_tms = list(tms)
        one(0, vectors=True)                                  # (untimed) the descriptors, for the float64 check below
        tms = timed_tms
    finally:
        D.force_collective(prev)
    total = runs * (n_db + n_q)
    out = None
    if rank == 0:
        res = result["res"]
        # the descriptors the flow produced, ranked by a float64 brute force on a sample (host, after the timed region): the
        # composed flow's neighbour lists must be the exact ones
        dbv, qv = res["database_vectors"], res["query_vectors"]
        top1 = []
        for n in range(0, runs, 5):
            m = (n + 1) % runs
            d = torch.cdist(torch.from_numpy(qv[n]).double(), torch.from_numpy(dbv[m]).double())
            top1.append(float((d.argmin(dim=1) == (7 * torch.arange(n_q)) % n_db).double().mean()))
        if min(top1) < 0.5:
            raise SystemExit("retrieval leg: the jittered copies are not their sources' nearest neighbours (%.3f): refusing to "
                             "report a number" % min(top1))
        ms = elapsed / steps * 1e3
        agg = lambda k: round(1e3 * sum(t[k] for t in tms) / len(tms), 3)
        gather_ms = agg("all_gather")
        out = {"workload": "Oxford-scale evaluate(): %d runs x (%d database + %d query) synthetic clouds x %d pts = %d clouds "
                           "extracted (EPC-Net, f32-equivalent arithmetic, batch 64) sharded over %d rank(s) -> one all-gather of the "
                           "256-d descriptors -> exact top-25 of every query against each other run's database, each rank "
                           "ranking AND booking its share of the queries on the device -> one all-reduce of the per-pair integer "
                           "counters (BASELINE.json configs[4]; evaluate.py:293-332, 351-537; retrieval.evaluate_sharded)" % (runs, n_db, n_q, N_POINTS, total, world),
               "rccl_exercised": bool(rccl), "backend": D.backend_name(), "rccl_ranks": world if rccl else 0, "ranks": world,
               "value": round(total * steps / elapsed, 1), "unit": "clouds/s evaluated end to end", "steps": steps,
               "ms_per_step": round(ms, 3),
               "phase_ms_rank0": {"extract": agg("extract"), "all_gather": gather_ms, "rank_book": agg("rank_book"),
                                  "reduce": agg("reduce"), "finish_host": agg("finish")},
               # what does NOT shrink with the number of ranks: the two collectives and the few numpy operations that turn the
               # reduced counters into the averages (extraction, ranking and booking are sharded)
               "serial_ms": round(gather_ms + agg("reduce") + agg("finish"), 3),
               "pack_truth_once_ms": round(t_pack * 1e3, 1),
               "all_gather_bytes": tms[-1]["all_gather_bytes"], "reduce_bytes": tms[-1]["reduce_bytes"],
               "all_gather_GBps": round(tms[-1]["all_gather_bytes"] * (world - 1) / max(world, 1) / (gather_ms * 1e-3) / 1e9, 3)
                                  if world > 1 and gather_ms > 0 else None,
               "ordered_pairs": runs * (runs - 1), "queries_ranked": runs * (runs - 1) * n_q,
               "jittered_copy_is_top1_float64_bruteforce": round(min(top1), 4),
               "ave_recall_at_1": round(float(res["ave_recall"][0]), 4),
               "ave_one_percent_recall": round(float(res["ave_one_percent_recall"]), 4),
               "average_similarity": round(float(res["average_similarity"]), 6),
               "note": "synthetic truth: query i of run n is a jittered copy of database cloud (7 i) %% 400 of run (n + 1) %% 23 "
                       "(its top-1 there, checked against a float64 brute force) plus random sets elsewhere -- the recall "
                       "figures check the bookkeeping, they are not Oxford Recall@1 %% (no dataset in this image)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--requested-gpus", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=400.0, help="extra untimed steps after the warm-up (0 = none)")
    ap.add_argument("--batch", type=int, default=64, help="clouds per step per GPU (configs[1]: 64)")
    ap.add_argument("--arch", default="epc-net", choices=["epc-net", "epc-net-l"])
    ap.add_argument("--precision", default="both", choices=["both", "f32", "fast"],
                    help="EPC-Net arithmetic(s) to time (include/epcnet.h EPC_PRECISION_*); `value` is always the f32-equivalent "
                         "one unless only `fast` is asked for (then the line says so in dtype)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="steps kept in flight on the engine's HIP streams (InferenceEngine.submit); 1 = one stream")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="record the stage-boundary HIP events on every n-th timed step (the events cost ~5 %% of a step)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--cpu-clouds", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the train_step / epc_net_l_b256 / retrieval legs")
    ap.add_argument("--no-rccl", action="store_true", help="N = 1 only: do not create the world-of-one RCCL process group")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend: nccl = RCCL (the product's); gloo = the dry run of the N > 1 code on a box with fewer GPUs")
    ap.add_argument("--same-device", action="store_true",
                    help="every rank on cuda:0 (with --backend gloo: RCCL refuses two ranks on one GPU); numbers from it are not credit")
    ap.add_argument("--regions", type=int, default=7,
                    help="fenced regions of --steps steps each; value / ms_per_step = the median region (min / max reported)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])                        # does not return

    # stdout carries ONE JSON line and nothing else.  Libraries write there behind Python's back (RCCL prints a five-line
    # version banner to the C stdout of every rank that creates a communicator), so file descriptor 1 is pointed at stderr for
    # the whole run and the line goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.same_device:
        local_rank = 0
        if world > 1:
            # Two ranks on ONE GPU: the persistent forward chain needs all its workgroups co-resident (one per CU) and two processes'
            # launches would hold each other's CUs -- each would sit in its first barrier until its spin budget ran out.  The dry run
            # keeps the launch chain; on real ranks (one GPU each) nothing shares the device with the forward.
            pkg("ops").CHAIN_PERSIST_FWD = False
    if args.backend == "nccl" and args.same_device and world > 1:
        raise SystemExit("--same-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # The process group is RCCL ("nccl") at every N -- at N = 1 too, a world of one rank (under torchrun --nproc-per-node 1 or
    # a plain `python bench.py`): the retrieval leg then drives its all-gather / index gather through RCCL on device tensors
    # exactly as the 8-GPU run does.  The extraction region's fences use the barrier only when there are ranks to wait for.
    import torch.distributed as tdist
    rccl, rccl_why = False, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 or not args.no_rccl:
        try:
            if args.backend == "nccl":
                tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
                rccl = True
            else:
                tdist.init_process_group("gloo", rank=rank, world_size=world)
                rccl_why = "--backend gloo: a dry run of the multi-rank code, not RCCL"
        except Exception as e:                                  # N = 1 only: the line then says so instead of dying
            if world > 1:
                raise
            rccl_why = "init_process_group(%r) failed at world 1: %r" % (args.backend, e)
    group_up = tdist.is_initialized()
    dist = tdist if world > 1 else None
    H = Harness(device, dist, world, rank)
    E = pkg("engine")
    every = max(1, args.profile_every)
    lanes = max(1, args.in_flight)

    store = build_store(args.arch, device, seed=0)             # same weights on every rank
    precisions = ["f32"] if args.arch == "epc-net-l" else (["f32", "fast"] if args.precision == "both" else [args.precision])
    legs = {}
    for prec in precisions:
        legs[prec], _ = extraction_leg(H, E, store, args.arch, prec, args.batch, args.steps, args.warmup, args.settle_ms,
                                       every, lanes, regions=max(1, args.regions))
    head_prec = precisions[0]
    head = legs[head_prec]

    configs = {}
    if not args.no_configs and args.arch == "epc-net":
        # The bounded legs may not take the headline down with them: a leg that raises is reported as {"error": ...} (on every rank the
        # same way: the legs' collectives raise on all ranks or on none).
        # A leg that raises is reported as {"error": ...} and the process exits non-zero AFTER the line is printed.  The
        # failure modes that matter raise before a leg's first collective or on every rank alike; the ranks then agree, through
        # one all-reduce of an ok flag behind every leg, on whether the leg counts (a rank that alone failed marks it failed everywhere).
        def guarded(name, fn):
            ok = True
            try:
                configs[name] = fn()
            except Exception as e:                              # noqa: BLE001  (reported on the line, not swallowed)
                import traceback
                ok = False
                configs[name] = {"error": repr(e), "where": traceback.format_exc().strip().splitlines()[-3:]}
                try:
                    import torch
                    torch.cuda.synchronize()
                except Exception:
                    pass
            if not H.all_ok(ok) and ok:
                configs[name] = {"error": "the leg failed on another rank"}

        tsteps = max(10, min(60, args.steps))
        guarded("train_step", lambda: train_step_leg(H, steps=tsteps, warmup=12, eager_steps=6))
        guarded("train_step_bf16", lambda: train_step_leg(H, steps=tsteps, warmup=12, precision="bf16"))
        # BASELINE.json configs[2] read literally: 1 query + 2 positives + 18 negatives (+ the other negative) = 22 clouds
        guarded("train_step_22", lambda: train_step_leg(H, steps=tsteps, warmup=12, n_neg=18))
        guarded("train_step_bf16_22", lambda: train_step_leg(H, steps=tsteps, warmup=12, precision="bf16", n_neg=18))

        def l_leg():
            stl = build_store("epc-net-l", device, seed=0)
            leg, _ = extraction_leg(H, E, stl, "epc-net-l", "f32", 256, max(10, min(50, args.steps)), min(args.warmup, 10), 0.0,
                                    every, 1)
            leg["workload"] = "EPC-Net-L inference, batch 256x4096x3 fp32 per GPU (BASELINE.json configs[3]; models/epc-net-l.py:29-102)"
            # `roofline` above is conv5 + max-pool's (the MFMA kernel); by TIME the leg is kNN + the two ProxyConv blocks (two thirds of
            # it), neither of which is MFMA- or HBM-bound: kNN saturates VALU issue (286.6 M vector instructions per launch = 0.98 of
            # the SIMDs' issue cycles), the blocks the vector L1 (gathered rows).  profiles/*_l_b256_pmc_compute.json (side collection).
            sm = leg["stage_ms"]
            tot = sum(sm.values())
            dom = max(sm, key=sm.get)
            leg["dominant_by_time"] = {
                "stage": dom, "ms": sm[dom], "share_of_step": round(sm[dom] / tot, 3),
                "knn_plus_blocks_share": round((sm["knn"] + sm["block1"] + sm["block2"]) / tot, 3),
                "knn_floor_ms": round(286.6e6 / 1024 * 4 / (VALU_GHZ * 1e9) * 1e3, 4),
                "knn_bound": "VALU issue: SQ_INSTS_VALU x 4 cycles / 1024 SIMDs (counters: 0.98 of the kernel's cycles)",
                "block_floor_ms_each": round((4 * 1.34e9 / (256 * 64) / (VALU_GHZ * 1e9) + 4 * 0.020e-3) * 1e3, 4),
                "block_bound": "vector L1: 20 gathered 256-byte rows per point at 64 B/clk/CU, plus the skeleton (as EPC-Net's blocks)"}
            # InferenceEngine.forward's DEFAULT for EPC-Net-L at this size: the 256 clouds as two 128-cloud halves in flight on two HIP
            # streams (the VALU-bound kNN of one half beside the L1-bound blocks / the matrix-bound conv5 of the other; engine.L_HALVES_FROM),
            # bit-identical descriptors (checked here).  `value` is THAT rate; the one-stream figures above it (stage_ms, roofline: the
            # kernels measured alone on the chip) stay under `one_stream`.
            import torch
            eng = E.InferenceEngine("epc-net-l", PARAMS, stl, outer=OUTER)
            ref = E.InferenceEngine("epc-net-l", PARAMS, stl, outer=OUTER, micro_batch=256, in_flight=1)
            g = torch.Generator(device="cpu")
            g.manual_seed(100 + H.rank)
            xyz = (torch.rand((256, N_POINTS, 3), generator=g) * 2.0 - 1.0).to(device)
            out = torch.empty((256, 256), dtype=torch.float32, device=device)
            ksteps = max(10, min(50, args.steps))
            for _ in range(4):
                eng.forward(xyz, out=out, check=False)
            el = median(H.timed_regions(lambda r, k: eng.forward(xyz, out=out, check=False), ksteps, 3))
            if not torch.equal(out, ref.forward(xyz)):
                raise SystemExit("EPC-Net-L: the two-halves pass disagrees with the one-stream descriptors: refusing to report a number")
            leg["one_stream"] = {"value": leg["value"], "ms_per_step": leg["ms_per_step"], "regions": leg.pop("regions")}
            leg["value"], leg["ms_per_step"] = round(H.world * 256 * ksteps / el, 2), round(el / ksteps * 1e3, 4)
            leg["pipeline_tflops"] = round(256 * ksteps / el * FLOPS_PER_CLOUD["epc-net-l"] / 1e12, 3)
            leg["how"] = ("InferenceEngine.forward's default for EPC-Net-L at >= 256 clouds: two 128-cloud halves in flight on two HIP streams, "
                          "same kernels, bit-identical descriptors; median of 3 regions of %d steps" % ksteps)
            leg["gain_over_one_stream"] = round(leg["value"] / leg["one_stream"]["value"], 4)
            return leg
        guarded("epc_net_l_b256", l_leg)
        guarded("retrieval", lambda: retrieval_leg(H, E, steps=3, warmup=1, rccl=rccl))
        if rank == 0 and rccl_why and isinstance(configs.get("retrieval"), dict):
            configs["retrieval"]["rccl_unavailable"] = rccl_why

    # HBM bytes of ONE step = the PMC bytes per launch of the pipeline's kernels (profiles/pmc_hbm_current.json, collected
    # by scripts/collect_profiles.sh on this configuration) x their launches per step: the figure behind "fraction of the
    # HBM roofline" (north_star); the path is MFMA / VALU-bound, so it is a secondary number.
    hbm_step = None
    pm = pmc_summary("pmc_hbm_current.json")
    if pm and args.arch == "epc-net" and args.batch == 64 and head_prec in pm.get("launches_per_step", {}):
        try:
            hbm_step = sum(v["hbm_bytes_per_launch_corrected"] * m for k, v in pm["kernels"].items()
                           for pre, m in pm["launches_per_step"][head_prec].items() if k.startswith(pre))
        except Exception:
            hbm_step = None

    if rank == 0:
        line = {
            "metric": "point-clouds/sec (4096 pts) descriptor extraction",
            "value": head["value"], "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
            "config": {"workload": "%s inference, batch %dx%dx3 fp32 per GPU, NetVLAD K=64, 256-D output "
                                   "(BASELINE.json configs[1])" % (args.arch, args.batch, N_POINTS),
                       "clouds_per_step_per_gpu": args.batch, "num_points": N_POINTS,
                       "precision": "EPC_PRECISION_%s (include/epcnet.h)" % head_prec.upper(),
                       "weights": "seeded random init of the architecture (no checkpoint payloads exist)",
                       "parallelism": "independent clouds sharded over %d GPU(s), no data-path collective" % world,
                       "stage_events_on_every_nth_step": every},
            "regions": head["regions"],
            "roofline": head["roofline"], "stage_ms": head["stage_ms"], "pipeline_tflops": head["pipeline_tflops"],
        }
        if "stage_floor_ms" in head:
            line["stage_floor_ms"], line["stage_floor"] = head["stage_floor_ms"], head["stage_floor"]
        if args.requested_gpus and args.requested_gpus != world:
            line["config"]["requested_gpus"] = args.requested_gpus
            line["config"]["note"] = "the box shows %d GPU(s): ran on those" % world
        if hbm_step:
            gbps = hbm_step / (head["ms_per_step"] * 1e-3) / 1e9
            line["pipeline_hbm"] = {"bytes_per_step": int(hbm_step), "achieved_GBps": round(gbps, 1), "peak_GBps": 8000.0,
                                    "frac": round(gbps / 8000.0, 4),
                                    # SURVEY.md 8d's stage-boundary model: 18.2 MB per cloud -- it assumes conv5, the assignment and
                                    # the aggregation fused, i.e. no feature map in HBM.  The map cannot go (the assignment needs a
                                    # point's whole 1024-channel row before the first aggregate product; recomputing conv5 instead
                                    # doubles the MFMA work) and 3-byte values are its narrowest form inside the 1e-4 bar
                                    # (profiles/r04_storage_format_numerics.txt): the ratio is a design limit, not waste to recover
                                    "model_bytes": int(18.2e6 * args.batch), "ratio": round(hbm_step / (18.2e6 * args.batch), 3),
                                    "ratio_is": "the 3-byte feature map's round trip (conv5 writes it, the aggregate reads it): a design limit",
                                    "how": "PMC bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/) x launches per step / ms_per_step"}
        elif args.arch == "epc-net" and args.batch == 64:
            line["pipeline_hbm"] = None
            line["pipeline_hbm_unavailable"] = _PMC_WHY.get("pmc_hbm_current.json", "no launch table for this arithmetic")
        line["lib_sha256"] = lib_sha256()
        if "overlapped" in head:
            line["overlapped"] = head["overlapped"]
        if "fast" in legs and head_prec != "fast":
            line["fast"] = legs["fast"]
        if configs:
            line["configs"] = configs
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.arch, store if args.arch != "epc-net" or not configs else build_store(args.arch, device, 0),
                                                args.cpu_budget_s, args.cpu_clouds)
        line["backend"] = {"name": args.backend if group_up else None, "rccl": bool(rccl), "ranks": world,
                           "same_device": bool(args.same_device)}
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if group_up:
        tdist.barrier()
        tdist.destroy_process_group()
    failed = [k for k, v in configs.items() if isinstance(v, dict) and "error" in v]
    if failed:
        sys.stderr.write("bench.py: legs failed: %s\n" % ", ".join(failed))
        raise SystemExit(3)


if __name__ == "__main__":
    main()
