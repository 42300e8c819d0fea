"""Reduce the rocprofv3 CSVs written by scripts/collect_profiles.sh to the small summaries kept under profiles/:
   profiles/<tag>_kernel_stats.csv   per-kernel calls / total / average duration (from --kernel-trace --stats)
   profiles/<tag>_pmc_hbm.json       per-kernel FETCH_SIZE / WRITE_SIZE means and the corrected HBM bytes per launch
                                      (gfx950: FETCH_SIZE counts 128-B requests as 64 B => x2; WRITE_SIZE exact;
                                      MI355X_MICROARCH.md, HBM section)
   profiles/<tag>_bench_line.json    the un-profiled bench line of the same command
   profiles/<tag>_pmc_compute.json   per-kernel means of the SQ / GRBM counters of the two compute passes and the ratios
                                      derived from them (mfma_util, VALU active-lane fraction, LDS conflict share, wait shares)
and refresh profiles/pmc_hbm_current.json / pmc_compute_current.json (what bench.py's roofline.traffic / mfma_util read)."""
import csv, glob, json, os, shutil, sys

import hashlib

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "cur"


def lib_sha256():
    """The library the profiled runs loaded (scripts/collect_profiles.sh runs this script right after them, on the same box,
    from the same tree): bench.py prints the counters only while the library it loads still has this hash."""
    h = hashlib.sha256()
    with open(os.environ.get("EPCNET_LIB") or os.path.join(root, "epc-net_amd", "libepcnet_hip.so"), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


LIB_SHA = lib_sha256()       # (no override: a summary carries the hash of the library in the tree it was made from, nothing else)
SIDE = len(sys.argv) > 2 and sys.argv[2] == "side"   # a side collection (another arch / batch): per-tag files only, *_current.json untouched
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(sub, suffix):
    hits = sorted(glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True))
    return hits[-1] if hits else None


stats = find("stats", "kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))


def counter_means(sub, counter):
    path = find(sub, "counter_collection.csv")
    acc = {}
    if not path:
        return acc
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            s, n = acc.get(k, (0.0, 0))
            acc[k] = (s + float(row["Counter_Value"]), n + 1)
    return {k: (s / n, n) for k, (s, n) in acc.items()}


fetch = counter_means("pmc_fetch", "FETCH_SIZE")
write = counter_means("pmc_write", "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f_kb, n = fetch.get(k, (0.0, 0))
    w_kb, n2 = write.get(k, (0.0, 0))
    kernels[k] = {"FETCH_SIZE_KB_mean": round(f_kb, 1), "WRITE_SIZE_KB_mean": round(w_kb, 1),
                  "hbm_bytes_per_launch_corrected": int(round((2.0 * f_kb + w_kb) * 1024)), "launches": max(n, n2)}
if kernels:
    doc = {"command": "rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 bench.py --steps 5 --warmup 2 "
                      "--no-cpu-baseline (separate passes for FETCH_SIZE and WRITE_SIZE; scripts/collect_profiles.sh)",
           "tag": tag, "lib_sha256": LIB_SHA, "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024)", "kernels": kernels}
    for name in (tag + "_pmc_hbm.json",) + (() if SIDE else ("pmc_hbm_current.json",)):
        with open(os.path.join(dst, name), "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
# ---- compute-side counters -------------------------------------------------------------------------------------------
# the launches of one bench step, per arithmetic (bench.py's pipeline_hbm sums bytes over them)
LAUNCHES = {"f32": {"morton_sort_kernel": 1, "void knn_topk_quad_kernel<20, true, 4>": 1, "proxyconv_block_kernel": 4,
                    "void conv5_vlad_f32_kernel<256>": 1, "vlad_aggregate_f32_kernel": 1, "void vlad_fold_kernel": 1,
                    "hidden_gemm_kernel": 1, "head_finish_kernel": 1},
            "fast": {"morton_sort_kernel": 1, "void knn_topk_quad_kernel<20, true, 4>": 1, "proxyconv_block_f16_kernel": 4,
                     "void conv5_kernel<256, 0, true, true>": 1, "vlad_aggregate_kernel": 1, "void vlad_fold_kernel": 1,
                     "hidden_gemm_kernel": 1, "head_finish_kernel": 1}}
for name in (tag + "_pmc_hbm.json",) + (() if SIDE else ("pmc_hbm_current.json",)):
    path = os.path.join(dst, name)
    if kernels and os.path.exists(path) and not SIDE:
        doc = json.load(open(path))
        doc["launches_per_step"] = LAUNCHES
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)


def all_counter_means(sub):
    path = find(sub, "counter_collection.csv")
    acc = {}
    if not path:
        return acc
    with open(path) as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"], row.get("Counter_Name"))
            s, n = acc.get(key, (0.0, 0))
            acc[key] = (s + float(row["Counter_Value"]), n + 1)
    out = {}
    for (k, c), (s, n) in acc.items():
        out.setdefault(k, {})[c] = s / n
        out[k]["launches"] = n
    return out


# GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (a 0.512-ms conv5 launch reads 8.37 M = 8 x 1.05 M cycles at ~2.04 GHz), the SQ
# counters summed over all SIMDs: busy cycles per SIMD = counter / 1024, kernel cycles = GUI_ACTIVE / 8
NUM_SIMD = 256 * 4 // 8  # SIMDs per unit of GRBM_GUI_ACTIVE
comp = {}
for sub in ("pmc_compA", "pmc_compB"):
    for k, d in all_counter_means(sub).items():
        comp.setdefault(k, {}).update(d)
for k, d in comp.items():
    g = d.get("GRBM_GUI_ACTIVE")
    r = {}
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is busy, summed over the SIMDs (32 per v_mfma_f32_32x32x16_*:
        # MI355X_MICROARCH.md, cycle constants); GRBM_GUI_ACTIVE: the kernel's duration in shader cycles
        r["mfma_util"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * NUM_SIMD), 4)
    if d.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in d:
        # both in quad-cycles: thread-cycles / (instruction-cycles x 64 lanes) = the share of lanes doing work per VALU issue
        r["valu_active_lane_fraction"] = round(d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
    if d.get("SQ_WAVE_CYCLES"):
        w = d["SQ_WAVE_CYCLES"]
        for c, nm in (("SQ_ACTIVE_INST_VALU", "valu_issue_share_of_wave_cycles"), ("SQ_WAIT_ANY", "wait_any_share"),
                      ("SQ_WAIT_INST_ANY", "wait_inst_share"), ("SQ_ACTIVE_INST_ANY", "active_inst_share")):
            if c in d:
                r[nm] = round(d[c] / w, 4)
    if d.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in d:
        r["lds_bank_conflict_share"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 4)
    d.update(r)
if comp:
    sets = ""
    try:
        sets = open(os.path.join(src, "pmc_compute_sets.txt")).read()
    except Exception:
        pass
    doc = {"command": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py --steps 5 --warmup 2 "
                      "--no-cpu-baseline --in-flight 1 --no-configs (two passes; scripts/collect_profiles.sh)",
           "tag": tag, "lib_sha256": LIB_SHA, "counter_sets": sets,
           "derived": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); valu_active_lane_fraction = "
                      "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); *_share = counter / SQ_WAVE_CYCLES; means per launch",
           "kernels": {k: {c: (round(v, 1) if isinstance(v, float) and v > 10 else v) for c, v in d.items()} for k, d in sorted(comp.items())}}
    for name in (tag + "_pmc_compute.json",) + (() if SIDE else ("pmc_compute_current.json",)):
        with open(os.path.join(dst, name), "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
line = os.path.join(src, "bench_line.json")
if os.path.exists(line) and os.path.getsize(line):
    shutil.copy(line, os.path.join(dst, tag + "_bench_line.json"))
print("summaries written for", tag, "kernels with HBM counters:", len(kernels), "with compute counters:", len(comp), "stats:", bool(stats))
