"""Reduce the rocprofv3 CSVs written by scripts/collect_profiles.sh to the small summaries kept under profiles/:
   profiles/<tag>_kernel_stats.csv   per-kernel calls / total / average duration (from --kernel-trace --stats)
   profiles/<tag>_pmc_hbm.json       per-kernel FETCH_SIZE / WRITE_SIZE means and the corrected HBM bytes per launch
                                      (gfx950: FETCH_SIZE counts 128-B requests as 64 B => x2; WRITE_SIZE exact;
                                      MI355X_MICROARCH.md, HBM section)
   profiles/<tag>_bench_line.json    the un-profiled bench line of the same command
and refresh profiles/pmc_hbm_current.json (what bench.py's roofline.traffic reads)."""
import csv, glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "cur"
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
           "tag": tag, "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024)", "kernels": kernels}
    for name in (tag + "_pmc_hbm.json", "pmc_hbm_current.json"):
        with open(os.path.join(dst, name), "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
line = os.path.join(src, "bench_line.json")
if os.path.exists(line) and os.path.getsize(line):
    shutil.copy(line, os.path.join(dst, tag + "_bench_line.json"))
print("summaries written for", tag, "kernels with counters:", len(kernels), "stats:", bool(stats))
