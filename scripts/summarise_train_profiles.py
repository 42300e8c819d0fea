"""Reduce the rocprofv3 CSVs of scripts/collect_train_profiles.sh to profiles/<tag>_train_pmc.json and refresh
profiles/pmc_train_current.json (read by bench.py's train legs while its lib_sha256 stamp is the loaded library's).

Per leg (<arithmetic>_<clouds>): steps profiled, launches per step, HBM bytes per step = sum over the step's dispatches of
2 x FETCH_SIZE + WRITE_SIZE (KB x 1024; gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes, MI355X_MICROARCH.md, HBM section),
the ALGORITHMIC bytes of the same step (model_bytes_per_step: every tensor a kernel of the step has to read or write, once, at its
stored width -- the table in DESIGN.md 4, "Training step"), and for the five kernels with the most time: average duration, launches
per step, bytes per launch, the matrix pipe's busy share (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 128 SIMDs))."""
import csv, glob, hashlib, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "cur"
src = os.path.join(root, "gpurun_out", "prof_train_" + tag)
dst = os.path.join(root, "profiles")
STEPS = 3 + 5      # WARM + STEPS of scripts/collect_train_profiles.sh: every profiled step is the same step


def lib_sha256():
    h = hashlib.sha256()
    with open(os.path.join(root, "epc-net_amd", "libepcnet_hip.so"), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def find(leg, sub, suffix):
    hits = sorted(glob.glob(os.path.join(src, leg, sub, "**", "*" + suffix), recursive=True))
    return hits[-1] if hits else None


def counter_sums(leg, sub, counter):
    """{kernel: (sum of the counter over its dispatches, dispatches)}"""
    path = find(leg, sub, "counter_collection.csv")
    acc = {}
    if not path:
        return acc
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            s, n = acc.get(row["Kernel_Name"], (0.0, 0))
            acc[row["Kernel_Name"]] = (s + float(row["Counter_Value"]), n + 1)
    return acc


def model_bytes(prec, ncl, n=4096):
    """Algorithmic bytes of one step: the tensors each launch must move once.  R = rows."""
    R = ncl * n
    f4 = 4
    w = 2 if prec == "bf16" else 4                       # stored width of the (rows, 1024) tensors and of the concat copy
    r64, r256, r1024 = R * 64, R * 256, R * 1024
    # backbone chain (11 layers of 64 channels, f32 tensors in both arithmetics): per forward linear launch read z + write z (+ xm, cat
    # slice on the block heads); per gather launch read the lists + z0, write xm, d, za; backward the mirror image with dy / dx
    # (round 6: the forward is ONE persistent launch -- the input's moments read z0 once; per block the gather reads the lists and z0 and
    # writes d and za, conv_b writes zb, the block's head writes its slice of the concat and the next z0: za, zb and the neighbour mean
    # are handed on in LDS / registers and never re-read; the launch chain it replaced moved 40 (rows, 64) tensors, this one 24)
    chain_fwd = r64 * f4 + 4 * (3 * r64 * f4 + R * 32 * 4) + 4 * (r64 * f4) + 4 * (r64 * f4) + 3 * (r64 * f4)
    chain_bwd = 11 * (4 * r64 * f4) + 4 * (3 * r64 * f4 + R * 32 * 4)
    cat16 = r256 * 2 if prec == "bf16" else 0
    if prec == "bf16":
        head_fwd = (r256 * 2 + r1024 * w) + (r1024 * w + r64 * f4) + 2 * r64 * f4 + (r1024 * w + r64 * f4)
        head_bwd = ((r1024 * w + r64 * f4) + 5 * r64 * f4 + (r1024 * w + r64 * f4) + (2 * r1024 * w + 2 * r64 * f4) + 3 * r1024 * w +
                    (r1024 * w + r256 * f4) + (r1024 * w + r256 * 2))
    else:
        # f32 tensors, the feature map f materialised: conv5 (cat -> z5), z5 -> f, f -> za, a, f -> vlad; backward: f -> da, softmax,
        # f -> dWc, (a, dz, z5) -> du, (du, z5) -> dz5, dz5 -> dcat, (cat, dz5) -> dW5
        head_fwd = (r256 * f4 + r1024 * f4) + 2 * r1024 * f4 + (r1024 * f4 + r64 * f4) + 2 * r64 * f4 + (r1024 * f4 + r64 * f4)
        head_bwd = ((r1024 * f4 + r64 * f4) + 5 * r64 * f4 + (r1024 * f4 + r64 * f4) + (2 * r1024 * f4 + 2 * r64 * f4) + 3 * r1024 * f4 +
                    (r1024 * f4 + r256 * f4) + (r1024 * f4 + r256 * f4))
    knn = R * 12 * 2 + R * 32 * 4 * 3                    # sort, kNN lists, transposed lists
    weights = 3 * 4 * 4.7e6 * 2                          # Adam: w, m, v read and written; gradients
    return int(chain_fwd + chain_bwd + cat16 + head_fwd + head_bwd + knn + weights)


doc = {"tag": tag, "lib_sha256": lib_sha256(), "steps_profiled": STEPS,
       "command": "rocprofv3 --kernel-trace [--pmc <C>] --output-format csv -- python3 scripts/time_train_step.py with GRAPH=0 WARM=3 "
                  "STEPS=5 PRECISION=<arithmetic> NEG=<negatives> in the environment (scripts/collect_train_profiles.sh)",
       "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024)", "legs": {}}
for leg_dir in sorted(glob.glob(os.path.join(src, "*_*"))):
    if not os.path.isdir(leg_dir):
        continue
    leg = os.path.basename(leg_dir)
    prec, ncl = leg.rsplit("_", 1)
    fetch, write = counter_sums(leg, "fetch", "FETCH_SIZE"), counter_sums(leg, "write", "WRITE_SIZE")
    busy, gui = counter_sums(leg, "mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), counter_sums(leg, "mfma", "GRBM_GUI_ACTIVE")
    if not fetch or not write:
        continue
    stats = {}
    path = find(leg, "stats", "kernel_stats.csv")
    if path:
        with open(path) as f:
            for row in csv.DictReader(f):
                stats[row["Name"]] = (int(row["Calls"]), float(row["TotalDurationNs"]))
    kernels, total = {}, 0.0
    for k in sorted(set(fetch) | set(write)):
        fs, fn = fetch.get(k, (0.0, 0))
        ws, wn = write.get(k, (0.0, 0))
        n = max(fn, wn)
        if n < STEPS:
            continue                                    # (one-off launches: initialisation, not the step)
        b = (2.0 * fs + ws) * 1024
        total += b / STEPS
        e = {"launches_per_step": round(n / STEPS, 2), "hbm_bytes_per_launch": int(b / n)}
        if k in stats and stats[k][0] > 0:
            e["avg_us"] = round(stats[k][1] / stats[k][0] / 1e3, 1)
            e["us_per_step"] = round(stats[k][1] / STEPS / 1e3, 1)
        if k in busy and k in gui and gui[k][0] > 0:
            e["mfma_util"] = round(busy[k][0] / (gui[k][0] / 8.0 * 128 * 8), 4)   # busy cycles summed over 1024 SIMDs / (cycles x 1024)
        kernels[k] = e
    top = sorted(kernels.items(), key=lambda kv: -kv[1].get("us_per_step", 0.0))[:5]
    mb = model_bytes(prec, int(ncl))
    doc["legs"][leg] = {"hbm_bytes_per_step": int(total), "model_bytes_per_step": mb, "traffic_over_model": round(total / mb, 3),
                        "launches_per_step": round(sum(v["launches_per_step"] for v in kernels.values()), 1),
                        "largest_kernels": [dict(kernel=k[:96], **v) for k, v in top], "kernels": kernels}
os.makedirs(dst, exist_ok=True)
for name in (tag + "_train_pmc.json", "pmc_train_current.json"):
    with open(os.path.join(dst, name), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
print(json.dumps({k: {kk: v[kk] for kk in ("hbm_bytes_per_step", "model_bytes_per_step", "traffic_over_model", "launches_per_step")}
                  for k, v in doc["legs"].items()}, indent=1))
