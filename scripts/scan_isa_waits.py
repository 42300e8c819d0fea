"""Compile every kernel file to gfx950 assembly (device only) and list the kernels whose loads are waited for ALONE -- a load followed
within a few instructions by `s_waitcnt vmcnt(0)` before the next load is issued: the signature of a per-lane guarded load in an unrolled
loop ("What the compiled code showed").  Runs here (hipcc cross-compiles without a GPU):
    python scripts/scan_isa_waits.py [min_chain]"""
import glob, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "epc-net_amd", "csrc")
min_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out = []
with tempfile.TemporaryDirectory() as tmp:
    for src in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        asm = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=on",
                        "-fno-slp-vectorize", "-I" + csrc, "--cuda-device-only", "-S", src, "-o", asm], check=True,
                       stderr=subprocess.DEVNULL)
        kern, lines = None, []
        for line in open(asm):
            if re.match(r"^[_A-Za-z]\w*:", line) and not line.startswith(".L"):
                kern, lines = line.split(":")[0], []
                continue
            if kern is None:
                continue
            t = line.strip()
            if not t or t[0] in ";.":
                continue
            lines.append(t)
            if t.startswith("s_endpgm"):
                idx = [i for i, x in enumerate(lines) if re.match(r"(global|buffer|flat)_load", x)]
                chain = best = total = 0
                for a, b in zip(idx, idx[1:] + [len(lines)]):
                    w = [j for j, x in enumerate(lines[a + 1:b]) if "vmcnt(0)" in x]
                    if w and w[0] <= 8:
                        chain += 1
                        total += 1
                        best = max(best, chain)
                    else:
                        chain = 0
                if best >= min_chain:
                    out.append((best, total, len(idx), os.path.basename(src), kern))
                kern = None
for best, total, n, f, k in sorted(out, reverse=True):
    print("longest chain %3d   %4d of %4d loads waited for alone   %-22s %s" % (best, total, n, f, k[:110]))
