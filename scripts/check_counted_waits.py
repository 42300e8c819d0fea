"""The kernels that stream an operand by LDS-DMA (h16_conv5_fwd, h16_df_tail, h32_conv5_fwd: DESIGN.md 4, "What the compiled code
showed" (ii)) wait for their pieces with a COUNTED `s_waitcnt vmcnt(N)` that leaves the N memory operations issued behind the request
in flight.  That is only right while the compiler really issues at least N such operations between the request and the wait.  This
script compiles the two files to gfx950 assembly here (hipcc cross-compiles without a GPU) and checks, for every hand-placed counted
wait, that the instructions between it and the closest preceding LDS-DMA request (walking back through the straight-line code and the
loop latch that precedes the wait) hold at least N stores / loads.  The walk is LINEAR in the file: where the compiler lays a loop's latch
block out in front of its body (h32_conv5_fwd_kernel today: 16 of the chunk's 32 stores sit in the latch, 16 at the end of the body) it
reports CHECK, and the control flow has to be followed by hand.  Run it after a compiler upgrade:
    python scripts/check_counted_waits.py"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "epc-net_amd", "csrc")
KERNELS = {"train_head16.hip": ("h16_conv5_fwd_kernel", "h16_df_tail_kernel"), "train_head32.hip": ("h32_conv5_fwd_kernel",)}
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for src, names in KERNELS.items():
        asm = os.path.join(tmp, src[:-4] + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=on",
                        "-fno-slp-vectorize", "-I" + csrc, "--cuda-device-only", "-S", os.path.join(csrc, src), "-o", asm], check=True,
                       stderr=subprocess.DEVNULL)
        text = open(asm).read()
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
            if not any(n in m.group(1) for n in names):
                continue
            body = [l.strip() for l in m.group(2).split("\n")]
            for k, line in enumerate(body):
                w = re.match(r"s_waitcnt vmcnt\((\d+)\)$", line)
                if not w or int(w.group(1)) == 0 or "ASMSTART" not in body[k - 1]:
                    continue
                n, ops, j = int(w.group(1)), 0, k - 1
                while j >= 0 and "global_load_lds" not in body[j]:
                    if re.match(r"(global|buffer|flat)_(store|load)", body[j]):
                        ops += 1
                    j -= 1
                ok = j >= 0 and ops >= n
                bad += not ok
                print("%-8s %s: vmcnt(%d) with %d memory operations behind the closest request above it" % ("ok" if ok else "CHECK", m.group(1)[:48], n, ops))
print("%d wait(s) to follow by hand" % bad if bad else "every counted wait is covered in file order")
