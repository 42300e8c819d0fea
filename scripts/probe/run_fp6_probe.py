"""GPU box: hipcc -O2 -fPIC --offload-arch=gfx950 -shared scripts/probe/mfma_fp6_probe.hip -o /tmp/fp6.so && python scripts/probe/run_fp6_probe.py"""
import ctypes, numpy as np, torch
lib = ctypes.CDLL("/tmp/fp6.so")
P = ctypes.c_void_p
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
out = torch.zeros(64 * 6, dtype=torch.int32, device=dev)
assert lib.run_onehot(P(out.data_ptr()), P(st)) == 0
torch.cuda.synchronize()
o = out.cpu().numpy().astype(np.uint32).reshape(64, 6)
ok = True
for j in range(32):
    bits = 0
    for w in range(6):
        bits |= int(o[j, w]) << (32 * w)
    pos = [b for b in range(192) if (bits >> b) & 1]
    # 1.0 in e2m3 = 0b001000: a single set bit, bit 3 of the element's 6-bit field
    elem = (pos[0] - 3) // 6 if len(pos) == 1 else None
    if elem != j:
        ok = False
        print("element", j, "set bits", pos)
print("conversion layout: element j at bits [6j, 6j+6), little-endian across the 6 dwords:", ok)

rng = np.random.default_rng(0)
A = rng.integers(-7, 8, size=(32, 64)).astype(np.float32)
B = rng.integers(-7, 8, size=(64, 32)).astype(np.float32)
tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
D = torch.zeros((32, 32), device=dev)
def run(sa, sb, opsel=0):
    a = torch.from_numpy(np.asarray(sa, dtype=np.int32)).to(dev)
    b = torch.from_numpy(np.asarray(sb, dtype=np.int32)).to(dev)
    assert lib.run_probe(P(tA.data_ptr()), P(tB.data_ptr()), P(D.data_ptr()), P(a.data_ptr()), P(b.data_ptr()), opsel, P(st)) == 0
    torch.cuda.synchronize()
    return D.cpu().numpy().copy()
ones = [127] * 64
d = run(ones, ones)
print("lane map (row/col = l & 31, k = 32 (l >> 5) + element) exact:", np.array_equal(d, A @ B))
# per-lane scales: lane l scales the k-block (l >> 5) of row l & 31
sa = [127 + (l & 1) + 2 * (l >> 5) for l in range(64)]
d = run(sa, ones)
ref = np.zeros((32, 32), dtype=np.float64)
for r in range(32):
    for h in range(2):
        ref[r] += (2.0 ** ((r & 1) + 2 * h)) * (A[r, 32 * h:32 * h + 32].astype(np.float64) @ B[32 * h:32 * h + 32])
print("scale_a is per lane = per (row, k-block):", np.array_equal(d, ref))
sb = [127 - (l & 3) - (l >> 5) for l in range(64)]
d = run(ones, sb)
ref = np.zeros((32, 32), dtype=np.float64)
for c in range(32):
    for h in range(2):
        ref[:, c] += (2.0 ** (-(c & 3) - h)) * (A[:, 32 * h:32 * h + 32].astype(np.float64) @ B[32 * h:32 * h + 32, c])
print("scale_b is per lane = per (column, k-block):", np.array_equal(d, ref))
# op_sel picks the byte of the scale dword
for sel in range(4):
    word = [(127 + 1) << (8 * sel) | sum(127 << (8 * q) for q in range(4) if q != sel) for _ in range(64)]
    word = [w - (1 << 32) if w >= (1 << 31) else w for w in word]
    d = run(word, [127 | 127 << 8 | 127 << 16 | 127 << 24] * 64, sel)
    print("op_sel", sel, "selects byte", sel, ":", np.array_equal(d, 2.0 * (A @ B)))
