"""Test-side helpers: build a variable store from oracle weights, drive the stage entry points of the C ABI."""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import epcnet_oracle as O  # noqa: E402

PARAMS = {"CLUSTER_SIZE": 64, "FEATURE_OUTPUT_DIM": 256, "KNN": 20, "INPUT_DIM": 3, "GROUPS": 4}
OUTER = "query_triplets"


def pkg(name=""):
    return importlib.import_module("epc-net_amd" + ("." + name if name else ""))


def make_store(arch, weights, device):
    """Variable store holding `weights` (oracle names, relative to OUTER) under the reference's full names."""
    V = pkg("variables")
    st = V.reset_default_store(device=device, seed=0)
    M = pkg("models." + arch)
    with V.variable_scope(OUTER):
        M.declare_variables(PARAMS, 4096)
    st.load_state_dict({OUTER + "/" + k: v for k, v in weights.items()}, strict=True)
    return st


def make_engine(arch, weights, device="cuda", micro_batch=0, precision=None, in_flight=None):
    """`precision`: None = the PRODUCT default (engine.py: 'f32', the f32-equivalent split arithmetic) / 'f32' / 'fast' (EPC-Net's
    f16 + f6 kernels, an explicit opt-in; EPC-Net-L ignores it).  Tests whose subject is the fast path pass 'fast'."""
    E = pkg("engine")
    st = make_store(arch, weights, device)
    return E.InferenceEngine(arch, PARAMS, st, outer=OUTER, micro_batch=micro_batch, precision=precision, in_flight=in_flight), st


def run_stages(eng, xyz):
    """Run the pipeline stage by stage through the C ABI, returning every intermediate as a torch tensor.
    EPC-Net's block chain runs on fp16 rows (x16 -> out16 / x_next16), EPC-Net-L's on f32 rows: `xs` / `cat` are
    returned in the dtype the stage wrote."""
    L = pkg("lib")
    E = pkg("engine")
    lib = L.lib()
    nc, n, _ = xyz.shape
    cfg = eng.cfg_for(n)
    packed = eng.packed(cfg)
    f16 = eng.arch == "epc-net" and cfg.precision == L.EPC_PRECISION_FAST
    base = packed.data_ptr()
    off = lambda s: base + lib.epc_net_packed_offset(ctypes.byref(cfg), s)
    dev = xyz.device
    st = L.current_stream()
    M = nc * n
    out = {}
    status = torch.zeros((nc,), dtype=torch.int32, device=dev)
    idx = torch.empty((nc, n, L.EPC_KNN_CAP), dtype=torch.int32, device=dev)
    cnt = torch.empty((nc, n), dtype=torch.int32, device=dev)
    kth = torch.empty((nc, n), dtype=torch.float32, device=dev)
    L.check(lib.epc_knn_topk(xyz.data_ptr(), nc, n, L.EPC_KNN_CAP, idx.data_ptr(), cnt.data_ptr(), kth.data_ptr(), st))
    out.update(idx=idx, cnt=cnt, kth=kth)
    idx16 = idx.clamp(0, 32767).to(torch.int16)     # the pipeline's 2-byte list format (entries past cnt are never read)
    nblocks = 4 if eng.arch == "epc-net" else 2
    ccat = 64 * nblocks
    dt = torch.float16 if f16 else torch.float32
    xs = [torch.empty((nc, n, 64), dtype=dt, device=dev) for _ in range(nblocks + 1)]
    cat = torch.empty((nc, n, ccat), dtype=dt, device=dev)
    a32 = lambda t: None if f16 else t.data_ptr()
    a16 = lambda t: t.data_ptr() if f16 else None
    L.check(lib.epc_conv1_fwd(xyz.data_ptr(), off(0), M, a32(xs[0]), a16(xs[0]), st))
    for b in range(1, nblocks + 1):
        has_next = 1 if b < nblocks else 0
        # (lists: the int32 form for odd blocks, the pipeline's uint16 form for even ones -- both must give the same rows)
        u16 = (b % 2 == 0) and n <= 8192
        L.check(lib.epc_proxyconv_block_fwd(a32(xs[b - 1]), a16(xs[b - 1]), xyz.data_ptr(),
                                            (idx16 if u16 else idx).data_ptr(), 1 if u16 else 0,
                                            cnt.data_ptr(), kth.data_ptr(), L.EPC_KNN_CAP, off(b), has_next, nc, n,
                                            cfg.knn, a32(cat), a16(cat), ccat, 64 * (b - 1), a32(xs[b]), a16(xs[b]),
                                            status.data_ptr(), st))
    out.update(xs=xs, cat=cat)
    desc = torch.empty((nc, 256), dtype=torch.float32, device=dev)
    if eng.arch == "epc-net":
        rnorm = torch.empty((nc, n), dtype=torch.float32, device=dev)
        assign = torch.empty((nc, n, 64), dtype=torch.float32, device=dev)
        apart = torch.empty((nc, n // 32, 64), dtype=torch.float32, device=dev)
        vlad = torch.empty((nc, 1024, 64), dtype=torch.float32, device=dev)
        colss = torch.empty((nc, 32, 64), dtype=torch.float32, device=dev)
        if f16:
            featf = torch.empty((M // 32, 32, 2, 64, 8), dtype=torch.float16, device=dev)
            assignf = torch.empty((M // 32, 2, 2, 64, 8), dtype=torch.float16, device=dev)
            L.check(lib.epc_conv5_assign_fwd(cat.data_ptr(), 1, ccat, off(5), M, n, featf.data_ptr(), rnorm.data_ptr(),
                                             assign.data_ptr(), assignf.data_ptr(), apart.data_ptr(), status.data_ptr(), st))
            L.check(lib.epc_vlad_aggregate_fwd(featf.data_ptr(), assignf.data_ptr(), rnorm.data_ptr(), apart.data_ptr(),
                                               off(6), nc, n, vlad.data_ptr(), colss.data_ptr(), st))
            # unpack the fragment order (include/epcnet.h): [tile g][chunk c][half s][lane l][q] ->
            # feat[32g + (l&31)][32c + 16s + 8(q>>2) + 4(l>>5) + (q&3)]   (fp16: 11 significant bits)
            ff = featf.float().reshape(M // 32, 32, 2, 2, 32, 2, 4)        # (g, c, s, h, j, q>>2, q&3)
            feat = ff.permute(0, 4, 1, 2, 5, 3, 6).reshape(nc, n, 1024)    # (g, j, c, s, q>>2, h, q&3) -> point-major
            af = assignf.float().reshape(M // 32, 2, 2, 2, 32, 8) / 16384.0
            aprime = af.permute(0, 2, 3, 5, 1, 4).reshape(nc, n, 64) * rnorm.reshape(nc, n, 1)   # assign * rnorm
        else:
            featf = torch.empty((M // 32, 32, 3, 64, 16), dtype=torch.uint8, device=dev)
            assignf = torch.empty((M // 32, 2, 2, 2, 64, 8), dtype=torch.bfloat16, device=dev)
            L.check(lib.epc_conv5_assign_f32_fwd(cat.data_ptr(), ccat, off(5), M, featf.data_ptr(), rnorm.data_ptr(),
                                                 assign.data_ptr(), assignf.data_ptr(), apart.data_ptr(), st))
            L.check(lib.epc_vlad_aggregate_f32_fwd(featf.data_ptr(), assignf.data_ptr(), rnorm.data_ptr(),
                                                   apart.data_ptr(), off(6), nc, n, vlad.data_ptr(), colss.data_ptr(), st))
            # 3-byte values (include/epcnet.h): the 48 bytes of lane l of (tile g, chunk c) -- its three 16-byte pieces
            # concatenated -- are 16 little-endian values; with li = l & 15, q = l >> 4, value 4t + r (t = 2g2 + p) ->
            # feat[32g + 16p + li][32c + 16g2 + 4q + r]   (conv5_f32.hip's 16x16x32 accumulator order)
            by = featf.permute(0, 1, 3, 2, 4).reshape(M // 32, 32, 64, 16, 3).to(torch.int32)       # (g, c, l, value, byte)
            bits = (by[..., 0] << 8) | (by[..., 1] << 16) | (by[..., 2] << 24)
            ff = bits.view(torch.float32).reshape(M // 32, 32, 4, 16, 2, 2, 4)                       # (g, c, q, li, g2, p, r)
            feat = ff.permute(0, 5, 3, 1, 4, 2, 6).reshape(nc, n, 1024)                              # (g, p, li, c, g2, q, r)
            # [tile g][t][s][part][lane l][q] -> a[32g + 16s + 8(l>>5) + q][32t + (l&31)], hi + lo
            af = assignf.float().sum(3).reshape(M // 32, 2, 2, 2, 32, 8)   # (g, t, s, h, j, q)
            aprime = af.permute(0, 2, 3, 5, 1, 4).reshape(nc, n, 64) * rnorm.reshape(nc, n, 1)
        wsb = lib.epc_vlad_head_workspace_bytes(nc, cfg.groups)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        L.check(lib.epc_vlad_head_fwd(vlad.data_ptr(), colss.data_ptr(), off(6), cfg.groups, nc, desc.data_ptr(),
                                      status.data_ptr(), ws.data_ptr(), wsb, st))
        out.update(feat=feat, rnorm=rnorm, assign=assign, aprime=aprime, vlad=vlad, colss=colss, apart=apart)
    else:
        pooled = torch.empty((nc, 1024), dtype=torch.float32, device=dev)
        L.check(lib.epc_conv5_maxpool_fwd(cat.data_ptr(), ccat, off(5), nc, n, pooled.data_ptr(), st))
        L.check(lib.epc_fc_head_fwd(pooled.data_ptr(), off(6), nc, desc.data_ptr(), status.data_ptr(), st))
        out.update(pooled=pooled)
    out["desc"] = desc
    out["status"] = status
    torch.cuda.synchronize()
    return out
