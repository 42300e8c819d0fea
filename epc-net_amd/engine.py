"""Inference engine: owns the packed (BN-folded, MFMA-ordered) weights and the workspace, and drives
``epc_net_forward`` of libepcnet_hip.so.  This is what ``forward(..., is_training=False)`` of the model
modules runs (reference: ``sess.run`` of ``models/epc-net.py:29-157`` / ``models/epc-net-l.py:29-102``).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import lib as L
from .variables import VariableStore

ARCH_IDS = {"epc-net": L.EPC_ARCH_EPC_NET, "epc-net-l": L.EPC_ARCH_EPC_NET_L}

DEFAULT_PARAMS = {"CLUSTER_SIZE": 64, "FEATURE_OUTPUT_DIM": 256, "KNN": 20, "INPUT_DIM": 3, "GROUPS": 4}
# Arithmetic of the inference path when neither the engine nor params["PRECISION"] names one (include/epcnet.h:
# EPC_PRECISION_*).  "f32" = the reference's class of arithmetic (float32 graph, models/epc-net.py:24-26) on split
# half-precision MFMA; "fast" = EPC-Net's f16 + f6 path: an explicit opt-in for checkpoints it has been validated on
# (tests/test_gpu_adversarial.py: it is wrong by 1e-2 on heavy-tailed weights while staying inside fp16's range, so
# nothing selects it automatically).
DEFAULT_PRECISION = "f32"
# EPC-Net-L (models/epc-net-l.py:29-102): batches of L_HALVES_FROM clouds or more run as L_HALF_BATCH-cloud passes on two HIP streams
L_HALVES_FROM = 256
L_HALF_BATCH = 128


def resolve_precision(params: Optional[dict], precision: Optional[str] = None) -> str:
    name = precision or (params or {}).get("PRECISION") or DEFAULT_PRECISION
    name = str(name).lower()
    if name not in ("f32", "fast"):
        raise ValueError("PRECISION must be 'f32' or 'fast', got %r" % (name,))
    return name


def make_cfg(arch: str, num_points: int, params: Optional[dict], micro_batch: int = 0,
             precision: Optional[str] = None) -> L.EpcCfg:
    p = dict(DEFAULT_PARAMS)
    p.update(params or {})
    if arch not in ARCH_IDS:
        raise ValueError("unknown ARCH %r" % arch)
    prec = resolve_precision(p, precision)
    return L.EpcCfg(arch=ARCH_IDS[arch], num_points=int(num_points), input_dim=int(p["INPUT_DIM"]),
                    knn=int(p["KNN"]), cluster_size=int(p["CLUSTER_SIZE"]), output_dim=int(p["FEATURE_OUTPUT_DIM"]),
                    groups=int(p.get("GROUPS", 4)), micro_batch=int(micro_batch), precision=L.PRECISION_IDS[prec])


class InferenceEngine:
    def __init__(self, arch: str, params: Optional[dict], store: VariableStore, outer: str = "query_triplets",
                 micro_batch: int = 0, backbone_scope: str = "fastdgcnn", in_flight: Optional[int] = None,
                 precision: Optional[str] = None):
        self.arch = arch
        self.precision = resolve_precision(params, precision)   # 'f32' | 'fast'
        self.resolved_precision: Optional[str] = None          # what the packed weights hold
        # passes kept in flight on separate HIP streams (submit / long calls).  Default: two in the `fast` arithmetic (+12 %),
        # ONE in the f32-equivalent one -- its kernels fill every CU's registers and LDS (the persistent block kernel the whole
        # chip): a second lane only delays the first (bench.py `overlapped`: 55.6 k against 57.5 k clouds/s)
        # EPC-Net-L: two lanes as well -- a batch of 256 clouds or more goes as 128-cloud halves in flight (the VALU-bound kNN of one
        # half beside the L1-bound blocks / the matrix-bound conv5 of the other: +3-4 %, the same bits; `_forward`)
        if in_flight is None:
            in_flight = 2 if ((self.precision == "fast" and arch == "epc-net") or arch == "epc-net-l") else 1
        self.in_flight = max(1, min(int(in_flight), 8))
        self._lanes = None                                 # [(torch.cuda.Stream, workspace tensor or None, last event or None)]
        self._next_lane = 0
        self.backbone_scope = backbone_scope   # 'BACKBONE' for the KD student (models/kd_epc-net-l.py:44)
        self.params = dict(params or {})
        self.store = store
        self.outer = outer
        self.micro_batch = micro_batch
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._ws: Optional[torch.Tensor] = None
        self._last_cfg: Optional[L.EpcCfg] = None
        self._last_overlap = None                          # (cfg, lanes, bytes per lane slice, clouds) of a one-pass-per-lane overlapped call
        self._fallback: "Optional[InferenceEngine]" = None   # f32-equivalent engine for clouds the fast path flags

    # ------------------------------------------------------------------------------------------------------
    def _relative_tensors(self) -> Dict[str, torch.Tensor]:
        prefix = self.outer + "/" if self.outer else ""
        out = {}
        for name, t in self.store.vars.items():
            if prefix and not name.startswith(prefix):
                continue
            rel = name[len(prefix):]
            if self.backbone_scope != "fastdgcnn" and rel.startswith(self.backbone_scope + "/"):
                rel = "fastdgcnn/" + rel[len(self.backbone_scope) + 1:]   # the library looks variables up by the EPC-Net names
            out[rel] = t
        return out

    def cfg_for(self, num_points: int) -> L.EpcCfg:
        """The epc_cfg of a call on clouds of ``num_points`` points, in the precision the packed weights hold."""
        return make_cfg(self.arch, num_points, self.params, self.micro_batch, precision=self.precision)

    def packed(self, cfg: L.EpcCfg) -> torch.Tensor:
        # keyed on the version of THIS model's variables: a frozen teacher that shares the store with a student being
        # trained (kd_train.py:467-497) is not re-folded every step
        key = (self.store.version_of(self.outer), cfg.num_points, cfg.groups, cfg.knn, self.precision)
        if self._packed is not None and self._packed_key == key:
            cfg.precision = L.PRECISION_IDS[self.resolved_precision]
            return self._packed
        L.require_gpu()
        nbytes = L.lib().epc_net_packed_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            raise L.EpcNetError(-1, "unsupported configuration (see include/epcnet.h: epc_cfg)")
        rel = self._relative_tensors()
        names = list(rel.keys())
        tensors = [rel[n] for n in names]
        for t in tensors:
            if not t.is_cuda:
                raise L.EpcNetError(-1, "variables must live on a ROCm device")
        c_names = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        c_ptrs = (ctypes.c_void_p * len(names))(*[t.data_ptr() for t in tensors])
        buf = torch.empty(nbytes, dtype=torch.uint8, device=tensors[0].device)
        rc = L.lib().epc_net_pack_weights(ctypes.byref(cfg), c_names, c_ptrs, len(names), buf.data_ptr(), nbytes,
                                          L.current_stream())
        L.check(rc)     # (fast precision: EPC_ERANGE when a folded weight does not fit fp16 -- never a packed Inf)
        if self._packed is not None:
            # batches submitted on the engine's lanes may still be reading the previous buffer: keep it alive until their
            # streams have passed this point (the caching allocator would otherwise hand it out again)
            for lane in self._lanes or []:
                self._packed.record_stream(lane[0])
        self.resolved_precision = "fast" if cfg.precision == L.EPC_PRECISION_FAST else "f32"
        self._packed, self._packed_key = buf, key
        return buf

    def workspace(self, cfg: L.EpcCfg, num_clouds: int, device) -> torch.Tensor:
        need = L.lib().epc_net_workspace_bytes(ctypes.byref(cfg), num_clouds)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def forward(self, xyz: torch.Tensor, out: Optional[torch.Tensor] = None, profile: "Optional[StageProfile]" = None,
                check: Optional[bool] = None) -> torch.Tensor:
        """xyz (num_clouds, N, 3) float32 on the GPU -> (num_clouds, FEATURE_OUTPUT_DIM).
        ``profile``: record HIP events at the stage boundaries of this pass (same launches, same stream).
        ``check`` (fast precision only, default off): look at the result (this synchronises) and re-extract in the
        f32-equivalent arithmetic every cloud the kernels flagged (NaN descriptor: an activation left fp16's range, include/
        epcnet.h EPC_STATUS_FP16_RANGE -- e.g. the zero-padding clouds of evaluate.py:425-430).  A cloud that is still NaN
        afterwards has a NaN / Inf coordinate -- the reference returns NaN for it as well."""
        out = self._forward(xyz, out, profile)
        if check and self.arch == "epc-net" and self.resolved_precision == "fast":
            bad = torch.isnan(out).any(dim=1).nonzero().flatten()
            if bad.numel():
                if self._fallback is None:
                    self._fallback = InferenceEngine(self.arch, self.params, self.store, outer=self.outer,
                                                     micro_batch=self.micro_batch, backbone_scope=self.backbone_scope,
                                                     in_flight=1, precision="f32")
                out[bad] = self._fallback._forward(xyz.contiguous()[bad], None, None)
        return out

    def _forward(self, xyz: torch.Tensor, out: Optional[torch.Tensor], profile: "Optional[StageProfile]") -> torch.Tensor:
        if xyz.dim() != 3 or xyz.shape[-1] != 3:
            raise L.EpcNetError(-1, "expected (num_clouds, N, 3) points, got %s" % (tuple(xyz.shape),))
        if xyz.dtype != torch.float32:
            raise L.EpcNetError(-1, "points must be float32")
        L.require_gpu()
        xyz = xyz.contiguous()
        nc, n = int(xyz.shape[0]), int(xyz.shape[1])
        cfg = self.cfg_for(n)
        packed = self.packed(cfg)
        if out is None:
            out = torch.empty((nc, cfg.output_dim), dtype=torch.float32, device=xyz.device)
        if self.arch == "epc-net-l" and self.micro_batch == 0 and self.in_flight > 1 and profile is None and nc >= L_HALVES_FROM:
            cfg.micro_batch = L_HALF_BATCH          # (passes of 128 clouds dealt over the lanes: results do not depend on the pass size)
        mb = L.micro_batch_of(cfg, nc)
        if profile is None and self.in_flight > 1 and nc > mb:
            # several passes: deal them over this stream and the auxiliary lanes (epc_net_forward_overlapped)
            lanes = min(self.in_flight, (nc + mb - 1) // mb)
            need = L.lib().epc_net_workspace_bytes(ctypes.byref(cfg), nc) * lanes
            if self._ws is None or self._ws.numel() < need or self._ws.device != xyz.device:
                self._ws = torch.empty(need, dtype=torch.uint8, device=xyz.device)
            aux = [ln[0] for ln in self._get_lanes(xyz.device)[:lanes - 1]]
            arr = (ctypes.c_void_p * max(len(aux), 1))(*[a.cuda_stream for a in aux])
            # several lanes, several workspace slices: last_status() can answer only while no slice has been reused (one pass per lane)
            self._last_cfg = None
            self._last_overlap = (cfg, lanes, L.lib().epc_net_workspace_bytes(ctypes.byref(cfg), nc), nc) if (nc + mb - 1) // mb <= lanes else None
            L.check(L.lib().epc_net_forward_overlapped(ctypes.byref(cfg), packed.data_ptr(), L.ptr(xyz), nc, L.ptr(out),
                                                       self._ws.data_ptr(), self._ws.numel(), L.current_stream(), arr,
                                                       len(aux)))
            return out
        ws = self.workspace(cfg, max(nc, 1), xyz.device)
        self._last_cfg = cfg
        self._last_overlap = None
        L.check(L.lib().epc_net_forward_profiled(ctypes.byref(cfg), packed.data_ptr(), L.ptr(xyz), nc, L.ptr(out),
                                                 ws.data_ptr(), ws.numel(), L.current_stream(),
                                                 profile.handle if profile is not None else None))
        return out


    def last_status(self, xyz_or_count) -> "list[int]":
        """Per-cloud EPC_STATUS_* words of the last pass of the most recent ``forward`` call on this engine's own
        workspace (include/epcnet.h: epc_net_last_status; synchronises the stream).  A non-zero word = that cloud's
        descriptor is NaN: bit 0 a NaN / Inf coordinate, bit 1 (fast precision) an activation outside fp16's range.
        Raises after a call that was dealt over several lanes (see below) -- never returns another call's words."""
        nc = int(xyz_or_count.shape[0]) if torch.is_tensor(xyz_or_count) else int(xyz_or_count)
        if nc <= 0:
            return []
        ov = getattr(self, "_last_overlap", None)
        if self._ws is not None and self._last_cfg is None and ov is not None and ov[3] == nc:
            # the last call went as one pass per lane (EPC-Net-L's two halves): every pass's words are still in its lane's slice
            cfg, lanes, slice_bytes, _ = ov
            mb = L.micro_batch_of(cfg, nc)
            words = []
            for p in range((nc + mb - 1) // mb):
                n_p = min(mb, nc - p * mb)
                arr = (ctypes.c_int32 * n_p)()
                L.check(L.lib().epc_net_last_status(ctypes.byref(cfg), self._ws.data_ptr() + (p % lanes) * slice_bytes, n_p, arr,
                                                    L.current_stream()))
                words += [int(v) for v in arr]
            return words
        if self._ws is None or self._last_cfg is None:
            # nothing ran yet, or the last call was dealt over the engine's lanes (num_clouds > micro_batch with in_flight > 1:
            # epc_net_forward_overlapped keeps one workspace slice per lane, so "the last pass's status words" do not exist in
            # one place): say so instead of returning stale words.  The NaN descriptor marks a flagged cloud in
            # every mode; for status words run the engine with in_flight=1.
            raise L.EpcNetError(-1, "last_status(): no single-workspace forward to report on (call forward() with "
                                    "num_clouds <= micro_batch or an engine built with in_flight=1)")
        mb = L.micro_batch_of(self._last_cfg, nc)
        last = nc % mb or mb
        arr = (ctypes.c_int32 * last)()
        L.check(L.lib().epc_net_last_status(ctypes.byref(self._last_cfg), self._ws.data_ptr(), nc, arr, L.current_stream()))
        return [int(v) for v in arr]

    # ---- throughput mode: independent batches in flight on the engine's own streams ---------------------------
    def _get_lanes(self, device):
        if self._lanes is None or self._lanes[0][0].device != device:
            self._lanes = [[torch.cuda.Stream(device=device), None, None] for _ in range(self.in_flight)]
        return self._lanes

    def submit(self, xyz: torch.Tensor, out: Optional[torch.Tensor] = None, profile: "Optional[StageProfile]" = None):
        """Asynchronous ``forward``: the batch runs on one of the engine's ``in_flight`` HIP streams (round-robin, own
        workspace each), ordered after everything already enqueued on the current stream.  Returns ``(out, event)``;
        ``out`` is valid once ``event`` has completed -- ``event.wait()`` orders the current stream after it,
        ``drain()`` after every submitted batch.  Batches submitted back to back overlap (the stages of a pass are
        bound by different units of the chip), which is where the throughput above a single stream comes from."""
        if xyz.dim() != 3 or xyz.shape[-1] != 3 or xyz.dtype != torch.float32:
            raise L.EpcNetError(-1, "expected float32 (num_clouds, N, 3) points, got %s %s" % (xyz.dtype, tuple(xyz.shape)))
        L.require_gpu()
        xyz = xyz.contiguous()
        nc, n = int(xyz.shape[0]), int(xyz.shape[1])
        cfg = self.cfg_for(n)
        packed = self.packed(cfg)
        if out is None:
            out = torch.empty((nc, cfg.output_dim), dtype=torch.float32, device=xyz.device)
        lanes = self._get_lanes(xyz.device)
        lane = lanes[self._next_lane % len(lanes)]
        self._next_lane += 1
        stream = lane[0]
        need = L.lib().epc_net_workspace_bytes(ctypes.byref(cfg), max(nc, 1))
        if lane[1] is None or lane[1].numel() < need:
            if lane[2] is not None:
                lane[2].synchronize()            # the lane's previous batch may still be using the old workspace
            lane[1] = torch.empty(need, dtype=torch.uint8, device=xyz.device)
        stream.wait_stream(torch.cuda.current_stream(xyz.device))
        xyz.record_stream(stream)
        out.record_stream(stream)
        packed.record_stream(stream)
        L.check(L.lib().epc_net_forward_profiled(ctypes.byref(cfg), packed.data_ptr(), L.ptr(xyz), nc, L.ptr(out),
                                                 lane[1].data_ptr(), lane[1].numel(), ctypes.c_void_p(stream.cuda_stream),
                                                 profile.handle if profile is not None else None))
        lane[2] = stream.record_event()
        return out, lane[2]

    def drain(self) -> None:
        """Order the current stream after every batch submitted so far."""
        for lane in self._lanes or []:
            if lane[2] is not None:
                lane[2].wait()


class StageProfile:
    """HIP-event stage timer (``epc_profile`` of include/epcnet.h)."""

    def __init__(self):
        h = ctypes.c_void_p()
        L.check(L.lib().epc_profile_create(ctypes.byref(h)))
        self.handle = h

    def elapsed_ms(self) -> Dict[str, float]:
        """Per-stage milliseconds; call after the stream is synchronised."""
        arr = (ctypes.c_float * L.EPC_NUM_STAGES)()
        L.check(L.lib().epc_profile_elapsed_ms(self.handle, arr))
        return {L.STAGE_NAMES[i]: float(arr[i]) for i in range(L.EPC_NUM_STAGES)}

    def __del__(self):
        try:
            if self.handle:
                L.lib().epc_profile_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
