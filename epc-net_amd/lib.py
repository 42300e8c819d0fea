"""ctypes binding of ``libepcnet_hip.so`` (declarations: ``include/epcnet.h``).

There is NO fallback: if the shared library has not been built (``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C epc-net_amd/csrc``) importing this module raises, and every op raises ``EpcNetError``
when the library reports a failure.  ``import torch`` happens first on purpose: the library's
``libamdhip64.so.7`` dependency then resolves to the HIP runtime torch already loaded, so device pointers and
streams are shared between torch (plumbing: allocation, streams, RCCL) and the kernels.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# EPCNET_LIB: tuning-only override (scripts/tune_*.sh build throw-away variants of the same library)
LIB_PATH = os.environ.get("EPCNET_LIB") or os.path.join(_HERE, "libepcnet_hip.so")

EPC_OK = 0
EPC_ARCH_EPC_NET = 0
EPC_ARCH_EPC_NET_L = 1
EPC_KNN_SELECT = 20
EPC_KNN_CAP = 32
EPC_ERANGE = -5
STATUS_NAMES = {0: "EPC_OK", -1: "EPC_EINVAL", -2: "EPC_ENOMEM", -3: "EPC_EHIP", -4: "EPC_ENOTFOUND", -5: "EPC_ERANGE"}
# epc_cfg.precision (include/epcnet.h): f32-equivalent arithmetic (conv layers scaled split-fp16 x3, assignment / aggregate
# split-bf16 x3, f32 tensors in HBM except the 3-byte `feat` map) / EPC-Net's f16 + f6 fast arithmetic
EPC_PRECISION_F32 = 0
EPC_PRECISION_FAST = 1
PRECISION_IDS = {"f32": EPC_PRECISION_F32, "fast": EPC_PRECISION_FAST}
# per-cloud status bits (a flagged cloud's descriptor is NaN)
EPC_STATUS_NONFINITE_INPUT = 1
EPC_STATUS_FP16_RANGE = 2

# every symbol include/epcnet.h declares (tests check the library exports exactly these)
EXPORTS = [
    "epc_last_error", "epc_version", "epc_net_packed_bytes", "epc_net_pack_weights", "epc_net_workspace_bytes",
    "epc_net_forward", "epc_net_forward_overlapped", "epc_net_last_status", "epc_conv5_assign_f32_fwd",
    "epc_vlad_aggregate_f32_fwd", "epc_knn_topk", "epc_knn_topk_conv1", "epc_knn_topk_form", "epc_knn_topk_conv1_form", "epc_knn_mask", "epc_conv1_fwd", "epc_proxyconv_block_fwd",
    "epc_conv5_assign_fwd", "epc_vlad_aggregate_fwd", "epc_vlad_head_workspace_bytes", "epc_vlad_head_fwd",
    "epc_conv5_maxpool_fwd", "epc_fc_head_fwd", "epc_pairwise_topk", "epc_pairwise_topk_workspace_bytes",
    "epc_pairwise_topk_ws", "epc_net_packed_offset",
    "epc_profile_create", "epc_profile_destroy", "epc_net_forward_profiled", "epc_profile_elapsed_ms",
    "epc_morton_sort",
    "epc_gemm_f32", "epc_gemm_f32_fast", "epc_gemm_bf16", "epc_gemm_splitk_det", "epc_linear_bn_bwd64", "epc_linear_bn_bwd64_ex", "epc_linear_stats64", "epc_linear_stats64_bn", "epc_bn_apply_add_fwd", "epc_neighbour_mean_diff_bwd_gather_sum", "epc_linear_smallk_fwd", "epc_linear_smallk_dw", "epc_linear_smallk_dw_partial_floats", "epc_linear_bn_bwd64_partial_floats", "epc_gemm_stats_tiles", "epc_gemm_f32_stats", "epc_gemm_f16x3_stats", "epc_gemm_bf16_stats", "epc_neighbour_mean_diff_fwd", "epc_neighbour_mean_diff_bwd_gather", "epc_bn_relu_rownorm_fwd", "epc_bn_relu_rownorm_bwd", "epc_bn_relu_rownorm_bwd_partial_floats", "epc_vlad_normalize_fwd", "epc_vlad_normalize_bwd",
    "epc_lazy_quadruplet_loss_fwd", "epc_lazy_quadruplet_loss_bwd", "epc_colreduce_workspace_bytes", "epc_col_moments", "epc_col_sum", "epc_bn_apply_fwd", "epc_bn_apply_bwd",
    "epc_neighbour_mean_fwd", "epc_knn_transpose", "epc_neighbour_mean_bwd_gather", "epc_rownorm_fwd", "epc_rownorm_bwd", "epc_softmax64_fwd",
    "epc_softmax64_bwd", "epc_softmax64_bwd_bcast", "epc_cloud_colsum64_partial_floats",
    "epc_assign_softmax_fwd", "epc_assign_softmax_bwd", "epc_gate_fwd",
    "epc_chain_parts", "epc_chain_stats", "epc_chain_fwd_linear", "epc_chain_fwd_gather", "epc_chain_bwd_linear",
    "epc_chain_bwd_gather", "epc_chain_sums", "epc_chain_bn_bwd", "epc_chain_dw_sum", "epc_knn_overflow_lists",
    "epc_chain_persist_ok", "epc_chain_persist_workspace_bytes", "epc_chain_persist_init", "epc_chain_fwd_persist", "epc_chain_persist_status", "epc_chain_persist_reset",
    "epc_vlad_df_packed_bytes", "epc_vlad_df", "epc_vlad_df_tail_partial_floats", "epc_vlad_df_tail", "epc_bn_apply_bwd_given", "epc_gate_bwd", "epc_sq_err_partial_floats", "epc_sq_err_fwd", "epc_sq_err_bwd", "epc_adam_step", "epc_adam_step_dev", "epc_ema_update", "epc_adam_multi", "epc_ema_multi", "epc_crc32c",
    "epc_h16_conv5_fwd_scratch_bytes", "epc_h16_conv5_fwd", "epc_h16_assign_scratch_bytes", "epc_h16_assign",
    "epc_h16_colgemm_scratch_bytes", "epc_h16_colgemm", "epc_h16_df_tail_scratch_bytes", "epc_h16_df_tail", "epc_h16_bn_bwd_apply",
    "epc_h16_dx_scratch_bytes", "epc_h16_conv5_dx", "epc_h16_conv5_dx_bn", "epc_h16_conv5_dw_scratch_bytes", "epc_h16_conv5_dw", "epc_h16_expand", "epc_gemm_splitk_det_b16",
    "epc_h32_conv5_fwd_scratch_bytes", "epc_h32_conv5_fwd", "epc_h32_assign_scratch_bytes", "epc_h32_assign",
    "epc_h32_colgemm_scratch_bytes", "epc_h32_colgemm", "epc_h32_dx_scratch_bytes", "epc_h32_conv5_dx", "epc_h32_conv5_dx_bn", "epc_h32_conv5_dw_scratch_bytes", "epc_h32_conv5_dw",
    "epc_hidden_proj_ok", "epc_hidden_proj_scratch_bytes", "epc_hidden_proj_fwd", "epc_hidden_proj_bwd",
    "epc_maxpool_points_fwd", "epc_maxpool_points_bwd", "epc_vlad_w2_grad", "epc_group_sum_fwd", "epc_group_sum_bwd",
    "epc_hidden_tail_ok", "epc_hidden_tail_fwd", "epc_hidden_tail_bwd",
]
EPC_NUM_STAGES = 10
STAGE_NAMES = ["sort", "knn", "conv1", "block1", "block2", "block3", "block4", "conv5", "aggregate", "head"]


class EpcNetError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__("%s (%d): %s" % (STATUS_NAMES.get(status, "?"), status, message))
        self.status = status


class EpcCfg(ctypes.Structure):
    """``struct epc_cfg`` of include/epcnet.h."""
    _fields_ = [("arch", c_int32), ("num_points", c_int32), ("input_dim", c_int32), ("knn", c_int32),
                ("cluster_size", c_int32), ("output_dim", c_int32), ("groups", c_int32),
                ("micro_batch", c_int32), ("precision", c_int32)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' or "
        "make -C epc-net_amd/csrc).  There is no CPU fallback." % LIB_PATH)

_lib = ctypes.CDLL(LIB_PATH)

_P = c_void_p
_lib.epc_last_error.restype = c_char_p
_lib.epc_last_error.argtypes = []
_lib.epc_version.restype = c_int
_lib.epc_crc32c.restype = ctypes.c_uint32
_lib.epc_crc32c.argtypes = [ctypes.c_uint32, c_void_p, c_size_t]
_lib.epc_net_packed_bytes.restype = c_size_t
_lib.epc_net_packed_bytes.argtypes = [POINTER(EpcCfg)]
_lib.epc_net_packed_offset.restype = c_size_t
_lib.epc_net_packed_offset.argtypes = [POINTER(EpcCfg), c_int]
_lib.epc_net_pack_weights.argtypes = [POINTER(EpcCfg), POINTER(c_char_p), POINTER(_P), c_int, _P, c_size_t, _P]
_lib.epc_net_workspace_bytes.restype = c_size_t
_lib.epc_net_workspace_bytes.argtypes = [POINTER(EpcCfg), c_int]
_lib.epc_net_forward.argtypes = [POINTER(EpcCfg), _P, _P, c_int, _P, _P, c_size_t, _P]
_lib.epc_net_forward_overlapped.argtypes = [POINTER(EpcCfg), _P, _P, c_int, _P, _P, c_size_t, _P, POINTER(_P), c_int]
_lib.epc_knn_topk.argtypes = [_P, c_int, c_int, c_int, _P, _P, _P, _P]
_lib.epc_knn_topk_conv1.argtypes = [_P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, _P]
_lib.epc_knn_topk_form.argtypes = [_P, c_int, c_int, c_int, _P, _P, _P, c_int, _P]
_lib.epc_knn_topk_conv1_form.argtypes = [_P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P]
_lib.epc_net_last_status.argtypes = [POINTER(EpcCfg), _P, c_int, POINTER(c_int32), _P]
_lib.epc_knn_mask.argtypes = [_P, _P, c_int, c_int, _P, _P]
_lib.epc_conv1_fwd.argtypes = [_P, _P, c_int, _P, _P, _P]
_lib.epc_proxyconv_block_fwd.argtypes = [_P, _P, _P, _P, c_int, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P,
                                         c_int, c_int, _P, _P, _P, _P]
_lib.epc_conv5_assign_fwd.argtypes = [_P, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]
_lib.epc_conv5_assign_f32_fwd.argtypes = [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P]
_lib.epc_vlad_aggregate_fwd.argtypes = [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P]
_lib.epc_vlad_aggregate_f32_fwd.argtypes = _lib.epc_vlad_aggregate_fwd.argtypes
_lib.epc_vlad_head_workspace_bytes.restype = c_size_t
_lib.epc_vlad_head_workspace_bytes.argtypes = [c_int, c_int]
_lib.epc_vlad_head_fwd.argtypes = [_P, _P, _P, c_int, c_int, _P, _P, _P, c_size_t, _P]
_lib.epc_conv5_maxpool_fwd.argtypes = [_P, c_int, _P, c_int, c_int, _P, _P]
_lib.epc_fc_head_fwd.argtypes = [_P, _P, c_int, _P, _P, _P]
_lib.epc_pairwise_topk.argtypes = [_P, c_int, _P, c_int, c_int, c_int, _P, _P, _P]
_lib.epc_pairwise_topk_workspace_bytes.restype = c_size_t
_lib.epc_pairwise_topk_workspace_bytes.argtypes = [c_int, c_int]
_lib.epc_pairwise_topk_ws.argtypes = [_P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]
_lib.epc_morton_sort.argtypes = [_P, c_int, c_int, _P, _P, _P]
from ctypes import c_long  # noqa: E402
_lib.epc_gemm_f32.argtypes = [_P, _P, _P, _P, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_int, c_int, c_long,
                              c_long, c_long, c_int, c_int, _P]
_lib.epc_gemm_f32_fast.argtypes = _lib.epc_gemm_f32.argtypes
_lib.epc_gemm_stats_tiles.argtypes = [c_int]
_lib.epc_gemm_f32_stats.argtypes = [_P, _P, _P, _P, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_int, _P, c_size_t, _P, _P, _P]
_lib.epc_gemm_f16x3_stats.argtypes = [_P, _P, _P, _P, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P]
_lib.epc_gemm_bf16.argtypes = _lib.epc_gemm_f32.argtypes
_lib.epc_gemm_bf16_stats.argtypes = _lib.epc_gemm_f32_stats.argtypes
_lib.epc_gemm_splitk_det.argtypes = _lib.epc_gemm_f32.argtypes[:-1] + [c_int, _P, c_size_t, _P]
_lib.epc_linear_smallk_fwd.argtypes = [_P, _P, _P, c_int, c_int, c_int, _P, _P]
_lib.epc_linear_smallk_dw_partial_floats.restype = c_size_t
_lib.epc_linear_smallk_dw_partial_floats.argtypes = [c_int, c_int]
_lib.epc_linear_smallk_dw.argtypes = [_P, _P, c_int, c_int, c_int, _P, _P, c_size_t, _P]
_lib.epc_linear_stats64.argtypes = [_P, _P, _P, c_int, _P, _P, _P, _P, c_size_t, _P]
_lib.epc_linear_stats64_bn.argtypes = [_P, _P, _P, _P, _P, c_float, _P, _P, c_int, _P, _P, _P, _P, c_size_t, _P]
_lib.epc_linear_bn_bwd64_ex.argtypes = [_P] * 12 + [c_float, c_int, c_int, _P, _P, _P, _P, _P, _P, c_size_t, _P, c_size_t, _P]
_lib.epc_bn_apply_add_fwd.argtypes = [_P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P]
_lib.epc_neighbour_mean_diff_bwd_gather_sum.argtypes = [_P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, _P, _P]
_lib.epc_linear_bn_bwd64_partial_floats.restype = c_size_t
_lib.epc_linear_bn_bwd64_partial_floats.argtypes = [c_int]
_lib.epc_linear_bn_bwd64.argtypes = [_P] * 8 + [c_float, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P, c_size_t, _P]
_lib.epc_bn_relu_rownorm_fwd.argtypes = [_P, _P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P]
_lib.epc_bn_relu_rownorm_bwd_partial_floats.restype = c_size_t
_lib.epc_bn_relu_rownorm_bwd_partial_floats.argtypes = [c_int]
_lib.epc_bn_relu_rownorm_bwd.argtypes = [_P] * 7 + [c_float, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]
_lib.epc_vlad_normalize_fwd.argtypes = [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]
_lib.epc_vlad_normalize_bwd.argtypes = [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]
_lib.epc_lazy_quadruplet_loss_fwd.argtypes = [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P]
_lib.epc_lazy_quadruplet_loss_bwd.argtypes = [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]
_lib.epc_colreduce_workspace_bytes.restype = c_size_t
_lib.epc_colreduce_workspace_bytes.argtypes = [c_int, c_int]
_lib.epc_col_moments.argtypes = [_P, c_int, c_int, _P, _P, _P, c_size_t, _P]
_lib.epc_col_sum.argtypes = [_P, c_int, c_int, _P, _P, c_size_t, _P]
_lib.epc_bn_apply_fwd.argtypes = [_P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P]
_lib.epc_bn_apply_bwd.argtypes = [_P, _P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]
_lib.epc_neighbour_mean_fwd.argtypes = [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]
_lib.epc_neighbour_mean_diff_fwd.argtypes = [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]
_lib.epc_neighbour_mean_diff_bwd_gather.argtypes = [_P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, _P, _P]
_lib.epc_knn_transpose.argtypes = [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P]
_lib.epc_neighbour_mean_bwd_gather.argtypes = [_P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, _P, _P]
_lib.epc_rownorm_fwd.argtypes = [_P, c_int, c_int, _P, _P, _P]
_lib.epc_rownorm_bwd.argtypes = [_P, _P, _P, c_int, c_int, _P, _P]
_lib.epc_softmax64_fwd.argtypes = [_P, c_int, _P, _P]
_lib.epc_softmax64_bwd.argtypes = [_P, _P, c_int, _P, _P]
_lib.epc_softmax64_bwd_bcast.argtypes = [_P, _P, c_int, _P, c_int, _P, _P]
_lib.epc_cloud_colsum64_partial_floats.argtypes = [c_int]
_lib.epc_cloud_colsum64_partial_floats.restype = ctypes.c_size_t
_lib.epc_assign_softmax_fwd.argtypes = [_P, _P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P, ctypes.c_size_t, _P]
_lib.epc_assign_softmax_bwd.argtypes = [_P, _P, _P, _P, _P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]
_lib.epc_vlad_df_tail_partial_floats.argtypes = [c_int, c_int]
_lib.epc_vlad_df_tail_partial_floats.restype = ctypes.c_size_t
_lib.epc_vlad_df_tail.argtypes = [_P, _P, _P, _P, c_int, c_int, c_int, _P, ctypes.c_size_t, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, ctypes.c_size_t, _P]
_lib.epc_bn_apply_bwd_given.argtypes = [_P, _P, _P, _P, _P, _P, _P, _P, c_float, c_int, c_int, _P, _P]
_lib.epc_gate_fwd.argtypes = [_P, _P, ctypes.c_long, _P, _P]
_lib.epc_gate_bwd.argtypes = [_P, _P, _P, ctypes.c_long, _P, _P, _P]
_lib.epc_hidden_tail_ok.argtypes = [c_int, c_int, c_int]
_lib.epc_hidden_tail_fwd.argtypes = [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_float, c_float, c_float] + [_P] * 9 + [_P]
_lib.epc_hidden_tail_bwd.argtypes = [_P, _P, c_int, c_int, c_int] + [_P] * 10 + [c_float] + [_P] * 7
_lib.epc_chain_parts.argtypes = [c_int]
_lib.epc_chain_stats.argtypes = [_P, c_int, _P, _P]
_lib.epc_chain_fwd_linear.argtypes = [_P] * 7 + [c_float, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P]
_lib.epc_chain_fwd_gather.argtypes = [_P] * 7 + [c_float, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P]
_lib.epc_chain_bwd_linear.argtypes = [_P, c_int] + [_P] * 5 + [c_float, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P] + \
    [_P] * 6 + [c_int, c_int, _P]
_lib.epc_chain_bwd_gather.argtypes = [_P, _P, c_int] + [_P] * 7 + [c_int, c_int, c_int] + [_P] * 5 + [c_float, _P, _P, _P]
_lib.epc_chain_sums.argtypes = [_P, c_int] + [_P] * 5 + [c_float, c_int, _P, _P]
_lib.epc_chain_bn_bwd.argtypes = [_P] * 6 + [c_float, _P, _P, _P, c_int, _P, _P]
_lib.epc_chain_dw_sum.argtypes = [c_int, _P, _P, c_int, _P]
EPC_CHAIN_MAX_BLOCKS = 4


class ChainFwdBlock(ctypes.Structure):
    """``struct epc_chain_fwd_block`` of include/epcnet.h."""
    _fields_ = [(n, _P) for n in ("gamma0", "beta0", "in_bias", "Wa", "ba", "gamma_a", "beta_a", "Wb", "bb", "gamma_b", "beta_b",
                                  "W0_next", "b0_next", "z0", "mean0", "var0", "mean_a", "var_a", "mean_b", "var_b",
                                  "d", "za", "zb", "z0_next")]


class ChainFwdArgs(ctypes.Structure):
    """``struct epc_chain_fwd_args`` of include/epcnet.h."""
    _fields_ = [("blk", ChainFwdBlock * EPC_CHAIN_MAX_BLOCKS), ("nblocks", c_int), ("xyz", _P), ("idx", _P), ("cnt", _P), ("kth", _P),
                ("cap", c_int), ("num_clouds", c_int), ("n", c_int), ("knn", c_int), ("cat", _P), ("cat_bf16", _P), ("eps", c_float),
                ("workspace", _P), ("spin_ticks", ctypes.c_longlong)]


_lib.epc_chain_persist_ok.argtypes = [c_int]
_lib.epc_chain_persist_workspace_bytes.restype = c_size_t
_lib.epc_chain_persist_workspace_bytes.argtypes = []
_lib.epc_chain_persist_init.argtypes = [_P, _P]
_lib.epc_chain_fwd_persist.argtypes = [POINTER(ChainFwdArgs), c_int, _P]
_lib.epc_chain_persist_status.argtypes = [_P, _P]
_lib.epc_chain_persist_reset.argtypes = [_P, _P]
_lib.epc_knn_overflow_lists.argtypes = [_P, c_int, c_int, c_int, _P, _P, _P]
_lib.epc_vlad_df_packed_bytes.restype = c_size_t
_lib.epc_vlad_df_packed_bytes.argtypes = [c_int, c_int]
_lib.epc_vlad_df.argtypes = [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]
_lib.epc_sq_err_partial_floats.restype = c_size_t
_lib.epc_sq_err_partial_floats.argtypes = [c_long]
_lib.epc_sq_err_fwd.argtypes = [_P, _P, c_long, c_int, _P, _P, c_size_t, _P]
_lib.epc_sq_err_bwd.argtypes = [_P, _P, c_long, c_int, _P, _P, _P]
_lib.epc_adam_step.argtypes = [_P, _P, _P, _P, c_long, c_float, c_float, c_float, c_float, c_int, _P]
_lib.epc_adam_step_dev.argtypes = [_P, _P, _P, _P, c_long, _P, c_float, c_float, c_float, _P]
_lib.epc_ema_update.argtypes = [_P, _P, c_long, c_float, _P, _P]
_lib.epc_adam_multi.argtypes = [c_int, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_int, _P, _P]
_lib.epc_ema_multi.argtypes = [c_int, _P, _P, _P, _P, c_float, c_float, _P, _P]
_lib.epc_h16_conv5_fwd_scratch_bytes.restype = c_size_t
_lib.epc_h16_conv5_fwd_scratch_bytes.argtypes = [c_int]
_lib.epc_h16_conv5_fwd.argtypes = [_P, c_int, _P, _P, c_int, _P, _P, _P, _P, c_size_t, _P]
_lib.epc_h16_assign_scratch_bytes.restype = c_size_t
_lib.epc_h16_assign_scratch_bytes.argtypes = [c_int, c_int, c_int]
_lib.epc_h16_assign.argtypes = [_P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]
_lib.epc_h16_colgemm_scratch_bytes.restype = c_size_t
_lib.epc_h16_colgemm_scratch_bytes.argtypes = [c_int, c_int]
_lib.epc_h16_colgemm.argtypes = [_P, _P, _P, _P, _P, c_float, _P, _P, c_int, c_int, c_int, _P, _P, c_size_t, _P]
_lib.epc_h16_df_tail_scratch_bytes.restype = c_size_t
_lib.epc_h16_df_tail_scratch_bytes.argtypes = [c_int, c_int]
_lib.epc_h16_df_tail.argtypes = [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, c_size_t, _P]
_lib.epc_h16_bn_bwd_apply.argtypes = [_P, _P, _P, _P, _P, _P, c_float, _P, _P, c_int, _P, _P]
_lib.epc_h16_dx_scratch_bytes.restype = c_size_t
_lib.epc_h16_dx_scratch_bytes.argtypes = []
_lib.epc_h16_conv5_dx.argtypes = [_P, _P, c_int, _P, _P, c_size_t, _P]
_lib.epc_h16_conv5_dx_bn.argtypes = [_P, _P, _P, _P, _P, c_float, _P, _P, _P, c_int, _P, _P, _P, c_size_t, _P]
_lib.epc_h16_conv5_dw_scratch_bytes.restype = c_size_t
_lib.epc_h16_conv5_dw_scratch_bytes.argtypes = [c_int]
_lib.epc_h16_conv5_dw.argtypes = [_P, c_int, _P, c_int, _P, _P, c_size_t, _P]
_lib.epc_h16_expand.argtypes = [_P, _P, _P, _P, _P, c_float, _P, c_int, _P, _P]
_lib.epc_gemm_splitk_det_b16.argtypes = _lib.epc_gemm_splitk_det.argtypes
_lib.epc_h32_conv5_fwd_scratch_bytes.restype = c_size_t
_lib.epc_h32_conv5_fwd_scratch_bytes.argtypes = [c_int]
_lib.epc_h32_conv5_fwd.argtypes = [_P, _P, _P, c_int, _P, _P, _P, _P, c_size_t, _P]
_lib.epc_h32_assign_scratch_bytes.restype = c_size_t
_lib.epc_h32_assign_scratch_bytes.argtypes = [c_int, c_int, c_int]
_lib.epc_h32_assign.argtypes = _lib.epc_h16_assign.argtypes
_lib.epc_h32_colgemm_scratch_bytes.restype = c_size_t
_lib.epc_h32_colgemm_scratch_bytes.argtypes = [c_int, c_int]
_lib.epc_h32_colgemm.argtypes = _lib.epc_h16_colgemm.argtypes
_lib.epc_h32_dx_scratch_bytes.restype = c_size_t
_lib.epc_h32_dx_scratch_bytes.argtypes = []
_lib.epc_h32_conv5_dx.argtypes = _lib.epc_h16_conv5_dx.argtypes
_lib.epc_h32_conv5_dx_bn.argtypes = _lib.epc_h16_conv5_dx_bn.argtypes
_lib.epc_h32_conv5_dw_scratch_bytes.restype = c_size_t
_lib.epc_h32_conv5_dw_scratch_bytes.argtypes = [c_int]
_lib.epc_h32_conv5_dw.argtypes = [_P, _P, c_int, _P, _P, c_size_t, _P]
_lib.epc_hidden_proj_ok.argtypes = [c_int, c_int, c_int]
_lib.epc_hidden_proj_scratch_bytes.restype = c_size_t
_lib.epc_hidden_proj_scratch_bytes.argtypes = [c_int, c_int]
_lib.epc_hidden_proj_fwd.argtypes = [_P, _P, c_int, c_int, c_int, _P, _P, c_size_t, _P]
_lib.epc_hidden_proj_bwd.argtypes = [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P]
_lib.epc_maxpool_points_fwd.argtypes = [_P, c_int, c_int, c_int, _P, _P, _P]
_lib.epc_maxpool_points_bwd.argtypes = [_P, _P, c_int, c_int, c_int, _P, _P]
_lib.epc_vlad_w2_grad.argtypes = [_P, _P, c_int, c_int, c_int, _P, _P]
_lib.epc_group_sum_fwd.argtypes = [_P, c_int, c_int, c_int, _P, _P]
_lib.epc_group_sum_bwd.argtypes = [_P, c_int, c_int, c_int, _P, _P]
_lib.epc_profile_create.argtypes = [POINTER(_P)]
_lib.epc_profile_destroy.argtypes = [_P]
_lib.epc_net_forward_profiled.argtypes = [POINTER(EpcCfg), _P, _P, c_int, _P, _P, c_size_t, _P, _P]
_lib.epc_profile_elapsed_ms.argtypes = [_P, POINTER(c_float)]
for _name in EXPORTS:
    getattr(_lib, _name)  # AttributeError here = the built library is stale (rebuild it)


def lib() -> ctypes.CDLL:
    return _lib


def check(status: int) -> None:
    if status != EPC_OK:
        raise EpcNetError(status, (_lib.epc_last_error() or b"").decode("utf-8", "replace"))


def ptr(t) -> int:
    """Device pointer of a contiguous CUDA/HIP tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EpcNetError(-1, "tensor must live on a ROCm device: the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise EpcNetError(-1, "tensor must be contiguous")
    return t.data_ptr()


def micro_batch_of(cfg: "EpcCfg", num_clouds: int) -> int:
    """Clouds per internal pass (mirrors micro_batch() of csrc/pipeline.hip)."""
    mb = cfg.micro_batch if cfg.micro_batch > 0 else (64 if cfg.arch == EPC_ARCH_EPC_NET else 256)
    return max(1, min(int(num_clouds), int(mb)))


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise EpcNetError(-3, "no ROCm device visible: the EPC-Net HIP path cannot run (no CPU fallback)")
