"""ctypes binding of the espnet_amd C ABI (include/espnet_amd.h).

The product path has no CPU fallback: if the HIP library is missing or a call is made with
tensors that are not on a GPU, this module raises.  torch is used only for device memory and
streams (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libespnet_amd_hip.so")


class EamdError(RuntimeError):
    pass


class GatherT(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("C", C.c_int32), ("ntap", C.c_int32), ("Ho", C.c_int32),
                ("Wo", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("sh", C.c_int32),
                ("sw", C.c_int32), ("dh", C.c_int32 * 9), ("dw", C.c_int32 * 9)]


class RowMapT(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Hc", C.c_int32),
                ("Wc", C.c_int32), ("sh", C.c_int32), ("oh", C.c_int32), ("sw", C.c_int32),
                ("ow", C.c_int32)]


class RowStatsT(C.Structure):
    _fields_ = [("part", C.c_void_p), ("col", C.c_void_p), ("zcol", C.c_void_p), ("zfix", C.c_void_p),
                ("fix", C.c_int32), ("reserved", C.c_int32),
                ("rowc", C.c_void_p), ("gscale", C.c_void_p), ("scale", C.c_float), ("reserved2", C.c_int32)]


class GemmT(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
                ("aux", C.c_void_p), ("R", C.c_void_p), ("colsum", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("transA", C.c_int32), ("transB", C.c_int32),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldaux", C.c_int64),
                ("ldr", C.c_int64),
                ("batch1", C.c_int32), ("batch2", C.c_int32),
                ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
                ("sC1", C.c_int64), ("sC2", C.c_int64),
                ("alpha", C.c_float), ("beta", C.c_float),
                ("a_act", C.c_int32), ("b_act", C.c_int32), ("epilogue", C.c_int32),
                ("splitk", C.c_int32), ("precision", C.c_int32), ("tile", C.c_int32),
                ("gather", GatherT), ("cmap", RowMapT),
                ("Cb", C.c_void_p), ("in_dtype", C.c_int32), ("aux_dtype", C.c_int32),
                ("drop_p", C.c_float), ("drop_salt", C.c_uint64), ("drop_step", C.c_void_p), ("Hb", C.c_void_p),
                ("h_act", C.c_int32),
                ("a_drop_p", C.c_float), ("b_drop_p", C.c_float), ("a_drop_salt", C.c_uint64), ("b_drop_salt", C.c_uint64),
                ("h_dtype", C.c_int32),
                ("stats", RowStatsT)]


class FfnT(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("R", C.c_void_p), ("out", C.c_void_p), ("f", C.c_void_p), ("h", C.c_void_p),
                ("M", C.c_int32), ("D", C.c_int32), ("F", C.c_int32), ("act", C.c_int32),
                ("alpha", C.c_float), ("p_in", C.c_float), ("salt_in", C.c_uint64), ("p_out", C.c_float),
                ("salt_out", C.c_uint64), ("drop_step", C.c_void_p), ("dtype", C.c_int32), ("hsplit", C.c_int32),
                ("ln_x", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("ln_mean", C.c_void_p),
                ("ln_rstd", C.c_void_p), ("ln_eps", C.c_float), ("reserved2", C.c_int32),
                ("lnb_x", C.c_void_p), ("lnb_gamma", C.c_void_p), ("lnb_mean", C.c_void_p), ("lnb_rstd", C.c_void_p),
                ("lnb_dres", C.c_void_p), ("lnb_ws", C.c_void_p), ("lnb_drop_out", C.c_void_p), ("lnb_drop_salt", C.c_uint64),
                ("lnb_drop_p", C.c_float), ("reserved3", C.c_int32)]


class RowProjT(C.Structure):
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int64), ("w", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
                ("ldr", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("M", C.c_int32), ("K", C.c_int32), ("N", C.c_int32), ("a_act", C.c_int32),
                ("alpha", C.c_float), ("p_out", C.c_float), ("salt_out", C.c_uint64), ("drop_step", C.c_void_p),
                ("ln_x", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("ln_mean", C.c_void_p),
                ("ln_rstd", C.c_void_p), ("ln_eps", C.c_float), ("lnb_drop_p", C.c_float),
                ("a_scale", C.c_void_p), ("a_shift", C.c_void_p), ("a_out", C.c_void_p),
                ("lnb_x", C.c_void_p), ("lnb_gamma", C.c_void_p), ("lnb_mean", C.c_void_p), ("lnb_rstd", C.c_void_p),
                ("lnb_dres", C.c_void_p), ("lnb_ws", C.c_void_p), ("lnb_drop_out", C.c_void_p), ("lnb_drop_salt", C.c_uint64)]


class FfnPackT(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("w2", C.c_void_p), ("fwd_first", C.c_void_p), ("fwd_second", C.c_void_p),
                ("bwd_first", C.c_void_p), ("bwd_second", C.c_void_p), ("D", C.c_int32), ("F", C.c_int32)]


class RowProjPackT(C.Structure):
    _fields_ = [("w", C.c_void_p), ("image", C.c_void_p), ("K", C.c_int32), ("N", C.c_int32), ("ldw", C.c_int32),
                ("trans", C.c_int32)]


class LstmSeqFwdT(C.Structure):
    _fields_ = [("gx", C.c_void_p), ("w_hh", C.c_void_p), ("b_hh", C.c_void_p), ("live", C.c_void_p),
                ("h_out", C.c_void_p), ("c_out", C.c_void_p), ("y", C.c_void_p), ("acts", C.c_void_p),
                ("reverse", C.c_int32), ("reserved", C.c_int32)]


class LstmSeqBwdT(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("w_t", C.c_void_p), ("acts", C.c_void_p), ("c_out", C.c_void_p),
                ("live", C.c_void_p), ("dgates", C.c_void_p), ("reverse", C.c_int32), ("reserved", C.c_int32)]


_lib = None

# every symbol include/espnet_amd.h declares (tests check they are all exported)
SYMBOLS = [
    "eamd_abi_version", "eamd_gemm", "eamd_gemm_multi", "eamd_gemm_group_plan", "eamd_gemm_group_launch", "eamd_ffn_fwd", "eamd_ffn_bwd", "eamd_ffn_pack_f32", "eamd_ffn_pack_bf16", "eamd_ffn_pack_f32_multi", "eamd_rowproj", "eamd_rowproj_pack_f32", "eamd_rowproj_lnb_workspace", "eamd_layernorm_fwd", "eamd_layernorm_bwd_workspace", "eamd_layernorm_bwd_drop_f32", "eamd_layernorm_bwd", "eamd_layernorm_bwd_reduce", "eamd_attn_fwd", "eamd_attn_bwd_q", "eamd_attn_fwd_f32", "eamd_attn_bwd_q_f32", "eamd_attn_bwd_kv_f32", "eamd_attn_bwd_kv", "eamd_softmax_fwd",
    "eamd_softmax_bwd", "eamd_lsm_loss", "eamd_argmax_rows", "eamd_reduce_sum", "eamd_log_softmax_rows", "eamd_topk_rows", "eamd_topk_rows_i32", "eamd_weighted_topk_rows", "eamd_beam_step", "eamd_beam_step_dyn", "eamd_decode_self_attn_dyn", "eamd_beam_slots_dyn", "eamd_ctc_prefix_psi_dyn", "eamd_ctc_prefix_state_dyn", "eamd_embed_pe_dyn", "eamd_copy_jobs", "eamd_embed_pe_ld", "eamd_linear_rows_f32", "eamd_linear_rows_ln_f32", "eamd_decode_self_attn", "eamd_beam_slots", "eamd_decode_src_attn", "eamd_decode_src_attn_group", "eamd_decode_src_attn_split", "eamd_decode_src_attn_split_workspace", "eamd_weighted_sum", "eamd_beam_select", "eamd_beam_finish",
    "eamd_axpby", "eamd_cast_bf16", "eamd_scale_dev", "eamd_act_fwd", "eamd_act_bwd", "eamd_glu_fwd", "eamd_glu_bwd",
    "eamd_add_bias2", "eamd_add_cast_bf16", "eamd_add_block_f32", "eamd_add_cast_colsum2", "eamd_add_colsum2_f32", "eamd_colsum", "eamd_embed_pe", "eamd_embed_bwd", "eamd_posenc", "eamd_posenc_scaled", "eamd_posenc_scaled_bwd", "eamd_permute4",
    "eamd_dropout", "eamd_rng_advance", "eamd_dwconv_fwd", "eamd_dwconv_glu_fwd", "eamd_dwconv_glu_bwd_x", "eamd_dwconv_glu_bwd_w", "eamd_dwconv_bwd_x", "eamd_dwconv_bwd_w", "eamd_bn_nslab",
    "eamd_bn_stats", "eamd_bn_finalize", "eamd_bn_apply", "eamd_bn_bwd", "eamd_bn_stats_bounded", "eamd_bn_bwd_bounded", "eamd_mask_time", "eamd_conv1_fwd", "eamd_conv1_bwd_w_workspace", "eamd_conv1_bwd_w",
    "eamd_conv2_weight_prep", "eamd_conv2_weight_grad", "eamd_add_sos_eos", "eamd_ctc_collapse",
    "eamd_ctc_workspace_bytes", "eamd_ctc_loss", "eamd_ctc_prefix_score", "eamd_ctc_prefix_score_batch", "eamd_ctc_prefix_psi", "eamd_ctc_prefix_state", "eamd_grad_norm", "eamd_sched_step", "eamd_adam_step", "eamd_adadelta_step", "eamd_add_gradient_noise",
    "eamd_specaug", "eamd_global_mvn", "eamd_utterance_mvn", "eamd_reflect_pad", "eamd_logmel", "eamd_unfold1d", "eamd_fold1d", "eamd_attloc_convmax_fwd", "eamd_attloc_convmax_bwd", "eamd_layernorm_bwd_drop",
    "eamd_lstm_cell_fwd", "eamd_lstm_cell_bwd", "eamd_lstm_step_fwd", "eamd_lstm_step_bwd", "eamd_lstm_seq_sync_bytes", "eamd_lstm_seq_fwd", "eamd_lstm_seq_bwd", "eamd_lstm_seq_status", "eamd_lstm_seq_status_merge", "eamd_gru_cell_fwd", "eamd_gru_cell_bwd", "eamd_maxpool2x2_fwd", "eamd_maxpool2x2_bwd", "eamd_mask_rows",
    "eamd_joint_fwd", "eamd_joint_bwd", "eamd_rnnt_workspace", "eamd_rnnt_loss", "eamd_rnnt_grad", "eamd_rnnt_node_stats", "eamd_rnnt_node_stats_part", "eamd_rnnt_row_coef", "eamd_rnnt_alpha_beta", "eamd_rnnt_node_grad",
    "eamd_conv3x3_c1_fwd", "eamd_conv3x3_c1_bwd_w_workspace", "eamd_conv3x3_c1_bwd_w", "eamd_attloc_fwd", "eamd_attloc_bwd_energy", "eamd_attloc_bwd_workspace", "eamd_attloc_bwd_energy_conv", "eamd_attloc_bwd_conv",
    "eamd_att_dot_energy_fwd", "eamd_att_dot_energy_bwd", "eamd_att_ctx_fwd", "eamd_att_ctx_bwd",
]


def lib():
    """Load the HIP shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if os.environ.get("EAMD_LIB"):          # diagnostic builds of tools/ (timing probes): never set in production
            globals()["LIB_PATH"] = os.environ["EAMD_LIB"]
        if not os.path.exists(LIB_PATH):
            raise EamdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C espnet_amd/csrc` (there is no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.eamd_ctc_workspace_bytes.restype = C.c_int64
        _lib.eamd_layernorm_bwd_workspace.restype = C.c_int64
        _lib.eamd_rnnt_workspace.restype = C.c_int64
        _lib.eamd_attloc_bwd_workspace.restype = C.c_int64
        _lib.eamd_conv1_bwd_w_workspace.restype = C.c_int64
        _lib.eamd_lstm_seq_sync_bytes.restype = C.c_int64
        _lib.eamd_conv3x3_c1_bwd_w_workspace.restype = C.c_int64
        for s in SYMBOLS:
            getattr(_lib, s)  # AttributeError here = header/library mismatch
    return _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """the current HIP stream of the current device.  torch.cuda.current_stream() builds a Stream object through several
    Python layers (2.8 us per call, paid by every launch of an eager decode step); the raw getter is one C call"""
    if _raw_stream is not None and _cur_device is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


EAMD_EUNSUPPORTED = -2   # include/espnet_amd.h


def check(rc, what):
    if rc != 0:
        raise EamdError(f"{what} failed with code {rc}")


def ptr(t, offset=0):
    """Device pointer of a tensor (+ element offset); None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EamdError("espnet_amd kernels need GPU tensors (no CPU fallback in the product path)")
    if offset == 0:
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.data_ptr() + offset * t.element_size())


def f32(t):
    if t.dtype != torch.float32:
        raise EamdError(f"expected float32 tensor, got {t.dtype}")
    return t
