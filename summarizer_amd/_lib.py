"""ctypes binding of libsumk.so (include/sumk.h).  The HIP library is the ONLY compute path: if it is missing
or a call fails, this module raises -- there is no CPU or eager-PyTorch fallback anywhere in summarizer_amd."""
import ctypes as C
import os

# torch FIRST: PyTorch-ROCm ships its own libamdhip64 and libsumk.so links the system one (same soname).  Whichever is loaded
# first serves the whole process; if libsumk.so came first, torch would later sit on a second HIP runtime and every call that
# mixes the two (our kernels on torch's streams and buffers) fails with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# SUMK_LIB_PATH: load another build of the same library (the diagnostic build libsumk_diag.so of `make DIAG=1`, for scripts/probes)
LIB_PATH = os.environ.get("SUMK_LIB_PATH") or os.path.join(_HERE, "libsumk.so")

c_f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
c_i32p = C.c_void_p
HOST_I32P = C.POINTER(C.c_int32)


class SumkError(RuntimeError):
    pass


class VasnetWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Wk", "Wq", "Wv", "Wo", "W1", "b1", "w2", "b2", "ln_w", "ln_b")]


class VasnetGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Wk", "Wq", "Wv", "Wo", "W1", "b1", "w2", "b2", "ln_w", "ln_b")]


class VasnetOpts(C.Structure):
    _fields_ = [("scale", C.c_float), ("eps", C.c_float), ("ignore_self", C.c_int32), ("aperture", C.c_int32),
                ("dropout_p", C.c_float), ("seed", C.c_uint64), ("precision", C.c_int32), ("seed_dev", C.c_void_p), ("tables", C.c_void_p), ("x16", C.c_void_p),
                ("xplanes", C.c_void_p), ("wplanes", C.c_void_p)]


class LstmDirWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_ih", "w_hh", "b_ih", "b_hh")]


class LstmDirGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_ih", "w_hh", "b_ih", "b_hh")]


class TfLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b",
                                          "lin2_w", "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b")]


class TfHeadWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_w", "ln_b", "k1_w", "k1_b", "k2_w", "k2_b")]


class TfOpts(C.Structure):
    _fields_ = [("layer_eps", C.c_float), ("final_eps", C.c_float), ("more_residuals", C.c_int32),
                ("layer_dropout_p", C.c_float), ("head_dropout_p", C.c_float), ("seed", C.c_uint64), ("precision", C.c_int32),
                ("wplanes", C.c_void_p)]


class EvalVideo(C.Structure):
    _fields_ = [("scores", C.c_void_p), ("n_steps", C.c_int32), ("picks", C.c_void_p), ("n_picks", C.c_int32),
                ("n_frames", C.c_int32), ("cps", C.c_void_p), ("nfps", C.c_void_p), ("n_segs", C.c_int32),
                ("user_summary", C.c_void_p), ("n_users", C.c_int32), ("user_ranks", C.c_void_p),
                ("machine_summary", C.c_void_p), ("summary_len", C.c_int32), ("corr", C.c_double), ("f_avg", C.c_double),
                ("f_max", C.c_double), ("seg_means", C.c_void_p)]


class EvalDevVideo(C.Structure):      # sumk_eval_dev_video: every pointer is a DEVICE pointer
    _fields_ = [("picks", C.c_void_p), ("n_picks", C.c_int32), ("n_frames", C.c_int32), ("n_steps", C.c_int32), ("row0", C.c_int32),
                ("frame0", C.c_int32), ("cps", C.c_void_p), ("n_segs", C.c_int32), ("seg0", C.c_int32), ("user_ranks", C.c_void_p),
                ("user_mean", C.c_void_p), ("user_ssq", C.c_void_p), ("n_users", C.c_int32)]


class LstmLayerWeights(C.Structure):
    _fields_ = [("w_ih", C.c_void_p * 2), ("w_hh", C.c_void_p * 2), ("b_ih", C.c_void_p * 2), ("b_hh", C.c_void_p * 2),
                ("x_planes", C.c_void_p), ("w_planes", C.c_void_p)]


class LstmLayerGrads(C.Structure):
    _fields_ = [("w_ih", C.c_void_p * 2), ("w_hh", C.c_void_p * 2), ("b_ih", C.c_void_p * 2), ("b_hh", C.c_void_p * 2),
                ("tail_ready_event", C.c_void_p)]


_SIGS = {
    # name: (restype, argtypes)
    "sumk_last_error": (C.c_char_p, []),
    "sumk_version": (C.c_int, []),
    "sumk_device_count": (C.c_int, []),
    "sumk_vasnet_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, HOST_I32P, C.c_int32]),
    "sumk_vasnet_workspace_bytes_for": (C.c_size_t, [C.c_int32, C.c_int32, HOST_I32P, C.c_int32, C.c_int32]),
    "sumk_vasnet_tables_bytes": (C.c_size_t, [C.c_int32, C.c_int32, HOST_I32P]),
    "sumk_vasnet_build_tables": (C.c_int, [C.c_int32, C.c_int32, HOST_I32P, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_vasnet_wplanes_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "sumk_vasnet_wplanes_build": (C.c_int, [C.c_int32, C.POINTER(VasnetWeights), c_f32p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_vasnet_forward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, HOST_I32P, c_i32p, C.POINTER(VasnetWeights),
                                      C.POINTER(VasnetOpts), c_f32p, c_i32p, c_f32p, C.c_void_p, C.c_size_t,
                                      C.c_int32, C.c_void_p]),
    "sumk_vasnet_forward_folded": (C.c_int, [c_f32p, C.c_int32, C.c_int32, HOST_I32P, c_i32p, C.POINTER(VasnetWeights), c_f32p,
                                             C.POINTER(VasnetOpts), c_f32p, c_i32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_vasnet_backward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, HOST_I32P, c_i32p, C.POINTER(VasnetWeights),
                                       C.POINTER(VasnetOpts), c_f32p, C.POINTER(VasnetGrads), c_f32p, C.c_void_p,
                                       C.c_size_t, C.c_void_p, C.c_void_p]),
    "sumk_bilstm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, HOST_I32P, C.c_int32]),
    "sumk_bilstm_wplanes_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "sumk_bilstm_wplanes_build": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(LstmLayerWeights), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_bilstm_layer_forward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, HOST_I32P, c_i32p,
                                            C.POINTER(LstmLayerWeights), c_f32p, C.c_void_p, C.c_size_t, C.c_int32,
                                            C.c_int32, C.c_void_p]),
    "sumk_bilstm_check": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, HOST_I32P, C.c_int32, C.c_int32, C.c_void_p]),
    "sumk_health_check": (C.c_int, [C.c_void_p]),
    "sumk_bilstm_layer_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, HOST_I32P,
                                             c_i32p, C.POINTER(LstmLayerWeights), C.POINTER(LstmLayerGrads), c_f32p,
                                             C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "sumk_lstm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, HOST_I32P, C.c_int32]),
    "sumk_lstm_layer_forward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, HOST_I32P, c_i32p, C.POINTER(LstmDirWeights),
                                          c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32,
                                          C.c_void_p]),
    "sumk_lstm_layer_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, HOST_I32P,
                                           c_i32p, C.POINTER(LstmDirWeights), c_f32p, C.POINTER(LstmDirGrads), c_f32p, c_f32p,
                                           c_f32p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "sumk_lstm_decoder_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, HOST_I32P]),
    "sumk_lstm_decoder_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, HOST_I32P, c_i32p, C.POINTER(LstmDirWeights), c_f32p,
                                            c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_lstm_decoder_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, HOST_I32P, c_i32p, C.POINTER(LstmDirWeights), c_f32p,
                                             c_f32p, c_f32p, C.POINTER(LstmDirGrads), c_f32p, c_f32p, C.c_void_p, C.c_size_t,
                                             C.c_void_p]),
    "sumk_gru_cell_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_void_p]),
    "sumk_gru_cell_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_void_p]),
    "sumk_linear_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "sumk_linear_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                      C.c_int32, C.c_void_p]),
    "sumk_linear_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p,
                                       C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "sumk_frame_head_forward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sumk_frame_head_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "sumk_frame_head_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p,
                                           c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_transformer_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, HOST_I32P, C.c_int32]),
    "sumk_transformer_workspace_bytes_for": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, HOST_I32P, C.c_int32, C.c_int32]),
    "sumk_transformer_wplanes_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sumk_transformer_wplanes_build": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_transformer_forward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, HOST_I32P, c_i32p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, c_f32p, c_i32p, c_f32p, C.c_void_p, C.c_size_t,
                                           C.c_int32, C.c_void_p]),
    "sumk_transformer_backward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, HOST_I32P, c_i32p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, c_f32p, C.c_void_p, C.c_void_p, c_f32p,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_dsn_reward_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, HOST_I32P, C.c_int32]),
    "sumk_dsn_reward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, HOST_I32P, c_i32p, c_f32p, C.c_int32, C.c_int32,
                                  C.c_int32, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_dsn_policy_loss_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_i32p, C.c_int32, C.c_float,
                                               C.c_float, c_f32p, c_f32p, C.c_void_p]),
    "sumk_dsn_policy_loss_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_i32p,
                                                C.c_int32, C.c_float, C.c_float, c_f32p, C.c_void_p]),
    "sumk_segment_mse_forward": (C.c_int, [c_f32p, c_f32p, C.c_int32, c_i32p, c_f32p, C.c_void_p]),
    "sumk_segment_mse_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, c_i32p, c_f32p, C.c_void_p]),
    "sumk_segment_mse_mean_forward": (C.c_int, [c_f32p, c_f32p, C.c_int32, c_i32p, C.c_float, c_f32p, c_f32p, C.c_void_p, C.c_void_p]),
    "sumk_segment_mse_mean_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_float, C.c_int32, c_i32p, c_f32p, C.c_void_p]),
    "sumk_adam_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_int32, C.c_float, C.c_void_p]),
    "sumk_adam_step_dev": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    "sumk_adam_step_dev_zero_grad": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                               C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    "sumk_gemm_splitk_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "sumk_gemm_splitk": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_sumsq_workspace_bytes": (C.c_size_t, []),
    "sumk_sumsq": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_void_p, C.c_void_p]),
    "sumk_cast_f32_bf16": (C.c_int, [c_f32p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sumk_cast_bf16_f32": (C.c_int, [C.c_void_p, c_f32p, C.c_int64, C.c_void_p]),
    "sumk_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "sumk_comm_init": (C.c_int, [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "sumk_allreduce_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "sumk_comm_destroy": (C.c_int, [C.c_void_p]),
    "sumk_gemm_nt": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sumk_eval_device": (C.c_int, [c_f32p, C.c_void_p, C.c_int32, c_f32p, c_f32p, C.c_void_p, C.c_void_p]),
    "sumk_eval_device_segments": (C.c_int, [c_f32p, C.c_void_p, C.c_int32, c_f32p, c_f32p, C.c_void_p]),
    "sumk_eval_device_spearman_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "sumk_eval_device_spearman": (C.c_int, [c_f32p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sumk_pack_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), HOST_I32P, C.c_int32, C.c_int32, C.c_int32]),
    "sumk_pack_rows_bf16": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), HOST_I32P, C.c_int32, C.c_int32, C.c_int32]),
    "sumk_gemm_prec": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sumk_gemm_bf16src": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sumk_planes_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "sumk_split_planes": (C.c_int, [c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sumk_gemm_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sumk_attn_planes_alpha_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "sumk_attn_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, HOST_I32P, C.c_float, C.c_int32, C.c_int32, c_f32p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "sumk_gemm_nn": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sumk_gemm_tn": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sumk_knapsack_dp": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, C.c_int64,
                                   C.POINTER(C.c_uint8)]),
    "sumk_eval_videos": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32]),
    "sumk_prof_enable": (C.c_int, [C.c_int32]),
    "sumk_prof_read": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
    "sumk_probe_mfma_rate": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "sumk_prof_gemm_stamps": (C.c_int, [C.POINTER(C.c_uint64), C.c_int32]),
}

PROF_GEMM_QKV, PROF_GEMM_ALL, PROF_LSTM_REC, PROF_GEMM_QKT, PROF_GEMM_PV, PROF_GEMM_OPROJ, PROF_GEMM_K1 = 0, 1, 2, 3, 4, 5, 6

_lib = None


def load():
    """Loads libsumk.so (once).  Raises SumkError -- loudly -- if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SumkError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C summarizer_amd/csrc`).  summarizer_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    missing = [n for n in _SIGS if not hasattr(lib, n)]
    if missing:
        raise SumkError(f"{LIB_PATH} does not export {missing}: stale build, rebuild with `make -C summarizer_amd/csrc`")
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().sumk_last_error().decode(errors="replace")
        raise SumkError(f"{what} failed (code {rc}): {msg}")


def host_i32(arr):
    """numpy int32 array -> ctypes pointer (array must stay alive for the call)."""
    return arr.ctypes.data_as(HOST_I32P)
