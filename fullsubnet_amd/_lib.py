"""ctypes binding of libfsn_hip.so (include/fsn_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, the caller
gets an exception.  PyTorch is used for device memory and streams only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfsn_hip.so")

NORM_TYPES = {"offline_laplace_norm": 0, "cumulative_laplace_norm": 1}  # the fused FullSubNet kernels (cfg.norm_type)
ALL_NORM_TYPES = {"offline_laplace_norm": 0, "cumulative_laplace_norm": 1, "offline_gaussian_norm": 2,
                  "cumulative_layer_norm": 3, "forgetting_norm": 4}  # fsn_norm

_f32p = ctypes.c_void_p
_lib = None


class FsnError(RuntimeError):
    pass


class Cfg(ctypes.Structure):
    _fields_ = [("num_freqs", ctypes.c_int), ("look_ahead", ctypes.c_int), ("sb_num_neighbors", ctypes.c_int),
                ("fb_hidden", ctypes.c_int), ("sb_hidden", ctypes.c_int), ("norm_type", ctypes.c_int),
                ("arith", ctypes.c_int)]


# FSN_ARITH_* of include/fsn_hip.h.  "f32" is the default and what every parity claim refers to; "f16x3" is
# the opt-in split-precision experiment (Model.arithmetic = "f16x3", or FSN_F16X3=1 in the environment of the
# HOST process - the library itself reads no environment variables).
ARITH = {"f32": 0, "f16x3": 1, "f16": 2, "bf16": 3}  # "f16" / "bf16": training entries only (autocast arithmetic)
ARITH_SAVES16 = 0x100  # FSN_ARITH_SAVES16: "f16+s16" / "bf16+s16" - the saved gates in the 16-bit type too
ARITH.update({"f16+s16": 2 | ARITH_SAVES16, "bf16+s16": 3 | ARITH_SAVES16})


def default_arith():
    return "f16x3" if os.environ.get("FSN_F16X3", "")[:1] == "1" else "f32"


PARAM_FIELDS = [
    "fb_w_ih_l0", "fb_w_hh_l0", "fb_b_ih_l0", "fb_b_hh_l0", "fb_w_ih_l1", "fb_w_hh_l1", "fb_b_ih_l1", "fb_b_hh_l1",
    "fb_fc_w", "fb_fc_b",
    "sb_w_ih_l0", "sb_w_hh_l0", "sb_b_ih_l0", "sb_b_hh_l0", "sb_w_ih_l1", "sb_w_hh_l1", "sb_b_ih_l1", "sb_b_hh_l1",
    "sb_fc_w", "sb_fc_b",
]
# reference state_dict key of each field
STATE_KEYS = [
    "fb_model.sequence_model.weight_ih_l0", "fb_model.sequence_model.weight_hh_l0",
    "fb_model.sequence_model.bias_ih_l0", "fb_model.sequence_model.bias_hh_l0",
    "fb_model.sequence_model.weight_ih_l1", "fb_model.sequence_model.weight_hh_l1",
    "fb_model.sequence_model.bias_ih_l1", "fb_model.sequence_model.bias_hh_l1",
    "fb_model.fc_output_layer.weight", "fb_model.fc_output_layer.bias",
    "sb_model.sequence_model.weight_ih_l0", "sb_model.sequence_model.weight_hh_l0",
    "sb_model.sequence_model.bias_ih_l0", "sb_model.sequence_model.bias_hh_l0",
    "sb_model.sequence_model.weight_ih_l1", "sb_model.sequence_model.weight_hh_l1",
    "sb_model.sequence_model.bias_ih_l1", "sb_model.sequence_model.bias_hh_l1",
    "sb_model.fc_output_layer.weight", "sb_model.fc_output_layer.bias",
]


class Lstm2Stack(ctypes.Structure):
    """fsn_lstm2_stack (include/fsn_hip.h)."""
    _fields_ = ([("x", ctypes.c_void_p), ("ldx", ctypes.c_long)] +
                [(n, ctypes.c_void_p) for n in ("w_ih0", "w_hh0", "b_ih0", "b_hh0", "w_ih1", "w_hh1", "b_ih1", "b_hh1")] +
                [("N", ctypes.c_int), ("I", ctypes.c_int), ("H0", ctypes.c_int), ("H1", ctypes.c_int),
                 ("hseq1", ctypes.c_void_p)])


class AdamCfg(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("max_norm", ctypes.c_float), ("step", ctypes.c_int)]


ADAM_MAX_TENSORS = 32  # FSN_ADAM_MAX_TENSORS


class TrainDims(ctypes.Structure):  # fsn_train_dims
    _fields_ = [("B", ctypes.c_int), ("F", ctypes.c_int), ("T", ctypes.c_int), ("look_ahead", ctypes.c_int),
                ("nb", ctypes.c_int), ("groups", ctypes.c_int), ("norm", ctypes.c_int)]


ABI_VERSION = 117  # FSN_ABI_VERSION of include/fsn_hip.h these signatures were written against


class MaskSection(ctypes.Structure):  # fsn_mask_section
    _fields_ = [("o", ctypes.c_void_p), ("Np", ctypes.c_int), ("ld", ctypes.c_int), ("lower", ctypes.c_int),
                ("units", ctypes.c_int), ("center", ctypes.c_int)]


class Params(ctypes.Structure):
    _fields_ = [(n, _f32p) for n in PARAM_FIELDS]


# name -> (restype, argtypes); must list every symbol declared in include/fsn_hip.h
_c = ctypes
SIGNATURES = {
    "fsn_last_error": (_c.c_char_p, []),
    "fsn_version": (_c.c_int, []),
    "fsn_stft": (_c.c_int, [_f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _f32p, _f32p, _f32p, _f32p,
                            _c.c_void_p]),
    "fsn_istft_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    "fsn_istft": (_c.c_int, [_f32p, _f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_int, _f32p,
                             _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_decompress_cirm": (_c.c_int, [_f32p, _f32p, _c.c_size_t, _c.c_void_p]),
    "fsn_compress_cirm": (_c.c_int, [_f32p, _f32p, _c.c_size_t, _c.c_void_p]),
    "fsn_build_cirm": (_c.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _c.c_size_t, _c.c_void_p]),
    "fsn_norm_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_norm": (_c.c_int, [_f32p, _f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float,
                            _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_fullsubnet_packed_bytes": (_c.c_size_t, [_c.POINTER(Cfg)]),
    "fsn_fullsubnet_pack": (_c.c_int, [_c.POINTER(Cfg), _c.POINTER(Params), _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_fullsubnet_workspace_bytes": (_c.c_size_t, [_c.POINTER(Cfg), _c.c_int, _c.c_int]),
    "fsn_fullsubnet_forward": (_c.c_int, [_c.POINTER(Cfg), _c.c_void_p, _f32p, _c.c_int, _c.c_int, _f32p, _c.c_void_p,
                                          _c.c_size_t, _c.c_void_p]),
    "fsn_fullsubnet_fullband": (_c.c_int, [_c.POINTER(Cfg), _c.c_void_p, _f32p, _c.c_int, _c.c_int, _f32p, _c.c_void_p,
                                           _c.c_size_t, _c.c_void_p]),
    "fsn_fullsubnet_rows_workspace_bytes": (_c.c_size_t, [_c.POINTER(Cfg), _c.c_int, _c.c_int, _c.c_long, _c.c_long]),
    "fsn_fullsubnet_forward_rows": (_c.c_int, [_c.POINTER(Cfg), _c.c_void_p, _f32p, _c.c_int, _c.c_int, _c.c_long,
                                               _c.c_long, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_enhance_workspace_bytes": (_c.c_size_t, [_c.POINTER(Cfg), _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_enhance": (_c.c_int, [_c.POINTER(Cfg), _c.c_void_p, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                               _f32p, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_fullsubnet_stream_state_bytes": (_c.c_size_t, [_c.POINTER(Cfg), _c.c_int]),
    "fsn_fullsubnet_stream_workspace_bytes": (_c.c_size_t, [_c.POINTER(Cfg), _c.c_int, _c.c_int]),
    "fsn_fullsubnet_stream_step": (_c.c_int, [_c.POINTER(Cfg), _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int, _f32p,
                                              _c.c_int, _c.c_int, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_debug_poison_if": (_c.c_int, [_c.c_void_p, _f32p, _c.c_size_t, _c.c_void_p]),
    "fsn_lstm_layer_save_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    "fsn_lstm_layer_fwd_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_lstm_layer_forward": (_c.c_int, [_f32p, _c.c_long, _f32p, _f32p, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_int,
                                          _c.c_int, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t,
                                          _c.c_void_p]),
    "fsn_lstm_layer_bwd_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_lstm2_train_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_lstm2_train_is_persistent": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_lstm2_forward_train": (_c.c_int, [_f32p, _c.c_long] + [_f32p] * 8 + [_c.c_int] * 4 + [_f32p, _f32p, _c.c_void_p,
                                           _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_void_p]),
    "fsn_lstm2_bwd_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_lstm2_backward": (_c.c_int, [_f32p, _f32p, _c.c_long, _f32p, _f32p, _f32p, _f32p] + [_c.c_int] * 4 +
                           [_f32p, _f32p, _c.c_void_p, _c.c_void_p, _f32p, _c.c_long] + [_f32p] * 6 +
                           [_c.c_void_p, _c.c_size_t, _c.c_int, _c.c_void_p]),
    "fsn_lstm_layer_plan_rows": (_c.c_int, [_c.c_int, _c.c_int]),
    "fsn_lstm_layer_fc_supported": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_long, _c.c_int, _c.c_int]),
    "fsn_lstm_layer_fc_workspace_bytes": (_c.c_size_t, [_c.c_int] * 4),
    "fsn_lstm_layer_forward_fc": (_c.c_int, [_f32p, _c.c_long, _f32p, _f32p, _f32p, _f32p] + [_c.c_int] * 4 +
                                  [_f32p, _f32p, _c.c_int, _f32p, _f32p, _c.c_long, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_lstm2_backward_phase": (_c.c_int, [_f32p, _f32p, _c.c_long, _f32p, _f32p, _f32p, _f32p] + [_c.c_int] * 4 +
                                 [_f32p, _f32p, _c.c_void_p, _c.c_void_p, _f32p, _c.c_long] + [_f32p] * 6 +
                                 [_c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_void_p]),
    "fsn_lstm_layer_backward": (_c.c_int, [_f32p, _f32p, _c.c_long, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_int, _f32p, _c.c_void_p, _f32p, _c.c_long, _f32p, _f32p, _f32p,
                                           _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_lstm2_forward_is_persistent": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_long, _c.c_int, _c.c_int]),
    "fsn_lstm2_fwd_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_lstm2_forward": (_c.c_int, [_f32p, _c.c_long] + [_f32p] * 8 + [_c.c_int] * 5 + [_f32p, _c.c_void_p, _c.c_size_t,
                                                                                         _c.c_void_p]),
    "fsn_improved_section_input_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "fsn_improved_section_input": (_c.c_int, [_f32p, _f32p] + [_c.c_int] * 11 + [_c.c_float, _f32p, _c.c_int, _c.c_int,
                                                                                _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_lstm2_multi_is_persistent": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_int]),
    "fsn_lstm2_multi_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_void_p, _c.c_int]),
    "fsn_lstm2_forward_multi": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_lstm_layer_packed_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "fsn_lstm_layer_pack": (_c.c_int, [_f32p, _f32p, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_size_t,
                                       _c.c_void_p]),
    "fsn_lstm_layer_state_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    "fsn_lstm_layer_forward_state": (_c.c_int, [_f32p, _c.c_long, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                                _f32p, _f32p, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_gru2_forward_supported": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    "fsn_gru2_fwd_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_gru2_forward": (_c.c_int, [_f32p, _c.c_long] + [_f32p] * 8 + [_c.c_int] * 4 + [_f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_gru_layer_is_persistent": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_long, _c.c_int]),
    "fsn_gru_layer_save_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    "fsn_gru_layer_fwd_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_gru_layer_forward": (_c.c_int, [_f32p, _c.c_long, _f32p, _f32p, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_int,
                                         _c.c_int, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t,
                                         _c.c_void_p]),
    "fsn_gru_layer_forward_state": (_c.c_int, [_f32p, _c.c_long, _f32p, _f32p, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_int,
                                               _c.c_int, _f32p, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_gru_layer_bwd_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "fsn_gru_layer_backward": (_c.c_int, [_f32p, _f32p, _c.c_long, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                          _f32p, _c.c_void_p, _f32p, _c.c_long, _f32p, _f32p, _f32p, _f32p,
                                          _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_linear_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    "fsn_linear_forward": (_c.c_int, [_f32p, _c.c_long, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _f32p,
                                      _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_linear_backward": (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_int, _c.c_int, _c.c_int, _f32p,
                                       _c.c_long, _f32p, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_mse_loss_workspace_bytes": (_c.c_size_t, [_c.c_size_t]),
    "fsn_mse_loss": (_c.c_int, [_f32p, _f32p, _c.c_size_t, _f32p, _f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_clip_adam_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.POINTER(_c.c_size_t)]),
    "fsn_clip_adam_step": (_c.c_int, [_c.c_int, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p),
                                      _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t),
                                      _c.c_void_p, _f32p, _f32p, _f32p, _c.c_void_p, _c.c_void_p, _c.c_size_t,
                                      _c.c_void_p]),
    "fsn_train_rows": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "fsn_train_glue_workspace_bytes": (_c.c_size_t, [_c.c_void_p]),
    "fsn_train_den_elems": (_c.c_size_t, [_c.c_void_p, _c.c_int]),
    "fsn_train_fb_input": (_c.c_int, [_c.c_void_p, _f32p, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_train_sb_input": (_c.c_int, [_c.c_void_p, _f32p, _f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _c.c_int, _f32p,
                                      _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_train_sb_input_backward": (_c.c_int, [_c.c_void_p, _f32p, _f32p, _c.c_int, _f32p, _f32p, _c.c_long, _c.c_int, _f32p,
                                               _c.c_long, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_train_mask_out": (_c.c_int, [_c.c_void_p, _f32p, _c.c_int, _f32p, _c.c_void_p]),
    "fsn_train_mask_grad": (_c.c_int, [_c.c_void_p, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_void_p]),
    "fsn_train_rows_pieces": (_c.c_int, [_f32p, _f32p, _c.c_int, _c.c_long, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "fsn_train_cirm_target": (_c.c_int, [_c.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, _c.c_void_p]),
    "fsn_scale_by_scalar": (_c.c_int, [_f32p, _f32p, _f32p, _c.c_size_t, _c.c_void_p]),
    "fsn_fast_low_rate_frames": (_c.c_int, [_c.c_int, _c.c_int]),
    "fsn_fast_glue_workspace_bytes": (_c.c_size_t, [_c.c_int] * 4),
    "fsn_fast_spec_rows": (_c.c_int, [_f32p] + [_c.c_int] * 4 + [_f32p, _c.c_int, _c.c_int, _c.c_void_p]),
    "fsn_fast_norm_rows": (_c.c_int, [_f32p] + [_c.c_int] * 4 + [_f32p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "fsn_fast_bottleneck_input": (_c.c_int, [_f32p, _f32p, _c.c_long] + [_c.c_int] * 7 + [_f32p, _c.c_int, _c.c_int, _c.c_void_p,
                                             _c.c_size_t, _c.c_void_p]),
    "fsn_fast_decoder_input": (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _c.c_long] + [_c.c_int] * 6 + [_f32p, _c.c_void_p]),
    "fsn_fast_mask_out": (_c.c_int, [_f32p, _c.c_long] + [_c.c_int] * 5 + [_f32p, _c.c_void_p]),
    "fsn_profile_enable": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "fsn_stream_timeout_policy": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "fsn_debug_g16_kernels": (_c.c_int, [_c.c_int]),
    "fsn_debug_tn16h_wide": (_c.c_int, [_c.c_int]),
    "fsn_set_persistent_mode": (_c.c_int, [_c.c_int]),
    "fsn_set_persistent_timeout_ms": (_c.c_int, [_c.c_int]),
    "fsn_stream_status": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_c.c_uint), _c.POINTER(_c.c_uint)]),
    "fsn_stream_status_clear": (_c.c_int, [_c.c_void_p]),
    "fsn_debug_hog": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_float, _f32p, _c.c_void_p]),
    "fsn_debug_persist_stats": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "fsn_debug_persist_set_fits": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p]),
    "fsn_debug_tn_plan": (_c.c_int, [_c.c_int, _c.c_int, _c.c_long, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "fsn_debug_core_chunks": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int]),
    "fsn_debug_core_plan": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int]),
    "fsn_improved_front": (_c.c_int, [_f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_void_p]),
    "fsn_bft_to_rows": (_c.c_int, [_f32p, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_int, _c.c_int, _c.c_void_p]),
    "fsn_rows_to_bft": (_c.c_int, [_f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_void_p]),
    "fsn_improved_mask_apply": (_c.c_int, [_c.c_int, _c.c_void_p, _f32p, _f32p, _c.c_int, _c.c_int, _c.c_int, _f32p, _f32p,
                                           _c.c_void_p]),
    "fsn_profile_num_stages": (_c.c_int, []),
    "fsn_profile_stage_name": (_c.c_char_p, [_c.c_int]),
    "fsn_profile_read": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_float), _c.c_int]),
}


def lib():
    """Load (once) and return the HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FsnError(
                f"{LIB_PATH} is missing: build it with `python -m fullsubnet_amd.build` (hipcc, gfx950). "
                "There is no CPU / PyTorch fallback for this path.")
        handle = ctypes.CDLL(LIB_PATH)
        handle.fsn_version.restype = ctypes.c_int
        built = handle.fsn_version()
        if built != ABI_VERSION:  # argument lists differ between revisions: a stale library would misread them silently
            raise FsnError(f"{LIB_PATH} was built for ABI revision {built}, this package binds revision {ABI_VERSION}: "
                           "rebuild it with `python -m fullsubnet_amd.build`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI drifted from include/fsn_hip.h
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class FsnTimeout(FsnError):
    """A persistent kernel ran out of time waiting for its partner workgroups (FSN_ERR_TIMEOUT, include/fsn_hip.h
    "residency contract"): its outputs are NaN and the stream refuses further persistent launches until
    `stream_status_clear()`."""


def check(rc):
    if rc == -4:
        raise FsnTimeout(f"libfsn_hip error {rc}: {lib().fsn_last_error().decode()}")
    if rc != 0:
        raise FsnError(f"libfsn_hip error {rc}: {lib().fsn_last_error().decode()}")


def stream_status(device=None, synchronize=True, raise_on_timeout=True):
    """(status, events) of the current stream of `device`: the sticky record of persistent launches that ran out of
    time (0, 0 = none).  Raises FsnTimeout when raised (unless raise_on_timeout is False)."""
    st, ev = ctypes.c_uint(0), ctypes.c_uint(0)
    rc = lib().fsn_stream_status(stream_ptr(device), 1 if synchronize else 0, ctypes.byref(st), ctypes.byref(ev))
    if rc != 0 and (raise_on_timeout or rc != -4):
        check(rc)
    return st.value, ev.value


def stream_status_clear(device=None):
    check(lib().fsn_stream_status_clear(stream_ptr(device)))


def stream_timeout_policy(policy, device=None):
    """"refuse" (default): a raised timeout record makes later persistent launches on the current stream of `device`
    fail with FsnTimeout until cleared.  "defer": they run regardless (a training step in flight; the poison is NaN,
    the fused optimizer skips, the caller reads `stream_status` once per step)."""
    check(lib().fsn_stream_timeout_policy(stream_ptr(device), {"refuse": 0, "defer": 1}[policy]))


def persist_stats():
    """(launches, waits, unreported) of the persistent-launch gate (test hook)."""
    v = [ctypes.c_uint(0) for _ in range(3)]
    check(lib().fsn_debug_persist_stats(*(ctypes.byref(x) for x in v)))
    return tuple(x.value for x in v)


def set_persistent_mode(mode):
    """"auto" (default) or "never": whether the kernels that need their whole grid resident may be used."""
    check(lib().fsn_set_persistent_mode({"auto": 0, "never": 1}[mode]))


def set_persistent_timeout_ms(ms):
    check(lib().fsn_set_persistent_timeout_ms(int(ms)))


def dev_ptr(t, name="tensor", allow_none=False):
    """Validated raw device pointer of a contiguous fp32 ROCm tensor."""
    if t is None:
        if allow_none:
            return None
        raise FsnError(f"{name} is None")
    if not t.is_cuda:
        raise FsnError(f"{name} must live on a ROCm device (got {t.device}); this path has no CPU implementation")
    if t.dtype != torch.float32:
        raise FsnError(f"{name} must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise FsnError(f"{name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_CANARY = 4096     # bytes of guard pattern behind every workspace when FSN_WS_CANARY is set (memory-safety runs)
_canaries = []      # (guarded tensor, payload bytes) not yet verified


def workspace(nbytes, device):
    """Workspace from PyTorch's caching allocator (the library never allocates).  With FSN_WS_CANARY in the environment
    (tools/gpu_run_nocache.sh) every workspace is followed by a guard pattern that `check_canaries` verifies: a kernel
    that writes beyond the size its `*_workspace_bytes` query promised is caught even when the overrun stays inside
    memory the process owns."""
    if nbytes <= 0:
        raise FsnError(f"workspace query failed: {lib().fsn_last_error().decode()}")
    if os.environ.get("FSN_WS_CANARY"):
        buf = torch.empty(nbytes + _CANARY, dtype=torch.uint8, device=device)
        buf[nbytes:].fill_(0xA5)
        _canaries.append((buf, nbytes))
        return buf[:nbytes]
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def check_canaries():
    """Verify and forget the guard patterns of the workspaces handed out since the last call (synchronises)."""
    bad = []
    while _canaries:
        buf, n = _canaries.pop()
        if not bool((buf[n:] == 0xA5).all()):
            bad.append(n)
    if bad:
        raise FsnError(f"workspace overrun: the guard behind workspaces of {bad} bytes was overwritten")


def profile_enable(on, device=None):
    """Per-stage hipEvent timing for calls made on the current stream of `device` (per stream: two callers do not
    race on it)."""
    check(lib().fsn_profile_enable(stream_ptr(device), 1 if on else 0))


def profile_stage_names():
    L = lib()
    return [L.fsn_profile_stage_name(i).decode() for i in range(L.fsn_profile_num_stages())]


def core_plan(cfg, B, T):
    """How the sub-band model of a B-utterance, T-frame call is spread over the device (fsn_debug_core_plan)."""
    buf = (ctypes.c_int * 9)()
    if lib().fsn_debug_core_plan(ctypes.byref(cfg), B, T, buf, 9) != 0:
        raise FsnError("fsn_debug_core_plan: bad arguments")
    keys = ("rows", "tiles", "row_tiles_per_workgroup", "persistent_workgroups", "left_over_tiles", "group_clusters",
            "fullband_chain", "chunks", "persistent_rows_all_chunks")  # the first seven: the FIRST chunk's plan
    return dict(zip(keys, list(buf)))


def profile_read(device=None):
    """Stage times (ms) of the last profiled call on the current stream of `device`."""
    L = lib()
    n = L.fsn_profile_num_stages()
    buf = (ctypes.c_float * n)()
    check(L.fsn_profile_read(stream_ptr(device), buf, n))
    return dict(zip(profile_stage_names(), list(buf)))
