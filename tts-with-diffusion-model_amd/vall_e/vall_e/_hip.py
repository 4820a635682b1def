"""ctypes binding of libd3pm_hip.so (C ABI in include/d3pm_hip.h).

There is deliberately no CPU fallback: if the shared library is missing the import of the
sampler fails loudly (build it with `python -c "import __graft_entry__ as g; g.build()"`).
PyTorch is used only to own device memory and to provide the current HIP stream.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import numpy as np
import torch

F32, F16, BF16 = 0, 1, 2
ABI_VERSION = 6
FLAG_GREEDY, FLAG_FORCE_GENERIC, FLAG_SEED_IN_HBM = 1, 2, 4
K_GEMM, K_ATTN, K_SAMPLE, K_LN, K_GEMM_LN, K_ALL = 0, 1, 2, 3, 4, 5

_DTYPES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
_LIB_DIR = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "lib"))
LIB_PATH = os.environ.get("D3PM_HIP_LIB") or os.path.join(_LIB_DIR, "libd3pm_hip.so")
AB_LIB_PATH = os.path.join(_LIB_DIR, "libd3pm_hip_ab.so")      # the experiment build (include/d3pm_hip_ab.h); tests/ab_*.py only


class Tuning(C.Structure):
    """d3pm_tuning (include/d3pm_hip.h): schedule choices, every value bit-identical.  The C library keeps no state: a
    pointer to one of these travels in the shape structs.  `TUNING` below is this module's default instance."""
    _fields_ = [(n, C.c_int32) for n in ("gemm_variant", "gemm_persist_slots", "lat_tile", "attn_query_groups",
                                         "attn_pair_sequential", "attn_cross_resident", "row_panel", "workspace_alias",
                                         "regime_batch", "ln_fold")] + \
               [("prof", C.c_void_p)]


class Shape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "n_heads", "n_layers", "canvas", "s_text", "s_prompt",
                                         "n_classes", "mask_id", "timesteps", "dtype", "n_q")] + [("tuning", C.POINTER(Tuning))]


_BLOCK_FIELDS = ("norm1_w", "norm1_b", "attn_in_w", "attn_in_b", "attn_out_w", "attn_out_b", "norm2_w", "norm2_b",
                 "norm22_w", "norm22_b", "cross_in_w", "cross_in_b", "cross_out_w", "cross_out_b", "norm3_w",
                 "norm3_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "tfc_w", "tfc_b")


class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _BLOCK_FIELDS]


class FoldBlock(C.Structure):
    """d3pm_fold_block: a block's LayerNorms folded into the projections they feed (filled by d3pm_fold_build)."""
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w", "qkv_s", "qkv_b", "q2_w", "q2_s", "q2_b")]


class Weights(C.Structure):
    _fields_ = [("resps_emb", C.c_void_p), ("time_emb", C.c_void_p), ("final_w", C.c_void_p),
                ("final_b", C.c_void_p), ("blocks", C.POINTER(BlockWeights)), ("fold", C.POINTER(FoldBlock))]


_ENC_LAYER_FIELDS = ("in_w", "in_b", "out_w", "out_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "norm1_w", "norm1_b",
                     "norm2_w", "norm2_b")


class EncoderLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _ENC_LAYER_FIELDS]


class EncoderWeights(C.Structure):
    _fields_ = [("layers", C.POINTER(EncoderLayerWeights)), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
                ("d_ff", C.c_int32), ("mlp_hidden", C.c_int32), ("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p),
                ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p)]


class CondWeights(C.Structure):
    _fields_ = [("text_emb", C.c_void_p), ("proms_emb", C.c_void_p), ("pe_text0", C.c_void_p),
                ("pe_prompt", C.c_void_p), ("n_levels", C.c_int32), ("text_encoder", EncoderWeights),
                ("prompt_encoder", EncoderWeights)]


class NarShape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "n_heads", "n_layers", "n_tokens", "n_prom_levels", "n_resp_levels",
                                         "dtype")] + [("tuning", C.POINTER(Tuning))]


_NAR_BLOCK_FIELDS = ("attn_norm_emb", "to_qkv_w", "to_out_w", "to_out_b", "ffn_norm_emb", "ffn0_w", "ffn0_b", "ffn3_w",
                     "ffn3_b")


class NarBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _NAR_BLOCK_FIELDS]


class NarWeights(C.Structure):
    _fields_ = [("text_emb", C.c_void_p), ("proms_emb", C.c_void_p), ("resps_emb", C.c_void_p), ("sep", C.c_void_p),
                ("classifier_w", C.c_void_p), ("classifier_b", C.c_void_p), ("pe", C.c_void_p), ("pe_rows", C.c_int32),
                ("blocks", C.POINTER(NarBlockWeights))]


class ScheduleC(C.Structure):
    _fields_ = [("timesteps", C.c_int32), ("d", C.POINTER(C.c_uint16)), ("c", C.POINTER(C.c_uint16)),
                ("dbar", C.POINTER(C.c_uint16)), ("cbar", C.POINTER(C.c_uint16))]


_lib = None

# name -> (restype, argtypes); every symbol include/d3pm_hip.h declares
SIGNATURES = {
    "d3pm_abi_version": (C.c_int, []),
    "d3pm_last_error": (C.c_char_p, []),
    "d3pm_schedule_build": (C.c_int, [C.c_int] + [C.POINTER(C.c_uint16)] * 5),
    "d3pm_workspace_bytes": (C.c_size_t, [C.POINTER(Shape), C.c_int]),
    "d3pm_film_table": (C.c_int, [C.POINTER(Shape), C.POINTER(Weights), C.c_void_p, C.c_void_p]),
    "d3pm_fold_bytes": (C.c_size_t, [C.POINTER(Shape)]),
    "d3pm_fold_build": (C.c_int, [C.POINTER(Shape), C.POINTER(Weights), C.c_void_p, C.c_size_t, C.POINTER(FoldBlock), C.c_void_p]),
    "d3pm_op_row_stats": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "d3pm_op_linear_stats": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Tuning), C.c_void_p]),
    "d3pm_op_linear_fold": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Tuning), C.c_void_p]),
    "d3pm_op_fold_weights": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "d3pm_cond_kv": (C.c_int, [C.POINTER(Shape), C.POINTER(Weights), C.c_int] + [C.c_void_p] * 5),
    "d3pm_cond_workspace_bytes": (C.c_size_t, [C.POINTER(Shape), C.POINTER(CondWeights), C.c_int]),
    "d3pm_encode_conditions": (C.c_int, [C.POINTER(Shape), C.POINTER(CondWeights), C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "d3pm_denoise_step": (C.c_int, [C.POINTER(Shape), C.POINTER(Weights), C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]),
    "d3pm_posterior_sample": (C.c_int, [C.POINTER(Shape), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int, C.POINTER(ScheduleC), C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p]),
    "d3pm_sample_loop": (C.c_int, [C.POINTER(Shape), C.POINTER(Weights), C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(ScheduleC), C.c_uint64,
                                   C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "d3pm_q_sample": (C.c_int, [C.POINTER(Shape), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.POINTER(ScheduleC), C.c_uint64, C.c_uint32, C.c_void_p]),
    "d3pm_denoise_step_fp8": (C.c_int, [C.POINTER(Shape), C.POINTER(Weights), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]),
    "d3pm_sample_loop_fp8": (C.c_int, [C.POINTER(Shape), C.POINTER(Weights), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(ScheduleC),
                                       C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "d3pm_op_quantize_mx": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "d3pm_op_layernorm_mx": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "d3pm_op_linear_mx": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.POINTER(Tuning), C.c_void_p]),
    "d3pm_ce_loss_rows": (C.c_int, [C.POINTER(Shape), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "d3pm_uniform": (C.c_int, [C.c_uint64, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "d3pm_nar_workspace_bytes": (C.c_size_t, [C.POINTER(NarShape), C.c_int, C.c_int]),
    "d3pm_nar_level": (C.c_int, [C.POINTER(NarShape), C.POINTER(NarWeights), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64,
                                 C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "d3pm_op_linear": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.POINTER(Tuning), C.c_void_p]),
    "d3pm_op_attention": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.POINTER(Tuning), C.c_void_p]),
    "d3pm_op_attention_pair": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_float, C.POINTER(Tuning), C.c_void_p]),
    "d3pm_op_layernorm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_float, C.c_void_p]),
    "d3pm_op_linear_rowpanel": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "d3pm_op_cond_embed": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]),
    "d3pm_op_matmul_f32": (C.c_int, [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "d3pm_op_colsum_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "d3pm_op_act_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "d3pm_op_mask_rows_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "d3pm_op_layernorm_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "d3pm_op_attention_bwd_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "d3pm_op_ce_bwd_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                     C.c_int, C.c_void_p]),
    "d3pm_op_embed_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "d3pm_op_embed_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    "d3pm_tuning_default": (None, [C.POINTER(Tuning)]),
    "d3pm_prof_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "d3pm_prof_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double)]),
    "d3pm_prof_read_class": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]),
    "d3pm_prof_destroy": (C.c_int, [C.c_void_p]),
}

# what libd3pm_hip_ab.so exports on top (include/d3pm_hip_ab.h): experiments that were measured and not shipped
AB_SIGNATURES = {
    "d3pm_ab_set": (C.c_int, [C.c_int, C.c_int]),
    "d3pm_op_linear_lnpro": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "d3pm_op_final_sample": (C.c_int, [C.POINTER(Shape), C.POINTER(Weights), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.POINTER(ScheduleC), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]),
    "d3pm_debug_gemm_clock": (C.c_int, [C.POINTER(C.c_uint64)]),
    "d3pm_debug_attn32_stamps": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
}
AB_GEMM_BIG_MODE, AB_ATTN_ARM, AB_GEMM_RING, AB_GELU_TABLE, AB_LN_PROLOGUE, AB_FUSED_FINAL_SAMPLE = range(6)
_is_ab = False
TUNING = Tuning()        # filled by d3pm_tuning_default when the library loads
TUNING_FIELDS = tuple(n for n, _ in Tuning._fields_ if n != "prof")


def _load(path, signatures):
    if not os.path.isfile(path):
        raise RuntimeError(f"{path} is missing: the HIP extension has not been built "
                           "(run __graft_entry__.build()); there is no CPU fallback")
    handle = C.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = res, args
    if handle.d3pm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{os.path.basename(path)} ABI version mismatch")
    return handle


def lib():
    """The loaded library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, SIGNATURES)
        _lib.d3pm_tuning_default(C.byref(TUNING))
    return _lib


def use_ab_library():
    """A/B scripts only (tests/ab_*.py): route this module to libd3pm_hip_ab.so -- the same sources built with
    -DD3PM_ABLATIONS (__graft_entry__.build_ab()) -- which also carries the experiments of include/d3pm_hip_ab.h."""
    global _lib, _is_ab
    _lib = _load(AB_LIB_PATH, {**SIGNATURES, **AB_SIGNATURES})
    _lib.d3pm_tuning_default(C.byref(TUNING))
    _is_ab = True
    return _lib


def _ab_set(knob: int, value: int):
    if not _is_ab:
        raise D3PMError("this knob belongs to an experiment that lives in libd3pm_hip_ab.so only: call _hip.use_ab_library() "
                        "(tests/ab_*.py); the product library has no such path")
    check(lib().d3pm_ab_set(knob, int(value)), "d3pm_ab_set")


class D3PMError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        raise D3PMError(f"{what} failed with code {rc}: {lib().d3pm_last_error().decode()}")


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise D3PMError(f"unsupported model dtype {dtype}") from None


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class Schedule:
    """Host-side fp16 schedule scalars (d3pm_schedule_build)."""

    def __init__(self, timesteps: int):
        self.timesteps = timesteps
        self.betas = np.zeros(timesteps + 1, np.uint16)
        self.d, self.c, self.dbar, self.cbar = (np.zeros(timesteps, np.uint16) for _ in range(4))
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint16))
        check(lib().d3pm_schedule_build(timesteps, p(self.betas), p(self.d), p(self.c), p(self.dbar), p(self.cbar)),
              "d3pm_schedule_build")
        self.c_struct = ScheduleC(timesteps, p(self.d), p(self.c), p(self.dbar), p(self.cbar))


def make_shape(cfg, dtype: torch.dtype, tuning: "Tuning | None" = None) -> Shape:
    lib()                                             # TUNING holds the library's defaults from here on
    return Shape(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.canvas, cfg.s_text, cfg.s_prompt, cfg.n_classes,
                 cfg.mask_id, cfg.timesteps, dtype_code(dtype), getattr(cfg, "n_q", 1),
                 C.pointer(tuning if tuning is not None else TUNING))


class DeviceWeights:
    """Pointer table over tensors that stay owned by the caller (an nn.Module's parameters)."""

    def __init__(self, tensors: dict, n_layers: int):
        self._keep = []

        def ptr(key):
            t = tensors[key]
            if not (t.is_cuda and t.is_contiguous()):
                raise D3PMError(f"weight {key} must be a contiguous device tensor")
            self._keep.append(t)
            return t.data_ptr()

        self.blocks = (BlockWeights * n_layers)()
        # n_q > 1: resps_emb.weight is [n_q, K, d] and final.{weight, bias} are [n_q * K, d] / [n_q * K], both contiguous -- the
        # layouts d3pm_weights documents
        names = {"norm1_w": "norm1.weight", "norm1_b": "norm1.bias", "attn_in_w": "attn.in_proj_weight",
                 "attn_in_b": "attn.in_proj_bias", "attn_out_w": "attn.out_proj.weight",
                 "attn_out_b": "attn.out_proj.bias", "norm2_w": "norm2.weight", "norm2_b": "norm2.bias",
                 "norm22_w": "norm22.weight", "norm22_b": "norm22.bias", "cross_in_w": "cross_attn.in_proj_weight",
                 "cross_in_b": "cross_attn.in_proj_bias", "cross_out_w": "cross_attn.out_proj.weight",
                 "cross_out_b": "cross_attn.out_proj.bias", "norm3_w": "norm3.weight", "norm3_b": "norm3.bias",
                 "fc1_w": "mlp.fc1.weight", "fc1_b": "mlp.fc1.bias", "fc2_w": "mlp.fc2.weight",
                 "fc2_b": "mlp.fc2.bias", "tfc_w": "timestep_fc.weight", "tfc_b": "timestep_fc.bias"}
        for i in range(n_layers):
            for field, key in names.items():
                setattr(self.blocks[i], field, ptr(f"blocks.{i}.{key}"))
        self.c_struct = Weights(ptr("resps_emb.weight"), ptr("time_emb.weight"), ptr("final.weight"),
                                ptr("final.bias"), self.blocks, None)
        self.fold_blocks = self.fold_storage = None

    def build_fold(self, shape: "Shape", device):
        """The LayerNorm-folded projection tables (d3pm_fold_build): W o gamma for the QKV and the merged cross-attention query
        projection of every block plus their fp32 column vectors (fc1's FiLM-folded copy depends on the timestep and is rebuilt
        inside every denoiser evaluation).  Shapes that do not qualify (fp32, d_model not a multiple of 256) keep fold = NULL."""
        need = lib().d3pm_fold_bytes(C.byref(shape))
        if need == 0:
            return False
        self.fold_storage = torch.empty(need, dtype=torch.uint8, device=device)
        self.fold_blocks = (FoldBlock * shape.n_layers)()
        check(lib().d3pm_fold_build(C.byref(shape), C.byref(self.c_struct), _p(self.fold_storage), need, self.fold_blocks,
                                    stream_ptr()), "d3pm_fold_build")
        self.c_struct.fold = self.fold_blocks
        return True


class Fp8BlockWeights(C.Structure):
    _fields_ = [("attn_in_w8", C.c_void_p), ("attn_in_scale", C.c_void_p), ("cross_in_w8", C.c_void_p),
                ("cross_in_scale", C.c_void_p), ("fc1_w8", C.c_void_p), ("fc1_scale", C.c_void_p),
                ("fc2_w8", C.c_void_p), ("fc2_scale", C.c_void_p)]


class DeviceFp8Weights:
    """Block-scaled e4m3 copies (quantize_mx) of the LayerNorm-fed projections -- and of fc2 unless `fc2=False` -- of every
    block for the fp8 fast path."""

    def __init__(self, tensors: dict, n_layers: int, d_model: int, fc2: bool = True):
        self._keep = []
        self.blocks = (Fp8BlockWeights * n_layers)()
        which = [("attn_in", "attn.in_proj_weight", None), ("cross_in", "cross_attn.in_proj_weight", d_model), ("fc1", "mlp.fc1.weight", None)]
        if fc2:
            which.append(("fc2", "mlp.fc2.weight", None))
        for i in range(n_layers):
            for field, key, rows in which:
                w = tensors[f"blocks.{i}.{key}"]
                codes, scale = quantize_mx(w[:rows] if rows else w)
                self._keep += [codes, scale]
                setattr(self.blocks[i], field + "_w8", codes.data_ptr())
                setattr(self.blocks[i], field + "_scale", scale.data_ptr())


class DeviceCondWeights:
    """Pointer tables of the two condition encoders (+ embeddings and position tables)."""

    def __init__(self, tensors: dict, cfg, pe_text0: torch.Tensor, pe_prompt: torch.Tensor):
        self._keep = [pe_text0, pe_prompt]

        def ptr(key):
            t = tensors[key]
            if not (t.is_cuda and t.is_contiguous()):
                raise D3PMError(f"weight {key} must be a contiguous device tensor")
            self._keep.append(t)
            return t.data_ptr()

        names = {"in_w": "self_attn.in_proj_weight", "in_b": "self_attn.in_proj_bias", "out_w": "self_attn.out_proj.weight",
                 "out_b": "self_attn.out_proj.bias", "lin1_w": "linear1.weight", "lin1_b": "linear1.bias",
                 "lin2_w": "linear2.weight", "lin2_b": "linear2.bias", "norm1_w": "norm1.weight", "norm1_b": "norm1.bias",
                 "norm2_w": "norm2.weight", "norm2_b": "norm2.bias"}

        def encoder(name, mult):
            layers = (EncoderLayerWeights * cfg.cond_layers)()
            for j in range(cfg.cond_layers):
                for field, key in names.items():
                    setattr(layers[j], field, ptr(f"{name}.0.layers.{j}.{key}"))
            self._keep.append(layers)
            return EncoderWeights(layers, cfg.cond_layers, cfg.cond_heads, cfg.cond_ff, mult * cfg.d_model,
                                  ptr(f"{name}.1.fc1.weight"), ptr(f"{name}.1.fc1.bias"), ptr(f"{name}.1.fc2.weight"),
                                  ptr(f"{name}.1.fc2.bias"))

        self.c_struct = CondWeights(ptr("text_emb.weight"), ptr("proms_emb.weight"), pe_text0.data_ptr(),
                                    pe_prompt.data_ptr(), cfg.n_levels, encoder("encodertext", 2), encoder("encoder2", 3))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _require(t, name, shape, dtypes, device):
    """The C ABI takes raw pointers and derives every extent from the shape struct: a tensor of another shape, dtype,
    layout or device would be read / written out of bounds without any error, so it is rejected here."""
    if not isinstance(t, torch.Tensor):
        raise D3PMError(f"{name}: expected a tensor, got {type(t).__name__}")
    if tuple(t.shape) != tuple(shape):
        raise D3PMError(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
    if t.dtype not in dtypes:
        raise D3PMError(f"{name}: dtype {t.dtype} not in {tuple(dtypes)}")
    if t.device != device:
        raise D3PMError(f"{name}: on {t.device}, the sampler lives on {device}")
    if not t.is_contiguous():
        raise D3PMError(f"{name}: must be contiguous")


class Sampler:
    """Thin object API over the C entry points for one (shape, weights) pair."""

    def __init__(self, cfg, tensors: dict, dtype: torch.dtype, device, pe_text0=None, pe_prompt=None):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.shape = make_shape(cfg, dtype)
        self.n_q = max(1, getattr(cfg, "n_q", 1))       # quantizer levels generated jointly (> 1: this build's extension)
        self.weights = DeviceWeights(tensors, cfg.n_layers)
        self._tensors, self._fp8 = tensors, None
        # fp8 fast path: also run fc2 on MX operands (fc1's GELU epilogue then writes the hidden layer in that format).  On: measured
        # on MI355X (profiles/round3_h_ab_mx_fp8_gemm.txt) fc1 + fc2 take 52.5 + 38.1 us per block against 60.1 + 45.0 us with a
        # 16-bit hidden layer and 65.8 + 45.0 us in bf16; the logits error vs the bf16 path grows from 3.1 % to 3.7 %, the
        # teacher-forced id agreement stays at 0.9993 (tests/test_gpu_parity.py); False keeps fc2 a 16-bit GEMM
        self.fp8_fc2 = True
        self.cond_weights = (DeviceCondWeights(tensors, cfg, pe_text0, pe_prompt)
                             if pe_text0 is not None and "encodertext.1.fc1.weight" in tensors else None)
        self._cond_ws = None
        self.schedule = Schedule(cfg.timesteps)
        self._ws = {}
        self._graphs = {}
        self.film = torch.empty((cfg.timesteps + 1, cfg.n_layers, 2 * cfg.d_model), dtype=dtype, device=self.device)
        check(lib().d3pm_film_table(C.byref(self.shape), C.byref(self.weights.c_struct), _p(self.film), stream_ptr()),
              "d3pm_film_table")
        # LayerNorm folded into the projections (include/d3pm_hip.h: d3pm_fold_block; tuning field ln_fold picks it per call)
        self.folded = self.weights.build_fold(self.shape, self.device)

    def fp8_weights(self) -> DeviceFp8Weights:
        """e4m3 weight copies for the fp8 fast path, quantised on first use."""
        if self._fp8 is None:
            if self.cfg.d_model != 512 or self.dtype == torch.float32:
                raise D3PMError("the fp8 fast path needs d_model = 512 and a 16-bit model dtype")
            self._fp8 = DeviceFp8Weights(self._tensors, self.cfg.n_layers, self.cfg.d_model, fc2=self.fp8_fc2)
        return self._fp8

    def workspace(self, batch: int, slot: int = 0) -> torch.Tensor:
        """Scratch for one in-flight call; `slot` separates calls that run concurrently on different streams."""
        need = lib().d3pm_workspace_bytes(C.byref(self.shape), batch)
        if need == 0:
            raise D3PMError("d3pm_workspace_bytes: " + lib().d3pm_last_error().decode())
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need:
            ws = self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def encode_conditions(self, text: torch.Tensor, prompt: torch.Tensor):
        """text int32 [B,S_t] (zero padded), prompt int32 [B,S_p,n_levels] (-1 = level absent) ->
        (cond_text [B,S_t,d], cond_prompt [B,S_p,d]) through d3pm_encode_conditions."""
        if self.cond_weights is None:
            raise D3PMError("this sampler was bound without condition-encoder weights")
        cfg, B = self.cfg, text.shape[0]
        assert text.shape == (B, cfg.s_text) and prompt.shape == (B, cfg.s_prompt, cfg.n_levels)
        text, prompt = text.to(torch.int32).contiguous(), prompt.to(torch.int32).contiguous()
        need = lib().d3pm_cond_workspace_bytes(C.byref(self.shape), C.byref(self.cond_weights.c_struct), B)
        if self._cond_ws is None or self._cond_ws.numel() < need:
            self._cond_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ct = torch.empty((B, cfg.s_text, cfg.d_model), dtype=self.dtype, device=self.device)
        cp = torch.empty((B, cfg.s_prompt, cfg.d_model), dtype=self.dtype, device=self.device)
        check(lib().d3pm_encode_conditions(C.byref(self.shape), C.byref(self.cond_weights.c_struct), B, _p(text),
                                           _p(prompt), _p(ct), _p(cp), _p(self._cond_ws), self._cond_ws.numel(),
                                           stream_ptr()), "d3pm_encode_conditions")
        return ct, cp

    def cond_kv(self, cond_text: torch.Tensor, cond_prompt: torch.Tensor):
        """cond_text [B,S_t,d], cond_prompt [B,S_p,d] -> per-layer K|V tensors [L,B,S,2d]."""
        cfg, B = self.cfg, cond_text.shape[0]
        assert cond_text.shape == (B, cfg.s_text, cfg.d_model) and cond_prompt.shape == (B, cfg.s_prompt, cfg.d_model)
        cond_text, cond_prompt = cond_text.to(self.dtype).contiguous(), cond_prompt.to(self.dtype).contiguous()
        kv_t = torch.empty((cfg.n_layers, B, cfg.s_text, 2 * cfg.d_model), dtype=self.dtype, device=self.device)
        kv_p = torch.empty((cfg.n_layers, B, cfg.s_prompt, 2 * cfg.d_model), dtype=self.dtype, device=self.device)
        check(lib().d3pm_cond_kv(C.byref(self.shape), C.byref(self.weights.c_struct), B, _p(cond_text),
                                 _p(cond_prompt), _p(kv_t), _p(kv_p), stream_ptr()), "d3pm_cond_kv")
        return kv_t, kv_p

    def _check_grid(self, x, frame_mask=None, name="x_t"):
        cfg = self.cfg
        tail = (cfg.canvas,) if self.n_q == 1 else (cfg.canvas, self.n_q)
        if not isinstance(x, torch.Tensor) or x.dim() != 1 + len(tail):
            raise D3PMError(f"{name}: expected an int32 [B, {', '.join(map(str, tail))}] token grid")
        _require(x, name, (x.shape[0],) + tail, (torch.int32,), self.device)
        if x.shape[0] < 1:
            raise D3PMError(f"{name}: empty batch")
        if frame_mask is not None:
            _require(frame_mask, "frame_mask", (cfg.canvas,), (torch.uint8,), self.device)
        return x.shape[0]

    def _check_kv(self, kv_t, kv_p, B):
        cfg = self.cfg
        _require(kv_t, "kv_text", (cfg.n_layers, B, cfg.s_text, 2 * cfg.d_model), (self.dtype,), self.device)
        _require(kv_p, "kv_prompt", (cfg.n_layers, B, cfg.s_prompt, 2 * cfg.d_model), (self.dtype,), self.device)

    def denoise(self, x_t, frame_mask, t, kv_t, kv_p, *, want_logits=True, want_hidden=False, only_layers=-1,
                flags=0, fp8=False):
        cfg = self.cfg
        B = self._check_grid(x_t, frame_mask)
        self._check_kv(kv_t, kv_p, B)
        ws = self.workspace(B)
        logits = torch.empty((B, cfg.canvas) + self._lvl() + (cfg.n_classes,), dtype=self.dtype, device=self.device) if want_logits else None
        hidden = torch.empty((B, cfg.canvas, cfg.d_model), dtype=self.dtype, device=self.device) if want_hidden else None
        if fp8:
            check(lib().d3pm_denoise_step_fp8(C.byref(self.shape), C.byref(self.weights.c_struct),
                                              C.cast(self.fp8_weights().blocks, C.c_void_p), B, _p(x_t), _p(frame_mask), int(t),
                                              _p(self.film), _p(kv_t), _p(kv_p), _p(ws), ws.numel(), _p(logits), _p(hidden),
                                              only_layers, flags, stream_ptr()), "d3pm_denoise_step_fp8")
            return logits, hidden
        check(lib().d3pm_denoise_step(C.byref(self.shape), C.byref(self.weights.c_struct), B, _p(x_t), _p(frame_mask),
                                      int(t), _p(self.film), _p(kv_t), _p(kv_p), _p(ws), ws.numel(), _p(logits),
                                      _p(hidden), only_layers, flags, stream_ptr()), "d3pm_denoise_step")
        return logits, hidden

    def _lvl(self):
        return () if self.n_q == 1 else (self.n_q,)

    def posterior_sample(self, logits, x_t, t, seed, utt0=0, flags=0, want_posterior=False):
        cfg = self.cfg
        B = self._check_grid(x_t)
        logits = logits.contiguous() if isinstance(logits, torch.Tensor) else logits
        _require(logits, "logits", (B, cfg.canvas) + self._lvl() + (cfg.n_classes,), tuple(_DTYPES), self.device)
        x_next = torch.empty_like(x_t)
        post = torch.empty((B, cfg.canvas) + self._lvl() + (cfg.n_classes,), dtype=torch.int16, device=self.device) if want_posterior else None
        check(lib().d3pm_posterior_sample(C.byref(self.shape), B, _p(logits), dtype_code(logits.dtype), _p(x_t),
                                          _p(x_next), int(t), C.byref(self.schedule.c_struct), seed, utt0, flags,
                                          _p(post), stream_ptr()), "d3pm_posterior_sample")
        return x_next, post

    def final_sample(self, hidden, x_t, t, seed, utt0=0, flags=0):
        """hidden [B, canvas, d] (masked) -> x_{t-1} int32 [B, canvas] through the fused final + sampler kernel."""
        cfg = self.cfg
        B = self._check_grid(x_t)
        _require(hidden, "hidden", (B, cfg.canvas, cfg.d_model), (self.dtype,), self.device)
        x_next = torch.empty_like(x_t)
        if not _is_ab:
            raise D3PMError("the fused final + sampler kernel is an experiment of libd3pm_hip_ab.so (use_ab_library())")
        check(lib().d3pm_op_final_sample(C.byref(self.shape), C.byref(self.weights.c_struct), B, _p(hidden), _p(x_t), _p(x_next),
                                         int(t), C.byref(self.schedule.c_struct), seed, utt0, flags, stream_ptr()),
              "d3pm_op_final_sample")
        return x_next

    def sample_loop(self, x, frame_mask, t_start, t_stop, kv_t, kv_p, seed, utt0=0, flags=0, trace=False, slot=0,
                    fp8=False):
        cfg = self.cfg
        B = self._check_grid(x, frame_mask, "x")
        self._check_kv(kv_t, kv_p, B)
        ws = self.workspace(B, slot)
        tr = torch.empty((t_start - t_stop, B, cfg.canvas) + self._lvl(), dtype=torch.int32, device=self.device) if trace else None
        if fp8:
            check(lib().d3pm_sample_loop_fp8(C.byref(self.shape), C.byref(self.weights.c_struct),
                                             C.cast(self.fp8_weights().blocks, C.c_void_p), B, _p(x), _p(frame_mask),
                                             int(t_start), int(t_stop), _p(self.film), _p(kv_t), _p(kv_p),
                                             C.byref(self.schedule.c_struct), seed, utt0, flags, _p(ws), ws.numel(), _p(tr),
                                             stream_ptr()), "d3pm_sample_loop_fp8")
            return tr
        check(lib().d3pm_sample_loop(C.byref(self.shape), C.byref(self.weights.c_struct), B, _p(x), _p(frame_mask),
                                     int(t_start), int(t_stop), _p(self.film), _p(kv_t), _p(kv_p),
                                     C.byref(self.schedule.c_struct), seed, utt0, flags, _p(ws), ws.numel(), _p(tr),
                                     stream_ptr()), "d3pm_sample_loop")
        return tr

    def ce_loss_rows(self, logits, targets, frame_mask):
        """fp32 [B, canvas] cross-entropy rows of masked logits against masked targets (d3pm_ce_loss_rows)."""
        B = self._check_grid(targets, frame_mask, "targets")
        logits = logits.contiguous() if isinstance(logits, torch.Tensor) else logits
        _require(logits, "logits", (B, self.cfg.canvas, self.cfg.n_classes), tuple(_DTYPES), self.device)
        out = torch.empty((B, self.cfg.canvas), dtype=torch.float32, device=self.device)
        check(lib().d3pm_ce_loss_rows(C.byref(self.shape), B, _p(logits), dtype_code(logits.dtype), _p(targets),
                                      _p(frame_mask), _p(out), stream_ptr()), "d3pm_ce_loss_rows")
        return out

    def training_forward(self, x0, frame_mask, kv_t, kv_p, seed, utt0=0, timesteps=None):
        """The training-side forward of AR.forward (ar_discrete.py:655-693) for utterances that share one frame mask:
        sum over t = 1 .. timesteps-1 of mean_rows CE(denoiser(q_sample(x0, t)), x0 * mask), divided by mask.sum().
        Returns (loss fp32 [B], logits of the last step [B, canvas, K] with padded frames zeroed)."""
        T = self.schedule.timesteps if timesteps is None else int(timesteps)
        fm = frame_mask.to(torch.uint8)
        live = fm.bool()
        targets = (x0 * fm.to(x0.dtype)[None, :]).to(torch.int32).contiguous()
        total = torch.zeros(x0.shape[0], dtype=torch.float32, device=self.device)
        logits = None
        for t in range(1, T):
            x_t = self.q_sample(x0, fm, t, seed, utt0)
            logits, _ = self.denoise(x_t, fm, t, kv_t, kv_p)
            total += self.ce_loss_rows(logits, targets, fm).mean(dim=1)
        if logits is not None:
            logits = logits * live[None, :, None].to(logits.dtype)
        return total / live.sum().to(torch.float32), logits

    def sample_loop_graphed(self, x, frame_mask, t_start, t_stop, kv_t, kv_p, seed, utt0=0, flags=0):
        """The same loop replayed from a captured HIP graph: one launch instead of ~50 per iteration, for the
        launch-bound regime (one or two utterances).  The graph is captured once per (batch, step range, utt0,
        flags) over static buffers; the seed lives in HBM (D3PM_FLAG_SEED_IN_HBM) so that a replay can draw new
        noise.  `x` is updated in place like sample_loop does."""
        B = x.shape[0]
        key = (B, int(t_start), int(t_stop), int(utt0), int(flags))
        ent = self._graphs.get(key)
        if ent is None:
            st = {"x": torch.empty_like(x), "mask": torch.empty_like(frame_mask), "kv_t": torch.empty_like(kv_t),
                  "kv_p": torch.empty_like(kv_p), "seed": torch.zeros(1, dtype=torch.int64, device=self.device)}
            for k, v in (("x", x), ("mask", frame_mask), ("kv_t", kv_t), ("kv_p", kv_p)):
                st[k].copy_(v)
            fl = int(flags) | FLAG_SEED_IN_HBM
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):          # eager warm-up on the capture stream: lazy module load, func attributes
                self.sample_loop(st["x"], st["mask"], t_start, max(t_start - 1, t_stop), st["kv_t"], st["kv_p"],
                                 st["seed"].data_ptr(), utt0=utt0, flags=fl, slot=("graph",) + key)
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.sample_loop(st["x"], st["mask"], t_start, t_stop, st["kv_t"], st["kv_p"], st["seed"].data_ptr(),
                                 utt0=utt0, flags=fl, slot=("graph",) + key)
            ent = self._graphs[key] = (graph, st)
        graph, st = ent
        st["x"].copy_(x)
        st["mask"].copy_(frame_mask)
        st["kv_t"].copy_(kv_t)
        st["kv_p"].copy_(kv_p)
        st["seed"].fill_(int(seed))
        graph.replay()
        x.copy_(st["x"])

    def q_sample(self, x0, frame_mask, t, seed, utt0=0):
        self._check_grid(x0, frame_mask, "x0")
        out = torch.empty_like(x0)
        check(lib().d3pm_q_sample(C.byref(self.shape), x0.shape[0], _p(x0), _p(out), _p(frame_mask), int(t),
                                  C.byref(self.schedule.c_struct), seed, utt0, stream_ptr()), "d3pm_q_sample")
        return out


FAMILY_AUTO, FAMILY_GENERIC, FAMILY_MFMA = 0, 1, 2


def op_linear(x, w, bias=None, *, act=0, r1=None, r2=None, row_mask=None, mask_period=1, family=0, out=None,
              ldy=None):
    """y = epilogue(x @ w.T + bias) through d3pm_op_linear; x [M,K], w [N,K] contiguous device tensors."""
    M, K = x.shape
    N = w.shape[0]
    ldy = N if ldy is None else ldy
    y = torch.zeros((M, ldy), dtype=x.dtype, device=x.device) if out is None else out
    check(lib().d3pm_op_linear(dtype_code(x.dtype), family, _p(x), x.stride(0), _p(w), _p(bias), _p(y), ldy, _p(r1),
                               _p(r2), 0 if r1 is None else r1.stride(0), _p(row_mask), mask_period, M, N, K, act,
                               C.byref(TUNING), stream_ptr()), "d3pm_op_linear")
    return y[:, :N]


def mx_scale_bytes(amax: torch.Tensor) -> torch.Tensor:
    """e8m0 byte of a block's absolute maximum (csrc/d3pm_mx.hip: mx_scale_byte): the smallest power of two 2^(byte - 127)
    with amax / scale <= 448, exact integer arithmetic on the fp32 bits."""
    bits = amax.float().contiguous().view(torch.int32)
    e = (bits >> 23) - 8 + ((bits & 0x7FFFFF) > 0x600000).to(torch.int32)
    return e.clamp(1, 254).to(torch.uint8)


def quantize_mx(w: torch.Tensor):
    """[N, K] (K a multiple of 128) -> (uint8 e4m3 codes [N, K], uint8 e8m0 scales [N, 4, K // 128]); blocks of 32 along K.
    The weight side of the fp8 path (host / torch arithmetic; torch's float8_e4m3fn is the OCP format gfx950 computes in); the
    device-side quantisers (d3pm_op_quantize_mx, d3pm_op_layernorm_mx, the MX epilogue) apply the same rule."""
    N, K = w.shape
    assert K % 128 == 0
    wf = w.float().reshape(N, K // 128, 4, 32)
    sb = mx_scale_bytes(wf.abs().amax(dim=-1))                       # [N, K/128, 4]
    inv = torch.exp2(127.0 - sb.float())
    codes = (wf * inv[..., None]).to(torch.float8_e4m3fn).view(torch.uint8).reshape(N, K)
    return codes.contiguous(), sb.permute(0, 2, 1).contiguous()


def dequantize_mx(codes: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """fp32 values of an MX tensor (codes [N, K], scales [N, 4, K // 128]): exact."""
    N, K = codes.shape
    sc = torch.exp2(scales.permute(0, 2, 1).float() - 127.0)        # [N, K/128, 4]
    return (codes.view(torch.float8_e4m3fn).float().reshape(N, K // 128, 4, 32) * sc[..., None]).reshape(N, K)


def op_quantize_mx(x):
    M, K = x.shape
    x8 = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    sx = torch.empty((M, 4, K // 128), dtype=torch.uint8, device=x.device)
    check(lib().d3pm_op_quantize_mx(dtype_code(x.dtype), _p(x), x.stride(0), _p(x8), _p(sx), M, K, stream_ptr()), "d3pm_op_quantize_mx")
    return x8, sx


def op_layernorm_mx(x, w, b, film=None, eps=1e-6):
    M, d = x.shape
    y8 = torch.empty((M, d), dtype=torch.uint8, device=x.device)
    sx = torch.empty((M, 4, d // 128), dtype=torch.uint8, device=x.device)
    check(lib().d3pm_op_layernorm_mx(dtype_code(x.dtype), _p(x), _p(y8), _p(sx), _p(w), _p(b), _p(film), M, d, eps,
                                     stream_ptr()), "d3pm_op_layernorm_mx")
    return y8, sx


def op_linear_mx(x8, sx, w8, sw, bias, out_dtype, *, act=0, r1=None, row_mask=None, mask_period=1, mx_out=False):
    """16-bit result [M, N], or with mx_out=True the result as (codes [M, N], scales [M, 4, N // 128])."""
    M, K = x8.shape
    N = w8.shape[0]
    if mx_out:
        y8 = torch.empty((M, N), dtype=torch.uint8, device=x8.device)
        sy = torch.empty((M, 4, N // 128), dtype=torch.uint8, device=x8.device)
        y = None
    else:
        y8 = sy = None
        y = torch.empty((M, N), dtype=out_dtype, device=x8.device)
    check(lib().d3pm_op_linear_mx(dtype_code(out_dtype), _p(x8), x8.stride(0), _p(sx), _p(w8), _p(sw), _p(bias), _p(y), N, _p(r1),
                                  0 if r1 is None else r1.stride(0), _p(row_mask), mask_period, _p(y8), _p(sy), M, N, K, act,
                                  C.byref(TUNING), stream_ptr()), "d3pm_op_linear_mx")
    return (y8, sy) if mx_out else y


def op_attention(q, k, v, n_heads, scale, *, family=0):
    """q [B,Tq,d], k/v [B,S,d] (views with arbitrary row stride allowed) -> [B,Tq,d]."""
    B, Tq, d = q.shape
    S = k.shape[1]
    assert k.stride(1) == v.stride(1) and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    assert q.stride(0) == Tq * q.stride(1) and k.stride(0) == S * k.stride(1) and v.stride(0) == S * v.stride(1)
    o = torch.empty((B, Tq, d), dtype=q.dtype, device=q.device)
    check(lib().d3pm_op_attention(dtype_code(q.dtype), family, _p(q), q.stride(1), _p(k), _p(v), k.stride(1), _p(o), d,
                                  B, Tq, S, n_heads, d // n_heads, float(scale), C.byref(TUNING), stream_ptr()),
          "d3pm_op_attention")
    return o


def op_attention_pair(q1, k1, v1, q2, k2, v2, n_heads, scale):
    """The text / prompt cross-attention pair of a block as one launch: q1, q2 [B,Tq,d]; k1 / v1 [B,S1,d], k2 / v2 [B,S2,d] (K / V
    views of one packed [.., 2d] cache row allowed) -> (o1, o2) [B,Tq,d]."""
    B, Tq, d = q1.shape
    S1, S2 = k1.shape[1], k2.shape[1]
    for t in (q1, q2, k1, v1, k2, v2):
        assert t.stride(2) == 1
    assert q1.stride(1) == q2.stride(1) and k1.stride(1) == v1.stride(1) == k2.stride(1) == v2.stride(1)
    assert q1.stride(0) == Tq * q1.stride(1) and q2.stride(0) == Tq * q2.stride(1)
    assert k1.stride(0) == S1 * k1.stride(1) and k2.stride(0) == S2 * k2.stride(1)
    o1 = torch.empty((B, Tq, d), dtype=q1.dtype, device=q1.device)
    o2 = torch.empty_like(o1)
    check(lib().d3pm_op_attention_pair(dtype_code(q1.dtype), _p(q1), _p(k1), _p(v1), _p(o1), S1, _p(q2), _p(k2), _p(v2), _p(o2), S2,
                                       q1.stride(1), k1.stride(1), d, B, Tq, n_heads, d // n_heads, float(scale), C.byref(TUNING),
                                       stream_ptr()), "d3pm_op_attention_pair")
    return o1, o2


def stats_buffer(M: int, parts: int, device) -> torch.Tensor:
    """Storage of the row moments in the library's layout [ceil(M / 16), parts, 16, 2] (include/d3pm_hip.h: d3pm_op_row_stats)."""
    return torch.zeros(((M + 15) // 16, parts, 16, 2), dtype=torch.float32, device=device)


def stats_rows(st: torch.Tensor, M: int) -> torch.Tensor:
    """The same moments as [M, parts, 2] (row-major view for tests)."""
    return st.permute(0, 2, 1, 3).reshape(-1, st.shape[1], 2)[:M]


def op_row_stats(x):
    """[M, d] 16-bit rows -> the fp32 partial (sum, sum of squares) per 32-column part, in the library's layout (stats_buffer)."""
    M, d = x.shape
    st = stats_buffer(M, d // 32, x.device)
    check(lib().d3pm_op_row_stats(dtype_code(x.dtype), _p(x), x.stride(0), M, d, _p(st), stream_ptr()), "d3pm_op_row_stats")
    return st


def op_linear_stats(x, w, bias, r1, *, r2=None, row_mask=None, mask_period=1):
    """(y, stats): y = residual epilogue of d3pm_op_linear, stats = the moments of the rows of y (layout of stats_buffer)."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    st = stats_buffer(M, N // 32, x.device)
    check(lib().d3pm_op_linear_stats(dtype_code(x.dtype), _p(x), x.stride(0), _p(w), _p(bias), _p(y), N, _p(r1), _p(r2), r1.stride(0),
                                     _p(row_mask), mask_period, M, N, K, _p(st), C.byref(TUNING), stream_ptr()), "d3pm_op_linear_stats")
    return y, st


def op_fold_weights(w, bias, gamma, beta, film=None):
    """(Wf [N, K] in w's dtype, fold_s [N] fp32, fold_b [N] fp32) of one LayerNorm-fed projection (d3pm_op_fold_weights)."""
    N, K = w.shape
    wf = torch.empty_like(w)
    fs = torch.empty(N, dtype=torch.float32, device=w.device)
    fb = torch.empty(N, dtype=torch.float32, device=w.device)
    check(lib().d3pm_op_fold_weights(dtype_code(w.dtype), _p(w), _p(bias), _p(gamma), _p(beta), _p(film), N, K, _p(wf), _p(fs), _p(fb),
                                     stream_ptr()), "d3pm_op_fold_weights")
    return wf, fs, fb


def op_linear_fold(x, wf, fold_s, fold_b, stats, *, act=0, eps=1e-6):
    """act(LN-folded projection) of the raw rows x [M, K] with their moments `stats` (d3pm_op_linear_fold)."""
    M, K = x.shape
    N = wf.shape[0]
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(lib().d3pm_op_linear_fold(dtype_code(x.dtype), _p(x), x.stride(0), _p(wf), _p(fold_s), _p(fold_b), _p(stats), eps, _p(y), N,
                                    M, N, K, act, C.byref(TUNING), stream_ptr()), "d3pm_op_linear_fold")
    return y


def op_layernorm(x, w, b, film=None, eps=1e-6):
    y = torch.empty_like(x)
    check(lib().d3pm_op_layernorm(dtype_code(x.dtype), _p(x), _p(y), _p(w), _p(b), _p(film), x.shape[0], x.shape[1],
                                  eps, stream_ptr()), "d3pm_op_layernorm")
    return y


def op_linear_lnpro(x, w, bias, ln_w, ln_b, *, ln2_w=None, ln2_b=None, film=None, act=0, eps=1e-6):
    """act(LN(x) @ w.T + bias) in one launch of the latency GEMM; x [m, 512] is the un-normalised stream.  With a second
    LayerNorm the result has 2 m rows (x under ln, then x under ln2)."""
    m = x.shape[0]
    if not (x.shape[1] == 512 and w.shape[1] == 512 and x.is_contiguous() and w.is_contiguous()):
        raise ValueError("op_linear_lnpro: x [m,512], w [N,512], contiguous")
    M, N = (2 * m if ln2_w is not None else m), w.shape[0]
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if not _is_ab:
        raise D3PMError("the LayerNorm-prologue GEMM is an experiment of libd3pm_hip_ab.so (use_ab_library())")
    check(lib().d3pm_op_linear_lnpro(dtype_code(x.dtype), _p(x), _p(w), _p(bias), _p(y), M, N, act, _p(ln_w), _p(ln_b), _p(ln2_w),
                                     _p(ln2_b), _p(film), eps, stream_ptr()), "d3pm_op_linear_lnpro")
    return y


def op_linear_rowpanel(x, w, bias, r1, ln_w, ln_b, *, x2=None, ln2_w=None, ln2_b=None, film=None, row_mask=None, eps=1e-6, mx=False):
    """Projection onto the residual stream + the LayerNorm(s) of the new rows in one launch (include/d3pm_hip.h).
    Returns (y, ln_y, ln2_y | None); with mx=True the LayerNorm rows come back in the block-scaled fp8 format:
    (y, (codes, scales), (codes2, scales2) | None)."""
    M, K = x.shape
    if not (tuple(w.shape) == (512, K) and tuple(r1.shape) == (M, 512) and x.stride(1) == 1 and w.is_contiguous() and r1.is_contiguous()):
        raise ValueError("op_linear_rowpanel: x [M,K], w [512,K], r1 [M,512]")
    if x2 is not None and (x2.shape != x.shape or x2.stride() != x.stride()):
        raise ValueError("op_linear_rowpanel: x2 must look like x")
    y = torch.empty_like(r1)
    if mx:
        ln_y = torch.empty((M, 512), dtype=torch.uint8, device=x.device)
        ln2_y = torch.empty_like(ln_y) if ln2_w is not None else None
        sx = torch.empty((M, 4, 4), dtype=torch.uint8, device=x.device)
        sx2 = torch.empty_like(sx) if ln2_w is not None else None
    else:
        ln_y = torch.empty_like(r1)
        ln2_y = torch.empty_like(r1) if ln2_w is not None else None
        sx = sx2 = None
    check(lib().d3pm_op_linear_rowpanel(dtype_code(x.dtype), _p(x), _p(x2), x.stride(0), _p(w), _p(bias), _p(y), _p(r1), _p(row_mask),
                                        0 if row_mask is None else row_mask.numel(), M, K, _p(ln_w), _p(ln_b), _p(ln_y),
                                        _p(ln2_w), _p(ln2_b), _p(ln2_y), _p(film), eps, _p(sx), _p(sx2), stream_ptr()),
          "d3pm_op_linear_rowpanel")
    if mx:
        return y, (ln_y, sx), ((ln2_y, sx2) if ln2_w is not None else None)
    return y, ln_y, ln2_y


class NarRunner:
    """Pointer tables + workspace of the stock NAR model (d3pm_nar_level)."""

    def __init__(self, cfg, tensors: dict, dtype: torch.dtype, device, pe: torch.Tensor):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        lib()
        self.shape = NarShape(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.n_tokens, cfg.n_prom_levels, cfg.n_resp_levels,
                              dtype_code(dtype), C.pointer(TUNING))
        self._keep = [pe]

        def ptr(key):
            t = tensors[key]
            if not (t.is_cuda and t.is_contiguous()):
                raise D3PMError(f"weight {key} must be a contiguous device tensor")
            self._keep.append(t)
            return t.data_ptr()

        self.blocks = (NarBlockWeights * cfg.n_layers)()
        names = {"attn_norm_emb": "attn.norm.emb.weight", "to_qkv_w": "attn.block.to_qkv.weight",
                 "to_out_w": "attn.block.to_out.weight", "to_out_b": "attn.block.to_out.bias",
                 "ffn_norm_emb": "ffn.norm.emb.weight", "ffn0_w": "ffn.block.0.weight", "ffn0_b": "ffn.block.0.bias",
                 "ffn3_w": "ffn.block.3.weight", "ffn3_b": "ffn.block.3.bias"}
        for i in range(cfg.n_layers):
            for field, key in names.items():
                setattr(self.blocks[i], field, ptr(f"blocks.{i}.{key}"))
        self.weights = NarWeights(ptr("text_emb.weight"), ptr("proms_emb.weight"), ptr("resps_emb.weight"), ptr("sep"),
                                  ptr("classifier.weight"), ptr("classifier.bias"), pe.data_ptr(), pe.shape[0], self.blocks)
        self._ws = None

    def level(self, lens, text, prom, resp, t_max, level, temperature, seed, utt0=0, flags=0, want_logits=False):
        """One quantizer level for the whole batch; `resp` [B, tr_max, n_resp_levels+1] int32 is updated in place."""
        cfg, B = self.cfg, lens.shape[0]
        need = lib().d3pm_nar_workspace_bytes(C.byref(self.shape), B, t_max)
        if need == 0:
            raise D3PMError("d3pm_nar_workspace_bytes: " + lib().d3pm_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        logits = torch.empty((B, t_max, cfg.n_tokens), dtype=self.dtype, device=self.device) if want_logits else None
        check(lib().d3pm_nar_level(C.byref(self.shape), C.byref(self.weights), B, t_max, _p(lens), _p(text), text.shape[1],
                                   _p(prom), prom.shape[1], _p(resp), resp.shape[1], int(level), float(temperature), seed,
                                   utt0, flags, _p(self._ws), self._ws.numel(), _p(logits), stream_ptr()), "d3pm_nar_level")
        return logits


def uniform(seed: int, t: int, row0: int, rows: int, n_classes: int, stream_id: int, device) -> torch.Tensor:
    out = torch.empty((rows, n_classes), dtype=torch.float32, device=device)
    check(lib().d3pm_uniform(seed, t, row0, rows, n_classes, stream_id, _p(out), stream_ptr()), "d3pm_uniform")
    return out


# ---- schedule choices: fields of this module's default Tuning (what every Sampler / NarRunner / op_* call passes) --------------
def set_gemm_variant(v: int):
    if v not in (0, 2, 3, 4, 5, 6, 7, 8):
        raise D3PMError(f"gemm_variant {v}: not a shipped schedule (include/d3pm_hip.h)")
    lib()
    TUNING.gemm_variant = v


def set_attn_query_groups(v: int):
    if v not in (0, 1, 2, 4, 32, 33) and not (v == 35 and _is_ab):      # 35: A/B library only (192-query workgroups)
        raise D3PMError(f"attn_query_groups {v}: not a shipped schedule (include/d3pm_hip.h)")
    lib()
    TUNING.attn_query_groups = v


def set_attn_pair_sequential(v):
    """0 / False never, 1 auto (default), 2 / True always."""
    lib()
    TUNING.attn_pair_sequential = 2 if v is True else int(v)


def set_lat_tile(v: int):
    lib()
    TUNING.lat_tile = int(v)


def set_row_panel(mask: int):
    lib()
    TUNING.row_panel = int(mask)


def set_attn_cross_resident(v):
    """0 / False never, 1 auto (default), 2 / True always, 3 one query block per workgroup."""
    lib()
    TUNING.attn_cross_resident = 2 if v is True else int(v)


def set_gemm_persist_slots(v: int):
    lib()
    TUNING.gemm_persist_slots = int(v)


def set_workspace_alias(v: bool):
    lib()
    TUNING.workspace_alias = 1 if v else 0


_TUNING_VALUES = {"gemm_variant": (0, 2, 3, 4, 5, 6, 7, 8), "attn_query_groups": (0, 1, 2, 4, 32, 33), "lat_tile": (0, 1, 2, 3),
                  "attn_pair_sequential": (0, 1, 2), "attn_cross_resident": (0, 1, 2, 3, 4, 5), "workspace_alias": (0, 1), "ln_fold": (0, 1)}


@contextlib.contextmanager
def tuning(**fields):
    """`with _hip.tuning(attn_query_groups=32, row_panel=0): ...` -- schedule choices for the calls made inside the block; the
    module's default Tuning is restored on exit, also when the body raises, so no test or caller can leak a schedule."""
    lib()
    for name, value in fields.items():
        if name not in TUNING_FIELDS:
            raise D3PMError(f"unknown tuning field {name!r}: one of {TUNING_FIELDS}")
        ok = _TUNING_VALUES.get(name)
        if ok is not None and int(value) not in ok and not (_is_ab and name == "attn_query_groups" and int(value) == 35):
            raise D3PMError(f"{name} = {value}: not a shipped schedule (include/d3pm_hip.h)")
    saved = {name: getattr(TUNING, name) for name in fields}
    try:
        for name, value in fields.items():
            setattr(TUNING, name, int(value))
        yield TUNING
    finally:
        for name, value in saved.items():
            setattr(TUNING, name, value)


def reset_tuning():
    """Back to the library's defaults (d3pm_tuning_default); an attached profiler stays attached."""
    prof = TUNING.prof
    lib().d3pm_tuning_default(C.byref(TUNING))
    TUNING.prof = prof



def set_tuning_field(name: str, value: int):
    """bench.py --tune name=value"""
    if name not in TUNING_FIELDS:
        raise D3PMError(f"unknown tuning field {name!r}: one of {TUNING_FIELDS}")
    lib()
    setattr(TUNING, name, int(value))


# ---- experiments (libd3pm_hip_ab.so only) ------------------------------------------------------------------------------------
def set_ln_prologue(v: bool):
    _ab_set(AB_LN_PROLOGUE, 1 if v else 0)


def set_gelu_table(v: bool):
    _ab_set(AB_GELU_TABLE, 1 if v else 0)


def set_fused_final_sample(v: bool):
    _ab_set(AB_FUSED_FINAL_SAMPLE, 1 if v else 0)


def set_gemm_big_mode(v: int):
    _ab_set(AB_GEMM_BIG_MODE, v)


def set_attn_arm(v: int):
    _ab_set(AB_ATTN_ARM, v)


def set_gemm_ring(v: bool):
    _ab_set(AB_GEMM_RING, 1 if v else 0)


def gemm_clock_ghz() -> float:
    """Shader clock (GHz) held during the last big-tile GEMM launched with big mode bit 8 set; synchronises."""
    if not _is_ab:
        raise D3PMError("d3pm_debug_gemm_clock lives in libd3pm_hip_ab.so (use_ab_library())")
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 2)()
    check(lib().d3pm_debug_gemm_clock(buf), "d3pm_debug_gemm_clock")
    return buf[0] / max(buf[1], 1) * 0.1


def attn32_stamps():
    """[2][12][8] shader-clock stamps of the last attn32 launch under set_attn_arm(320) (include/d3pm_hip_ab.h); synchronises."""
    if not _is_ab:
        raise D3PMError("d3pm_debug_attn32_stamps lives in libd3pm_hip_ab.so (use_ab_library())")
    buf = (C.c_uint64 * 192)()
    check(lib().d3pm_debug_attn32_stamps(buf, 192), "d3pm_debug_attn32_stamps")
    return [[[int(buf[(s * 12 + t) * 8 + p]) for p in range(8)] for t in range(12)] for s in range(2)]


# ---- timing hooks: a d3pm_prof handle attached to the default Tuning ------------------------------------------------------
def prof_enable(kclass: int, max_events: int):
    prof_disable()
    h = C.c_void_p()
    check(lib().d3pm_prof_create(kclass, max_events, C.byref(h)), "d3pm_prof_create")
    TUNING.prof = h.value


def prof_read():
    if not TUNING.prof:
        return 0, 0.0, 0.0, 0.0
    n, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
    check(lib().d3pm_prof_read(TUNING.prof, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "d3pm_prof_read")
    return n.value, ms.value, fl.value, by.value


def prof_read_class(kclass: int):
    """(launches, total ms, algorithmic flops, algorithmic bytes) of one kernel class; does not reset.  Zeros when no
    profiler is attached (bench.py --no-kernel-events)."""
    if not TUNING.prof:
        return 0, 0.0, 0.0, 0.0
    n, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
    check(lib().d3pm_prof_read_class(TUNING.prof, kclass, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "d3pm_prof_read_class")
    return n.value, ms.value, fl.value, by.value


def prof_disable():
    if TUNING.prof:
        lib().d3pm_prof_destroy(TUNING.prof)
        TUNING.prof = None


def profiling() -> bool:
    return bool(TUNING.prof)


import atexit  # noqa: E402

atexit.register(lambda: prof_disable() if _lib is not None else None)      # a process that exits with a profiler attached frees its events
