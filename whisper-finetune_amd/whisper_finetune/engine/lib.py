"""ctypes binding of libwft.so (the C ABI declared in include/wft.h).

Only plain pointers and sizes cross this boundary: tensors are passed as
``tensor.data_ptr()`` and the stream as ``torch.cuda.current_stream().cuda_stream``.
There is NO fallback: if the shared library is missing or a call fails, an exception
is raised (the product path never routes through the CPU oracle).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parents[2]  # whisper-finetune_amd/
LIB_PATH = Path(os.environ.get("WFT_LIB", _PKG_ROOT / "libwft.so"))

c_i64 = C.c_int64
c_vp = C.c_void_p


class GemmArgs(C.Structure):
    """Mirror of wft_gemm_args (include/wft.h)."""

    _fields_ = [
        ("A", c_vp), ("lda", c_i64), ("strideA", c_i64),
        ("B", c_vp), ("ldb", c_i64), ("strideB", c_i64),
        ("C", c_vp), ("ldc", c_i64), ("strideC", c_i64), ("c_is_f32", C.c_int), ("accumulate", C.c_int),
        ("bias", c_vp),
        ("residual", c_vp), ("ldr", c_i64), ("strideR", c_i64),
        ("aux", c_vp), ("ldaux", c_i64), ("strideAux", c_i64),
        ("epilogue", C.c_int), ("alpha", C.c_float),
        ("M", c_i64), ("N", c_i64), ("K", c_i64), ("batch", C.c_int),
        ("valid_rows_period", C.c_int), ("valid_rows", C.c_int),
        ("residual_first", C.c_int),
        ("workspace", c_vp), ("workspace_bytes", c_i64),
        ("beta", C.c_float), ("p_valid", C.c_int),
        ("colsum", c_vp),
        ("tn_col_scale", c_vp), ("tn_scale_rows", C.c_int), ("tn_block_n", C.c_int), ("tn_block_r", C.c_int),
        ("tn_seg_count", C.c_int), ("tn_seg_end", C.c_int * 4), ("tn_seg_ptr", c_vp * 4),
        ("launch_mode", C.c_int), ("variant", C.c_int),
    ]


class AttnArgs(C.Structure):
    """Mirror of wft_attn_args (include/wft.h)."""

    _fields_ = [
        ("q", c_vp), ("ldq", c_i64), ("q_bs", c_i64),
        ("k", c_vp), ("ldk", c_i64), ("k_bs", c_i64),
        ("v", c_vp), ("ldv", c_i64), ("v_bs", c_i64),
        ("o", c_vp), ("ldo", c_i64), ("o_bs", c_i64),
        ("lse", c_vp),
        ("B", C.c_int), ("H", C.c_int), ("Tq", C.c_int), ("Tk", C.c_int), ("causal", C.c_int), ("scale", C.c_float),
        ("d_o", c_vp), ("lddo", c_i64), ("do_bs", c_i64),
        ("delta", c_vp),
        ("dq", c_vp), ("lddq", c_i64), ("dq_bs", c_i64),
        ("dk", c_vp), ("lddk", c_i64), ("dk_bs", c_i64),
        ("dv", c_vp), ("lddv", c_i64), ("dv_bs", c_i64),
        ("dq_colsum", c_vp), ("dv_colsum", c_vp), ("colsum_ws", c_vp),
        ("launch_mode", C.c_int), ("variant", C.c_int), ("q_prescaled", C.c_int),
    ]


class GemmF32Args(C.Structure):
    """Mirror of wft_gemm_f32_args (include/wft.h): the fp32 compute mode's strided GEMM."""

    _fields_ = [
        ("A", c_vp), ("a_rs", c_i64), ("a_cs", c_i64), ("a_bs", c_i64),
        ("B", c_vp), ("b_rs", c_i64), ("b_cs", c_i64), ("b_bs", c_i64),
        ("C", c_vp), ("ldc", c_i64), ("c_bs", c_i64),
        ("bias", c_vp),
        ("M", c_i64), ("N", c_i64), ("K", c_i64), ("batch", C.c_int),
        ("alpha", C.c_float), ("beta", C.c_float),
    ]


EPI_NONE, EPI_GELU, EPI_DGELU, EPI_GELU_GRAD, EPI_MUL_AUX, EPI_GELU_GRAD8, EPI_MUL_AUX8 = 0, 1, 2, 3, 4, 5, 6

# name -> argtypes (restype is int unless listed in _RESTYPES); this table is also what
# tests/test_abi.py checks against the declarations in include/wft.h.
SIGNATURES = {
    "wft_cast_f32_bf16": [c_vp, c_vp, c_i64, c_vp],
    "wft_cast_bf16_f32": [c_vp, c_vp, c_i64, c_vp],
    "wft_cast_pad_transpose_f32_bf16": [c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, C.c_float, c_vp],
    "wft_lora_merge": [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, C.c_int, C.c_float, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, C.c_float, c_vp],
    "wft_lora_refresh_mt": [c_vp, c_vp, C.c_int, C.c_int, c_vp],
    "wft_lora_pack": [c_vp, c_vp, c_vp, C.c_int, c_i64, c_i64, C.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp],
    "wft_add_bf16": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "wft_axpby_bf16": [C.c_float, c_vp, C.c_float, c_vp, c_vp, c_i64, c_vp],
    "wft_dgelu_mul_bf16": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "wft_colsum_bf16": [c_vp, c_i64, c_i64, c_i64, c_vp, C.c_int, c_vp],
    "wft_mt_copy_f32": [c_vp, C.c_int, c_vp],
    "wft_colsum_bf16_ws": [c_vp, c_i64, c_i64, c_i64, c_vp, C.c_int, c_vp, c_i64, c_vp],
    "wft_colsum_workspace_bytes": [c_i64, c_i64],
    "wft_layernorm_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_float,
                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp],
    "wft_layernorm_bwd_workspace": [c_i64, C.c_int],
    "wft_layernorm_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int,
                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp],
    "wft_gemm_nt_bf16": [C.POINTER(GemmArgs), c_vp],
    "wft_gemm_nt_variant": [C.POINTER(GemmArgs)],
    "wft_gemm_tn_segments_ok": [C.POINTER(GemmArgs)],
    "wft_gemm_nt_colsum_workspace_bytes": [C.POINTER(GemmArgs)],
    "wft_gemm_nt_splitk_workspace_bytes": [C.POINTER(GemmArgs)],
    "wft_gemm_nt_aux8_bytes": [C.POINTER(GemmArgs)],
    "wft_gemm_tn_bf16": [C.POINTER(GemmArgs), c_vp],
    "wft_gemm_nt_rank_pair_bf16": [C.POINTER(GemmArgs), C.POINTER(GemmArgs), c_vp],
    "wft_gemm_tn_rank_pair_bf16": [C.POINTER(GemmArgs), C.POINTER(GemmArgs), c_vp],
    "wft_gemm_tn_workspace_bytes": [C.POINTER(GemmArgs)],
    "wft_attn_fwd_bf16": [C.POINTER(AttnArgs), c_vp],
    "wft_attn_bwd_bf16": [C.POINTER(AttnArgs), c_vp],
    "wft_attn_bwd_colsum_workspace_bytes": [C.POINTER(AttnArgs)],
    "wft_attn_variant": [C.POINTER(AttnArgs), C.c_int],
    "wft_embed_fwd": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, C.c_int, c_i64, c_vp],
    "wft_embed_bwd": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, C.c_int, c_i64, c_vp],
    "wft_ce_fwd": [c_vp, c_i64, c_vp, c_i64, c_i64, C.c_float, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wft_ce_bwd": [c_vp, c_i64, c_vp, c_i64, c_i64, C.c_float, c_vp, c_vp, c_vp, c_vp, c_vp],
    "wft_token_stats": [c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp],
    "wft_logmel": [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_vp],
    "wft_specaug": [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp],
    "wft_mel_to_tmajor_bf16": [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_vp],
    "wft_adamw_step": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_float, C.c_float, C.c_float, C.c_float,
                       C.c_float, C.c_float, C.c_float, c_vp, c_vp],
    "wft_sumsq_f32": [c_vp, c_i64, c_vp, c_vp],
    "wft_mt_sumsq_f32": [c_vp, c_vp, c_vp, C.c_int, C.c_int, c_vp, c_vp, c_vp],
    "wft_mt_adamw": [c_vp, c_vp, c_vp, C.c_int, C.c_int] + [C.c_float] * 7 + [c_vp, C.c_float, c_vp],
    "wft_mt_adamw8": [c_vp, c_vp, c_vp, C.c_int, C.c_int, c_vp, c_vp] + [C.c_float] * 7 + [c_vp, C.c_float, c_vp],
    "wft_muon_momentum_mt": [c_vp, C.c_int, c_i64, C.c_float, C.c_int, c_vp, c_vp, c_vp, C.c_float, c_vp],
    "wft_muon_prepare": [c_vp, C.c_int, C.c_int, c_vp, C.c_int, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp],
    "wft_transpose_bf16": [c_vp, C.c_int, C.c_int, c_vp, C.c_int, c_vp],
    "wft_muon_apply_mt": [c_vp, C.c_int, C.c_int, C.c_int, c_vp, c_i64, c_i64, C.c_float, C.c_float, C.c_float, c_vp],
    "wft_gemm_f32": [C.POINTER(GemmF32Args), c_vp],
    "wft_softmax_fwd_f32": [c_vp, c_i64, c_i64, c_i64, C.c_float, C.c_int, c_i64, c_vp],
    "wft_softmax_bwd_f32": [c_vp, c_vp, c_i64, c_i64, c_i64, C.c_float, c_vp],
    "wft_layernorm_fwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_float, c_vp, c_vp],
    "wft_layernorm_bwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.c_int, c_vp, c_vp],
    "wft_gelu_fwd_f32": [c_vp, c_vp, c_i64, c_vp],
    "wft_gelu_bwd_f32": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "wft_axpby_f32": [C.c_float, c_vp, C.c_float, c_vp, c_vp, c_i64, c_vp],
    "wft_colsum_f32": [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp],
    "wft_embed_fwd_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, C.c_int, c_vp],
    "wft_embed_bwd_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, C.c_int, c_vp],
    "wft_ce_fwd_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, C.c_float, c_vp, c_vp, c_vp, c_vp],
    "wft_ce_bwd_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, C.c_float, c_vp, c_vp, c_vp, c_vp],
    "wft_last_error": [],
    "wft_version": [],
}
_RESTYPES = {"wft_last_error": C.c_char_p, "wft_version": C.c_char_p, "wft_layernorm_bwd_workspace": c_i64,
             "wft_gemm_tn_workspace_bytes": c_i64, "wft_gemm_nt_colsum_workspace_bytes": c_i64, "wft_gemm_nt_splitk_workspace_bytes": c_i64, "wft_gemm_nt_aux8_bytes": c_i64, "wft_colsum_workspace_bytes": c_i64,
             "wft_attn_bwd_colsum_workspace_bytes": c_i64}

_lib = None


class WftError(RuntimeError):
    pass


def load():
    """Load libwft.so (once). Raises WftError if it is missing — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise WftError(
            f"HIP extension {LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C whisper-finetune_amd/csrc). There is no CPU fallback for the product path."
        )
    # Launch modes of a multi-GPU job: persistent grids by default; model_utils.train_step asks for one workgroup per tile / item
    # (wft_gemm_args / wft_attn_args launch_mode = 1, a per-call field) in the backward pass that runs beside the gradient all-reduce
    # (runtime.exchange_launch_mode; measured in bench.py's ddp_mode_1gpu block).  The library holds no mutable launch state;
    # WFT_NT256_PERSISTENT=0 / WFT_ATTN_PERSISTENT=0 (read ONCE when the library is loaded) force per-tile launches everywhere.
    lib = C.CDLL(str(LIB_PATH))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def last_error() -> str:
    return load().wft_last_error().decode()


def check(status: int, what: str):
    if status != 0:
        raise WftError(f"{what} failed with status {status}: {last_error()}")


_RAW_STREAM = None


def stream_ptr():
    """torch's CURRENT stream on the current device as a void* (every libwft call is enqueued there).  Through the raw
    accessors Inductor uses: `torch.cuda.current_stream().cuda_stream` builds a Stream object per call (9 us — 2 ms of a
    19 ms whisper-base step at 230 launches)."""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch

        raw, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if raw is not None and dev is not None:
            torch.cuda.init()
            _RAW_STREAM = lambda: raw(dev())
        else:
            _RAW_STREAM = lambda: torch.cuda.current_stream().cuda_stream
    return C.c_void_p(_RAW_STREAM())
