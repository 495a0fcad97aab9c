"""ctypes view of the C ABI declared in include/svdq_amd.h (struct layouts must match it)."""

import ctypes as C
import os

# torch must initialise ITS bundled HIP runtime first: libsvdq_amd.so then binds to the already
# loaded libamdhip64 (same soname) instead of pulling a second runtime from /opt/rocm, which on a
# GPU box ends in "no ROCm-capable device is detected" for our launches.
import torch  # noqa: F401

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsvdq_amd.so")

SVDQ_BF16, SVDQ_FP16 = 0, 1
FUSE_NONE, FUSE_SILU, FUSE_GELU_QUANT, FUSE_RMSNORM_ROPE = 0, 1, 2, 3
ABI_VERSION = 22
LORA_ACT_F32, LORA_ACT_Q32, LORA_ACT_Q32_RUNS = 0, 1, 2


class QuantizeArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("smooth", C.c_void_p), ("lora_down", C.c_void_p),
        ("act", C.c_void_p), ("ascales", C.c_void_p), ("lora_act", C.c_void_p),
        ("M", C.c_int32), ("M_pad", C.c_int32), ("K", C.c_int32), ("R", C.c_int32),
        ("ldx", C.c_int32), ("dtype", C.c_int32), ("fuse_glu", C.c_int32), ("fp4", C.c_int32),
        ("ln_stats", C.c_void_p), ("mod_scale", C.c_void_p), ("mod_shift", C.c_void_p),
        ("x2", C.c_void_p), ("smooth2", C.c_void_p), ("lora_down2", C.c_void_p), ("mod_scale2", C.c_void_p),
        ("mod_shift2", C.c_void_p), ("ln_stats2", C.c_void_p), ("M2", C.c_int32), ("ldx2", C.c_int32),
        ("split_rows", C.c_int32), ("lora_act_zeroed", C.c_int32), ("lora_act_format", C.c_int32), ("reserved", C.c_int32),
    ]


class ResidualArgs(C.Structure):
    _fields_ = [
        ("res", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p), ("gate", C.c_void_p), ("out", C.c_void_p),
        ("stats", C.c_void_p), ("M", C.c_int32), ("C", C.c_int32), ("ld", C.c_int32), ("dtype", C.c_int32),
        ("eps", C.c_float), ("clamp_fp16", C.c_int32), ("zero_ptr", C.c_void_p), ("zero_bytes", C.c_int64),
        ("res2", C.c_void_p), ("a2", C.c_void_p), ("b2", C.c_void_p), ("gate2", C.c_void_p), ("out2", C.c_void_p),
        ("stats2", C.c_void_p), ("M2", C.c_int32), ("reserved2", C.c_int32),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("act", C.c_void_p), ("wgt", C.c_void_p), ("ascales", C.c_void_p), ("wscales", C.c_void_p),
        ("bias", C.c_void_p), ("lora_act_in", C.c_void_p), ("lora_up", C.c_void_p),
        ("lora_scales", C.POINTER(C.c_float)), ("out", C.c_void_p),
        ("qout", C.c_void_p), ("oscales", C.c_void_p), ("next_smooth", C.c_void_p),
        ("next_lora_down", C.c_void_p), ("lora_act_out", C.c_void_p),
        ("norm_q", C.c_void_p), ("norm_k", C.c_void_p), ("rotary_emb", C.c_void_p),
        ("M", C.c_int32), ("M_pad", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("R", C.c_int32), ("R2", C.c_int32), ("ldo", C.c_int32), ("dtype", C.c_int32),
        ("act_unsigned", C.c_int32), ("fuse", C.c_int32), ("variant", C.c_int32), ("reserved", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("out_vt", C.c_void_p), ("ldvt", C.c_int32), ("geometry", C.c_int32),
        ("wgt2", C.c_void_p), ("wscales2", C.c_void_p), ("bias2", C.c_void_p), ("lora_up2", C.c_void_p),
        ("next_smooth2", C.c_void_p), ("next_lora_down2", C.c_void_p), ("norm_q2", C.c_void_p), ("norm_k2", C.c_void_p),
        ("split_rows", C.c_int32), ("lora_act_format", C.c_int32), ("status", C.c_void_p),
        ("q_scale", C.c_float), ("reserved2", C.c_int32),
        ("next_lora_down_packed", C.c_void_p), ("next_lora_down_packed2", C.c_void_p), ("lora_up_packed", C.c_void_p),
    ]


class AttentionArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("out", C.c_void_p),
        ("q_hs", C.c_int64), ("k_hs", C.c_int64), ("vt_hs", C.c_int64), ("o_hs", C.c_int64),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldvt", C.c_int32), ("ldo", C.c_int32),
        ("L", C.c_int32), ("H", C.c_int32), ("head_dim", C.c_int32), ("dtype", C.c_int32),
        ("scale", C.c_float), ("reserved", C.c_int32), ("zero_ptr", C.c_void_p), ("zero_bytes", C.c_int64),
        ("qact", C.c_void_p), ("qscales", C.c_void_p), ("qlora_act", C.c_void_p), ("qsmooth", C.c_void_p),
        ("qlora_down", C.c_void_p), ("qsmooth2", C.c_void_p), ("qlora_down2", C.c_void_p), ("qR", C.c_int32),
        ("qsplit_rows", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("qlora_act_format", C.c_int32), ("q_prescaled", C.c_int32), ("status", C.c_void_p),
        ("kv_len0", C.c_int32), ("kv_start1", C.c_int32), ("kv_end1", C.c_int32), ("geometry", C.c_int32),
        ("qlora_down_packed", C.c_void_p), ("qlora_down_packed2", C.c_void_p),
    ]


class GemvAwqArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("qweight", C.c_void_p), ("scales", C.c_void_p), ("zeros", C.c_void_p),
        ("bias", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ldx", C.c_int32),
        ("group_size", C.c_int32), ("dtype", C.c_int32), ("out_chunks", C.c_int32), ("reserved", C.c_int32),
    ]


EXPORTS = {
    "svdq_quantize_w4a4_act_fuse_lora": (C.c_int, [C.POINTER(QuantizeArgs), C.c_void_p]),
    "svdq_gemm_w4a4": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "svdq_attention": (C.c_int, [C.POINTER(AttentionArgs), C.c_void_p]),
    "svdq_gemv_awq": (C.c_int, [C.POINTER(GemvAwqArgs), C.c_void_p]),
    "svdq_gemv_awq_batched": (C.c_int, [C.POINTER(GemvAwqArgs), C.c_int32, C.c_void_p]),
    "svdq_residual_gate_stats": (C.c_int, [C.POINTER(ResidualArgs), C.c_void_p]),
    "svdq_gemm_workspace_bytes": (C.c_int64, []),
    "svdq_gemm_workspace_bytes_for": (C.c_int64, [C.POINTER(GemmArgs)]),
    "svdq_gemm_workspace_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "svdq_gemm_last_plan": (C.c_int, [C.POINTER(C.c_int32)]),
    "svdq_attention_workspace_bytes": (C.c_int64, []),
    "svdq_attention_workspace_bytes_for": (C.c_int64, [C.POINTER(AttentionArgs)]),
    "svdq_attention_workspace_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "svdq_attention_schedule": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "svdq_attention_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "svdq_attention_last_plan": (C.c_int, [C.POINTER(C.c_int32)]),
    "svdq_gemm_schedule": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "svdq_gemm_schedule_ex": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "svdq_repack_qweight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_repack_wscales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_repack_vec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "svdq_repack_lowrank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_unrepack_qweight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_unrepack_wscales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_unrepack_vec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "svdq_unrepack_lowrank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_unpack_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_unpack_scales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_pack_lora_down_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "svdq_pack_lora_down": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_pack_lora_up_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "svdq_pack_lora_up": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "svdq_prof_enable": (C.c_int, [C.c_int32]),
    "svdq_prof_reset": (C.c_int, []),
    "svdq_prof_select": (C.c_int, [C.c_uint32]),
    "svdq_prof_read": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "svdq_last_error": (C.c_char_p, []),
    "svdq_abi_version": (C.c_int, []),
}

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libsvdq_amd.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} is missing: build it with `python -m nunchaku_amd.build` "
            "(or __graft_entry__.build()).  nunchaku_amd has no CPU/PyTorch fallback for its kernels."
        )
    lib = C.CDLL(_LIB_PATH)
    for name, (res, argt) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, argt
    if lib.svdq_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libsvdq_amd.so ABI {lib.svdq_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().svdq_last_error().decode("utf-8", "replace")
        if rc == 1:
            raise ValueError(f"{what}: {msg}")
        if rc == 2:
            raise NotImplementedError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: {msg}")
