"""The C-ABI library builds, loads and exports every symbol include/svdq_amd.h declares; argument
validation returns error codes + messages instead of aborting.  No kernel is launched here."""

import ctypes as C
import os
import re

import pytest

from nunchaku_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "svdq_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svdq_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    lib = C.CDLL(built_lib)
    names = _declared_symbols()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/svdq_amd.h but not exported"
    assert set(names) == set(_lib.EXPORTS), "ctypes binding and header disagree"


def test_abi_version(built_lib):
    lib = _lib.load()
    assert lib.svdq_abi_version() == _lib.ABI_VERSION


_STRUCTS = {"svdq_quantize_args": _lib.QuantizeArgs, "svdq_gemm_args": _lib.GemmArgs, "svdq_attention_args": _lib.AttentionArgs,
            "svdq_residual_args": _lib.ResidualArgs, "svdq_gemv_awq_args": _lib.GemvAwqArgs}


def test_struct_layouts_match_header(built_lib, tmp_path):
    """sizeof and every field offset of the ctypes twins against what the C compiler makes of include/svdq_amd.h
    (a host program that includes the header and prints them: no hand-kept arithmetic)."""
    import subprocess

    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "svdq_amd.h")}"', "int main(void) {"]
    for cname, cls in _STRUCTS.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in _STRUCTS.items():
        assert int(got[cname]) == C.sizeof(cls), f"sizeof({cname}): header {got[cname]} != ctypes {C.sizeof(cls)}"
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"offsetof({cname}, {fname})"


def test_attention_validation(built_lib):
    lib = _lib.load()
    assert lib.svdq_attention(None, None) == 1
    a = _lib.AttentionArgs()
    a.q = a.k = a.vt = a.out = 4096
    a.L, a.H, a.head_dim, a.ldq, a.ldk, a.ldo, a.ldvt = 256, 2, 64, 768, 768, 256, 256
    assert lib.svdq_attention(C.byref(a), None) == 2 and b"head_dim" in lib.svdq_last_error()
    a.head_dim, a.L = 128, 200
    assert lib.svdq_attention(C.byref(a), None) == 1 and b"multiple of 128" in lib.svdq_last_error()
    a.L, a.ldvt = 256, 128
    assert lib.svdq_attention(C.byref(a), None) == 1 and b"ldvt" in lib.svdq_last_error()


def test_validation_errors_are_returned_not_aborted(built_lib):
    lib = _lib.load()
    assert lib.svdq_quantize_w4a4_act_fuse_lora(None, None) == 1
    assert b"NULL" in lib.svdq_last_error()
    a = _lib.QuantizeArgs()
    a.x, a.act, a.ascales = 16, 16, 16
    a.M, a.M_pad, a.K, a.ldx = 10, 100, 128, 128  # M_pad not a multiple of 256
    assert lib.svdq_quantize_w4a4_act_fuse_lora(C.byref(a), None) == 1
    assert b"M_pad" in lib.svdq_last_error()
    a.M_pad, a.fp4 = 256, 1
    assert lib.svdq_quantize_w4a4_act_fuse_lora(C.byref(a), None) == 2  # unsupported
    # fuse_glu reads rows of 2K (value, gate) pairs and does not combine with the LayerNorm front end
    a.fp4, a.fuse_glu = 0, 1
    assert lib.svdq_quantize_w4a4_act_fuse_lora(C.byref(a), None) == 1  # ldx = 128 < 2K
    assert b"fuse_glu" in lib.svdq_last_error() and b"2K" in lib.svdq_last_error()
    a.ldx, a.ln_stats, a.mod_scale, a.mod_shift = 256, 16, 16, 16
    assert lib.svdq_quantize_w4a4_act_fuse_lora(C.byref(a), None) == 1
    assert b"fuse_glu" in lib.svdq_last_error()
    g = _lib.GemmArgs()
    g.act, g.wgt, g.ascales, g.wscales = 16, 16, 16, 16
    g.M, g.M_pad, g.N, g.K = 256, 256, 100, 128
    assert lib.svdq_gemm_w4a4(C.byref(g), None) == 1
    assert b"N=100" in lib.svdq_last_error()
    g.N, g.fuse = 128, 9
    assert lib.svdq_gemm_w4a4(C.byref(g), None) == 1
    assert lib.svdq_repack_qweight(16, 32, 100, 128, None) == 1
    assert lib.svdq_repack_lowrank(16, 16, 128, 32, 0, None) == 1  # aliasing


def test_python_wrappers_raise_without_gpu(built_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nunchaku_amd.models.linear import SVDQW4A4Linear

    m = SVDQW4A4Linear(128, 128)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 128, dtype=torch.bfloat16))
    with pytest.raises(NotImplementedError):
        SVDQW4A4Linear(128, 128, precision="nvfp4")


def test_ctypes_field_order_matches_header():
    """Every struct of include/svdq_amd.h against its ctypes twin: same field NAMES in the same ORDER (sizes are checked
    above) -- a transposed pair of pointers would otherwise go unnoticed until a kernel reads the wrong tensor."""
    import re

    src = open(os.path.join(ROOT, "include", "svdq_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # drop comments
    twins = {"svdq_quantize_args": _lib.QuantizeArgs, "svdq_gemm_args": _lib.GemmArgs, "svdq_attention_args": _lib.AttentionArgs,
             "svdq_gemv_awq_args": _lib.GemvAwqArgs, "svdq_residual_args": _lib.ResidualArgs}
    for name, cls in twins.items():
        body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", src, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            # "const void *a, *b" / "int32_t M, N" / "float *stats": names are the identifiers before ',' or the end
            for part in decl.split(","):
                fields.append(re.findall(r"[A-Za-z_][A-Za-z_0-9]*", part)[-1])
        assert fields == [f[0] for f in cls._fields_], f"{name}: header {fields} != ctypes {[f[0] for f in cls._fields_]}"
