"""SVDQuant W4A4 linear layer for MI355X (reference: nunchaku/models/linear.py:13-274).

Same constructor, parameter names/shapes/dtypes and methods as the reference's ``SVDQW4A4Linear``
so reference checkpoints ``load_state_dict`` unchanged.  Checkpoint tensors arrive in NVIDIA mma
fragment order; ``repack_()`` converts them ONCE (same ``nn.Parameter`` objects) into the CDNA4
operand images the HIP kernels read: ``qweight`` becomes the [out, 3*in/4]-byte FP6 (e2m3) register
image of v_mfma_scale_f32_32x32x64_f8f6f4 (a 4-bit code is exactly an FP6 value), the other tensors
are permuted in place.  It runs lazily before the first kernel call and again after every
``load_state_dict``, PER TENSOR: only the tensors a (possibly partial) state dict actually brought are
converted again.  ``state_dict()`` of a repacked layer returns CHECKPOINT-layout tensors (the exact inverse
permutations, ``svdq_unrepack_*``): load -> forward -> ``state_dict()`` -> load into a fresh module reproduces the
layer bit for bit.  ``_amd_names`` (which parameters hold the kernel layout) is the single source of truth; the
per-tensor marks ``nunchaku_amd._C`` reads are re-stamped from it before every launch, so ``copy.deepcopy`` of a
repacked layer (which drops tensor attributes) stays correct.
"""

from __future__ import annotations

import math

import torch
from torch import nn

from .. import layout
from ..ops.gemm import svdq_gemm_w4a4_cuda
from ..ops.quantize import svdq_quantize_w4a4_act_fuse_lora_cuda


def synthetic_codes_(qweight: torch.Tensor, wscales: torch.Tensor, in_features: int, generator: torch.Generator, codes: str = "uniform"):
    """Random 4-bit weight codes + group scales in the checkpoint layout (no checkpoints in this environment), in place.

    ``codes="uniform"``: every nibble uniform in [-8, 7] (std 4.6).  ``codes="residual"``: the distribution of an SVDQuant residual as SURVEY.md 8(d)
    prescribes synthetic layers -- a Gaussian weight quantised symmetrically per group of 64 with scale = amax / 7: code = rne(7 x / amax), and amax
    of 64 standard normal samples is 2.41 on average, so the codes are rne(N(0, 2.9^2)) clamped to +-7.  Both fill the packed tensor nibble by nibble
    (the checkpoint's permutation does not change an i.i.d. distribution).  The scales make the dequantised rows ~ 1 / sqrt(K) either way."""
    dev = qweight.device
    if codes == "uniform":
        qweight.copy_(torch.randint(-128, 128, qweight.shape, generator=generator, device=dev, dtype=torch.int16))
        std = 4.6
    elif codes == "residual":
        std = 7.0 / 2.41
        lo = torch.randn(qweight.shape, generator=generator, device=dev).mul_(std).round_().clamp_(-7, 7).to(torch.int16)
        hi = torch.randn(qweight.shape, generator=generator, device=dev).mul_(std).round_().clamp_(-7, 7).to(torch.int16)
        qweight.copy_((((lo & 15) | ((hi & 15) << 4)) ^ 128) - 128)  # two's-complement nibbles, low nibble first, as a signed byte
    else:
        raise ValueError(f"codes must be 'uniform' or 'residual', got {codes!r}")
    wscales.copy_((torch.rand(wscales.shape, generator=generator, device=dev) * 0.5 + 0.75) * (1.0 / (std * math.sqrt(in_features))))


class SVDQW4A4Linear(nn.Module):
    """y = W4A4(x) + (x @ proj_down) @ proj_up^T + bias, SVDQuant INT4 (group size 64).

    Parameters: ``qweight`` int8 [out, in/2]; ``wscales`` [in/64, out]; ``bias`` [out] or None;
    ``smooth_factor`` / ``smooth_factor_orig`` [in]; ``proj_down`` [in, rank]; ``proj_up`` [out, rank].
    ``precision`` must be "int4" on MI355X ("nvfp4" needs Blackwell's block-scaled mma).
    """

    def __init__(
        self,
        in_features: int,
        out_features: int,
        rank: int = 32,
        bias: bool = True,
        precision: str = "int4",
        act_unsigned: bool = False,
        torch_dtype: torch.dtype = torch.bfloat16,
        device: str | torch.device | None = None,
    ):
        super().__init__()
        if precision == "nvfp4":
            raise NotImplementedError("nvfp4 checkpoints are Blackwell-only; use the int4 checkpoint on MI355X")
        if precision != "int4":
            raise ValueError(f"Invalid precision: {precision}")
        if device is None:
            device = torch.device("cpu")
        self.in_features = in_features
        self.out_features = out_features
        self.rank = rank
        self.precision = precision
        self.torch_dtype = torch_dtype
        self.group_size = 64

        def p(*shape, dtype=torch_dtype, grad=False):
            return nn.Parameter(torch.empty(*shape, dtype=dtype, device=device), requires_grad=grad)

        self.qweight = p(out_features, in_features // 2, dtype=torch.int8)
        self.bias = p(out_features, grad=True) if bias else None
        self.wscales = p(in_features // self.group_size, out_features)
        self.smooth_factor = p(in_features)
        self.smooth_factor_orig = p(in_features)
        self.proj_down = p(in_features, rank, grad=True)
        self.proj_up = p(out_features, rank, grad=True)
        self.wtscale = None  # nvfp4 only
        self.wcscales = None  # nvfp4 only
        self.act_unsigned = act_unsigned
        # runtime LoRA (set_lora): one scale per 16 ranks, applied to lora_act in the GEMM epilogue
        # (reference: GEMM_W4A4::lora_scales, src/Linear.h:98, Linear.cpp:131; ops/gemm.py:125-127)
        self.lora_scales: list[float] | None = None
        self._base_lowrank = None
        self._offloaded = False  # set by CPUOffloadManager while the owning block lives in host memory (set_lora then raises)

        # names of the parameters that currently hold the MI355X kernel layout (empty: everything is in the reference /
        # checkpoint layout).  Tracked per tensor: a partial load_state_dict (strict=False, only the 16-bit tensors, ...)
        # brings SOME tensors back in checkpoint layout and must not touch the others.
        self._amd_names: set[str] = set()
        self._incoming: tuple[str, ...] = ()
        self._register_load_state_dict_pre_hook(self._before_load, with_module=True)
        self.register_load_state_dict_post_hook(self._after_load)
        self._register_state_dict_hook(self._export_checkpoint_layout)

    # ------------------------------------------------------------------ layout
    _LAYOUT_PARAMS = ("qweight", "wscales", "smooth_factor", "bias", "proj_down", "proj_up")

    def _layout_params(self):
        return [n for n in self._LAYOUT_PARAMS if getattr(self, n, None) is not None and not (n.startswith("proj_") and self.rank == 0)]

    @property
    def _amd_layout(self) -> bool:
        """True when every parameter is in the kernel layout (nothing left for repack_() to do)."""
        return all(n in self._amd_names for n in self._layout_params())

    @_amd_layout.setter
    def _amd_layout(self, value: bool):
        self._amd_names = set(self._layout_params()) if value else set()
        for n in self._layout_params():
            getattr(self, n)._svdq_amd = bool(value)

    def _set_amd_names(self, names):
        """Adopt a layout state decided elsewhere (replica broadcast): which parameters hold the kernel layout."""
        self._amd_names = set(names)
        for n in self._layout_params():
            getattr(self, n)._svdq_amd = n in self._amd_names

    @staticmethod
    def _before_load(module, state_dict, prefix, local_metadata, strict, missing, unexpected, errors):
        # only the tensors actually present under `prefix` come back in checkpoint layout
        incoming = tuple(n for n in module._LAYOUT_PARAMS if prefix + n in state_dict)
        module._incoming = incoming
        if "proj_down" in incoming or "proj_up" in incoming:
            module.reset_lora()  # a checkpoint brings the base rank again: drop a runtime LoRA first
        if "qweight" in incoming:
            qw = module.qweight
            if qw.shape[-1] != module.in_features // 2:  # undo the FP6 image shape so that the [out, in/2] int8 tensor copies in
                qw.data = torch.empty(module.out_features, module.in_features // 2, dtype=torch.int8, device=qw.device)
                module._amd_names.discard("qweight")

    @staticmethod
    def _after_load(module, incompatible_keys):
        module._amd_names.difference_update(module._incoming)
        for n in module._incoming:
            getattr(module, n)._svdq_amd = False
        module._incoming = ()

    @staticmethod
    def _export_checkpoint_layout(module, state_dict, prefix, local_metadata):
        """state_dict hook: entries of parameters that hold the kernel layout are replaced by their checkpoint-layout form
        (new tensors on the same device; the module's own parameters are untouched).  A runtime LoRA (``set_lora``) is not
        part of the checkpoint: the base low-rank factors are exported."""
        if not module._amd_names:
            return
        with torch.no_grad():
            for n in module._layout_params():
                key = prefix + n
                if n not in module._amd_names or key not in state_dict:
                    continue
                t = getattr(module, n).data
                if n.startswith("proj_") and module._base_lowrank is not None:
                    t = module._base_lowrank[0 if n == "proj_down" else 1]
                dev = t.device
                if not t.is_cuda:  # a host-resident copy in the kernel layout (offloaded block): convert on the GPU, hand back on the host
                    if not torch.cuda.is_available():
                        raise RuntimeError(f"{key}: holds the MI355X operand layout; converting it back to the checkpoint layout "
                                           "needs the GPU (svdq_unrepack_*: there is no CPU path)")
                    t = t.cuda()
                if n == "qweight":
                    conv = layout.unrepack_qweight(t)
                elif n == "wscales":
                    conv = layout.unrepack_wscales(t)
                elif n in ("smooth_factor", "bias"):
                    conv = layout.unrepack_vec(t)
                else:
                    conv = layout.unrepack_lowrank(t, down=(n == "proj_down"))
                state_dict[key] = conv.to(dev)

    @torch.no_grad()
    def repack_(self, skip: tuple[str, ...] = ()) -> "SVDQW4A4Linear":
        """Permute the checkpoint-layout parameters into the kernel layout, in place (idempotent, per tensor).  ``skip``: names
        left as they are (the host-offload manager keeps ``qweight`` in nibble form on the host)."""
        if self._amd_layout:
            return self
        if not self.qweight.is_cuda:
            raise RuntimeError("SVDQW4A4Linear.repack_(): move the layer to the GPU first (no CPU path)")
        todo = [n for n in self._layout_params() if n not in self._amd_names and n not in skip]
        for n in self._amd_names:  # marks may be stale after a deepcopy: _param() must not convert these a second time
            getattr(self, n)._svdq_amd = True
        for n in todo:
            t = getattr(self, n)
            if n == "qweight":
                t.data = layout.repack_qweight(t.data)
            elif n == "wscales":
                t.data.copy_(layout.repack_wscales(t.data))
            elif n in ("smooth_factor", "bias"):
                t.data.copy_(layout.repack_vec(t.data))
            else:
                t.data.copy_(layout.repack_lowrank(t.data, down=(n == "proj_down")))
            t._svdq_amd = True  # nunchaku_amd._C: this Parameter holds the kernel layout, pass it through unconverted
            self._amd_names.add(n)
        return self

    def _ensure_layout(self):
        if not self._amd_layout:
            self.repack_()
        # the per-tensor marks nunchaku_amd._C reads: derived state, re-stamped from _amd_names (deepcopy / pickling of a
        # Parameter drops python attributes, `_amd_names` survives them)
        for n in self._amd_names:
            getattr(self, n)._svdq_amd = True

    # ------------------------------------------------------------------ API of the reference
    @classmethod
    def from_linear(cls, linear: nn.Linear, **kwargs):
        """Uninitialised quantised twin of ``linear`` (same shapes, dtype, device)."""
        in_features = kwargs.pop("in_features", linear.in_features)
        return cls(
            in_features=in_features,
            out_features=linear.out_features,
            bias=linear.bias is not None,
            torch_dtype=linear.weight.dtype,
            device=linear.weight.device,
            **kwargs,
        )

    def forward(self, x: torch.Tensor, output: torch.Tensor | None = None, ln=None, pool=None) -> torch.Tensor:
        """x [B, S, in] 16-bit -> [B, S, out]: quantise (+ low-rank down) then the fused GEMM."""
        B, S, C_in = x.shape
        x2 = x.reshape(B * S, C_in)
        if output is None:
            output = torch.empty(B * S, self.out_features, dtype=x.dtype, device=x.device)
        qx, ascales, lora_act = self.quantize(x2, ln=ln, pool=pool)
        output = self.forward_quant(qx, ascales, lora_act, output)
        return output.reshape(B, S, -1)

    def quantize(self, x: torch.Tensor, pad_size: int = 256, ln=None, pool=None):
        """x [N, in] -> (FP6 code image [N_pad, 3*in/4] uint8, ascales [in/64, N_pad], lora_act [N_pad, rank] f32).
        ``ln = (stats, scale, shift)``: quantise ``layer_norm(x) * scale + shift`` (fused AdaLayerNormZero, scale incl. +1);
        ``pool``: a ``ZeroPool`` of pre-cleared fp32 scratch for the low-rank accumulator (ops/elementwise.py)."""
        self._ensure_layout()
        return svdq_quantize_w4a4_act_fuse_lora_cuda(
            x, lora_down=self.proj_down, smooth=self.smooth_factor, fp4=False, pad_size=pad_size, ln=ln, pool=pool
        )

    def forward_quant(self, quantized_x, ascales, lora_act, output: torch.Tensor | None = None) -> torch.Tensor:
        """GEMM on pre-quantised input (codes from :meth:`quantize` or from a fused GELU epilogue)."""
        self._ensure_layout()
        if output is None:
            output = torch.empty(
                quantized_x.shape[0], self.out_features, dtype=self.proj_up.dtype, device=quantized_x.device
            )
        svdq_gemm_w4a4_cuda(
            act=quantized_x, wgt=self.qweight, out=output, ascales=ascales, wscales=self.wscales,
            lora_act_in=lora_act, lora_up=self.proj_up, bias=self.bias, fp4=False, alpha=self.wtscale,
            wcscales=self.wcscales, act_unsigned=self.act_unsigned, lora_scales=self.lora_scales,
        )
        return output

    # ------------------------------------------------------------------ runtime LoRA
    @torch.no_grad()
    def set_lora(self, down: torch.Tensor, up: torch.Tensor, strength: float = 1.0) -> "SVDQW4A4Linear":
        """Attach a user LoRA ``W += strength * up @ down`` (``down`` [r, in], ``up`` [out, r], logical layout) by
        widening the low-rank branch to rank ``R + r`` -- the mechanism of the reference's ``update_lora_params`` /
        ``set_lora_strength`` (transformer_flux.py:783-855: concatenate along the rank axis, per-16-rank scales).
        The 4-bit weights are untouched; ``strength`` can be changed later with :meth:`set_lora_strength` for free."""
        if self._offloaded:
            raise RuntimeError("set_lora: this layer belongs to a block that lives in host memory (CPUOffloadManager): its device slots are sized "
                               "for the checkpoint's rank.  Attaching a LoRA to an offloaded block is unsupported whether it happens before or after "
                               "set_offload(True): merge it into the checkpoint, or keep the block resident (num_blocks_on_gpu)")
        self._ensure_layout()
        self.reset_lora()
        r, K = down.shape
        if K != self.in_features or tuple(up.shape) != (self.out_features, r):
            raise ValueError("set_lora: expected down [r, in_features] and up [out_features, r]")
        rp = (r + 15) // 16 * 16
        R = self.rank
        if R + rp > 256:
            raise ValueError(f"set_lora: total rank {R + rp} exceeds 256")
        dt, dev = self.proj_up.dtype, self.proj_up.device
        d = torch.zeros(rp, K, dtype=dt, device=dev)
        d[:r] = down.to(dt)
        u = torch.zeros(self.out_features, rp, dtype=dt, device=dev)
        u[:, :r] = up.to(dt)
        self._base_lowrank = (self.proj_down.data, self.proj_up.data, R)
        # kernel layouts: proj_down rank-major [R][K] behind a [K, R]-shaped parameter, proj_up natural [N][R]
        self.proj_down.data = torch.cat([self.proj_down.data.view(R, K), d], dim=0).reshape(K, R + rp)
        self.proj_up.data = torch.cat([self.proj_up.data, u], dim=1).contiguous()
        self.rank = R + rp
        self.lora_scales = [1.0] * (R // 16) + [float(strength)] * (rp // 16)
        return self

    def set_lora_strength(self, strength: float):
        if self._base_lowrank is None:
            raise RuntimeError("set_lora_strength: no LoRA attached")
        R = self._base_lowrank[2]
        self.lora_scales = [1.0] * (R // 16) + [float(strength)] * ((self.rank - R) // 16)

    @torch.no_grad()
    def reset_lora(self):
        if self._base_lowrank is not None:
            self.proj_down.data, self.proj_up.data, self.rank = self._base_lowrank
            self._base_lowrank, self.lora_scales = None, None

    def __repr__(self):
        return (
            f"SVDQW4A4Linear(in_features={self.in_features}, out_features={self.out_features}, "
            f"rank={self.rank}, precision={self.precision}, act_unsigned={self.act_unsigned})"
        )


class AWQW4A16Linear(nn.Module):
    """AWQ W4A16 linear for M <= 8 rows (reference: nunchaku/models/linear.py:277-414): the AdaLayerNormZero
    modulation projections.  Same constructor, parameter names, shapes and dtypes, so checkpoints load unchanged:
    ``qweight`` int32 [out/4, in/2] (tinychat ``pack_w4`` order, consumed as stored -- no repack),
    ``wscales`` / ``wzeros`` [in/group, out] (zeros pre-scaled: w = q * scale + zero), ``bias`` [out] or None.
    ``forward`` is one launch: the reference's separate ``output.add_(bias)`` is fused into the GEMV
    (same 16-bit rounding point)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, group_size: int = 64,
                 torch_dtype: torch.dtype = torch.bfloat16, device: str | torch.device | None = None):
        super().__init__()
        if device is None:
            device = torch.device("cpu")
        self.in_features, self.out_features, self.group_size = in_features, out_features, group_size
        self.qweight = nn.Parameter(torch.empty(out_features // 4, in_features // 2, dtype=torch.int32, device=device),
                                    requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, dtype=torch_dtype, device=device), requires_grad=True) if bias else None
        self.wscales = nn.Parameter(torch.empty(in_features // group_size, out_features, dtype=torch_dtype, device=device),
                                    requires_grad=False)
        self.wzeros = nn.Parameter(torch.empty(in_features // group_size, out_features, dtype=torch_dtype, device=device),
                                   requires_grad=False)
        # extension: c > 1 makes forward() return the output de-interleaved into c contiguous [out/c] vectors
        # (what ``emb.view(B, -1, c).permute(2, 0, 1)`` of the AdaLayerNormZero modules reads), saving the copy
        self.out_chunks = 1

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ..ops.gemv import awq_gemv_w4a16_cuda

        m = x.numel() // x.shape[-1]
        return awq_gemv_w4a16_cuda(in_feats=x, kernel=self.qweight, scaling_factors=self.wscales, zeros=self.wzeros,
                                   m=m, n=self.out_features, k=self.in_features, group_size=self.group_size, bias=self.bias,
                                   out_chunks=self.out_chunks)

    @classmethod
    def from_linear(cls, linear: nn.Linear, group_size: int = 64, torch_dtype: torch.dtype = torch.bfloat16,
                    device: str = "cpu", **kwargs):
        """Uninitialised layer of the same shape (reference :379-411)."""
        return cls(in_features=linear.in_features, out_features=linear.out_features, bias=linear.bias is not None,
                   group_size=group_size, torch_dtype=torch_dtype, device=device)

    def __repr__(self):
        return f"AWQW4A16Linear(in_features={self.in_features}, out_features={self.out_features}, group_size={self.group_size})"
