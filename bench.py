#!/usr/bin/env python
"""Headline benchmark: denoise steps/s of a FLUX.1-dev-shaped 4-bit (SVDQuant W4A4 + rank-32)
transformer at 1024x1024, bs=1 per GPU, on N GPUs of one node (independent replicas).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one transformer forward (19 joint + 38 single blocks, 4096 image + 512 text tokens,
guidance embedding on) followed by the Euler scheduler update, on synthetic inputs and random-init
weights of the FLUX.1-dev architecture (no checkpoints exist in this environment).  All 57 blocks
and every operator run inside the timed region; inputs are resident in HBM before it starts.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel (gemm_w4a4): algorithmic ops / summed kernel time, measured
                  live with HIP events recorded around every launch on the launch stream
                  (svdq_prof_*, include/svdq_amd.h) during the timed steps, against the dense INT8
                  MFMA peak of MI355X (BASELINE.json's yardstick; the kernel itself runs the 4-bit
                  codes on the FP6 matrix path);
  cpu_baseline -- the CPU oracle (a numpy port; the reference ships no CPU path) timed on a bounded
                  sample of the same workload on this box's host cores (rank 0, N=1 only).
"""

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL / CUDA-tensor sharing fail with the legacy mode): set it before
# torch initialises the HIP runtime, so that a bare `python -m torch.distributed.run ... bench.py --gpus N` works
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

# MI355X dense INT8 MFMA peak: 256 CU x 4 SIMD x 1024 MAC/clk x 2 op/MAC x 2.4 GHz
# (MI355X_MICROARCH.md: i8 = 2x the 2.5 PF bf16 dense peak)
INT8_PEAK_TOPS = 256 * 4 * 1024 * 2 * 2.4e9 / 1e12
# the pipe the kernel actually runs on: FP6/FP4 MX MFMA, dense, 2x the INT8 rate (MI355X_MICROARCH.md: ~10 PF)
FP6_PEAK_TOPS = 2 * INT8_PEAK_TOPS
BF16_PEAK_TFLOPS = INT8_PEAK_TOPS / 2
# W4A4 + low-rank work of one FLUX.1-dev 1024^2 step: 59.5 TOP + 0.83 TFLOP
FLUX_STEP_GOP = 59.5e3 + 0.83e3
# HBM-side traffic of the dominant kernel, bytes per launch averaged over the gemm_w4a4 dispatches of THIS command:
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_profile_bench.sh), FETCH_SIZE doubled as
# MI355X_MICROARCH.md prescribes for gfx950.  PMC counters cannot be read from inside the timed run: the line carries the
# COMMITTED measurement of the same command and says so ("traffic_source"); null when the profile file is absent.
# Both files carry "csrc_sha16": the hash of the kernel sources they were measured on (kernel_sources_sha16 below, stamped by
# tools/gpu/r4_profile_bench.sh); a line printed from a tree whose kernels changed since says "stale": true next to the number.
TRAFFIC_PROFILES = [os.path.join(ROOT, "profiles", n) for n in ("r6_bench_gemm_hbm_counters.json", "r5_bench_gemm_hbm_counters.json", "r4_bench_gemm_hbm_counters.json")]
MFMA_PROFILES = [os.path.join(ROOT, "profiles", n) for n in ("r6_bench_gemm_mfma_util.json", "r5_bench_gemm_mfma_util.json", "r4_bench_gemm_mfma_util.json")]
# per-variant share of a tile spent behind the main loop (shader-cycle stamps of the probe build, tools/gpu/r5_gemm_trace.sh + tools/epilogue_share.py): committed, stamped
EPILOGUE_SHARE_PROFILE = os.path.join(ROOT, "profiles", "r6_gemm_epilogue_share.json")


def kernel_sources_sha16():
    """sha256 (first 16 hex digits) over the kernel sources of the library: nunchaku_amd/csrc/*.{hip,h,inc} in name order"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "nunchaku_amd", "csrc", "*"))):
        if path.endswith((".hip", ".h", ".inc")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def committed_traffic():
    """-> (bytes per launch, file, measured on these kernel sources?)"""
    for path in TRAFFIC_PROFILES:
        try:
            d = json.load(open(path))
            return ((2 * d["FETCH_SIZE"]["avg_per_dispatch_KB"] + d["WRITE_SIZE"]["avg_per_dispatch_KB"]) * 1024, os.path.relpath(path, ROOT),
                    d.get("csrc_sha16") == kernel_sources_sha16())
        except Exception:
            continue
    return None, None, False


def committed_mfma_util():
    """SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over the gemm_w4a4 dispatches of this command (tools/gpu/r4_profile_bench.sh), with the
    staleness stamp"""
    for path in MFMA_PROFILES:
        try:
            d = json.load(open(path))
            d["file"] = os.path.relpath(path, ROOT)
            d["stale"] = d.get("csrc_sha16") != kernel_sources_sha16()
            return d
        except Exception:
            continue
    return None


def committed_epilogue_share():
    try:
        d = json.load(open(EPILOGUE_SHARE_PROFILE))
        d["file"] = os.path.relpath(EPILOGUE_SHARE_PROFILE, ROOT)
        d["stale"] = d.get("csrc_sha16") != kernel_sources_sha16()
        return d
    except Exception:
        return None


class ClockSampler:
    """The shader clock during the timed region: the `*` line of the device's pp_dpm_sclk (what `rocm-smi --showclocks` prints), read every
    50 ms by a thread.  profiles/r4_clock_instrument.txt: this reading agrees with GRBM_GUI_ACTIVE / dispatch duration within 1 % for the GEMM
    launches (s_memtime does not: it undercounts in issue-sparse kernels)."""

    def __init__(self, device_index: int):
        import glob
        import threading
        self.path = None
        # the sysfs card of THIS HIP device: matched by PCI address (a box shows every card of the node, the process sees one of them as device 0)
        bdf = self._pci_address(device_index)
        for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            if bdf and os.path.basename(os.path.realpath(os.path.dirname(f))).lower() == bdf:
                self.path = f
        self.samples, self._stop, self._thread = [], threading.Event(), None
        if self.path:
            self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _pci_address(device_index: int):
        """'0000:bb:dd.f' of a HIP device (hipDeviceGetPCIBusId), or None"""
        import ctypes
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0 and buf.value:
                return buf.value.decode().strip().lower()
        except Exception:
            pass
        try:
            pr = torch.cuda.get_device_properties(device_index)
            return "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            return None

    def _run(self):
        import re
        while not self._stop.is_set():
            try:
                m = re.search(r"(\d+)Mhz\s*\*", open(self.path).read())
                if m:
                    self.samples.append(int(m.group(1)))
            except Exception:
                pass
            self._stop.wait(0.05)

    def start(self):
        if self._thread:
            self._thread.start()

    def mark(self):
        """end of a region: returns the median GHz of the samples since the previous mark (None: unreadable file / fewer than three samples -- the
        first sample of a short region can still show the idle clock) and starts the next region.  The sampler thread keeps running."""
        if not self._thread:
            return None
        xs, self.samples = sorted(self.samples), []
        return xs[len(xs) // 2] / 1e3 if len(xs) >= 3 else None

    def stop(self):
        """median GHz of the samples since the last mark; ends the thread"""
        if not self._thread:
            return None
        self._stop.set()
        self._thread.join()
        return self.mark()


def cpu_baseline(max_seconds: float = 30.0):
    """Time the numpy oracle (oracle/svdq_oracle.py, fp32 mode) on BASELINE config 1: one SVDQuant linear 3072->3072,
    rank 32 (activation quantisation + low-rank + int4 GEMM + bias), at M = 512 tokens (median of <= 3 runs) and once at
    M = 4096 (SURVEY.md section 8d asks for both); the reported value is the M = 4096 rate when it was measured."""
    import numpy as np

    from oracle import svdq_oracle as O

    K, N, R = 3072, 3072, 32
    L = O.make_svdq_layer(K, N, R, seed=0, cheap=True)
    x = O.make_activations(4096, K, seed=0)
    O.svdq_linear(x[:64], L)  # warm-up (BLAS threads, caches)
    gop = lambda M: (2.0 * M * N * K + 2.0 * M * R * (K + N)) / 1e9
    ts = []
    t_all = time.perf_counter()
    while len(ts) < 3 and time.perf_counter() - t_all < max_seconds / 3:
        t0 = time.perf_counter()
        O.svdq_linear(x[:512], L)
        ts.append(time.perf_counter() - t0)
    t512 = float(np.median(ts))
    gops = gop(512) / t512
    sample = f"numpy oracle, 1 SVDQuant linear 3072->3072 r=32: M=512 median of {len(ts)} runs {t512:.2f} s = {gops:.1f} GOP/s"
    if t512 * 8 < max_seconds:  # M = 4096 costs ~8x: only when it fits the budget
        t0 = time.perf_counter()
        O.svdq_linear(x, L)
        t4096 = time.perf_counter() - t0
        gops = gop(4096) / t4096
        sample += f"; M=4096 one run {t4096:.2f} s = {gops:.1f} GOP/s (reported)"
    return {
        "value": gops / FLUX_STEP_GOP,
        "unit": "steps/s (equivalent: oracle GOP/s / 60.3 TOP of W4A4+low-rank work per step)",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": sample + f" (numpy/BLAS threads = all {os.cpu_count()} cores; the reference ships no CPU path)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # the headline config is a 50-step denoise loop (~3 s at N=1)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["dev1024", "schnell512", "qwen1024"], default="dev1024",
                    help="dev1024: BASELINE config 3 (FLUX.1-dev, 1024^2, guidance embedding, 4096+512 tokens; the headline). "
                         "schnell512: BASELINE config 2 (FLUX.1-schnell, 512^2, no guidance embedding, 1024+512 tokens, "
                         "4 timed steps per image by default).  qwen1024: BASELINE config 5 (Qwen-Image, 60 dual-stream blocks, "
                         "1024^2 = 4096 image + 512 text tokens); with --offload N the blocks live in pinned host memory and N stay resident")
    ap.add_argument("--offload", type=int, default=0, metavar="N",
                    help="qwen1024 only: layer-wise host offload with N blocks resident on the GPU (0 = everything resident)")
    ap.add_argument("--offload-slots", type=int, default=4, metavar="S",
                    help="qwen1024 --offload: device slots of the offload ring (CPUOffloadManager num_slots: one computes while S - 1 fill)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="torch.distributed backend for --gpus > 1 (default: nccl = RCCL; gloo lets several ranks share one GPU: smoke tests)")
    ap.add_argument("--resolution", type=int, nargs="+", default=None, metavar="PIXELS",
                    help="image size: one number (square) or WIDTH HEIGHT, multiples of 16 -- e.g. 1664 928, the reference's Qwen-Image quality "
                         "gate (tests/v1/qwenimage/test_qwenimage.py:21): token counts that are not a multiple of 256 run the same fused path "
                         "on padded streams")
    ap.add_argument("--txt-tokens", type=int, default=512)
    ap.add_argument("--weight-codes", choices=["uniform", "residual"], default="residual",
                    help="distribution of the random 4-bit weight codes: that of a quantised Gaussian residual (SURVEY 8d's synthetic layers; the default "
                         "since round 5) or uniform (the round 1-4 default; profiles/r4_weight_codes_ab.txt: no measurable difference)")
    ap.add_argument("--rank", type=int, default=32, metavar="R",
                    help="rank of every low-rank branch (a multiple of 16): 32 = the SVDQuant default; 128 = the reference's r128 Qwen-Image / "
                         "FLUX checkpoints (tests/v1/qwenimage/test_qwenimage.py:20-26)")
    ap.add_argument("--lora", type=int, default=0, metavar="r",
                    help="attach a runtime LoRA of rank r to every SVDQuant linear (set_lora: the low-rank branch becomes rank R + r with per-16-rank "
                         "scales -- the reference's update_lora_params, transformer_flux.py:783-855)")
    ap.add_argument("--prof-steps", type=int, default=10, metavar="P",
                    help="steps bracketed with per-launch HIP events AFTER the timed region (the roofline object); 0 = none")
    ap.add_argument("--no-prof", action="store_true", help="same as --prof-steps 0")
    ap.add_argument("--layers", type=int, nargs=2, default=(19, 38), help=argparse.SUPPRESS)  # debugging only
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--attention-geometry", type=int, default=0, choices=[0, 1, 2],
                    help="workgroup geometry of the attention kernel (svdq_attention_args.geometry): 0 = the library's choice (4 waves x 64 "
                         "rows on the prescaled Q the QKV GEMM emits), 1 = 8 waves x 32 rows, 2 = 4 x 64 with its persistent schedule (same-box A/B)")
    ap.add_argument("--geometry", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 7],
                    help="workgroup geometry of the W4A4 GEMM (svdq_gemm_args.geometry): 0 = the library's choice, 1 = 256x128 "
                         "tiles / one workgroup per CU, 2 = 128x128 tiles / two per CU out of phase, 3 = 2 without the phase offset; "
                         "6 / 7 = GELU_QUANT launches with a next-layer rank beyond 32 on the solo-carry kernel / with the split low-rank "
                         "down projection (A/B of what 0 picks from rank 96), everything else as 0")
    ap.add_argument("--no-fragment-cache", action="store_true",
                    help="A/B (rank 48 .. 160): the launches pack their weight-side low-rank operands themselves, as ABI 20 did (default: one cached fragment image per parameter, ABI 21)")
    ap.add_argument("--no-attention-split", action="store_true",
                    help="A/B: keep the low-rank down projection of the attention epilogue's quantiser inside the epilogue at every rank (rank 48 .. 160 "
                         "runs it as a contraction kernel behind the attention kernel by default)")
    ap.add_argument("--deterministic", nargs="?", const="strict", default=False, choices=["strict", "runs"],
                    help="fixed-point low-rank accumulation (nunchaku_amd.mode): bit-reproducible steps; 'strict' (default of the flag): independent of the "
                         "launch configuration; 'runs': a GELU_QUANT workgroup sums its run of column tiles in fp32 in a fixed order first (ABI 22)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one captured HIP graph (no per-launch events: the roofline object is then "
                         "measured on one extra eager step after the timed region)")
    args = ap.parse_args()
    schnell = args.config == "schnell512"
    qwen = args.config == "qwen1024"
    if args.offload and not qwen:
        ap.error("--offload needs --config qwen1024")
    if qwen and "--steps" not in " ".join(sys.argv):
        args.steps, args.warmup = 10, 2
    if args.resolution is None:
        args.resolution = [512 if schnell else 1024]
    if len(args.resolution) > 2 or any(r <= 0 or r % 16 for r in args.resolution):
        ap.error("--resolution takes one or two positive multiples of 16")
    width, height = args.resolution[0], args.resolution[-1]
    res_name = f"{width}x{height}"
    if schnell and "--steps" not in " ".join(sys.argv):
        args.steps, args.warmup = 4 * 10, 4  # ten 4-step images back to back

    from nunchaku_amd import _lib, replica
    from nunchaku_amd.models.flux import FluxTransformerAMD

    from nunchaku_amd import mode
    from nunchaku_amd._C import _Ops

    _Ops.gemm_geometry = args.geometry
    _Ops.attention_geometry = args.attention_geometry
    _Ops.attention_split_lowrank = not args.no_attention_split
    _Ops.cache_packed_lowrank = not args.no_fragment_cache
    mode.set_deterministic(args.deterministic)
    rank, local_rank, world = replica.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    if args.backend == "gloo":  # several ranks may share a device (single-GPU smoke test of the N > 1 path)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = _lib.load()

    # ---- model: rank 0 initialises, RCCL broadcasts the parameters (the only collective) ------
    if qwen:
        from nunchaku_amd.models.qwenimage import NunchakuQwenImageTransformer2DModel

        n_blocks = args.layers[0] if "--layers" in " ".join(sys.argv) else 60
        model = NunchakuQwenImageTransformer2DModel(num_layers=n_blocks, rank=args.rank, device=dev)
    else:
        model = FluxTransformerAMD(num_layers=args.layers[0], num_single_layers=args.layers[1], guidance_embeds=not schnell, rank=args.rank, device=dev)
    if rank == 0:
        model.init_synthetic_(seed=0, codes=args.weight_codes)
    bcast_bytes = replica.broadcast_module_(model, src=0)
    model.eval()
    if args.lora:
        # the same synthetic LoRA on every rank (seeded): W += 0.8 * up @ down on every SVDQuant linear, through the widened low-rank branch
        gl = torch.Generator(device=dev).manual_seed(99)
        for m in model.svdq_layers():
            m.set_lora(torch.randn(args.lora, m.in_features, generator=gl, device=dev) * (0.5 / m.in_features ** 0.5),
                       torch.randn(m.out_features, args.lora, generator=gl, device=dev) * (0.5 / args.lora ** 0.5), strength=0.8)
    if qwen and args.offload:
        model.set_offload(True, num_blocks_on_gpu=args.offload, num_slots=args.offload_slots)

    # ---- one independent image per rank ------------------------------------------------------
    gh, gw = height // 16, width // 16   # the grid of 2 x 2 latent patches
    t_img, t_txt = gh * gw, args.txt_tokens
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    latents = torch.randn(1, t_img, 64, generator=g, device=dev, dtype=torch.bfloat16)
    enc = torch.randn(1, t_txt, 3584 if qwen else 4096, generator=g, device=dev, dtype=torch.bfloat16)
    pooled = torch.randn(1, 768, generator=g, device=dev, dtype=torch.bfloat16)
    img_ids = torch.zeros(t_img, 3, device=dev)
    img_ids[:, 1] = torch.arange(gh, device=dev).repeat_interleave(gw)
    img_ids[:, 2] = torch.arange(gw, device=dev).repeat(gh)
    txt_ids = torch.zeros(t_txt, 3, device=dev)
    guidance = None if schnell else torch.full((1,), 3.5, device=dev)
    txt_mask = torch.ones(1, t_txt, dtype=torch.long, device=dev)
    total = args.steps + args.warmup
    sigmas = torch.linspace(1.0, 0.0, total + 1, device=dev)

    def step(i, lat):
        with torch.no_grad():
            if qwen:
                # the pipeline's call: an all-ones text mask and the text lengths ride along (the reference's processor ignores the mask)
                v = model(lat, enc, txt_mask, sigmas[i].reshape(1), [(1, gh, gw)], txt_seq_lens=[t_txt], return_dict=False)[0]
            else:
                v = model(lat, enc, pooled, sigmas[i].reshape(1), img_ids, txt_ids, guidance)
            return lat + (sigmas[i + 1] - sigmas[i]).to(v.dtype) * v  # Euler / flow-matching update

    for i in range(args.warmup):
        latents = step(i, latents)

    if args.graph:
        from nunchaku_amd.graph import CapturedStep

        def graph_fn(lat, sig, dsig):
            v = model(lat, enc, pooled, sig, img_ids, txt_ids, guidance)
            return lat + dsig.to(v.dtype) * v

        captured = CapturedStep(graph_fn, [latents, sigmas[0].reshape(1), (sigmas[1] - sigmas[0]).reshape(1)])
        eager_step = step

        def step(i, lat):  # noqa: F811
            return captured(lat, sigmas[i].reshape(1), (sigmas[i + 1] - sigmas[i]).reshape(1))

    n_gemm = sum(1 for _ in model.svdq_layers())
    # The timed region carries NO instrumentation (an event pair around a launch serialises the queue for ~3 us; 228 GEMM launches
    # per step were ~1.3 % of the reported step).  The roofline object is measured right behind it: P more steps of the same loop with HIP events
    # around every gemm_w4a4 launch (and nothing else), timed as well -> ms_per_step_instrumented.
    prof_steps = 0 if (args.no_prof or args.graph) else max(0, args.prof_steps)
    clock = ClockSampler(local_rank)
    replica.barrier()
    torch.cuda.synchronize()
    clock.start()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        latents = step(i, latents)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0          # this replica's own time for its K steps (before it waits for the others)
    clock_ghz = clock.mark()                # the shader clock of the TIMED region only (the instrumented steps behind it are a region of their own)
    replica.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = replica.max_over_ranks(elapsed, dev)
    own_max, own_min = replica.max_over_ranks(own, dev), -replica.max_over_ranks(-own, dev)
    finite = bool(torch.isfinite(latents.float()).all())
    ms_instrumented = None
    if prof_steps:
        _lib.check(lib.svdq_prof_select(1 << 0), "svdq_prof_select")
        _lib.check(lib.svdq_prof_enable(max(1, 2 * n_gemm * (prof_steps + 1) + 64)), "svdq_prof_enable")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lat_p = latents
        for i in range(prof_steps):
            lat_p = step(total - 1 - (i % max(1, args.steps)), lat_p)
        torch.cuda.synchronize()
        ms_instrumented = (time.perf_counter() - t1) / prof_steps * 1e3
    clock_ghz_prof = clock.stop()  # ... of the event-bracketed steps the roofline object was measured on

    def prof(cls):
        n, ms, work = C.c_int64(), C.c_double(), C.c_double()
        _lib.check(lib.svdq_prof_read(cls, C.byref(n), C.byref(ms), C.byref(work)), "svdq_prof_read")
        return n.value, ms.value, work.value

    if args.graph:  # per-launch events cannot live inside a graph: bracket one extra eager step instead
        step = eager_step
        _lib.check(lib.svdq_prof_select(1 << 0), "svdq_prof_select")
        _lib.check(lib.svdq_prof_enable(max(1, 4 * n_gemm + 64)), "svdq_prof_enable")
        step(total - 1, latents.clone())
        torch.cuda.synchronize()
    n_g, ms_g, ops_g = prof(0)
    prof_steps = 1 if args.graph else max(prof_steps, 1)
    per_variant = {}
    for name, fuse in (("default", 0), ("silu", 1), ("gelu_quant", 2), ("rmsnorm_rope", 3)):
        n_v_, ms_v_, ops_v_ = prof(0 | ((fuse + 1) << 8))  # SVDQ_PROF_GEMM_VARIANT(fuse), include/svdq_amd.h
        if n_v_:
            tops = ops_v_ / (ms_v_ * 1e-3) / 1e12
            per_variant[name] = {"launches_per_step": n_v_ / prof_steps, "avg_launch_us": ms_v_ * 1e3 / n_v_, "ms_per_step": ms_v_ / prof_steps,
                                 "TOPs": tops, "frac": tops / INT8_PEAK_TOPS}
    # the quantiser's numbers come from ONE extra, untimed step bracketed on its class only
    lib.svdq_prof_select(1 << 1)
    lib.svdq_prof_reset()
    step(total - 1, latents)
    torch.cuda.synchronize()
    n_q, ms_q, bytes_q = prof(1)
    # ... and the attention kernel's from another one
    lib.svdq_prof_select(1 << 2)
    lib.svdq_prof_reset()
    step(total - 1, latents)
    torch.cuda.synchronize()
    n_a, ms_a, flops_a = prof(2)
    # ... and the AWQ GEMV's (the modulation projections: one batched launch per step) from a third
    lib.svdq_prof_select(1 << 3)
    lib.svdq_prof_reset()
    step(total - 1, latents)
    torch.cuda.synchronize()
    n_v, ms_v, bytes_v = prof(3)
    lib.svdq_prof_enable(0)
    lib.svdq_prof_select(0xFFFFFFFF)

    if rank == 0:
        achieved = ops_g / (ms_g * 1e-3) / 1e12 if ms_g > 0 else 0.0
        traffic, traffic_file, traffic_fresh = committed_traffic()
        if qwen:
            workload = (f"Qwen-Image-shaped transformer step, {res_name} ({t_img} image + {t_txt} text tokens), bs=1 "
                        f"per GPU, {len(model.transformer_blocks)} dual-stream blocks, int4 rank-{args.rank}" + (f" + rank-{args.lora} runtime LoRA" if args.lora else "") + ", random-init weights, " +
                        (f"layer-wise host offload with {args.offload} blocks resident ({model.offload_manager.host_bytes_per_block() / 1e6:.0f} MB "
                         f"per block over PCIe, ring of {args.offload_slots} device slots)" if args.offload else "all blocks resident"))
        else:
            workload = (f"FLUX.1-{'schnell' if schnell else 'dev'}-shaped transformer step, {res_name} "
                        f"({t_img} image + {t_txt} text tokens), bs=1 per GPU, {args.layers[0]} joint + "
                        f"{args.layers[1]} single blocks, guidance embedding {'off' if schnell else 'on'}, int4 rank-{args.rank}" +
                        (f" + rank-{args.lora} runtime LoRA" if args.lora else "") + ", random-init weights")
        line = {
            "metric": (f"denoise steps/sec Qwen-Image {res_name} bs=1 (4-bit SVDQuant W4A4 + rank-{args.rank})" if qwen else
                       f"denoise steps/sec FLUX.1-schnell {res_name} bs=1 (4-bit SVDQuant W4A4 + rank-{args.rank})" if schnell else
                       f"denoise steps/sec FLUX.1-dev {res_name} bs=1 (4-bit SVDQuant W4A4 + rank-{args.rank})").replace("1024x1024", "1024^2").replace("512x512", "512^2"),
            "value": world * args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # the same loop with HIP events around every gemm_w4a4 launch (the steps the roofline object was measured on; null: --no-prof / --graph)
            "ms_per_step_instrumented": ms_instrumented,
            "higher_is_better": True,
            "scaling": "weak",
            # aggregate steps/s (value) and the slowest / fastest replica's own rate over its K steps
            "per_replica": {"min_steps_per_s": args.steps / own_max, "max_steps_per_s": args.steps / own_min},
            "vs_baseline": None,
            "dtype": "int4 codes as FP6 (e2m3) operands of the f8f6f4 MFMA (exact products; per-group scales applied in fp32), fp32 accumulate, bf16 I/O",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "gemm_geometry": args.geometry, "attention_geometry": args.attention_geometry, "deterministic": args.deterministic,
                "weights": ("uniform-random 4-bit codes (--weight-codes uniform: the round 1-4 default)" if args.weight_codes == "uniform"
                            else "4-bit codes distributed like a quantised Gaussian residual, rne(N(0, 2.9^2)) clamped to +-7 (SURVEY 8d)"),
                "rank": args.rank, "runtime_lora_rank": args.lora,
                "parallelism": f"{world} independent replica(s), one image each; weights broadcast once over RCCL "
                               f"({bcast_bytes / 1e9:.2f} GB)" + ("; step replayed as one HIP graph" if args.graph else ""),
                "output_finite": finite,
            },
            "roofline": {
                "kernel": "svdq::gemm_w4a4_kernel (all epilogue variants)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": INT8_PEAK_TOPS,
                "unit": "TOP/s",
                "frac": achieved / INT8_PEAK_TOPS,
                "frac_int8": achieved / INT8_PEAK_TOPS,   # BASELINE.json's yardstick
                "frac_fp6": achieved / FP6_PEAK_TOPS,     # the matrix pipe the 4-bit product actually runs on
                "traffic": traffic,
                "traffic_source": f"committed: {traffic_file} (rocprofv3 PMC passes of this command: "
                                  "2*FETCH_SIZE + WRITE_SIZE per gemm_w4a4 dispatch); not measured in this run",
                "traffic_stale": not traffic_fresh,   # true: the kernel sources changed since that file was measured
                "mfma_util": committed_mfma_util(),
                # share of a tile's cycles behind the main loop per epilogue variant (committed probe measurement): what the launch-level
                # fraction loses against the loop's own
                "epilogue_share": committed_epilogue_share(),
                # shader clock during the timed region (pp_dpm_sclk sampled every 50 ms on rank 0): the chip runs these kernels at its power
                # limit; profiles/r4_clock_instrument.txt ties this reading to GRBM_GUI_ACTIVE / dispatch duration
                "effective_clock_ghz": clock_ghz,
                "effective_clock_ghz_prof_steps": clock_ghz_prof,   # the same reading over the instrumented steps (where `achieved` was measured)
                "per_variant": per_variant,
                # the three per-variant fractions once more as flat keys (the driver's record keeps scalars of this object and drops nested ones)
                **{f"frac_{k}": v["frac"] for k, v in per_variant.items()},
                **{f"avg_launch_us_{k}": v["avg_launch_us"] for k, v in per_variant.items()},
                "launches": n_g,
                "prof_steps": prof_steps,
                "avg_launch_us": ms_g * 1e3 / max(n_g, 1),
                "gemm_ms_per_step": ms_g / prof_steps,
                "measured_on": "one extra eager step (graph replay in the timed region)" if args.graph else
                               f"{prof_steps} steps run right behind the timed region with HIP events around every gemm_w4a4 launch (the timed region itself carries none)",
                "quantize": {"launches": n_q, "ms_per_step": ms_q, "measured_on": "one extra untimed step",
                             "GBps": bytes_q / (ms_q * 1e-3) / 1e9 if ms_q > 0 else 0.0, "bound": "hbm"},
                "attention": {"launches": n_a, "ms_per_step": ms_a, "measured_on": "one extra untimed step",
                              "TFLOPs": flops_a / (ms_a * 1e-3) / 1e12 if ms_a > 0 else 0.0, "bound": "mfma",
                              "frac_bf16": (flops_a / (ms_a * 1e-3) / 1e12 / BF16_PEAK_TFLOPS) if ms_a > 0 else 0.0},
                "gemv_awq": {"launches": n_v, "ms_per_step": ms_v, "measured_on": "one extra untimed step",
                             "GBps": bytes_v / (ms_v * 1e-3) / 1e9 if ms_v > 0 else 0.0, "bound": "hbm by bytes, VALU in practice"},
            },
        }
        # ... and at the top level of the contract line
        for k, v in per_variant.items():
            line[f"roofline_frac_{k}"] = v["frac"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
