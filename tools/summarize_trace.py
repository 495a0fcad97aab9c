#!/usr/bin/env python3
"""Turn the probe's phase-stamp lines (tools/gpu/r5_gemm_trace.sh -> gpurun_out/<dir>/trace.jsonl) into the text table kept under profiles/.

    python tools/summarize_trace.py gpurun_out/tr1 profiles/r5_gemm_phase_trace.txt
"""
import json
import sys

HEADER = """Round 5 (final build): phase stamps of workgroup 0 (probe build of the library, tools/ablate, tools/gpu/r5_gemm_trace.sh) for the four production launches of a
FLUX block at rank 32 and at rank 128 (next-layer rank = rank), geometry 0 (the library's choice) / 1 (256 x 128 tiles) / 2 (128 x 128 queues); k cycles per segment.
Rank 128: lora_act_in arrives as packed 16-bit fragments (one pack launch per GEMM), lora_up staged in LDS for every rank (256 x 128) or packed too (128 x 128).
fuse=2 (GELU_QUANT) at rank 128: geo=0 is the SPLIT low-rank down projection -- the all-rank kernel stores the 16-bit GELU output as fragments ("lowrank-down" 0.1 k
cycles: there is none in the tile) and lowrank_down_split_kernel contracts them behind it; the launch time printed includes the pack kernels and that contraction.
geo=1 / 2 keep the per-tile passes (hybrid carry / atomics): 30-60 k cycles per tile.  (profiles/r5_gemm_phase_trace.txt is the same trace BEFORE the fragments
were packed: "bias + low-rank up" 17 k cycles per tile at rank 128.)

"""


def main(src, dst):
    out = []
    for line in open(src + "/trace.jsonl"):
        try:
            r = json.loads(line)
        except Exception:
            continue
        if "case" in r:
            out.append(r["case"])
        elif "segments" in r:
            out.append("  trace of geometry %s" % r.get("trace_variant"))
            for s in r["segments"][:4]:
                d, prev = [], s[1]
                for x in s[2:]:
                    if x > 0:
                        d.append((x - prev) / 1e3)
                        prev = x
                    else:
                        d.append(0)
                out.append("     loop %.1f | bias+lowrank %.1f  fuse-math %.1f  lowrank-down %.1f  stores %.1f" % ((s[1] - s[0]) / 1e3, d[0], d[1], d[2], d[3]))
        elif "us" in r:
            out.append("  geo=%s %.1f us %.0f TOPS %.3f GHz" % (r.get("geometry"), r["us"], r["TOPS"], r.get("eff_GHz", 0)))
    open(dst, "w").write(HEADER + "\n".join(out) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
