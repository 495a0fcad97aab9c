#!/usr/bin/env python3
"""Turn the probe's phase-stamp lines (tools/gpu/r5_gemm_trace.sh -> gpurun_out/<dir>/trace.jsonl) into the text table kept under profiles/.

    python tools/summarize_trace.py gpurun_out/tr1 profiles/r5_gemm_phase_trace.txt
"""
import json
import sys

HEADER = """Round 5: phase stamps of workgroup 0 (probe build of the library, tools/ablate, tools/gpu/r5_gemm_trace.sh) for the four production launches of a FLUX block
at rank 32 and at rank 128 (next-layer rank = rank), geometry 0 (the library's choice) / 1 (256 x 128 tiles) / 2 (128 x 128 queues); k cycles per segment.
Taken BEFORE the low-rank activations were packed: at rank 128 the phase "bias + low-rank up" is 17 k cycles per 256 x 128 tile (2.5 k at rank 32 with the
staged operands) and 16-18 k per 128 x 128 tile -- the row-per-lane fp32 loads of lora_act_in (256 KB per tile, 16 bytes used of every cache line touched,
both column waves of a row block fetching the same rows).  With the fragments packed once per launch the bench lines of the same shapes give, on one box
(profiles/r5_rank_ab.txt): default 129.4 -> 122.0 us, QKV 199.6 -> 182.3 us.  geo=0 at rank 128, fuse=2 is the solo-carry kernel (128 x 128 tiles, one
workgroup per CU): its low-rank-down phase is 6.9 k cycles where the per-tile atomics of the other geometries take 32-68 k.

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
