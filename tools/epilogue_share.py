#!/usr/bin/env python3
"""Per-variant epilogue share of a GEMM tile from the probe's phase stamps (tools/gpu/r5_gemm_trace.sh -> trace.jsonl): the rank-32 launches of a FLUX block under
the geometry the library picks (0).  share = cycles behind the main loop / cycles of the whole segment, median over workgroup 0's whole-tile segments.
Writes the JSON bench.py quotes as roofline.epilogue_share (stamped with the hash of the kernel sources it was measured on).

    python tools/epilogue_share.py gpurun_out/<dir> profiles/r5_gemm_epilogue_share.json
"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "default", 2: "gelu_quant", 3: "rmsnorm_rope"}


def main(src, dst):
    import bench

    sha_file = os.path.join(ROOT, "tools", "ablate", "libsvdq_amd_probe.so.sha16")
    probe_sha = open(sha_file).read().strip() if os.path.exists(sha_file) else "unknown (probe library built before the stamp existed)"
    out = {"csrc_sha16": probe_sha,  # the sources the PROBE library was built from (tools/ablate/build.py), not whatever the tree holds now
           "source": "tools/gpu/r5_gemm_trace.sh: shader-cycle stamps of workgroup 0 (probe build of the library), rank 32, geometry 0, M = 4608; "
                     "share = (segment - loop) / segment, median over its whole-tile segments", "per_variant": {}}
    case, shares = None, {}
    for line in open(os.path.join(src, "trace.jsonl")):
        try:
            r = json.loads(line)
        except Exception:
            continue
        if "case" in r:
            case = r["case"]
        elif "segments" in r and case and case.startswith("R=32") and r.get("trace_variant") == 0:
            fuse = int(case.split("fuse=")[1])
            K = int(case.split("K=")[1].split()[0])
            rows = []
            for s in r["segments"]:
                if s[1] <= s[0] or s[2] <= 0:   # a stream-K partial segment (no epilogue) or an unused slot
                    continue
                end = max(x for x in s[1:] if x > 0)
                rows.append(((s[1] - s[0]), end - s[1]))
            if rows:
                loop = statistics.median(x[0] for x in rows)
                epi = statistics.median(x[1] for x in rows)
                shares.setdefault(NAMES[fuse], []).append({"K": K, "loop_kcycles": loop / 1e3, "epilogue_kcycles": epi / 1e3, "share": epi / (loop + epi)})
    for name, lst in shares.items():
        out["per_variant"][name] = lst if len(lst) > 1 else lst[0]
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["per_variant"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
