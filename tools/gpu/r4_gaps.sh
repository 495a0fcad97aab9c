#!/bin/bash
# round 4: where is the step's time BETWEEN kernels?  rocprofv3 --kernel-trace of a short bench run; per step: kernel time, idle time between consecutive
# kernels on the stream, and the idle time by (previous kernel -> next kernel) pair.   usage: r4_gaps.sh <outdir> [bench args]
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof "$@" > $R/$O/trace.log 2>&1
cd $R
python - $O <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
f = glob.glob(O + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    n = n.replace('void ', '')
    for k in ('gemm_w4a4_kernel', 'attention_kernel64', 'attention_kernel', 'quantize_kernel_v2', 'residual_kernel', 'gemv_awq_batched_kernel', 'quantize_kernel'):
        if k in n:
            return k + (n.split(k)[1].split('(')[0] if 'gemm' in k else '')
    return 'torch/other'
# the last 4 steps: find the gemv launches (one per step) as step markers
marks = [i for i, r in enumerate(rows) if 'gemv_awq_batched' in r[2]]
out = {}
if len(marks) >= 3:
    a, b = marks[-3], marks[-1]   # two whole steps
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0] + 0  # ns (up to the end of the last kernel before the next marker)
    span = rows[b][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = collections.Counter(); cnt = collections.Counter()
    for (s0, e0, n0), (s1, e1, n1) in zip(seg, rows[a + 1:b + 1]):
        g = max(0, s1 - e0)
        gaps[(short(n0), short(n1))] += g; cnt[(short(n0), short(n1))] += 1
    tot_gap = sum(gaps.values())
    out = {'steps': 2, 'launches_per_step': len(seg) / 2, 'ms_per_step_span': span / 2e6, 'ms_per_step_in_kernels': busy / 2e6, 'ms_per_step_idle_between_kernels': tot_gap / 2e6,
           'pairs': [{'prev': k[0], 'next': k[1], 'per_step': cnt[k] / 2, 'avg_gap_us': gaps[k] / cnt[k] / 1e3, 'ms_per_step': gaps[k] / 2e6} for k in sorted(gaps, key=lambda k: -gaps[k])[:25]]}
    kt = collections.Counter(); kc = collections.Counter()
    for s, e, n in seg:
        kt[short(n)] += e - s; kc[short(n)] += 1
    out['kernels'] = [{'kernel': k, 'per_step': kc[k] / 2, 'ms_per_step': kt[k] / 2e6} for k in sorted(kt, key=lambda k: -kt[k])]
# the step boundary in detail: every kernel from 45 launches before the step's GEMV to 6 after it
if len(marks) >= 2:
    m = marks[-2]
    t0 = rows[m - 45][0]
    out['boundary'] = [{'us_since': (s - t0) / 1e3, 'dur_us': (e - s) / 1e3, 'gap_before_us': (s - rows[i - 1][1]) / 1e3, 'kernel': n.replace('void ', '')[:90]}
                       for i, (s, e, n) in enumerate(rows) if m - 45 <= i <= m + 6]
json.dump(out, open(O + '/gaps.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ('pairs', 'kernels', 'boundary')}))
for b in out.get('boundary', []): print('%9.1f us  dur %7.1f  gap %7.1f  %s' % (b['us_since'], b['dur_us'], b['gap_before_us'], b['kernel']))
for p in out.get('pairs', [])[:16]: print(p)
for k in out.get('kernels', []): print(k)
PY
rm -rf $O/trace
