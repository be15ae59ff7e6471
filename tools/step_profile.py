#!/usr/bin/env python
"""Per-training-step kernel budget from a rocprofv3 kernel trace (csv): sums kernel durations between consecutive
optimiser kernels and prints the per-kernel totals of a representative step.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline
    python tools/step_profile.py out/*/*_kernel_trace.csv
"""
import collections
import csv
import sys


def main(path, marker='adam_ema_kernel'):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    if len(idx) < 12:
        idx = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name']]
    spans = []
    for a, b in zip(idx[:-1], idx[1:]):
        seg = rows[a + 1:b + 1]
        if len(seg) < 50:
            continue
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
        span = int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])
        spans.append((busy, span, a, b))
    spans = spans[5:]                      # skip warm-up
    cnt = sorted(s[3] - s[2] for s in spans)[len(spans) // 2]
    spans = [s for s in spans if abs((s[3] - s[2]) - cnt) <= 6]     # training steps only: drop spans that straddle another
                                                                   # phase of the command (forward-only loops, probes)
    busy = sorted(s[0] for s in spans)
    print('steps %d  kernel-time per step: median %.3f ms  min %.3f  max %.3f   (span median %.3f ms)' % (
        len(spans), busy[len(busy) // 2] / 1e6, busy[0] / 1e6, busy[-1] / 1e6,
        sorted(s[1] for s in spans)[len(spans) // 2] / 1e6))
    tot = collections.defaultdict(lambda: [0, 0])
    for _, _, a, b in spans:
        for r in rows[a + 1:b + 1]:
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:78]
            tot[k][0] += 1
            tot[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    n = float(len(spans))
    print('%-80s %7s %9s %8s' % ('kernel', 'calls', 'us/step', 'avg us'))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
        print('%-80s %7.1f %9.1f %8.2f' % (k, v[0] / n, v[1] / n / 1e3, v[1] / v[0] / 1e3))


if __name__ == '__main__':
    main(sys.argv[1])
