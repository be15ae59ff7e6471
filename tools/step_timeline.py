#!/usr/bin/env python
"""Timeline of one training step from a rocprofv3 kernel trace: per queue busy time, and the main queue's kernels in
order (name, start, duration, gap to the previous kernel).  usage: step_timeline.py trace.csv [step_index]"""
import collections
import csv
import sys


def main(path, which=20):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_ema' in r['Kernel_Name']]
    a, b = idx[which], idx[which + 1]
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]['Start_Timestamp'])
    busy = collections.Counter()
    for r in seg:
        busy[r['Queue_Id']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    mainq = max(busy, key=busy.get)
    span = int(seg[-1]['End_Timestamp']) - t0
    print('step span %.1f us; busy per queue: %s' % (span / 1e3, {q: round(v / 1e3, 1) for q, v in busy.items()}))
    side = collections.Counter()
    calls = collections.Counter()
    for r in seg:
        if r['Queue_Id'] != mainq:
            nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
            side[nm] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            calls[nm] += 1
    print('side queue(s), by kernel:')
    for nm, v in side.most_common():
        print('   %7.1f us  x%-2d %s' % (v / 1e3, calls[nm], nm))
    print('side queue(s), in order (queue, start, duration):')
    for r in seg:
        if r['Queue_Id'] != mainq:
            s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
            nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:58]
            print('  q%s %8.1f %7.1f  %s' % (r['Queue_Id'], s / 1e3, (e - s) / 1e3, nm))
    print('main queue, in order (start, duration, gap to the previous kernel):')
    prev = None
    for r in seg:
        if r['Queue_Id'] != mainq:
            continue
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:58]
        print('%8.1f %7.1f gap %6.1f  %s' % (s / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev is not None else 0.0, nm))
        prev = e


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)
