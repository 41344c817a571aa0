#!/usr/bin/env python3
"""Per-launch timeline of ONE static-stage pass (equi -> cube -> ResNet-50 -> CAM) from a rocprofv3
kernel trace of `bench.py --static-only`: launch order, duration, gap, kernel, grid - the table the
ResNet-stage work in DESIGN.md section 3 is read from.

    (on the GPU box)  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv \\
        -d $R/gpurun_out/static_prof -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2
    python tools/static_timeline.py gpurun_out/static_prof [out.md]
"""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    f = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    starts = [i for i, r in enumerate(rows) if 'equi2cube' in r['Kernel_Name']]
    lo = starts[-1]
    hi = len(rows)
    t0 = int(rows[lo]['Start_Timestamp'])
    prev = t0
    lines = ['| # | start us | gap us | dur us | kernel | grid (threads) | LDS |', '|---|---|---|---|---|---|---|']
    tot = {}
    for k, r in enumerate(rows[lo:hi]):
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        lines.append('| %d | %.1f | %.1f | %.1f | `%s` | %s | %s |' % (k, (st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3,
                                                                   name[:60], r['Grid_Size_X'], r.get('LDS_Block_Size', '')))
        tot[name[:40]] = tot.get(name[:40], 0.0) + (en - st) / 1e3
        prev = en
    lines.append('')
    lines.append('pass: %.1f us from the first launch to the end of the last' % ((prev - t0) / 1e3))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        lines.append('  %8.1f us  %s' % (v, k))
    text = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text)
    print(text)


if __name__ == '__main__':
    main()
