"""Per-launch timeline of ONE static-stage pass (equi -> cube -> ResNet-50 -> CAM) from a rocprofv3
kernel trace of `bench.py --static-only`: launch order, duration, gap, kernel, workgroups (and how many
of the 256 CUs they can cover), LDS - the table the ResNet-stage work in DESIGN.md section 3 is read from.

    (on the GPU box)  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv \\
        -d $R/gpurun_out/static_prof -- python3 $R/bench.py --static-only --no-cpu-baseline --no-secondary --steps 3 --warmup 2
    python3 tools/static_timeline.py gpurun_out/static_prof [out.md] [--flops GFLOP] [--peak TFLOP/s]

--flops: algorithmic GFLOP of the pass (50.25 per frame at cube 224) -> the stage's fraction of --peak.
"""
import csv
import glob
import os
import sys


def main():
    args = [a for a in sys.argv[1:]]
    flops = peak = None
    for key in ('--flops', '--peak'):
        if key in args:
            i = args.index(key)
            val = float(args[i + 1])
            del args[i:i + 2]
            if key == '--flops':
                flops = val
            else:
                peak = val
    src = args[0]
    f = max(glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getsize)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    starts = [i for i, r in enumerate(rows) if 'equi2cube' in r['Kernel_Name']]
    lo = starts[-1]
    hi = len(rows)
    for i in range(lo + 1, len(rows)):                     # the pass ends where torch's own kernels (bench.py's checks) begin
        nm = rows[i]['Kernel_Name']
        if nm.startswith('void at::') or nm.startswith('at::') or '__amd_rocclr' in nm or 'clock_probe' in nm:
            hi = i
            break
    t0 = int(rows[lo]['Start_Timestamp'])
    prev = t0
    lines = ['| # | start us | gap us | dur us | kernel | workgroups (x threads) | workgroups / 256 CUs | LDS |',
             '|---|---|---|---|---|---|---|---|']
    tot = {}
    busy = 0.0
    slowest = None
    for k, r in enumerate(rows[lo:hi]):
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        wg_size = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1)
        wgs = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) // max(wg_size, 1)
        lines.append('| %d | %.1f | %.1f | %.1f | `%s` | %d x %d | %.2f | %s |'
                     % (k, (st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3, name[:60], wgs, wg_size, wgs / 256.0,
                        r.get('LDS_Block_Size', '')))
        if slowest is None or (en - st) / 1e3 > slowest['us']:
            slowest = {'index': k, 'kernel': name[:60], 'us': round((en - st) / 1e3, 1), 'workgroups': wgs, 'threads': wg_size}
        tot[name[:40]] = tot.get(name[:40], 0.0) + (en - st) / 1e3
        busy += (en - st) / 1e3
        prev = en
    span = (prev - t0) / 1e3
    lines.append('')
    lines.append('pass: %.1f us from the first launch to the end of the last (%d launches, %.1f us inside kernels, %.1f us of gaps)'
                 % (span, hi - lo, busy, span - busy))
    if flops:
        ach_t = flops / (span * 1e-6) / 1e3
        lines.append('algorithmic %.2f GFLOP per pass -> %.1f TFLOP/s%s'
                     % (flops, ach_t, (' = %.3f of the %.1f TFLOP/s MFMA peak' % (ach_t / peak, peak)) if peak else ''))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        lines.append('  %8.1f us  %s' % (v, k))
    text = '\n'.join(lines) + '\n'
    if len(args) > 1:
        open(args[1], 'w').write(text)
        import json
        json.dump({'source': os.path.basename(args[1]), 'launches': hi - lo, 'pass_us': round(span, 1), 'kernel_us': round(busy, 1),
                   'slowest_launch': slowest}, open(os.path.splitext(args[1])[0] + '.json', 'w'))
    print(text)


if __name__ == '__main__':
    main()
