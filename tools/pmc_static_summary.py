"""Per-kernel table of the counter passes tools/pmc_static.sh collected over bench.py --static-only (per-launch averages, chip-wide sums).
    python3 tools/pmc_static_summary.py gpurun_out/pmc_static out.txt"""
import collections, csv, glob, os, sys


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:58]


def main():
    src, txt = sys.argv[1], sys.argv[2]
    agg = collections.OrderedDict()                     # kernel -> counter -> [values]
    dur = collections.OrderedDict()
    for d in sorted(glob.glob(os.path.join(src, 'p*'))):
        if not os.path.isdir(d):
            continue
        f = newest(os.path.join(d, '**', '*counter_collection.csv'))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            agg.setdefault(short(r['Kernel_Name']), collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    f = newest(os.path.join(src, 'p1', '**', '*kernel_trace.csv'))
    for r in csv.DictReader(open(f)):
        dur.setdefault(short(r['Kernel_Name']), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    div = lambda x, y: x / y if y else float('nan')
    lines = ['static stage, 64 frames: rocprofv3 --pmc passes per kernel (per-launch averages)', '',
             '%-58s %5s %8s %5s %6s %6s %6s %6s %6s %6s %6s %8s' % ('kernel', 'n', 'us', 'GHz', 'mfma', 'park', 'stall', 'issue', 'valu', 'ldsbc', 'l2hit', 'HBM MB')]
    for k, c in agg.items():
        a = {n: sum(v) / len(v) for n, v in c.items()}
        g = lambda n: a.get(n, float('nan'))
        us = sum(dur.get(k, [0])) / max(1, len(dur.get(k, [])))
        if us < 3.0 or 'at::' in k or 'rocclr' in k:
            continue
        kcyc = g('SQ_BUSY_CYCLES') / 32.0
        wc = g('SQ_WAVE_CYCLES')
        lines.append('%-58s %5d %8.1f %5.2f %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f %8.1f' % (
            k, len(dur.get(k, [])), us, kcyc / us / 1e3, g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0 / kcyc, g('SQ_WAIT_ANY') / wc,
            g('SQ_WAIT_INST_ANY') / wc, g('SQ_ACTIVE_INST_ANY') / wc, g('SQ_ACTIVE_INST_VALU') / 4.0 / 256.0 / kcyc if 'SQ_ACTIVE_INST_VALU' in a else float('nan'),
            div(g('SQ_LDS_BANK_CONFLICT'), g('SQ_LDS_IDX_ACTIVE')), div(g('TCC_HIT_sum'), g('TCC_HIT_sum') + g('TCC_MISS_sum')),
            (2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / 1e6))
    lines += ['', 'mfma = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / kernel cycles (matrix pipe busy); park / stall / issue = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY',
              'over SQ_WAVE_CYCLES; valu = SQ_ACTIVE_INST_VALU / (4 x 256 SIMD-ish normalisation) / kernel cycles (relative); ldsbc = LDS bank-conflict cycles / LDS active cycles;',
              'GHz = SQ_BUSY_CYCLES / 32 / launch time; HBM MB = (2 FETCH_SIZE + WRITE_SIZE) KiB per launch.  Names shared by several launch shapes are averaged together.']
    open(txt, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
