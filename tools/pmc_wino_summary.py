"""Per-launch averages of the counter passes tools/pmc_wino.sh collected for one kernel (name substring = argv[3]) and the
ratios DESIGN.md quotes (units as in tools/pmc_clip_summary.py).

    python3 tools/pmc_wino_summary.py gpurun_out/pmc_wino profiles/r05_pmc_wino.txt wino_gemm [profiles/pmc_clock_bf16_b4_w7.json]
"""
import collections, csv, glob, os, sys


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def main():
    src, txt, name = sys.argv[1], sys.argv[2], sys.argv[3]
    agg, dur = collections.OrderedDict(), []
    for d in sorted(glob.glob(os.path.join(src, 'p*'))):
        f = newest(os.path.join(d, '**', '*counter_collection.csv')) if os.path.isdir(d) else None
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if name in r['Kernel_Name']:
                agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    f = newest(os.path.join(src, 'p1', '**', '*kernel_trace.csv'))
    if f:
        for r in csv.DictReader(open(f)):
            if name in r['Kernel_Name']:
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    g = lambda k: a.get(k, float('nan'))
    lines = ['kernel *%s*: rocprofv3 --pmc passes (per-launch averages; chip-wide sums)' % name, '']
    for k, v in a.items():
        lines.append('%-36s n=%-3d avg=%.5g' % (k, len(agg[k]), v))
    us = sum(dur) / len(dur) if dur else float('nan')
    kcyc = g('SQ_BUSY_CYCLES') / 32.0
    mfma = g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0
    lines += ['', 'launch duration under the profiler: %.1f us (n=%d)' % (us, len(dur)),
              'kernel cycles (SQ_BUSY_CYCLES / 32): %.0f -> clock held: %.2f GHz' % (kcyc, kcyc / us / 1e3),
              'MFMA pipe busy: %.3f (%.0f busy cycles per SIMD)' % (mfma / kcyc, mfma),
              'waves parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES):        %.3f' % (g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES')),
              'waves issue-stalled (SQ_WAIT_INST_ANY / ...):       %.3f' % (g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES')),
              'waves issuing (SQ_ACTIVE_INST_ANY / ...):           %.3f' % (g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')),
              'LDS bank conflicts / LDS active cycles:             %.3f' % (g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')),
              'LDS active / kernel cycles (per CU):                %.3f' % (g('SQ_LDS_IDX_ACTIVE') / 256.0 / kcyc),
              'L2 hit rate TCC_HIT / (TCC_HIT + TCC_MISS):         %.3f (%.3g requests)' % (g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')), g('TCC_REQ_sum')),
              'HBM traffic per launch (2 * FETCH_SIZE + WRITE_SIZE) KiB: %.1f MB (fetch %.1f MB, write %.1f MB)'
              % ((2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / 1e6, 2 * g('FETCH_SIZE') * 1024 / 1e6, g('WRITE_SIZE') * 1024 / 1e6)]
    open(txt, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    if len(sys.argv) > 4:      # the clock / pipe-busy pair bench.py quotes next to its register-only clock probe
        import json
        json.dump({'kernel': name, 'clock_ghz': round(kcyc / us / 1e3, 3), 'mfma_pipe_busy': round(mfma / kcyc, 3), 'launch_us': round(us, 1),
                   'source': os.path.basename(txt)}, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
