"""Summary of the counter passes tools/pmc_clip_final.sh collected for the dominant kernel (conv_clip_kernel on the
ConvLSTM Conv2 / Gates shape): per-launch averages and the derived ratios DESIGN.md quotes.

    python tools/pmc_clip_summary.py gpurun_out/pmc_clip_final profiles/r02c_pmc_clip.txt

Units (checked against the launch duration): SQ_BUSY_CYCLES sums the 32 shader engines (8 XCDs x 4), so
SQ_BUSY_CYCLES / 32 = kernel cycles and kernel cycles / duration = the clock the chip held; SQ_VALU_MFMA_BUSY_CYCLES
sums the 1024 SIMDs; SQ_WAVE_CYCLES and the SQ_WAIT_* / SQ_ACTIVE_* counters tick once per 4 cycles per wave.
Only the newest CSV of each pass directory is read (the directory may hold older runs).
"""
import collections
import csv
import glob
import os
import sys


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def main():
    src, txt = sys.argv[1], sys.argv[2]
    agg = collections.OrderedDict()
    dur = []
    for d in sorted(glob.glob(os.path.join(src, 'p*'))):
        if not os.path.isdir(d):
            continue
        f = newest(os.path.join(d, '**', '*counter_collection.csv'))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if 'conv_clip' in r['Kernel_Name']:
                agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    f = newest(os.path.join(src, 'p1', '**', '*kernel_trace.csv'))
    if f:
        for r in csv.DictReader(open(f)):
            if 'conv_clip' in r['Kernel_Name']:
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    # shape of the launch (tools/pmc_clip_final.sh SHAPE=...): useful rows, padded rows, slabs, label
    shape = 'w8.Conv2' if 'w8.Conv2' in src else 'clstm.Conv2'
    M, MP, SL, what = {'clstm.Conv2': (1176, 1216, 4, 'clip tile>: ConvLSTM Conv2 / Gates, M = 1176 (4 clips of 7x7 faces), N = 4000, K = 36000, 4 K-splits, 256 workgroups'),
                       'w8.Conv2': (1536, 1536, 2, 'HALF tile>: ConvLSTM Conv2 / Gates at 8x8 faces (cube 256), M = 1536 (4 clips, 8 half-cube tiles), N = 4000, K = 36000, 2 K-splits, 256 workgroups')}[shape]
    lines = ['conv_clip_kernel<bf16, ' + what,
             'rocprofv3 --pmc passes over tools/bench_conv.py --only %s --clips 4 (per-launch averages; chip-wide sums)' % shape, '']
    for k, v in a.items():
        lines.append('%-36s n=%-3d avg=%.5g' % (k, len(agg[k]), v))
    lines.append('')
    g = lambda k: a.get(k, float('nan'))
    us = sum(dur) / len(dur) if dur else float('nan')
    kcyc = g('SQ_BUSY_CYCLES') / 32.0
    mfma = g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0
    ideal = 2.0 * MP * 4096 * 36000 / (1024 * 1024.0)       # MFMA cycles per SIMD: padded tile flops / (1024 flop/clk/SIMD x 1024 SIMDs)
    lines.append('launch duration under the profiler: %.1f us (n=%d)' % (us, len(dur)))
    lines.append('kernel cycles (SQ_BUSY_CYCLES / 32 shader engines): %.0f -> clock held during the launch: %.2f GHz' % (kcyc, kcyc / us / 1e3))
    lines.append('MFMA pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel cycles):                 %.3f  (%.0f busy cycles per SIMD; '
                 '%.0f = the MFMAs of the padded %d x 4096 tile at 16 cycles each)' % (mfma / kcyc, mfma, ideal, MP))
    lines.append('  => fraction of the 2.5 PFLOP/s peak = MFMA busy x clock / 2.4 GHz x useful / padded flops = %.3f'
                 % (mfma / kcyc * (kcyc / us / 1e3) / 2.4 * (M * 4000.0) / (MP * 4096.0)))
    lines.append('wave cycles parked in s_waitcnt / barrier (SQ_WAIT_ANY / SQ_WAVE_CYCLES):       %.3f' % (g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES')))
    lines.append('wave cycles stalled on issue (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES):               %.3f' % (g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES')))
    lines.append('wave cycles issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES):                      %.3f' % (g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')))
    lines.append('LDS bank-conflict cycles / LDS active cycles:                                   %.3f' % (g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')))
    lines.append('LDS active / kernel cycles (SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles):       %.3f' % (g('SQ_LDS_IDX_ACTIVE') / 256.0 / kcyc))
    lines.append('L2 hit rate TCC_HIT / (TCC_HIT + TCC_MISS):                                     %.3f' % (g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum'))))
    tr = (2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024
    lines.append('HBM traffic per launch (2 * FETCH_SIZE + WRITE_SIZE) KiB:                        %.1f MB (fetch %.1f MB, write %.1f MB)'
                 % (tr / 1e6, 2 * g('FETCH_SIZE') * 1024 / 1e6, g('WRITE_SIZE') * 1024 / 1e6))
    lines.append('algorithmic: packed weights 295 MB + activations 2 x %.1f MB; the %d f32 split-K slabs add %.0f MB of writes' % (M * 4000 * 2 / 1e6, SL, SL * M * 4000 * 4 / 1e6))
    open(txt, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
