#!/bin/bash
# PMC summary of the FINAL ConvLSTM kernel (conv_clip_kernel, K = 36000, M = 1176: the dominant kernel of bench.py)
# on the GPU box: MFMA busy, wait cycles, LDS bank conflicts, L2 hit rate, HBM fetch / write.  Counter passes are
# separate rocprofv3 runs (--pmc with --kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass).
#   tools/pmc_clip_final.sh [out_txt]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_clip_final
TXT=${1:-$R/gpurun_out/pmc_clip_final.txt}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_conv.py --only clstm.Conv2 --iters 3 --clips 4 > $OUT/p$i.log 2>&1
done
cd $R
python3 - "$TXT" <<'PY'
import csv, glob, collections, sys
agg = collections.OrderedDict()
dur = []
for f in sorted(glob.glob('gpurun_out/pmc_clip_final/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'conv_clip' not in r['Kernel_Name']:
            continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for f in sorted(glob.glob('gpurun_out/pmc_clip_final/p1/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'conv_clip' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
a = {k: sum(v) / len(v) for k, v in agg.items()}
lines = ['conv_clip_kernel<bf16, clip tile>: ConvLSTM Conv2 / Gates, M = 1176 (4 clips), N = 4000, K = 36000, 4 K-splits, 256 workgroups',
         'rocprofv3 --pmc passes over tools/bench_conv.py --only clstm.Conv2 --clips 4 (per-launch averages; chip-wide sums)', '']
for k, v in a.items():
    lines.append('%-36s n=%-3d avg=%.5g' % (k, len(agg[k]), v))
lines.append('')
g = lambda k: a.get(k, float('nan'))
if dur:
    lines.append('launch duration under the profiler: %.1f us (n=%d)' % (sum(dur) / len(dur), len(dur)))
lines.append('wave cycles parked in s_waitcnt / barrier (SQ_WAIT_ANY / SQ_WAVE_CYCLES):       %.3f' % (g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES')))
lines.append('wave cycles stalled on issue (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES):               %.3f' % (g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES')))
lines.append('wave cycles issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES):                      %.3f' % (g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')))
lines.append('MFMA pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / 4 SIMDs... raw ratio): %.3f' % (g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES')))
lines.append('LDS bank-conflict cycles / LDS active cycles:                                     %.3f' % (g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')))
lines.append('L2 hit rate TCC_HIT / (TCC_HIT + TCC_MISS):                                       %.3f' % (g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum'))))
tr = (2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024
lines.append('HBM traffic per launch (2 * FETCH_SIZE + WRITE_SIZE) KiB:                          %.1f MB (fetch %.1f MB, write %.1f MB)' % (tr / 1e6, 2 * g('FETCH_SIZE') * 1024 / 1e6, g('WRITE_SIZE') * 1024 / 1e6))
lines.append('algorithmic: packed weights 295 MB + activations 2 x 9.4 MB; the 4 f32 split-K slabs add 75 MB of writes')
open(sys.argv[1], 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
