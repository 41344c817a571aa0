#!/bin/bash
# PMC counters of conv_small_f32_kernel on layer3's conv2 of ONE frame (f32, 72 K steps, 76 workgroups; GPU box).
# One counter set per pass (--pmc with --kernel-trace only).   tools/pmc_small.sh [out_txt]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_small
TXT=${1:-$R/gpurun_out/pmc_small.txt}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_conv.py --precision fp32 --frames 1 --iters 5 --tile-px 6464 --splits 1 --only "l3.conv2" > $OUT/p$i.log 2>&1
done
cd $R
python3 - "$TXT" <<'PY'
import csv, glob, collections, sys, os
agg = collections.OrderedDict()
dur = []
for d in sorted(glob.glob('gpurun_out/pmc_small/p*')):
    if not os.path.isdir(d): continue
    fs = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not fs: continue
    f = max(fs, key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if 'conv_small' not in r['Kernel_Name']: continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
fs = glob.glob('gpurun_out/pmc_small/p1/**/*kernel_trace.csv', recursive=True)
if fs:
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if 'conv_small' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
lines = ['conv_small_f32_kernel, layer3 conv2 of one frame (f32, M = 1176, N = 256, K = 2304: 72 K steps, 76 workgroups x 4 waves), per-launch averages']
for k, v in agg.items():
    lines.append('%-36s n=%-3d avg=%.5g' % (k, len(v), sum(v) / len(v)))
a = {k: sum(v) / len(v) for k, v in agg.items()}
g = lambda k: a.get(k, float('nan'))
if dur:
    us = sum(dur) / len(dur)
    lines.append('launch %.1f us under the profiler' % us)
wc = g('SQ_WAVE_CYCLES')
lines.append('waves parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)          %.3f' % (g('SQ_WAIT_ANY') / wc))
lines.append('waves stalled on issue (SQ_WAIT_INST_ANY / ...)      %.3f' % (g('SQ_WAIT_INST_ANY') / wc))
lines.append('waves issuing (SQ_ACTIVE_INST_ANY / ...)             %.3f' % (g('SQ_ACTIVE_INST_ANY') / wc))
lines.append('MFMA busy cycles per active SIMD (304 SIMDs)         %.0f  (72 steps x 32 MFMAs x 32 cycles = 73728)' % (g('SQ_VALU_MFMA_BUSY_CYCLES') / 304.0))
lines.append('wave cycles per wave (x4: counter ticks per 4 cycles) %.0f' % (4.0 * wc / 304.0))
open(sys.argv[1], 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
