#!/bin/bash
# PMC comparison: clip-resident ConvLSTM kernel vs the generic 256x304 ring kernel (GPU box).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_clip
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CL=${CLIPS:-13}
for cr in 0 1; do
 i=0
 for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/c${cr}_p$i -- python3 $R/tools/bench_conv.py --only clstm.Conv2 --iters 3 --clips $CL --clip-resident $cr > $OUT/c${cr}_p$i.log 2>&1
 done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for cr in (0, 1):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob('gpurun_out/pmc_clip/c%d_p*/**/*counter_collection.csv' % cr, recursive=True)):
        for r in csv.DictReader(open(f)):
            if 'conv_' not in r['Kernel_Name'] or 'finish' in r['Kernel_Name']:
                continue
            agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for cn, v in agg.items():
        print('clip_resident %d  %-36s n=%d avg=%.4g' % (cr, cn, len(v), sum(v) / len(v)))
PY
