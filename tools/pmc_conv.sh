#!/bin/bash
# PMC counters for the ConvLSTM conv micro-benchmark (GPU box).  One counter set per pass.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_conv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_conv.py --only clstm.Conv2 --iters 3 "$@" > $OUT/p$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_conv/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' not in r['Kernel_Name']:
            continue
        k = (r['Kernel_Name'][:40], r['Counter_Name'])
        agg.setdefault(k, []).append(float(r['Counter_Value']))
for (kn, cn), v in agg.items():
    print('%-42s %-32s n=%d avg=%.4g' % (kn, cn, len(v), sum(v) / len(v)))
PY
