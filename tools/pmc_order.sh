#!/bin/bash
# PMC comparison of the two packed-K orders of the ConvLSTM convolution (GPU box).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_order
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ko in 0 1; do
 i=0
 for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TA_BUSY_sum TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/k${ko}_p$i -- python3 $R/tools/bench_conv.py --only clstm.Conv2 --iters 3 --clips 13 --k-order $ko > $OUT/k${ko}_p$i.log 2>&1
 done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for ko in (0, 1):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob('gpurun_out/pmc_order/k%d_p*/**/*counter_collection.csv' % ko, recursive=True)):
        for r in csv.DictReader(open(f)):
            if 'conv_igemm' not in r['Kernel_Name']:
                continue
            agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for cn, v in agg.items():
        print('k_order %d  %-36s n=%d avg=%.4g' % (ko, cn, len(v), sum(v) / len(v)))
PY
