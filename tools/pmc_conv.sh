#!/bin/bash
# MFMA / LDS counters of the convolution kernels (dev tool; separate --pmc passes, kernel-trace only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pc
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc -- python $ROOT/bench.py --no-cpu-baseline --no-pipeline --steps 2 --warmup 1 > /tmp/pc.log 2>&1
  f=$(find /tmp/pc -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'k_conv5_stream' in n or 'k_conv1_mfma' in n or 'k_fc1_split' in n:
        key = (n.split('(')[0].replace('void trexhip::', '')[:40], r['Counter_Name'])
        acc[key].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for k, v in sorted(acc.items()):
    print(k[0].ljust(42), k[1].ljust(34), '%.4g' % (sum(a for a, _ in v) / len(v)), 'dur_us %.1f' % (sum(b for _, b in v) / len(v) / 1e3))
PY
done
