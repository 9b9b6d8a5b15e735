#!/bin/bash
# per-kernel SQ counters for compact vs dense, project on
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_cmp; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for lay in compact dense; do
  EVC_LAYOUT=$lay rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/$lay -o bench -- python $REPO/bench.py --steps 16 --warmup 150 --no-cpu-baseline --kernel-timing-steps 1 > /dev/null 2> $OUT/$lay.err
done
cd $REPO
python - <<'PY'
import pandas as pd, glob
for lay in ('compact','dense'):
    f = glob.glob(f'gpurun_out/pmc_cmp/{lay}/**/bench_counter_collection.csv', recursive=True)
    df = pd.read_csv(f[0])
    df = df[df['Kernel_Name'].str.contains('step_kernel_c?quad')]
    print(lay, df.groupby('Counter_Name')['Counter_Value'].mean().round(0).to_dict(), 'VGPR', df['VGPR_Count'].iloc[0], 'scratch', df['Scratch_Size'].iloc[0])
PY
