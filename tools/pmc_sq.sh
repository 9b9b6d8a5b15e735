#!/bin/bash
# SQ counters of the streaming kernel (two passes), per env-step.  Usage: tools/pmc_sq.sh [layout]
LAY=${1:-compact}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_sq_$LAY; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
EVC_LAYOUT=$LAY rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o bench -- python $REPO/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-secondary --pipeline 1 --kernel-timing-steps 1 > /dev/null 2> $OUT/a.err
EVC_LAYOUT=$LAY rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/b -o bench -- python $REPO/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-secondary --pipeline 1 --kernel-timing-steps 1 > /dev/null 2> $OUT/b.err
cd $REPO
python - $LAY <<'PY'
import pandas as pd, glob, sys, json
lay = sys.argv[1]
res = {}
for part in 'ab':
    f = glob.glob(f'gpurun_out/pmc_sq_{lay}/{part}/**/bench_counter_collection.csv', recursive=True)
    if not f: continue
    df = pd.read_csv(f[0])
    df = df[df['Kernel_Name'].str.contains('step_kernel_c?quad<true, ., false, true')]
    res.update((df.groupby('Counter_Name')['Counter_Value'].mean() / 65536).round(2).to_dict())
    res['VGPR'] = int(df['VGPR_Count'].iloc[0]); res['scratch'] = int(df['Scratch_Size'].iloc[0]); res['LDS'] = int(df['LDS_Block_Size'].iloc[0])
print(lay, json.dumps(res))
PY
