#!/bin/bash
# PC sampling of the streaming kernel (beta feature of rocprofv3; bounded by timeouts).  tools/scratch/pcsample.sh [method]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pcsample; mkdir -p $OUT
METHOD=${1:-host_trap}
cd /tmp && export TMPDIR=/tmp
if [ "$METHOD" = stochastic ]; then UNIT=cycles; INT=1048576; else UNIT=time; INT=1; fi
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INT \
   --output-format csv -d $OUT/$METHOD -o pcs -- python $REPO/bench.py --steps 2000 --warmup 300 --no-cpu-baseline --no-secondary --kernel-timing-steps 2 > $OUT/$METHOD.json 2> $OUT/$METHOD.err
echo "rc=$?"
tail -3 $OUT/$METHOD.err | cut -c1-300
find $OUT/$METHOD -type f | head; 
