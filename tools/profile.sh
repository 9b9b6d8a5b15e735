#!/bin/bash
# Usage (on the GPU box, from the repo root): [SITE=jpl] tools/profile.sh <tag>
# Kernel trace + the two PMC passes (separate runs, MI355X_MICROARCH.md §HBM) of bench.py, then the
# summaries under gpurun_out/<tag>_*; copy them into profiles/ to commit.
set -u
TAG=${1:-prof}
SITE=${SITE:-caltech}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
# the plain run FIRST: counter collection leaves the GPU in a lower, fixed clock state for a while (MI355X_MICROARCH.md, DVFS:
# "never compare a profiled arm with an un-profiled one") — a plain bench run after the PMC passes read 33 us per step on a
# box whose undisturbed figure was 26.5
python bench.py --site $SITE --full-out $OUT/bench_plain_full.json > $OUT/bench_plain.json 2> $OUT/bench_plain.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $REPO/bench.py --site $SITE --full-out $OUT/bench_traced_full.json --steps 288 --warmup 96 --no-cpu-baseline --no-secondary --no-single-launch --kernel-timing-steps 1 > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- \
    python $REPO/bench.py --site $SITE --full-out '' --steps 16 --warmup 8 --no-cpu-baseline --no-secondary --no-single-launch --kernel-timing-steps 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- \
    python $REPO/bench.py --site $SITE --full-out '' --steps 16 --warmup 8 --no-cpu-baseline --no-secondary --no-single-launch --kernel-timing-steps 1 > /dev/null 2> $OUT/pmc_write.err
cd $REPO
python tools/summarize_profile.py $OUT $REPO/gpurun_out/${TAG} $SITE
cat $OUT/bench_plain.json
