set -u
T=r2e
tools/profile.sh $T > gpurun_out/${T}_profile.log 2>&1
O=gpurun_out
python bench.py --steps 20 --warmup 5 > $O/${T}_bench_driver.json 2>/dev/null
python bench.py --phase sync --no-secondary --no-cpu-baseline > $O/${T}_bench_sync.json 2>/dev/null
python bench.py --site jpl --no-secondary --no-cpu-baseline > $O/${T}_bench_jpl.json 2>/dev/null
python bench.py --no-project --no-secondary --no-cpu-baseline > $O/${T}_bench_noproject.json 2>/dev/null
python bench.py --battery stepwise --no-secondary --no-cpu-baseline > $O/${T}_bench_stepwise.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --single-device --envs-per-gpu 32768 --no-secondary --no-cpu-baseline > $O/${T}_bench_2rank.json 2>/dev/null
ls -la $O/${T}*
