mkdir -p gpurun_out/r6c
for rep in 1 2; do
for cap in 768 384 512 640 1024 1536; do
  EVC_GRID_CAP=$cap python bench.py --no-secondary --no-cpu-baseline --full-out gpurun_out/r6c/b_$cap.json 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('cap $cap', r['ms_per_step'], ro['step_period_ms'], ro['single_launch']['ms_per_step'], ro['single_launch']['avg_kernel_ms'])"
done; done > gpurun_out/r6c/grid_sweep.txt 2>&1
cat gpurun_out/r6c/grid_sweep.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r6c/gputest_tail.txt; cat gpurun_out/r6c/gputest_tail.txt
