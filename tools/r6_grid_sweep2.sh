mkdir -p gpurun_out/r6d
for rep in 1 2 3; do
for cap in 768 688 704 736 696; do
  EVC_GRID_CAP=$cap python bench.py --no-secondary --no-cpu-baseline --full-out gpurun_out/r6d/b_$cap.json 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('cap $cap', r['ms_per_step'], ro['step_period_ms'], ro['single_launch']['ms_per_step'], ro['single_launch']['avg_kernel_ms'])"
done; done > gpurun_out/r6d/grid_sweep2.txt 2>&1
cat gpurun_out/r6d/grid_sweep2.txt
