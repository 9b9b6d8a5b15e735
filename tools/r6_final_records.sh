# round 6: every record of the final build in one GPU call (one box): GPU suite, profile.sh for both sites, driver window,
# SQ counters, soaks.  Outputs under gpurun_out/; copy what is to be judged into profiles/.
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r6x_gputest_tail.txt
tools/profile.sh r6x > gpurun_out/r6x_profile.log 2>&1
SITE=jpl tools/profile.sh r6x_jpl > gpurun_out/r6x_jpl_profile.log 2>&1
python bench.py --steps 20 --warmup 5 --full-out gpurun_out/r6x_bench_driver_full.json > gpurun_out/r6x_bench_driver.json 2> gpurun_out/r6x_bench_driver.err
tools/pmc_sq.sh > gpurun_out/r6_pmc_sq_step_kernel.txt 2>&1
bash tools/r6_soaks.sh > gpurun_out/r6x_soaks.txt 2>&1
