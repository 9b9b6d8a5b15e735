# round 6: the soaks of the final kernels (GPU box): oracle comparison over whole GMM days (lean streaming kernels), random networks,
# the bench workload at full size, pipelined halves against single launches, the fused rollout against the oracle's episode loop,
# the KKT certificate's long form
for site in caltech jpl; do timeout 600 python tests/soak/oracle_soak.py $site 16384 2 2>&1 | tail -1; done
for site in caltech jpl; do timeout 700 python tests/soak/oracle_soak.py $site 16384 2 1 2>&1 | tail -1; done      # autoreset over two days: the ALIVE copies
timeout 600 python tests/soak/network_fuzz.py 40 6 2>&1 | tail -2
timeout 600 python tests/soak/bench_workload_parity.py 2>&1 | tail -2
timeout 600 python tests/soak/pipeline_soak.py caltech 4 2>&1 | tail -1
for site in caltech jpl; do timeout 900 python tests/soak/rollout_soak.py $site 16384 2 2>&1 | tail -1; done
timeout 900 python tests/soak/kkt_soak.py 2>&1 | tail -4
