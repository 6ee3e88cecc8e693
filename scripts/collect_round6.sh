#!/bin/bash
# Copy what scripts/profile_round6.sh wrote on the GPU box (merged back into gpurun_out/r06/) into the tracked profiles/r06_* files.
set -eu
cd "$(dirname "$0")/.."
O=gpurun_out/r06
for w in ns b c d e ns_ens10 ns_share8; do [ -s $O/bench_$w.json ] && cp $O/bench_$w.json profiles/r06_bench_$w.json; done
for w in ns b d e; do [ -s $O/${w}_kernel_stats.csv ] && cp $O/${w}_kernel_stats.csv profiles/r06_${w}_kernel_stats.csv; done
for f in roofline.json pmc_traffic.json pmc_traffic_b.json pmc_fetch_size.csv pmc_write_size.csv chol_taskgraph.txt chol_soak.txt sweep_schedules_ab.log sweep_schedules_traffic.txt \
         loglik_rate.txt gpu_suite.log pmc_sq_summary.txt pmc_sq_v19.csv pmc_sq_v27.csv bench_ns_2ranks_gloo_shared_gpu.json bench_d_2ranks_gloo_shared_gpu.json bench_gpus2_rccl_on_one_gpu.log; do
  [ -s $O/$f ] && cp $O/$f profiles/r06_$f
done
ls -la profiles/r06_* | awk '{print $5, $9}'
