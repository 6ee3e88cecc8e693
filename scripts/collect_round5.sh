#!/bin/bash
# Copy what scripts/profile_round5.sh wrote on the GPU box (merged back into gpurun_out/r05/) into the tracked profiles/r05_* files.
set -eu
cd "$(dirname "$0")/.."
O=gpurun_out/r05
for w in ns ns2 b c d e ns_ens10 ns_share8; do [ -s $O/bench_$w.json ] && cp $O/bench_$w.json profiles/r05_bench_$w.json; done
for w in ns b d e; do [ -s $O/${w}_kernel_stats.csv ] && cp $O/${w}_kernel_stats.csv profiles/r05_${w}_kernel_stats.csv; done
for f in roofline.json pmc_traffic.json pmc_traffic_b.json pmc_fetch_size.csv pmc_write_size.csv chol_taskgraph.txt chol_soak.txt trtri_ahead_ab.txt sweep_ablation.log rff_kernels_ab.txt loglik_rate.txt pmc_sq_summary.txt \
         bench_ns_2ranks_gloo_shared_gpu.json bench_d_2ranks_gloo_shared_gpu.json bench_ns_sharded_2handles_shared_gpu.json bench_gpus2_rccl_on_one_gpu.log; do
  [ -s $O/$f ] && cp $O/$f profiles/r05_$f
done
[ -s $O/pmc_sq/r05_pmc_sq_v27.csv ] && cp $O/pmc_sq/r05_pmc_sq_v27.csv profiles/r05_pmc_sq.csv
[ -s $O/pmc_sq/r05_pmc_sq_v23.csv ] && cp $O/pmc_sq/r05_pmc_sq_v23.csv profiles/r05_pmc_sq_round4_kloop.csv
ls -la profiles/r05_* | awk '{print $5, $9}'
