#!/bin/bash
# Copy what scripts/profile_round4.sh wrote on the GPU box (merged back into gpurun_out/r04/) into the tracked profiles/r04_*
# files.  Run in the build container after the gpurun call has returned:  bash scripts/collect_round4.sh
set -eu
cd "$(dirname "$0")/.."
O=gpurun_out/r04
for w in ns b c d e; do cp $O/bench_$w.json profiles/r04_bench_$w.json; done
for w in ns d e; do cp $O/${w}_kernel_stats.csv profiles/r04_${w}_kernel_stats.csv; done
cp $O/roofline.json profiles/r04_roofline.json
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json
cp $O/pmc_fetch_size.csv profiles/r04_pmc_fetch_size.csv
cp $O/pmc_write_size.csv profiles/r04_pmc_write_size.csv
cp $O/chol_taskgraph.txt profiles/r04_chol_taskgraph.txt
cp $O/grad_forms.txt profiles/r04_grad_forms.txt
cp $O/plugin_phases.txt profiles/r04_plugin_phases.txt
cp $O/bench_ns_2ranks_gloo_shared_gpu.json profiles/r04_bench_ns_2ranks_gloo_shared_gpu.json
cp $O/bench_d_2ranks_gloo_shared_gpu.json profiles/r04_bench_d_2ranks_gloo_shared_gpu.json
cp $O/bench_ns_sharded_2handles_shared_gpu.json profiles/r04_bench_ns_sharded_2handles_shared_gpu.json
cp $O/bench_gpus2_rccl_on_one_gpu.log profiles/r04_bench_gpus2_rccl_on_one_gpu.log
ls -la profiles/r04_* | awk '{print $5, $9}'
