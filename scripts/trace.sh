#!/bin/bash
# rocprofv3 kernel-trace summary of one bench.py invocation, bounded.  Usage (via gpurun):
#   bash scripts/trace.sh <tag> <bench.py args...>      -> gpurun_out/<round>/trace_<tag>/..., prints the top kernels
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${GPX_ROUND:-r02}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$TAG -o $TAG -- python $R/bench.py "$@" --no-cpu-baseline > $O/trace_$TAG.log 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
f=$(find $O/trace_$TAG -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f $O/${TAG}_kernel_stats.csv; head -${TOPN:-16} $f | cut -c1-170; else echo "no kernel_stats.csv"; tail -5 $O/trace_$TAG.log; fi
