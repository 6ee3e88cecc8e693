#!/bin/bash
# one rocprofv3 --pmc pass (kernel-trace only) over an arbitrary python command, bounded.
# usage: pmc_cmd.sh <tag> "<counters>" <script.py> args...
set -u
TAG=$1; shift
CTR=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${GPX_ROUND:-r02}
mkdir -p $O
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $O/pmc_$TAG -o $TAG -- python $SCRIPT "$@" > $O/pmc_$TAG.log 2>&1 < /dev/null
echo "rocprofv3 rc=$?"; tail -2 $O/pmc_$TAG.log; ls $O/pmc_$TAG
