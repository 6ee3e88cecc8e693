#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python command, bounded.  usage: trace_cmd.sh <tag> <script.py> args...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${GPX_ROUND:-r02}
mkdir -p $O
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$TAG -o $TAG -- python $SCRIPT "$@" > $O/trace_$TAG.log 2>&1 < /dev/null
echo "rocprofv3 rc=$?"; tail -3 $O/trace_$TAG.log
