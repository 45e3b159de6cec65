#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool: tools/profile_cmd.sh <tag> <script> [args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
S=$GRAFT_REPO_ROOT/$1; shift
rocprofv3 --kernel-trace -d $OUT -o trace -- python3 $S "$@" > $OUT/run.log 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB grid > $OUT/stats.txt
