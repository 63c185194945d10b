#!/bin/bash
# Kernel trace of one rank of a sharded proof alone on the GPU (loopback communicator), condensed like trace_proof.sh.
# usage: tools/trace_shard.sh <out_file> <log_n> <world> <grouped>     (run through gpurun; writes under gpurun_out/)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-shard_timeline.txt}
cd /tmp
export TMPDIR=/tmp PYTHONPATH=$REPO
rm -rf /tmp/trace_shard
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_shard -- python3 $REPO/tools/dev_shard_timing.py ${2:-20} ${3:-8} 6 0 ${4:-1} > /tmp/trace_shard.log 2>&1 || { tail -5 /tmp/trace_shard.log; exit 1; }
tail -1 /tmp/trace_shard.log
F=$(find /tmp/trace_shard -name "*kernel_trace.csv" | head -1)
python3 $REPO/tools/kernel_sums.py $F > $OUT
python3 $REPO/tools/timeline.py $F >> $OUT
echo "written $OUT"
