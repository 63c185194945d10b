#!/bin/bash
# Kernel trace of a few proofs on the GPU box, condensed to per-kernel totals and the timeline of one proof period.
# usage: tools/trace_proof.sh <out_file> [log_n]     (run through gpurun; writes under gpurun_out/)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-timeline.txt}
K=${2:-20}
cd /tmp
export TMPDIR=/tmp PYTHONPATH=$REPO
rm -rf /tmp/trace_proof
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_proof -- python3 $REPO/tools/dev_prove_loop.py $K 6 > /tmp/trace_proof.log 2>&1 || { tail -5 /tmp/trace_proof.log; exit 1; }
grep "^prove" /tmp/trace_proof.log
F=$(find /tmp/trace_proof -name "*kernel_trace.csv" | head -1)
python3 $REPO/tools/kernel_sums.py $F > $OUT
python3 $REPO/tools/timeline.py $F >> $OUT
echo "written $OUT"
