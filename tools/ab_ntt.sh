#!/bin/bash
# A/B of two builds of the library on the GPU box: per-sweep kernel times of a coset transform.
# usage: tools/ab_ntt.sh <log_n> <sweeps>   (the other build is .ab/libplonk_base.so)
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
K=${1:-22}; P=${2:-3}
cd /tmp
export TMPDIR=/tmp PYTHONPATH=$REPO
one() {
    rm -rf /tmp/ab_$1
    timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_$1 -- python3 $REPO/tools/dev_ntt_bench.py $K 20 2 | grep "^ntt"
    python3 $REPO/tools/ntt_sweep_times.py $(find /tmp/ab_$1 -name "*kernel_trace.csv" | head -1) $P
}
echo "== new"; one new
if [ -f $REPO/.ab/libplonk_base.so ]; then
    cp $REPO/plonk_amd/libplonk_hip.so /tmp/new.so
    cp $REPO/.ab/libplonk_base.so $REPO/plonk_amd/libplonk_hip.so
    echo "== base"; one base
    cp /tmp/new.so $REPO/plonk_amd/libplonk_hip.so
fi
