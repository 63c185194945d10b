#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box: kernel-trace statistics, HBM-traffic
# counters (FETCH_SIZE and WRITE_SIZE in separate passes) and SQ issue counters, each pass a run of
# its own, the program directly after "--".  usage: tools/profile_round.sh <tag>   (e.g. r2)
# Output under gpurun_out/prof_<tag>/; condense with tools/pmc_summary.py and copy into profiles/.
set -e
TAG=${1:-r3}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp PYTHONPATH=$REPO
run() {   # run <subdir> <rocprofv3 options...> -- <program...>
    local dir=$1; shift
    echo "== $dir" | tee -a "$OUT/log.txt"
    timeout -s KILL 240 rocprofv3 "$@" >> "$OUT/log.txt" 2>&1
}
run stats_bench --kernel-trace --stats --output-format csv -d "$OUT/stats_bench" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-sweep
run stats_msm --kernel-trace --stats --output-format csv -d "$OUT/stats_msm" -- python3 "$REPO/tools/dev_msm_bench.py" 20 5
run stats_ntt --kernel-trace --stats --output-format csv -d "$OUT/stats_ntt" -- python3 "$REPO/tools/dev_ntt_bench.py" 22 20 2
run msm_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/msm_fetch" -- python3 "$REPO/tools/dev_msm_bench.py" 20 3
run msm_write --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/msm_write" -- python3 "$REPO/tools/dev_msm_bench.py" 20 3
run ntt_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/ntt_fetch" -- python3 "$REPO/tools/dev_ntt_bench.py" 22 5 2
run ntt_write --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/ntt_write" -- python3 "$REPO/tools/dev_ntt_bench.py" 22 5 2
run sq_msm --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_msm" -- python3 "$REPO/tools/dev_msm_bench.py" 20 4
run sq_ntt --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_ntt" -- python3 "$REPO/tools/dev_ntt_bench.py" 22 6 2
# keep what is merged back small: statistics and counter tables only
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
