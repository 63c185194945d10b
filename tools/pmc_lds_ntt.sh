#!/bin/bash
# SQ counters of ntt_pass_kernel in a few passes (development aid).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-22}
pass() {
  name=$1; shift
  OUT=$REPO/gpurun_out/pmc_ntt/$name
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 "$REPO/tools/dev_ntt_bench.py" $K 2 2 > "$OUT/run.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/run.log"; return 1; }
  python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(collections.Counter)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'ntt_pass' not in r['Kernel_Name']: continue
        key=r['Kernel_Name'][:60]+' lds='+r.get('LDS_Block_Size','')
        agg[key][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[key][r['Counter_Name']]+=1
for k,v in agg.items():
    print(k, {a:int(b/max(1,cnt[k][a])) for a,b in v.items()}, 'launches', max(cnt[k].values()))
PY
}
pass a SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_CYCLES &&
pass b SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU &&
pass c SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
