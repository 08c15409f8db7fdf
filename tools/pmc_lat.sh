#!/bin/bash
# Memory latency / address translation PMC passes.  Usage (through gpurun, repo root): bash tools/pmc_lat.sh <tag> [bench args...]
set -o pipefail
tag=${1:-pmclat}; shift
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="$root/bench.py"
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-stats $*"
cd /tmp
pmc() {
  local dir=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" -d "$out/$dir" -o r -- python3 "$B" $ARGS > "$out/$dir.log" 2>&1 || { echo "PMC pass $dir failed"; tail -5 "$out/$dir.log"; exit 1; }
}
pmc pmc_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
pmc pmc_tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
cd "$root"
mkdir -p "$out/summary"
python3 - "$out" <<'PY'
import sys, glob, os, json
sys.path.insert(0, "tools")
from summarize_profiles import counters, first_db
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if os.path.isdir(d):
        db = first_db(d)
        if db:
            c = counters(db)
            res[os.path.basename(d)] = c[-1] if c else None
json.dump(res, open(os.path.join(out, "summary", "pmc_lat.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$out" -name "*.db" -delete
find "$out" -type d -empty -delete
