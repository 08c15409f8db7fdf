#!/bin/bash
# Extra PMC passes on the memory path of the default bench workload (each pass in its own run): instruction mix, TA / TD /
# TCP stalls, LDS.  Usage (through gpurun, repo root): bash tools/pmc_mem.sh <tag> [bench args...]
set -o pipefail
tag=${1:-pmcmem}; shift
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="$root/bench.py"
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-stats $*"
cd /tmp
pmc() {
  local dir=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$out/$dir" -o r -- python3 "$B" $ARGS > "$out/$dir.log" 2>&1 || { echo "PMC pass $dir failed"; tail -5 "$out/$dir.log"; exit 1; }
}
pmc pmc_mix SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pmc pmc_stall SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pmc pmc_tcp TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_SPI_STALL_sum TD_COALESCABLE_WAVEFRONT_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
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
json.dump(res, open(os.path.join(out, "summary", "pmc_mem.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$out" -name "*.db" -delete
find "$out" -type d -empty -delete
