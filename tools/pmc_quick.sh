#!/bin/bash
# Quick per-kernel PMC look at the config-5 launch (run on the GPU box through gpurun): bash tools/pmc_quick.sh <tag> [perf_matrix args]
# One --pmc pass per counter group, each in its own run; prints the counters of the last dispatch of every kernel.
tag=${1:-pmcq}; shift
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
run() { local dir=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" -d "$out/$dir" -o r -- python3 "$root/tools/perf_matrix.py" --cases c5 --spp 64 --launches 2 $EXTRA > "$out/$dir.log" 2>&1 || { echo "pass $dir failed"; tail -5 "$out/$dir.log"; exit 1; }; }
run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
cd "$root"
python3 - "$out" <<'PY'
import sys, glob, os, sqlite3, collections, json
out = sys.argv[1]
for d in ("sq", "sq2"):
    dbs = sorted(glob.glob(os.path.join(out, d, "**", "*.db"), recursive=True))
    if not dbs: continue
    con = sqlite3.connect(dbs[0])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for name, disp, cname, val in con.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        agg[(name.split("(")[0][-60:], disp)][cname] += val
    last = {}
    for (name, disp), c in sorted(agg.items(), key=lambda kv: kv[0][1]):
        last[name] = c
    for name, c in last.items():
        if "render_kernel" in name or "first_kernel" in name:
            print(d, name, json.dumps({k: v for k, v in c.items()}))
PY
find "$out" -name "*.db" -delete
