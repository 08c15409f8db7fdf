#!/bin/bash
# Collects the round's GPU evidence on the MI355X box (run through gpurun from the repo root):
#   tests, bench line, rocprofv3 kernel-trace stats of the same bench command, and PMC passes (each in its own run).
# Usage: bash tools/profile_round.sh <tag>      -> everything lands in gpurun_out/<tag>/
set -o pipefail
tag=${1:-prof}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/pytest_gpu.log" 2>&1 || { tail -20 "$out/pytest_gpu.log"; exit 1; }
timeout -k 10 300 python3 bench.py > "$out/bench.json.log" 2> "$out/bench.err" || { tail "$out/bench.err"; exit 1; }
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o r -- python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_under_rocprof.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$out/pmc_sq" -o r -- python3 "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-stats > "$out/pmc_sq.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d "$out/pmc_fetch" -o r -- python3 "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-stats > "$out/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d "$out/pmc_write" -o r -- python3 "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-stats > "$out/pmc_write.log" 2>&1 || exit 1
cd "$root"
tail -1 "$out/bench.json.log" | cut -c1-600
find "$out" -name "*.csv" | head -20
