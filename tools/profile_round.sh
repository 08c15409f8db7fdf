#!/bin/bash
# Collects the round's GPU evidence on the MI355X box (run through gpurun from the repo root):
#   tests, bench line, rocprofv3 kernel-trace stats of the same bench command, PMC passes (each in its own run), the
#   config-1/config-2 baselines, and the same for the 10 M-triangle synthetic workload.
# Usage: bash tools/profile_round.sh <tag> [quick]     -> everything lands in gpurun_out/<tag>/
#   then here: python tools/summarize_profiles.py gpurun_out/<tag> profiles/<prefix>
set -o pipefail
tag=${1:-prof}
quick=${2:-}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="$root/bench.py"
if [ -z "$quick" ]; then
timeout -k 10 800 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$out/pytest_gpu.log" 2>&1 || { tail -20 "$out/pytest_gpu.log"; exit 1; }
echo "pytest: $(tail -1 "$out/pytest_gpu.log")"
fi
timeout -k 10 300 python3 "$B" > "$out/bench.json.log" 2> "$out/bench.err" || { tail "$out/bench.err"; exit 1; }
cd /tmp
pmc() {  # pmc <dir> <counters...> -- <bench args...>
  local dir=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -s KILL 240 rocprofv3 --pmc "${ctr[@]}" -d "$out/$dir" -o r -- python3 "$B" "$@" > "$out/$dir.log" 2>&1 || { echo "PMC pass $dir failed"; tail -5 "$out/$dir.log"; exit 1; }
}
C5="--steps 1 --warmup 1 --no-cpu-baseline --no-stats"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o r -- python3 "$B" --steps 16 --warmup 2 --no-cpu-baseline > "$out/bench_under_rocprof.log" 2>&1 || exit 1
pmc pmc_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $C5
pmc pmc_fetch FETCH_SIZE TCC_HIT_sum -- $C5
pmc pmc_write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -- $C5
pmc pmc_ta TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -- $C5
cd "$root"
if [ -z "$quick" ]; then
# the other configurations (config 2 at r = 4 / 1.7, config 4 at its start-up camera and with the sphere filling the frame, the
# config-5 scene in chapter-3 mode, one frame per call): rates + the kernel trace of the same command
timeout -k 10 300 python3 -u tools/perf_matrix.py --cases c5,c3,c4,c4fill,one,prim --spp 64 --stats > "$out/matrix.log" 2>&1 || { tail "$out/matrix.log"; exit 1; }
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/matrix_stats" -o r -- python3 "$root/tools/perf_matrix.py" --cases c5,c3,c4,c4fill,prim --spp 64 > "$out/matrix_under_rocprof.log" 2>&1 || exit 1
cd "$root"
# a sustained run: 512 launches of 64 frames (>= 10 s), kernel time and measured shader clock over the whole run
timeout -k 10 300 python3 "$B" --steps 512 --warmup 2 --no-cpu-baseline > "$out/sustained.json.log" 2> "$out/sustained.err" || { tail "$out/sustained.err"; exit 1; }
timeout -k 10 600 python3 -u tools/baselines.py > "$out/baselines.log" 2>&1 || { tail "$out/baselines.log"; exit 1; }
# the synthetic ~10 M-triangle workload at 4K (BASELINE configs[4]); the scene build is cached under /tmp for the passes
SYN="--scene synthetic --width 3840 --height 2160 --spp 8 --no-cpu-baseline"
timeout -k 10 600 python3 "$B" $SYN --steps 16 --warmup 2 > "$out/syn_bench.json.log" 2> "$out/syn_bench.err" || { tail "$out/syn_bench.err"; exit 1; }
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$out/syn_stats" -o r -- python3 "$B" $SYN --steps 4 --warmup 1 --no-overlap > "$out/syn_bench_under_rocprof.log" 2>&1 || exit 1
pmc syn_pmc_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $SYN --steps 1 --warmup 1 --no-stats
pmc syn_pmc_fetch FETCH_SIZE TCC_HIT_sum -- $SYN --steps 1 --warmup 1 --no-stats
pmc syn_pmc_write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -- $SYN --steps 1 --warmup 1 --no-stats
cd "$root"
fi
# summarise on the box and drop the rocprofv3 databases (gpurun copies back at most 64 MiB)
mkdir -p "$out/summary"
cp profiles/hbm_traffic.json "$out/summary/hbm_traffic.json" 2>/dev/null
python3 tools/summarize_profiles.py "$out" "$out/summary/$tag" > "$out/summary/summarize.log" 2>&1 || { tail "$out/summary/summarize.log"; exit 1; }
find "$out" -name "*.db" -delete
find "$out" -type d -empty -delete
tail -1 "$out/bench.json.log" | cut -c1-300
du -sh "$out"
