"""Turn the rocprofv3 databases collected by tools/profile_round.sh into the small text summaries kept under profiles/,
and refresh profiles/hbm_traffic.json (the per-workload PMC figures bench.py's roofline block reads).

usage: python tools/summarize_profiles.py gpurun_out/<tag> profiles/<prefix>
"""
import collections
import glob
import json
import os
import shutil
import sqlite3
import sys


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    return [{"kernel": r[0], "calls": r[1], "total_us": r[2], "avg_us": r[3], "percent": r[4]} for r in rows]


def counters(db, kernel_substr="render_kernel"):
    con = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for name, disp, cname, val in con.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        if kernel_substr in name:
            agg[(name, disp)][cname] += val  # summed over XCDs / shader engines
    out = []
    for (name, disp), c in sorted(agg.items(), key=lambda kv: kv[0][1]):
        out.append({"kernel": name, "dispatch": disp, **c})
    return out


def first_db(d):
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    return dbs[0] if dbs else None


def traffic_entry(pmc, prefix="", stats=None, head=None):
    """bytes per launch etc. of the LAST render_kernel dispatch of each pass (the timed launch of a --steps 1 --warmup 1 run);
    the same for first_kernel (two-kernel form of the path loop) under first_kernel_*."""
    def last(name, first=False):
        v = pmc.get(prefix + name + (":first" if first else ""))
        return v[-1] if v else {}
    f, w, q = last("pmc_fetch"), last("pmc_write"), last("pmc_sq")
    if not f or not w:
        return None
    e = {"fetch_kib": f.get("FETCH_SIZE"), "write_kib": w.get("WRITE_SIZE"),
         # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB summed over the XCDs; on gfx950 FETCH_SIZE tallies each
         # memory-side read at 64 B although a 128-B line moves (calibrated on this kernel's gathers: profiles/r01h_fetch_calibration.md)
         "bytes_per_launch": int((2.0 * f.get("FETCH_SIZE", 0.0) + w.get("WRITE_SIZE", 0.0)) * 1024),
         "bytes_per_launch_raw_counters": int((f.get("FETCH_SIZE", 0.0) + w.get("WRITE_SIZE", 0.0)) * 1024)}
    if w.get("TCC_REQ_sum"):
        e["l2_hit_rate"] = 1.0 - w.get("TCC_MISS_sum", 0.0) / w["TCC_REQ_sum"]
    if q:
        e["valu_wave_instructions_per_launch"] = int(q.get("SQ_INSTS_VALU", 0))
        e["salu_wave_instructions_per_launch"] = int(q.get("SQ_INSTS_SALU", 0))
        if q.get("SQ_INSTS_VALU"):
            e["valu_lane_util"] = q.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * q["SQ_ACTIVE_INST_VALU"]) if q.get("SQ_ACTIVE_INST_VALU") else None
        if q.get("SQ_WAVE_CYCLES"):
            e["wave_wait_fraction"] = q.get("SQ_WAIT_ANY", 0.0) / q["SQ_WAVE_CYCLES"]
        e["grbm_gui_active_cycles_sum"] = q.get("GRBM_GUI_ACTIVE")
    f1, w1, q1 = last("pmc_fetch", True), last("pmc_write", True), last("pmc_sq", True)
    if f1 and w1:
        e["first_kernel_bytes_per_launch"] = int((2.0 * f1.get("FETCH_SIZE", 0.0) + w1.get("WRITE_SIZE", 0.0)) * 1024)
    if q1.get("SQ_INSTS_VALU"):
        e["first_kernel_valu_wave_instructions_per_launch"] = int(q1["SQ_INSTS_VALU"])
        if q1.get("SQ_ACTIVE_INST_VALU"):
            e["first_kernel_valu_lane_util"] = q1.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * q1["SQ_ACTIVE_INST_VALU"])
        if q1.get("SQ_WAVE_CYCLES"):
            e["first_kernel_wave_wait_fraction"] = q1.get("SQ_WAIT_ANY", 0.0) / q1["SQ_WAVE_CYCLES"]
    # what the figures belong to: bench.py drops them when the kernel it times is no longer this one (> 3 % off)
    for r in stats or []:  # (the product kernels: second template argument STATS = false; the statistics passes run the others)
        args = r["kernel"].split("<", 1)[1].split(",") if "<" in r["kernel"] else []
        product = len(args) > 1 and args[1].strip().startswith("false")
        if "render_kernel" in r["kernel"] and product and "kernel_ms_when_profiled" not in e:
            e["kernel_ms_when_profiled"] = r["avg_us"] / 1e3
        if "first_kernel" in r["kernel"] and product and "first_kernel_ms_when_profiled" not in e:
            e["first_kernel_ms_when_profiled"] = r["avg_us"] / 1e3
    e["git_head"] = head
    return e


def bench_key(logpath):
    try:
        line = [l for l in open(logpath) if l.startswith("{")][-1]
        j = json.loads(line)
        w = j["config"]["workload"]
        scene = "synthetic" if w.startswith("synthetic") else ("chap4" if "4chap" in w else ("small" if w.startswith("small") else "chap5"))
        mode = w.split("mode chap")[1][0]
        res = w.split(", ")[2] if "x" in w.split(", ")[2] else [p for p in w.split(", ") if "x" in p and p[0].isdigit()][0]
        spp = [p for p in w.split(", ") if "spp per step" in p][0].split()[0]
        return f"{scene}_chap{mode}_{res}_spp{spp}", j
    except Exception as ex:  # noqa: BLE001
        print("could not derive the workload key from", logpath, ex)
        return None, None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    head = None
    try:  # written before the gpurun call (the box has no .git): git rev-parse --short HEAD > .git_head
        head = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".git_head")).read().strip() or None
    except OSError:
        pass
    all_stats = {}
    for sub, suffix in (("stats", ""), ("syn_stats", "_synthetic"), ("matrix_stats", "_matrix")):
        db = first_db(os.path.join(src, sub))
        if not db:
            continue
        stats = kernel_stats(db)
        all_stats[{"": "", "_synthetic": "syn_", "_matrix": "matrix_"}[suffix]] = stats
        with open(dst + suffix + "_kernel_stats.csv", "w") as f:
            f.write("kernel,calls,total_us,avg_us,percent\n")
            for r in stats:
                f.write('"%s",%d,%.3f,%.3f,%.4f\n' % (r["kernel"], r["calls"], r["total_us"], r["avg_us"], r["percent"]))
        print(json.dumps(stats[:3], indent=1))
    pmc = {}
    for d in sorted(glob.glob(os.path.join(src, "*pmc_*"))):
        if os.path.isdir(d):
            db = first_db(d)
            if db:
                pmc[os.path.basename(d)] = counters(db)
                first = counters(db, "first_kernel")
                if first:
                    pmc[os.path.basename(d) + ":first"] = first
    with open(dst + "_pmc.json", "w") as f:
        json.dump(pmc, f, indent=1)
    for name in ("bench.json.log", "bench_under_rocprof.log", "pytest_gpu.log", "baselines.log", "syn_bench.json.log", "syn_bench_under_rocprof.log",
                 "matrix.log", "sustained.json.log"):
        p = os.path.join(src, name)
        if os.path.isfile(p):
            shutil.copy(p, dst + "_" + name)
    # refresh the figures bench.py reads
    tpath = os.path.join(os.path.dirname(dst) or ".", "hbm_traffic.json")
    try:
        table = json.load(open(tpath))
    except (OSError, ValueError):
        table = {}
    for prefix, log in (("", "pmc_sq.log"), ("syn_", "syn_pmc_sq.log")):
        key, _ = bench_key(os.path.join(src, log))
        e = traffic_entry(pmc, prefix, all_stats.get(prefix), head)
        if key and e:
            e["source"] = os.path.basename(dst) + "_pmc.json"
            table[key] = e
            print(key, json.dumps(e))
    json.dump(table, open(tpath, "w"), indent=1)
    for k, v in pmc.items():
        print(k, json.dumps(v[-1] if v else None))


if __name__ == "__main__":
    main()
