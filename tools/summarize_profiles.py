"""Turn the rocprofv3 databases collected by tools/profile_round.sh into the small text summaries kept under profiles/.

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


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats = kernel_stats(os.path.join(src, "stats", "r_results.db"))
    with open(dst + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for r in stats:
            f.write('"%s",%d,%.3f,%.3f,%.4f\n' % (r["kernel"], r["calls"], r["total_us"], r["avg_us"], r["percent"]))
    pmc = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        db = os.path.join(d, "r_results.db")
        if os.path.isfile(db):
            pmc[os.path.basename(d)] = counters(db)
    with open(dst + "_pmc.json", "w") as f:
        json.dump(pmc, f, indent=1)
    for name in ("bench.json.log", "bench_under_rocprof.log", "pytest_gpu.log"):
        p = os.path.join(src, name)
        if os.path.isfile(p):
            shutil.copy(p, dst + "_" + name)
    print(json.dumps(stats[:3], indent=1))
    for k, v in pmc.items():
        print(k, json.dumps(v[-1] if v else None))


if __name__ == "__main__":
    main()
