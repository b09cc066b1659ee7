#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of ROCm 7.2's rocprofv3) as text.

  python tools/rocprof_summary.py gpurun_out/prof_r1/stats/r1_results.db            # kernel-trace --stats summary
  python tools/rocprof_summary.py gpurun_out/prof_r1/pmc_fetch/r1_results.db --pmc  # per-kernel PMC counters
  python tools/rocprof_summary.py <db> --skip-first 1     # adds avg_us_timed: the average WITHOUT each kernel's first N dispatches -- the
                                                          # bench's warm-up step(s), which bench.py's own launch_ms (HIP events over the timed
                                                          # region) does not contain either; the two then measure the same launches

The outputs committed under profiles/ are produced with this script from the databases the GPU runs leave in
gpurun_out/ (scratch).
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pmc = "--pmc" in sys.argv
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    if not pmc:
        rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
        timed = {}
        if skip:
            per = {}
            for name, start, end in db.execute("select name, start, end from kernels order by start").fetchall():
                per.setdefault(name, []).append((end - start) / 1000.0)
            timed = {k: (sum(v[skip:]) / len(v[skip:]) if len(v) > skip else None) for k, v in per.items()}
        print("%-72s %6s %14s %14s %7s%s" % ("kernel", "calls", "total_us", "avg_us", "pct", ("   avg_us_timed (without the first %d)" % skip) if skip else ""))
        for name, calls, total, avg, pct in rows:
            short = name.split("(")[0][-72:]
            t = timed.get(name)
            print("%-72s %6d %14.3f %14.3f %7.2f%s" % (short, calls, total, avg, pct, ("   %14.3f" % t) if t is not None else ""))
        regs = db.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                          "max(grid_x), max(grid_y), max(workgroup_x) from kernels group by name").fetchall()
        print()
        print("%-40s %5s %5s %5s %7s %8s %10s %6s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid_x", "grid_y", "wg"))
        for r in regs:
            if r[0].startswith("k_") or "gpv" in r[0]:
                print("%-40s %5d %5d %5d %7d %8d %10d %6d %5d" % ((r[0].split("(")[0][:40],) + tuple(r[1:])))
    else:
        # min / max over the rows of one (kernel, counter): a row is one dispatch on one XCD (GRBM_*, TCC_*) or shader engine (SQ_*), so for a
        # kernel launched the same way every time the spread IS the imbalance between XCDs / shader engines (round 6; appended columns --
        # tools/make_traffic_json.py reads the first six only)
        rows = db.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value), avg(duration), min(counter_value), max(counter_value) "
                          "from pmc_events group by name, counter_name order by 4 desc").fetchall()
        print("%-56s %-12s %6s %18s %18s %14s %18s %18s" % ("kernel", "counter", "n", "sum", "avg_per_dispatch", "avg_dur_us", "min_row", "max_row"))
        for name, ctr, n, total, avg, dur, lo, hi in rows:
            print("%-56s %-12s %6d %18.3f %18.3f %14.3f %18.3f %18.3f" % (str(name).split("(")[0][-56:], ctr, n, total, avg, dur / 1e3, lo, hi))


if __name__ == "__main__":
    main()
