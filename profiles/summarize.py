#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite outputs (gpurun_out/<dir>/...) into the small CSV summaries kept here.

  python profiles/summarize.py stats gpurun_out/r01prof/kt/kt_results.db  profiles/r01_bench_kernel_stats.csv
  python profiles/summarize.py pmc   gpurun_out/r01prof/pmc_rd/pmc_results.db profiles/r01_pmc_read.csv
"""
import collections
import csv
import sqlite3
import sys


def stats(db, out):
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name.split("(")[0], calls, "%.3f" % tot, "%.3f" % avg, "%.3f" % pct])


def pmc(db, out):
    acc = collections.OrderedDict()
    for k, c, v in db.execute("select kernel_name,counter_name,value from counters_collection"):
        e = acc.setdefault((k.split("(")[0], c), [0, 0.0])
        e[0] += 1
        e[1] += v
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "avg_per_launch"])
        for (k, c), (n, v) in acc.items():
            w.writerow([k, c, n, "%.1f" % (v / n)])


if __name__ == "__main__":
    mode, src, dst = sys.argv[1:4]
    {"stats": stats, "pmc": pmc}[mode](sqlite3.connect(src), dst)
