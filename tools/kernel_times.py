#!/usr/bin/env python3
"""Prints per-kernel average durations from a rocprofv3 results .db (any output format has one).
usage: kernel_times.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, calls, avg, mn, mx in db.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
        "where name like '%svc::%' group by name order by 3 desc"):
    print(f"{avg / 1e3:10.1f} us  x{calls:<4d} min {mn / 1e3:9.1f} max {mx / 1e3:9.1f}  {name}")
