#!/usr/bin/env python3
"""Turns rocprofv3's rocpd sqlite outputs (gpurun_out/prof/*/rNN_results.db) into the small CSV
summaries committed under profiles/.  Usage: summarize_rocpd.py <prof_dir> <out_prefix>"""
import glob
import os
import sqlite3
import sys


def main(prof_dir, out_prefix):
    rows = []
    for db in sorted(glob.glob(os.path.join(prof_dir, "*", "*_results.db"))):
        tag = os.path.basename(os.path.dirname(db))
        con = sqlite3.connect(db)
        try:
            for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
                rows.append(("kernel_stats", tag) + tuple(r))
        except sqlite3.Error:
            pass
        try:
            q = ("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(grid_size), avg(workgroup_size), avg(vgpr_count), "
                 "avg(accum_vgpr_count), avg(sgpr_count), avg(lds_block_size), avg(scratch_size) from counters_collection group by kernel_name, counter_name")
            for r in con.execute(q):
                rows.append(("pmc", tag) + tuple(r))
        except sqlite3.Error:
            pass
    with open(out_prefix + "_kernel_stats.csv", "w") as f:
        f.write("pass,kernel,calls,total_us,avg_us,percent\n")
        for r in rows:
            if r[0] == "kernel_stats":
                f.write("%s,\"%s\",%d,%.3f,%.3f,%.3f\n" % (r[1], r[2], r[3], r[4], r[5], r[6]))
    with open(out_prefix + "_pmc.csv", "w") as f:
        f.write("pass,kernel,counter,dispatches,sum,avg_per_dispatch,grid,workgroup,vgpr,agpr,sgpr,lds_bytes,scratch_bytes\n")
        for r in rows:
            if r[0] == "pmc":
                f.write("%s,\"%s\",%s,%d,%.6g,%.6g,%d,%d,%d,%d,%d,%d,%d\n" % ((r[1], r[2], r[3], r[4], r[5], r[6]) + tuple(int(x or 0) for x in r[7:])))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
