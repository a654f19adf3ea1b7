#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (ROCm 7.2 default format) into the small text summaries kept under profiles/.

  python tools/rocpd_summary.py stats <results.db>          -> per-kernel calls / total / avg / min / max (us)
  python tools/rocpd_summary.py pmc   <results.db> [...]    -> per-kernel mean counter values
"""
import sqlite3
import sys


def short(name):
    name = name.replace("HIP_vector_type<float, 4u>", "float4")
    return name if len(name) < 60 else name[:name.index("(")] if "(" in name else name[:60]


def stats(path):
    db = sqlite3.connect(path)
    q = ("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
         "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc")
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows)
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct,vgpr,sgpr,lds_bytes,grid_x,wg_x")
    for r in rows:
        print("%s,%d,%.1f,%.3f,%.3f,%.3f,%.2f,%d,%d,%d,%d,%d" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))


def pmc(paths):
    print("kernel,counter,dispatches,mean_value")
    for path in paths:
        db = sqlite3.connect(path)
        q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name order by 4 desc"
        for r in db.execute(q):
            print("%s,%s,%d,%.3f" % (short(r[0]), r[1], r[2], r[3]))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
