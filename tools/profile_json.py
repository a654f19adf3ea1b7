#!/usr/bin/env python3
"""rocprofv3 rocpd databases -> profiles/<tag>_counters.json, the file bench.py quotes PMC figures from.

  python tools/profile_json.py <tag> <out.json> <kernel-trace.db> <pmc.db> [<pmc.db> ...]

Per (kernel, grid size): calls and average duration from the kernel-trace pass, the mean of every counter from the separate PMC
passes (FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them -- bench.py applies the gfx950 x2 correction of the read side),
plus the register / LDS / scratch figures of the dispatch.  `source_hash` is bench.py's hash of csrc/: a bench run only quotes
these counters while the device sources are the ones that were profiled.
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    name = name.replace("HIP_vector_type<float, 4u>", "float4")
    return name if len(name) < 60 else name[:name.index("(")] if "(" in name else name[:60]


def main():
    tag, out, kt, pmcs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
    import bench
    kernels = {}
    db = sqlite3.connect(kt)
    q = ("select name, grid_x, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x) "
         "from kernels group by name, grid_x")
    for r in db.execute(q):
        kernels["%s@%d" % (short(r[0]), r[1])] = dict(grid_x=r[1], calls=r[2], avg_us=round(r[3], 3), min_us=round(r[4], 3), max_us=round(r[5], 3),
                                                    vgpr=r[6], sgpr=r[7], lds_bytes=r[8], wg_x=r[9])
    for path in pmcs:
        db = sqlite3.connect(path)
        q = "select kernel_name, grid_size_x, counter_name, count(*), avg(value), max(scratch_size), max(accum_vgpr_count) from counters_collection group by kernel_name, grid_size_x, counter_name"
        for r in db.execute(q):
            k = kernels.setdefault("%s@%d" % (short(r[0]), r[1]), dict(grid_x=r[1]))
            k[r[2]] = round(r[4], 3)
            k["pmc_dispatches"] = r[3]
            k["scratch_bytes_per_lane"] = r[5]
            k["agpr"] = r[6]
    # keep the file small: kernels that matter to the bench line
    keep = ("gdpt_tr::k_render", "gdpt::kp_cg", "gdpt::kf_", "gdpt_tr::k_resolve", "gdpt_tr::k_develop", "gdpt_tr::k_gather", "gdpt_tr::k_bounce", "gdpt_tr::k_", "gdpt_bdk::k_")
    kernels = {k: v for k, v in kernels.items() if any(s in k for s in keep)}
    json.dump(dict(tag=tag, source_hash=bench._source_hash(), units={"FETCH_SIZE": "KiB", "WRITE_SIZE": "KiB", "avg_us": "us"},
                   command="tools/prof_r02.sh (rocprofv3 --kernel-trace --stats | --pmc ..., separate passes) -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline",
                   kernels=kernels), open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, len(kernels), "kernel entries")


if __name__ == "__main__":
    main()
