"""Turn a rocprofv3 rocpd database (``rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`` writes
DIR/NAME_results.db on this image) into the per-kernel summary CSV that is committed next to it.

    python profiles/extract_stats.py gpurun_out/prof_r01/bench_results.db profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    extra = {}
    for name, vg, ag, sg, lds, gx, wx in con.execute(
            "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), "
            "max(workgroup_x) from kernels group by name"):
        extra[name] = (vg, ag, sg, lds, gx, wx)
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent", "vgpr", "agpr", "sgpr", "lds_bytes", "grid_x",
                    "workgroup_x"])
        for name, calls, total, avg, pct in rows:
            short = name if len(name) < 160 else name[:157] + "..."
            w.writerow([short, calls, "%.3f" % total, "%.3f" % avg, "%.3f" % pct] + list(extra.get(name, [""] * 6)))
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
