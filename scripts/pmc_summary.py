"""Summarise the rocprofv3 counter passes written by scripts/pmc_passes.sh into a small CSV + derived metrics.
    python scripts/pmc_summary.py gpurun_out/pmc_<tag> [profiles/<name>.csv]
"""
import csv
import os
import sqlite3
import sys


def main(d, out=None):
    rows = {}
    for name in ("mfma", "waves", "fetch", "write"):
        path = os.path.join(d, name + "_results.db")
        if not os.path.exists(path):
            continue
        con = sqlite3.connect(path)
        q = ("select kernel_name, counter_name, avg(value), count(*), avg(end-start) from counters_collection "
             "where kernel_name like '%cde::%' group by kernel_name, counter_name")
        for kern, ctr, val, n, dur in con.execute(q):
            short = kern.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            rows.setdefault(short, {})[ctr] = val
            rows[short].setdefault("dur_ns_" + name, dur)
    lines = []
    for kern, c in sorted(rows.items()):
        dur = c.get("dur_ns_mfma", 0)
        info = dict(kernel=kern, dur_us=dur / 1e3)
        if "GRBM_GUI_ACTIVE" in c and dur:
            clk = c["GRBM_GUI_ACTIVE"] / 8 / (dur * 1e-9)       # counter is summed over the 8 XCDs
            info["clock_GHz"] = clk / 1e9
            if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                info["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (clk * dur * 1e-9)   # 1024 SIMDs
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            wc = c["SQ_WAVE_CYCLES"]
            info["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / wc
            info["wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / wc
            info["active_inst_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0) / wc
            info["insts_valu"] = c.get("SQ_INSTS_VALU")
            info["insts_mfma"] = c.get("SQ_INSTS_MFMA")
            info["insts_lds"] = c.get("SQ_INSTS_LDS")
            info["lds_bank_conflict_cycles"] = c.get("SQ_LDS_BANK_CONFLICT")
        if "FETCH_SIZE" in c:
            info["fetch_KB_raw"] = c["FETCH_SIZE"]
        if "WRITE_SIZE" in c:
            info["write_KB_raw"] = c["WRITE_SIZE"]
        lines.append(info)
    keys = []
    for l in lines:
        for k in l:
            if k not in keys:
                keys.append(k)
    w = csv.DictWriter(open(out, "w", newline="") if out else sys.stdout, fieldnames=keys)
    w.writeheader()
    for l in lines:
        w.writerow({k: ("%.4g" % v if isinstance(v, float) else v) for k, v in l.items()})


if __name__ == "__main__":
    main(*sys.argv[1:])
