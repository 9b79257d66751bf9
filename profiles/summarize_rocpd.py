#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (ROCm 7.2 default output) into small text summaries.

    python profiles/summarize_rocpd.py <trace.db> [<pmc.db> ...] > profiles/<name>.md

The first database is a `--kernel-trace --stats` capture (per-kernel time table); any
further ones are `--pmc <COUNTER> --kernel-trace` captures (per-kernel mean counter value
per launch).  FETCH_SIZE / WRITE_SIZE are reported in bytes (rocprofv3 unit: KiB); for the
16-B-per-lane coalesced streaming kernels the gfx950 correction of MI355X_MICROARCH.md
(FETCH_SIZE reads exactly half of the real bytes) is applied in the `corrected` column.
"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main():
    trace = sqlite3.connect(sys.argv[1])
    print("## kernel time (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in trace.execute("select * from top_kernels"):
        print(f"| {short(name)} | {calls} | {total / 1e3:.3f} | {avg:.1f} | {pct:.2f} |")
    for path in sys.argv[2:]:
        db = sqlite3.connect(path)
        rows = db.execute(
            "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
            "group by kernel_name, counter_name order by sum(duration) desc").fetchall()
        if not rows:
            continue
        print(f"\n## PMC pass: {rows[0][1]} ({path})\n")
        print("| kernel | launches | mean value/launch (KiB) | bytes/launch | corrected bytes (x2 for 16-B/lane reads) | avg us |")
        print("|---|---:|---:|---:|---:|---:|")
        for name, ctr, n, val, dur in rows[:8]:
            b = val * 1024
            corr = b * 2 if ctr == "FETCH_SIZE" and ("_v2" in name) else b
            print(f"| {short(name)} | {n} | {val:.1f} | {b:.4g} | {corr:.4g} | {dur / 1e3:.1f} |")


if __name__ == "__main__":
    main()
