#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (ROCm 7.2 default output) into small text summaries.

    python profiles/summarize_rocpd.py <trace.db> [<pmc.db> ...] > profiles/<name>.md

The first database is a `--kernel-trace --stats` capture (per-kernel time table); any
further ones are `--pmc <COUNTER> --kernel-trace` captures (per-kernel mean counter value
per launch, collected in SEPARATE passes).  FETCH_SIZE / WRITE_SIZE are reported in bytes
(rocprofv3 unit: KiB).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
reads exactly half of the bytes of a coalesced streaming read, so the `corrected` column
doubles it.  Calibration in this repo's own access patterns: k_finish reads two [D][C]
arrays = 1.0737e9 B and FETCH_SIZE reports 5.369e8 B (8 B/lane); k_kick_drift_v2 reads three
= 1.6106e9 B and FETCH_SIZE reports 8.054e8 B (16 B/lane) -- exactly 1/2 in both.
WRITE_SIZE matches the written bytes 1:1 (k_kick_drift_v2 writes two arrays = 1.0737e9 B).

With --json <file> the first PMC rows of k_kick_drift_v2 are also written as the per-launch
HBM traffic that bench.py reports in roofline.traffic.
"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


TRAFFIC = {}
TRAFFIC_GRAD = {}


def main():
    json_out = None
    if "--json" in sys.argv:
        i = sys.argv.index("--json")
        json_out = sys.argv[i + 1]
        del sys.argv[i:i + 2]
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
        print("| kernel | launches | mean value/launch (KiB) | bytes/launch | corrected bytes (FETCH x2, gfx950) | avg us |")
        print("|---|---:|---:|---:|---:|---:|")
        for name, ctr, n, val, dur in rows[:8]:
            b = val * 1024
            corr = b * 2 if ctr == "FETCH_SIZE" else b
            if "k_kick_drift_v2" in name:
                TRAFFIC[ctr] = corr
            if "k_gauss_grad_v2" in name:
                TRAFFIC_GRAD[ctr] = corr
            print(f"| {short(name)} | {n} | {val:.1f} | {b:.4g} | {corr:.4g} | {dur / 1e3:.1f} |")
    if json_out and "FETCH_SIZE" in TRAFFIC and "WRITE_SIZE" in TRAFFIC:
        import json

        rec = {"kernel": "k_kick_drift_v2", "chains": 65536, "dims": 1024,
               "fetch_bytes_corrected": TRAFFIC["FETCH_SIZE"], "write_bytes": TRAFFIC["WRITE_SIZE"],
               "traffic_bytes_per_launch": TRAFFIC["FETCH_SIZE"] + TRAFFIC["WRITE_SIZE"],
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH x2 (gfx950)"}
        if "FETCH_SIZE" in TRAFFIC_GRAD and "WRITE_SIZE" in TRAFFIC_GRAD:
            rec["gradient_kernel"] = {"kernel": "k_gauss_grad_v2", "fetch_bytes_corrected": TRAFFIC_GRAD["FETCH_SIZE"],
                                      "write_bytes": TRAFFIC_GRAD["WRITE_SIZE"],
                                      "traffic_bytes_per_launch": TRAFFIC_GRAD["FETCH_SIZE"] + TRAFFIC_GRAD["WRITE_SIZE"]}
        with open(json_out, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
