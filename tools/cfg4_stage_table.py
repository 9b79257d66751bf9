"""Per-draw kernel table of a config-4 run from a rocprofv3 --kernel-trace CSV
(`rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/cfg4_profile_run.py`):
kernels of the LAST n draws (a draw starts at each k_zig_parallel / k_refresh launch), mean time per draw by
kernel, the trajectory kernel by its position in the schedule, and the span of a draw on the device.
usage: cfg4_stage_table.py <dir or kernel_trace.csv> [draws=150]"""
import csv, glob, os, sys
from collections import defaultdict

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


starts = [i for i, r in enumerate(rows) if "k_zig_parallel" in r["Kernel_Name"]]
starts = starts[-(want + 1):]
draws = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
TAGS = {7: ["P0", "P1", "G0(P1)", "P2", "G0(P2)", "G1(P2)", "G0(G1(P2))"],
        4: ["P0", "P1+G0(P1)", "P2+G0(P2)", "G1(P2)+G0(G1(P2))"]}  # (a launch runs its lanes' first ghost too)
tags = TAGS.get(sum("k_lane_traj" in r["Kernel_Name"] for r in draws[-1]), TAGS[7])
per_kernel, per_traj, spans, busy, launches = defaultdict(float), defaultdict(float), [], [], []
for d in draws:
    t = 0
    for r in d:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        per_kernel[short(r["Kernel_Name"])] += dur
        if "k_lane_traj" in r["Kernel_Name"]:
            per_traj[tags[t] if t < len(tags) else f"traj{t}"] += dur
            t += 1
    spans.append((int(d[-1]["End_Timestamp"]) - int(d[0]["Start_Timestamp"])) / 1e3)
    busy.append(sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in d))
    launches.append(len(d))
n = len(draws)
print(f"{n} draws; launches per draw {sum(launches) / n:.1f}; kernels busy {sum(busy) / n:.1f} us per draw; "
      f"first start -> last end {sum(spans) / n:.1f} us per draw\n")
print("| kernel | us per draw | % of busy |\n|---|---:|---:|")
tot = sum(per_kernel.values())
for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]):
    print(f"| {k} | {v / n:.1f} | {100 * v / tot:.1f} |")
print("\nk_lane_traj by trajectory (us, mean): " + " | ".join(f"{t} {per_traj[t] / n:.1f}" for t in tags if t in per_traj))
