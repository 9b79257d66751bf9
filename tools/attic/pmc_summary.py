"""Per-kernel mean of one rocprofv3 PMC counter (FETCH_SIZE / WRITE_SIZE, in KiB) from a
`--pmc X --kernel-trace --output-format csv` run.  usage: pmc_summary.py <dir> <COUNTER> [x2]
x2: the gfx950 correction for FETCH_SIZE on wide coalesced streams (/opt/skills/guides/MI355X_MICROARCH.md)."""
import csv, glob, os, sys
from collections import defaultdict
d, counter = sys.argv[1], sys.argv[2]
mul = 2.0 if len(sys.argv) > 3 and sys.argv[3] == "x2" else 1.0
path = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
acc = defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] == counter:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print(f"| kernel | launches | mean {counter}/launch (KiB) | bytes/launch{' (x2)' if mul == 2 else ''} |\n|---|---:|---:|---:|")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    m = sum(v) / len(v)
    print(f"| {name} | {len(v)} | {m:.1f} | {m * 1024 * mul:.4g} |")
