import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{n[:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {r["Percentage"]}')
