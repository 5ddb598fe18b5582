"""Sum rocprofv3 counter_collection.csv per kernel: python scripts/pmc_sum.py <csv> <name-substr> [launches]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
for k, v in acc.items():
    if sys.argv[2] in k:
        n = max(len(disp[k]), 1)
        print(k[:60], "launches", n)
        for a, b in sorted(v.items()):
            print(f"   {a:32s} {b / n:16.1f} per launch")
