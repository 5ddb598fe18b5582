"""Summarise rocprofv3 CSV output directories into one text table.
usage: python scripts/prof_summary.py OUT_DIR [OUT_DIR ...]   (each holds *_kernel_trace.csv and/or *_counter_collection.csv)
Per kernel (name + grid size): launches, average duration (us), and the per-launch average of every counter."""
import collections, csv, glob, os, re, sys

def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("mi355x::", "")
    return name[:110]

for out in sys.argv[1:]:
    print(f"== {out}")
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = r.get("Grid_Size", r.get("Grid_Size_X", ""))
            dur[(short(r["Kernel_Name"]), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = r.get("Grid_Size", "")
            ctr[(short(r["Kernel_Name"]), grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    keys = sorted(set(dur) | set(ctr), key=lambda k: -sum(dur.get(k, [0])))
    for k in keys:
        d = dur.get(k, [])
        line = f"{k[0]:110s} grid={k[1]:>9s} launches={len(d) or max((len(v) for v in ctr[k].values()), default=0):5d}"
        if d:
            line += f" avg_us={sum(d) / len(d):10.1f} total_ms={sum(d) / 1e3:9.2f}"
        for c, v in sorted(ctr.get(k, {}).items()):
            line += f" {c}={sum(v) / len(v):.1f}"
        print(line)
