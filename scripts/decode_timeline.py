"""Timeline of one decode step from a rocprofv3 kernel trace of bench.py: per kernel of the step, start offset,
duration and the gap to the previous kernel's end; totals of busy time and gaps.
usage: python scripts/decode_timeline.py OUT_DIR   (rocprofv3 --kernel-trace --output-format csv -d OUT_DIR -- python3 bench.py --steps 1 --warmup 0 --skip-cpu)"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a decode step = the kernels between two consecutive lm_head GEMMs (the only hipBLASLt "Cijk" launches) late in the job
lm = [i for i, r in enumerate(rows) if r[2].startswith("Cijk")]
a, b = lm[-3], lm[-2]
step = rows[a + 1:b + 1]
t0 = step[0][0]
busy = sum(e - s for s, e, _ in step)
gaps = [step[i][0] - step[i - 1][1] for i in range(1, len(step))]
print(f"kernels in the step: {len(step)}   wall {(step[-1][1] - t0) / 1e3:.1f} us   busy {busy / 1e3:.1f} us   gaps {sum(gaps) / 1e3:.1f} us "
      f"(mean {sum(gaps) / len(gaps) / 1e3:.2f} us, max {max(gaps) / 1e3:.2f} us)")
def short(n):
    n = n.replace("mi355x::", "")
    return n[:60]
print("first layer of the step:")
for i, (s, e, n) in enumerate(step[:12]):
    gap = (s - step[i - 1][1]) / 1e3 if i else 0.0
    print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:5.2f}  {short(n)}")
by = {}
for i, (s, e, n) in enumerate(step):
    k = short(n).split("<")[0].split("(")[0]
    d = by.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1; d[1] += (e - s) / 1e3; d[2] += (gaps[i - 1] / 1e3 if i else 0.0)
print("per kernel name: launches, total us, total gap in front")
for k, d in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:60s} {d[0]:4d} {d[1]:9.1f} {d[2]:8.1f}")
