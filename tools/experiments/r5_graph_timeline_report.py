import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last replay = the kernels after the largest idle gap near the end
starts = [int(r["Start_Timestamp"]) for r in rows]
ends = [int(r["End_Timestamp"]) for r in rows]
cut = max(range(len(rows) - 200, len(rows)), key=lambda i: starts[i] - max(ends[:i]) if i else 0)
last = rows[cut:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = {}
tot = 0
print(f"{len(last)} kernels in the last replay; wall {(max(int(r['End_Timestamp']) for r in last) - t0) / 1e3:.1f} us")
for r in last:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    tot += e - s
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:58]
    print(f"q{q:>3} +{(s - t0) / 1e3:7.1f} us  {(e - s) / 1e3:6.1f} us  gap {gap:5.1f}  {name}")
print(f"sum of kernel durations {tot / 1e3:.1f} us")
