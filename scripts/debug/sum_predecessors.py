"""Which kernels do the immediate k_sum_partials launches follow?   python scripts/debug/sum_predecessors.py <kernel_trace.csv> [kernel]
(rocprofv3 --kernel-trace --output-format csv of a few EAGER steps: `bench.py --no-graph`)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
target = sys.argv[2] if len(sys.argv) > 2 else "k_sum_partials"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
c = collections.Counter()
for i, n in enumerate(names):
    if n.startswith(target):
        j = i - 1
        while j >= 0 and names[j].startswith(target):
            j -= 1
        c[names[j][:110] if j >= 0 else "?"] += 1
total = sum(c.values())
print(f"{total} runs of {target} launches")
for k, v in c.most_common(30):
    print(f"{v:5d}  {k}")
