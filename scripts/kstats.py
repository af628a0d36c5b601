"""Per-kernel table from a rocprofv3 `*_kernel_stats.csv` (ms per training step, calls per step, average µs)."""
import csv
import sys

path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
marker = sys.argv[3] if len(sys.argv) > 3 else "k_adamw"
rows = list(csv.DictReader(open(path)))
steps = int(marker) if marker.isdigit() else int([r for r in rows if marker in r["Name"]][0]["Calls"])   # or the step count itself
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"steps {steps}  total {tot:.3f} ms/step  launches/step {sum(int(r['Calls']) for r in rows) / steps:.0f}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    name = r["Name"]
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms {int(r['Calls']) / steps:7.1f} calls {float(r['AverageNs']) / 1e3:8.1f} us  {name[:120]}")
