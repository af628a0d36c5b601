"""Where a replayed step's time goes BETWEEN kernels (rocprofv3 --kernel-trace CSV).

    python scripts/gap_analysis.py <..._kernel_trace.csv> [marker-substring] [steps]

Sorts the dispatches by start time, cuts the trace into steps at the marker kernel (default: the AdamW kernel, the
last node of a step), and for the last `steps` steps prints the step span, the sum of kernel durations, the time with
no kernel running (gaps), the overlap, and the gaps attributed to the kernel that FOLLOWS them (i.e. whose start was
late) — by kernel name, largest first.  Used to decide whether the launch tail or the kernels bound a small-batch step.
"""

from __future__ import annotations

import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:70]


def main() -> None:
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_adamw"
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(cuts) < nsteps + 1:
        print(f"only {len(cuts)} marker kernels ({marker}) in {len(rows)} dispatches")
        return
    cuts = cuts[-(nsteps + 1):]
    gap_by = defaultdict(lambda: [0.0, 0])
    dur_by = defaultdict(lambda: [0.0, 0])
    spans, durs, gaps, novs = [], [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        step = rows[a + 1:b + 1]
        t_end = rows[a][1]                       # end of the previous step's last kernel
        span = step[-1][1] - t_end
        dsum = gsum = 0.0
        busy_until = t_end
        for s, e, n in step:
            dsum += e - s
            dur_by[short(n)][0] += e - s
            dur_by[short(n)][1] += 1
            if s > busy_until:
                g = s - busy_until
                gsum += g
                gap_by[short(n)][0] += g
                gap_by[short(n)][1] += 1
            busy_until = max(busy_until, e)
        spans.append(span)
        durs.append(dsum)
        gaps.append(gsum)
        novs.append(len(step))
    n = len(spans)
    print(f"{n} steps, {sum(novs) / n:.0f} dispatches per step")
    print(f"step span {sum(spans) / n / 1e3:9.1f} us   kernel durations {sum(durs) / n / 1e3:9.1f} us   "
          f"idle gaps {sum(gaps) / n / 1e3:9.1f} us   overlap {(sum(durs) + sum(gaps) - sum(spans)) / n / 1e3:9.1f} us")
    print(f"\n{'gap us/step':>12} {'count/step':>10} {'avg gap us':>10}  follower kernel")
    for name, (g, c) in sorted(gap_by.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"{g / n / 1e3:12.1f} {c / n:10.1f} {g / c / 1e3:10.2f}  {name}")
    print(f"\n{'dur us/step':>12} {'count/step':>10} {'avg us':>10}  kernel")
    for name, (d, c) in sorted(dur_by.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"{d / n / 1e3:12.1f} {c / n:10.1f} {d / c / 1e3:10.2f}  {name}")


if __name__ == "__main__":
    main()
