"""Print the headline numbers of bench.py's JSON line(s) read from stdin: value / ms per step of the default model and of `models`."""
import json
import sys

for line in sys.stdin.read().strip().splitlines():
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    extra = " | ".join(f"{k} {v['value']} img/s {v['ms_per_step']} ms" for k, v in (d.get("models") or {}).items())
    print(f"{d['config'].get('workload', '')[:24]:<24} {d['value']} img/s {d['ms_per_step']} ms" + (" | " + extra if extra else ""))
