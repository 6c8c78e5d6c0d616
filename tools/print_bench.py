"""One line per bench.py JSON file: value, ms per step, achieved GB/s, roofline fraction (development helper)."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f"{f}: {d['value']:.0f} {d['unit']}  {d['ms_per_step'] * 1e3:.1f} us/step  kernel {r['achieved']:.0f} {r['unit']} frac {r['frac']:.3f}  hint {d['config'].get('next_batch_hint')}")
    except Exception as e:   # noqa: BLE001
        print(f"{f}: unreadable ({e})")
