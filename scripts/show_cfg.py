"""print the per-direction kernel summary of the configs in a bench.py JSON line: python scripts/show_cfg.py FILE"""
import json
import sys

d = json.load(open(sys.argv[1]))
for k, c in d.get("configs", {}).items():
    if isinstance(c, dict):
        for direction in ("encode", "decode"):
            if direction in c:
                print(k, direction, json.dumps(c[direction]))
