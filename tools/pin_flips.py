#!/usr/bin/env python
"""Turn the flip counts a GPU collection run observed into pins.

    LITEGS_COLLECT_FLIPS=1 python -m pytest tests -m gpu -q        (on the GPU box; writes gpurun_out/flip_counts.jsonl)
    python tools/pin_flips.py [--merge]                            (here; writes tests/golden/flip_pins.json)

pin = 2 x the largest count observed for the (test, tensor) + 2: the tests' flip_frac stays a ceiling, the pin is what a regression
is measured against.  --merge keeps existing pins for keys the log does not mention (and never lowers a pin below a new observation)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "flip_counts.jsonl")
PINS = os.path.join(ROOT, "tests", "golden", "flip_pins.json")


def main():
    merge = "--merge" in sys.argv
    seen = {}
    for line in open(LOG):
        r = json.loads(line)
        seen[r["key"]] = max(seen.get(r["key"], 0), r["flips"])
    pins = {}
    if merge and os.path.exists(PINS):
        pins = json.load(open(PINS))
    for k, v in seen.items():
        pins[k] = max(2 * v + 2, pins.get(k, 0) if merge else 0)
    with open(PINS, "w") as f:
        json.dump(dict(sorted(pins.items())), f, indent=0)
        f.write("\n")
    print(f"{len(seen)} observed, {len(pins)} pinned -> {PINS}")
    for k, v in sorted(seen.items()):
        print(f"  {v:8d} -> pin {pins[k]:8d}  {k}")


if __name__ == "__main__":
    main()
