#!/usr/bin/env python3
"""parity_margins.jsonl (tests/margins.py) -> the table of DESIGN.md §2: per (test, workload, arithmetic, quantity) the worst error any
run of the comparison saw, its gate, and gate / worst.

    python tools/margins_table.py gpurun_out/parity_margins.jsonl > profiles/round6/r6_parity_margins.txt
"""
import json
import sys


def main(path):
    rows = {}
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        r = json.loads(line)
        key = (r["test"], r["workload"], r.get("arith") or "-", r["quantity"])
        cur = rows.get(key)
        if cur is None or r["worst"] > cur["worst"]:
            rows[key] = dict(r, n=(cur["n"] + 1 if cur else 1))
        else:
            cur["n"] += 1
    print("%-44s %-46s %-7s %-58s %10s %9s %8s  %s" % ("test", "workload / case", "arith", "quantity", "worst", "gate", "gate/w", "where (runs)"))
    for key in sorted(rows):
        r = rows[key]
        ratio = ("%8.1f" % (r["gate"] / r["worst"])) if r["worst"] > 0 else "     inf"
        print("%-44s %-46s %-7s %-58s %10.2e %9.1e %s  %s (%d)" % (key[0], key[1], key[2], key[3], r["worst"], r["gate"], ratio, r.get("where") or "", r["n"]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_margins.jsonl")
