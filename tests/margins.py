"""Parity MARGINS: every parity gate that runs on the GPU box also records how far from its tolerance the comparison landed
(a gate whose margin nobody knows cannot tell a regression from noise).  One JSON line per comparison, appended to
$RAT_MARGINS_FILE (default gpurun_out/parity_margins.jsonl when that directory exists) and printed (pytest -s / -rP shows it);
tools/margins_table.py turns the file into the table of DESIGN.md §2."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _path():
    p = os.environ.get("RAT_MARGINS_FILE")
    if p:
        return p
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d, "parity_margins.jsonl") if os.path.isdir(d) else None


def record(test, workload, quantity, worst, gate, arith=None, where=None):
    """worst: the largest error the comparison saw, in the gate's own unit (absolute, or relative to the tensor's largest element)"""
    line = {"test": test, "workload": workload, "arith": arith, "quantity": quantity, "worst": float(worst), "gate": float(gate),
            "margin": (float(gate) / float(worst)) if worst > 0 else None, "where": where}
    print("parity margin: " + json.dumps(line))
    p = _path()
    if p:
        try:
            with open(p, "a") as f:
                f.write(json.dumps(line) + "\n")
        except OSError:
            pass
    return line
