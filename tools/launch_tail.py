#!/usr/bin/env python3
"""Diagnostic: the small-launch tail of a training step from a rocprofv3 `--kernel-trace --stats` kernel_stats.csv of a bench.py run:
launches per step, how many of them run for under 20 us on average and what they add up to.  `python tools/launch_tail.py <csv> [...]`
(steps = the number of rat_clip_adam_fused launches; the run's warm-up and its eager event-timing pass are part of the averages)."""
import csv
import sys


def summarize(path, limit_us=20.0):
    rows = list(csv.DictReader(open(path)))
    steps = [int(r["Calls"]) for r in rows if "clip_adam_fused" in r["Name"] or "clip_other" in r["Name"]][0]
    total = n = small = nsmall = 0.0
    lines = []
    for r in rows:
        per_step, avg = float(r["TotalDurationNs"]) / steps / 1e3, float(r["AverageNs"]) / 1e3
        calls = int(r["Calls"]) / steps
        total, n = total + per_step, n + calls
        if avg < limit_us:
            small, nsmall = small + per_step, nsmall + calls
            lines.append((per_step, "  %-72s %5.1f x %6.1f us = %7.1f us" % (r["Name"][:72], calls, avg, per_step)))
    print("%s: %d steps; %.1f launches per step, %.1f us of kernels per step; under %.0f us: %.1f launches = %.1f us"
          % (path, steps, n, total, limit_us, nsmall, small))
    for _, line in sorted(lines, reverse=True):
        print(line)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarize(p)
