#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel.

usage: summarize_pmc.py OUT.json DIR [DIR ...]
Each DIR is the -d directory of one `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv` run (one counter per
pass, as MI355X_MICROARCH.md prescribes).  Output: {kernel: {counter: {"calls": n, "sum_kb": s}}} with the kernel's
argument list stripped.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB."""
import csv, json, pathlib, sys, collections

out = collections.defaultdict(dict)
for d in sys.argv[2:]:
    for f in pathlib.Path(d).rglob("*counter_collection.csv"):
        acc = collections.defaultdict(lambda: [0, 0.0])
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
                acc[k][0] += 1
                acc[k][1] += float(r["Counter_Value"])
        for (kern, ctr), (n, s) in acc.items():
            out[kern][ctr] = {"calls": n, "sum_kb": s}
pathlib.Path(sys.argv[1]).write_text(json.dumps(out, indent=1, sort_keys=True))
print(json.dumps({k: {c: round(v["sum_kb"] / 1024.0, 1) for c, v in d.items()} for k, d in out.items()}, indent=1), "(MB)")
