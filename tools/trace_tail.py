"""Cut a rocprofv3 kernel trace down to the LAST factorization (from the last k_amax to the end) and write a compact CSV:
name, queue, start_us, end_us (relative).  usage: trace_tail.py <p_kernel_trace.csv> <out.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key_s = [k for k in rows[0] if k.lower().startswith("start")][0]
key_e = [k for k in rows[0] if k.lower().startswith("end")][0]
key_n = [k for k in rows[0] if "kernel_name" in k.lower() or k.lower() == "name"][0]
key_q = [k for k in rows[0] if "queue" in k.lower()][0]
rows.sort(key=lambda r: int(r[key_s]))
last = max(i for i, r in enumerate(rows) if r[key_n].startswith("k_amax"))
t0 = int(rows[last][key_s])
with open(sys.argv[2], "w") as f:
    for r in rows[last:]:
        f.write("%s,%s,%.2f,%.2f\n" % (r[key_n].split("(")[0], r[key_q], (int(r[key_s]) - t0) / 1e3, (int(r[key_e]) - t0) / 1e3))
print("rows", len(rows) - last)
