"""Per-kernel totals (ms, calls) of the LAST factorization in a rocprofv3 kernel trace.  usage: kernel_sums.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [k for k in rows[0] if k.lower().startswith("start")][0]
ke = [k for k in rows[0] if k.lower().startswith("end")][0]
kn = [k for k in rows[0] if "kernel_name" in k.lower() or k.lower() == "name"][0]
rows.sort(key=lambda r: int(r[ks]))
last = max(i for i, r in enumerate(rows) if r[kn].startswith("k_amax"))
t = collections.defaultdict(float); n = collections.Counter()
for r in rows[last:]:
    k = r[kn].split("(")[0]; t[k] += (int(r[ke]) - int(r[ks])) / 1e6; n[k] += 1
for k in sorted(t, key=lambda k: -t[k])[:14]: print("%-16s %6d calls %9.2f ms" % (k, n[k], t[k]))
print("span %.2f ms" % ((int(rows[-1][ke]) - int(rows[last][ks])) / 1e6))
