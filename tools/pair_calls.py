"""Per-call durations of the pair-update kernels in a rocprofv3 kernel trace (last factorization), with the launch's grid:
usage: pair_calls.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [k for k in rows[0] if k.lower().startswith("start")][0]
ke = [k for k in rows[0] if k.lower().startswith("end")][0]
kn = [k for k in rows[0] if "kernel_name" in k.lower() or k.lower() == "name"][0]
gx = [k for k in rows[0] if k.lower() in ("grid_size_x", "grid_size")][0]
gy = [k for k in rows[0] if k.lower() == "grid_size_y"]
wx = [k for k in rows[0] if k.lower() in ("workgroup_size_x", "workgroup_size")][0]
rows.sort(key=lambda r: int(r[ks]))
last = max(i for i, r in enumerate(rows) if r[kn].startswith("k_amax"))
for r in rows[last:]:
    n = r[kn].split("(")[0]
    if n in ("k_upd_w2", "k_upd_c2", "k_upd_c4x"):
        x = int(r[gx]) // int(r[wx]); y = int(r[gy[0]]) if gy else 0
        print(n, x, y, "%.1f" % ((int(r[ke]) - int(r[ks])) / 1e3))
