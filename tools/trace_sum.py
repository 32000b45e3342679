"""Per-kernel sums of a trace cut by trace_tail.py.  usage: trace_sum.py <trace_last.csv>"""
import sys, collections
rows=[l.strip().split(',') for l in open(sys.argv[1])]
d=collections.defaultdict(lambda:[0,0.0])
for n,q,s,e in rows:
    d[n][0]+=1; d[n][1]+=float(e)-float(s)
tot=float(rows[-1][3])-float(rows[0][2])
print("span %.1f us, %d kernels"%(tot,len(rows)))
for n,(c,t) in sorted(d.items(), key=lambda x:-x[1][1])[:16]:
    print("%-14s %5d %9.1f us  avg %7.2f"%(n,c,t,t/c))
rows2=sorted(rows,key=lambda r:float(r[2]))
gap=sum(max(0.0,float(b[2])-float(a[3])) for a,b in zip(rows2,rows2[1:]))
print("gaps %.1f"%gap)
