import csv,sys,glob
def load(d,steps):
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)[0]
    return {r['Name']:(int(r['Calls'])/steps,float(r['TotalDurationNs'])/1e6/steps) for r in csv.DictReader(open(f))}
a=load(sys.argv[1],7); b=load(sys.argv[2],7)
keys=sorted(set(a)|set(b),key=lambda k:-abs(a.get(k,(0,0))[1]-b.get(k,(0,0))[1]))
print("total new %.2f old %.2f"%(sum(v[1] for v in a.values()),sum(v[1] for v in b.values())))
for k in keys[:25]:
    print("%-80s new %6.1f calls %6.3f ms | old %6.1f calls %6.3f ms | d %+.3f"%(k[:80],*a.get(k,(0,0)),*b.get(k,(0,0)),a.get(k,(0,0))[1]-b.get(k,(0,0))[1]))
