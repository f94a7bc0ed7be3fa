"""Same-box comparison of two rocprofv3 kernel-stats tables (7 steps each: bench.py --steps 5 --warmup 2).
usage: compare_stats.py NEW OLD   (each a *_kernel_stats.csv file or a directory containing one)"""
import csv, sys, glob, os, re
def load(d, steps):
    f = d if os.path.isfile(d) else glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r['Name']).replace("void dei2i::", "").replace("dei2i::", "")
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + int(r['Calls']) / steps, t + float(r['TotalDurationNs']) / 1e6 / steps)
    return out
a = load(sys.argv[1], 7); b = load(sys.argv[2], 7)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
print("total new %.2f ms %.0f launches | old %.2f ms %.0f launches" % (sum(v[1] for v in a.values()), sum(v[0] for v in a.values()), sum(v[1] for v in b.values()), sum(v[0] for v in b.values())))
for k in keys[:n]:
    print("%-70s new %6.1f calls %6.3f ms | old %6.1f calls %6.3f ms | d %+.3f" % (k[:70], *a.get(k, (0, 0)), *b.get(k, (0, 0)), a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
