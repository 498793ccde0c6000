import glob, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = list(c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels where name like '%lstm_step2%' order by start"))
import collections
g = collections.defaultdict(list)
for n, s, e, gx, gy, gz in rows: g[(gx, gy, gz)].append((e - s) / 1e3)
for k, v in g.items():
    v = sorted(v); print(k, len(v), "median %.1f us  p10 %.1f  p90 %.1f" % (v[len(v)//2], v[len(v)//10], v[9*len(v)//10]))
# gaps between consecutive step2 launches of the same grid
prev = None; gaps = collections.defaultdict(list)
for n, s, e, gx, gy, gz in rows:
    if prev and prev[0] == (gx, gy, gz): gaps[(gx, gy, gz)].append((s - prev[1]) / 1e3)
    prev = ((gx, gy, gz), e)
for k, v in gaps.items():
    v = sorted(v); print("gap", k, "median %.2f us p90 %.2f" % (v[len(v)//2], v[9*len(v)//10]))
