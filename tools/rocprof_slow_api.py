"""Slowest HIP runtime API calls in a rocprofv3 --hip-trace database (which call held the host?).
usage: rocprof_slow_api.py <rocprofv3 output dir> [N]"""
import glob
import sqlite3
import sys


def main(outdir, top=15):
    db = sorted(glob.glob(outdir + "/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")]
    cand = [n for n in names if "region" in n.lower() or "api" in n.lower()]
    print("tables/views:", ", ".join(cand)[:600])
    for view in ("regions", "regions_and_samples"):
        if view in names:
            cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
            print(view, cols)
            rows = list(c.execute(f"select name, start, end from {view} order by (end - start) desc limit {top}"))
            t0 = c.execute(f"select min(start) from {view}").fetchone()[0]
            for n, s, e in rows:
                print(f"{(e - s) / 1e6:9.3f} ms  at {(s - t0) / 1e6:10.3f} ms  {n}")
            break


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 15)
