"""Last N kernel dispatches of a rocprofv3 --kernel-trace database: start (us, relative), duration, stream, queue, name.
usage: rocprof_tail.py <rocprofv3 output dir> [N]"""
import glob
import sqlite3
import sys


def main(outdir, last=120):
    db = sorted(glob.glob(outdir + "/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    extra = [k for k in ("stream_id", "queue_id") if k in cols]
    rows = list(c.execute(f"select name, start, end, {', '.join(extra)} from kernels order by start"))[-last:]
    t0 = rows[0][1]
    for r in rows:
        name = r[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        print(f"{(r[1] - t0) / 1e3:9.1f} us  +{(r[2] - r[1]) / 1e3:7.1f}  " + " ".join(f"{k[0]}{v}" for k, v in zip(extra, r[3:])) + f"  {name}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120)
