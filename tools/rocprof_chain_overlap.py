"""Do the persistent launches of different streams overlap?  Lists the fb_chain / group kernel dispatches of the LAST
iteration in a rocprofv3 --kernel-trace database with start / end relative to the first one, and their stream / queue.
usage: rocprof_chain_overlap.py <rocprofv3 output dir> [n last dispatches]"""
import glob
import sqlite3
import sys


def main(outdir, last=5):
    db = sorted(glob.glob(outdir + "/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    extra = [k for k in ("stream_id", "queue_id", "grid_x") if k in cols]
    rows = list(c.execute(f"select name, start, end, {', '.join(extra)} from kernels order by start"))
    rows = [r for r in rows if "fb_chain_kernel" in r[0] or "lstm2_group" in r[0]][-last:]
    t0 = rows[0][1]
    for r in rows:
        name = r[0].split("(")[0][-40:]
        print(f"{name:42s} start {(r[1] - t0) / 1e3:8.1f} us  end {(r[2] - t0) / 1e3:8.1f} us  dur {(r[2] - r[1]) / 1e3:7.1f} us  "
              + "  ".join(f"{k}={v}" for k, v in zip(extra, r[3:])))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)
