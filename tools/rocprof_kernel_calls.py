"""Every dispatch of the kernels whose name contains a pattern, in a rocprofv3 --kernel-trace database: duration and grid.
usage: rocprof_kernel_calls.py <rocprofv3 output dir> <pattern> [N last]"""
import glob
import sqlite3
import sys


def main(outdir, pat, last=12):
    db = sorted(glob.glob(outdir + "/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    rows = [r for r in c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start") if pat in r[0]]
    for n, s, e, gx, gy, gz in rows[-last:]:
        print(f"{(e - s) / 1e3:9.1f} us  grid {gx} x {gy} x {gz}  {n[:70]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12)
