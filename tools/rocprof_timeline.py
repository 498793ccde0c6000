"""Timeline check of the left-over-tile concurrency on a rocprofv3 --kernel-trace database:
for every persistent lstm_rec_kernel dispatch, how many lstm_step1_kernel launches ran INSIDE its window and how
many only started after it (= they did not fit beside it and queued: costs ~1.7 ms per layer).
usage: rocprof_timeline.py <rocprofv3 output dir>"""
import glob
import sqlite3
import sys


def main(outdir):
    db = sorted(glob.glob(outdir + "/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, vgpr_count, lds_size, scratch_size from kernels order by start"))
    recs = [r for r in rows if "lstm_rec_kernel" in r[0] or "lstm_rec_in_kernel" in r[0] or "lstm_rec_x_kernel" in r[0]]
    steps = [r for r in rows if "lstm_step1_kernel" in r[0]]
    for n, s, e, vg, lds, scr in recs:
        inside = [x for x in steps if x[1] >= s and x[2] <= e]
        after = [x for x in steps if x[2] > e and x[1] < e + 5e6]
        kind = "layer 0" if ("true>" in n or "rec_in" in n) else "layer 1"
        tail = f", last one ends {(e - max(x[2] for x in inside)) / 1e6:.1f} ms before the persistent kernel" if inside else ""
        print(f"{kind}: persistent kernel {(e - s) / 1e6:.2f} ms (vgpr field {vg}, lds {lds}, scratch {scr}); "
              f"{len(inside)} step launches inside its window{tail}; {len(after)} finished after it")


if __name__ == "__main__":
    main(sys.argv[1])
