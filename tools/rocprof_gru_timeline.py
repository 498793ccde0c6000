"""Timeline of one GRU FullSubNet call on a rocprofv3 --kernel-trace database (tools/bench_family.py gru B): for every persistent
many-row launch, how many gru_step_kernel launches of the left-over tiles ran inside its window / finished after it, and the
gaps between the large kernels of the last call.  usage: rocprof_gru_timeline.py <rocprofv3 output dir>"""
import glob
import sqlite3
import sys


def main(outdir):
    db = sorted(glob.glob(outdir + "/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    recs = [r for r in rows if "lstm_rec_in_kernel" in r[0] or "lstm_rec_x_kernel" in r[0]]
    steps = [r for r in rows if "gru_step_kernel" in r[0]]
    for n, s, e in recs[-2:]:
        inside = [x for x in steps if x[1] >= s and x[2] <= e]
        after = [x for x in steps if x[2] > e and x[1] < e + 20e6 and x[1] >= s]
        last = max((x[2] for x in after), default=e)
        print(f"{n[:24]}: {(e - s) / 1e6:.2f} ms; {len(inside)} step launches inside its window, {len(after)} end after it "
              f"(the last one {(last - e) / 1e6:.2f} ms after)")
    # the last call: every kernel longer than 50 us, with the idle gap before it
    s0 = recs[-2][1] - 3e6
    big = [r for r in rows if r[1] >= s0 and (r[2] - r[1]) > 50e3 and "gru_step" not in r[0]]
    prev = None
    for n, s, e in big:
        gap = (s - prev) / 1e6 if prev else 0.0
        print(f"  +{(s - s0) / 1e6:8.3f} ms  {n[:60]:60s} {(e - s) / 1e6:8.3f} ms  (gap before: {gap:.3f})")
        prev = e


if __name__ == "__main__":
    main(sys.argv[1])
