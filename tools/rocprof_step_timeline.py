"""Kernel timeline of the LAST training step in a rocprofv3 --kernel-trace database: start / end (us, from the step's first
kernel), duration, queue and name of every kernel after the last clip_adam_kernel but one - to see what ran beside what.
usage: rocprof_step_timeline.py <rocprofv3 output dir> [min duration us]"""
import glob
import sqlite3
import sys


def main(outdir, min_us=20.0):
    db = sorted(glob.glob(outdir + "/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(c.execute(f"select name, start, end, {q} from kernels order by start"))
    adam = [i for i, r in enumerate(rows) if "clip_adam_kernel" in r[0]]
    lo = adam[-2] + 1 if len(adam) >= 2 else 0
    hi = adam[-1] + 1
    t0 = rows[lo][1]
    for n, s, e, qid in rows[lo:hi]:
        d = (e - s) / 1e3
        if d >= min_us:
            print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  {d:8.1f} us  q{qid}  {n[:70]}")
    # idle time of the device inside the step: the gaps of the union of all kernel intervals
    ivs = sorted((s, e) for _, s, e, _ in rows[lo:hi])
    idle, gaps, end = 0, [], ivs[0][1]
    for s, e in ivs[1:]:
        if s > end:
            idle += s - end
            gaps.append(((s - end) / 1e3, (end - t0) / 1e3))
        end = max(end, e)
    gaps.sort(reverse=True)
    print(f"step: {(rows[hi - 1][2] - t0) / 1e6:.3f} ms, no kernel running for {idle / 1e6:.3f} ms of it; largest gaps (us @ us): "
          + ", ".join(f"{g:.0f} @ {at:.0f}" for g, at in gaps[:8]))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 20.0)
