#!/usr/bin/env python3
"""Per basic block of one kernel in a `hipcc -S` listing: MFMA, vector, LDS and memory instruction counts and the block's
loop depth - the view that found lstm_rec_x_kernel's fill loop and the 64-bit division in its tail.
usage: isa_blocks.py <file.s> <kernel name substring> [min vector instructions to list a block without MFMAs]"""
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    on, lines = False, []
    for l in open(path):
        if not on and re.match(r"^[A-Za-z_][\w$.]*:", l) and want in l:
            on = True
            print(l.split(":")[0])
        if on:
            lines.append(l.rstrip())
            if "s_endpgm" in l:
                break
    blk, stats, order = "entry", {}, []
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blk = m.group(1)
        s = stats.setdefault(blk, {"mfma": 0, "valu": {}, "ds": 0, "vmem": 0, "salu": 0, "depth": ""})
        if blk not in order:
            order.append(blk)
        if "Depth=" in l:
            s["depth"] = l.split(";")[-1].strip()
        if l.strip().startswith(";"):
            continue
        t = l.strip().split()
        if not t:
            continue
        op = t[0]
        if op.startswith("v_mfma"):
            s["mfma"] += 1
        elif op.startswith("v_"):
            s["valu"][op] = s["valu"].get(op, 0) + 1
        elif op.startswith("ds_"):
            s["ds"] += 1
        elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
            s["vmem"] += 1
        elif op.startswith("s_"):
            s["salu"] += 1
    for b in order:
        s = stats[b]
        nv = sum(s["valu"].values())
        if s["mfma"] or nv >= floor:
            top = ", ".join(f"{k} {v}" for k, v in sorted(s["valu"].items(), key=lambda x: -x[1])[:6])
            print(f"{b:12s} {s['depth'][:34]:34s} mfma {s['mfma']:4d}  vector {nv:4d}  lds {s['ds']:3d}  mem {s['vmem']:3d}  scalar {s['salu']:4d}  | {top}")


if __name__ == "__main__":
    main()
