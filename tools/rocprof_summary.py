"""Turn a rocprofv3 rocpd sqlite database (default output of `rocprofv3 --kernel-trace --stats`
on ROCm 7.2) into a small text summary for profiles/.  usage: rocprof_summary.py in.db out.md [title]"""
import sqlite3
import sys


def main(db, out, title="rocprofv3 --kernel-trace --stats"):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    det = {}
    for name, vg, ag, sg, lds, gx, wx in c.execute(
            "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), "
            "max(workgroup_x) from kernels group by name"):
        det[name] = (vg, ag, sg, lds, gx, wx)
    pmc = {}
    try:
        for name, cname, val in c.execute(
                "select k.name, p.counter_name, avg(p.value) from pmc_events p join kernels k on "
                "p.dispatch_id = k.dispatch_id group by k.name, p.counter_name"):
            pmc.setdefault(name, {})[cname] = val
    except sqlite3.Error:
        pass
    with open(out, "w") as f:
        f.write(f"# {title}\n\n")
        f.write("| kernel | calls | total ms | avg us | % | vgpr | agpr | sgpr | lds B | grid_x | wg |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")
            short = short.split("(")[0]
            d = det.get(name, ("",) * 6)
            f.write(f"| {short} | {calls} | {tot / 1e3:.3f} | {avg:.2f} | {pct:.2f} | "
                    + " | ".join(str(x) for x in d) + " |\n")
        if pmc:
            f.write("\n## counters (average per dispatch)\n\n")
            for name, d in pmc.items():
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                f.write(f"- {short}: " + ", ".join(f"{k}={v:.4g}" for k, v in sorted(d.items())) + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:4])
