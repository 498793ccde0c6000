"""Collect the per-kernel PMC figures bench.py's `roofline` quotes from rocprofv3 counter passes.

usage: rocprof_pmc.py <dir with one sub-directory per pass> <out.json> [kernel regex ...]

Every pass directory holds the output of one
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <pass dir> -- python bench.py --steps 1 ...
(separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
Counters are averaged per dispatch of each matching kernel.  Derived figures (guide corrections applied):
  hbm_bytes   = FETCH_SIZE x 1024 x 2 (gfx950: FETCH_SIZE reports half of a wide coalesced read) + WRITE_SIZE x 1024
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): fraction of SIMD-cycles the
                matrix pipe is busy.  Calibration on lstm_rec_x_kernel (profiles/r02_pmc.json): SQ_INSTS_MFMA =
                3.577e9 = the algorithmic count (16 384 rows x 190 steps x 1536 x 768 / 1024 MAC per instruction);
                SQ_VALU_MFMA_BUSY_CYCLES = 32.0 x that (32 cycles per v_mfma_f32_16x16x4_f32);
                SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = its FLOPs exactly; GRBM_GUI_ACTIVE = 1.007e9 is the SUM over the
                8 XCDs (1.259e8 cycles each = the 52.9 ms launch at 2.38 GHz).  For an fp32-MFMA kernel the figure
                therefore equals achieved / peak FLOP rate at the clock the kernel actually ran at.
  mfma_flops  = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 (per launch)
"""
import csv
import glob
import json
import os
import re
import sys

CUS, SIMDS, XCDS = 256, 4, 8
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_stamp(root=ROOT):
    """What the counters were taken on: sha256 over the kernel sources WITHOUT their comments and blank lines (the GPU box has no
    .git; a reworded comment is not another build), and the commit when there is one.  bench.py replays these files' figures and
    compares the stamp with the tree it runs in."""
    import hashlib
    import subprocess
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "fullsubnet_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "fullsubnet_amd", "csrc", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        text = open(f, encoding="utf-8", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)            # block comments
        text = re.sub(r"//[^\n]*", "", text)                          # line comments (no string literal of the sources holds "//")
        code = "\n".join(ln.rstrip() for ln in text.splitlines() if ln.strip())
        h.update(code.encode())
    stamp = {"csrc_sha256": h.hexdigest()[:16], "files": len(files)}
    try:
        stamp["commit"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                         timeout=10).stdout.strip() or None
    except Exception:
        stamp["commit"] = None
    return stamp



def read_pass(d):
    """kernel name -> counter -> [values] over the dispatches of one pass directory."""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("kernel_name") or ""
                cn = row.get("Counter_Name") or row.get("counter_name")
                cv = row.get("Counter_Value") or row.get("counter_value")
                if cn is None or cv is None:
                    continue
                out.setdefault(name, {}).setdefault(cn, []).append(float(cv))
    return out


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(root, out_json, patterns):
    pats = [re.compile(p) for p in (patterns or ["lstm_rec", "gemm_kernel"])]
    merged = {}
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(d):
            continue
        for name, ctr in read_pass(d).items():
            if not any(p.search(name) for p in pats):
                continue
            for cn, vals in ctr.items():
                merged.setdefault(short(name), {}).setdefault(cn, []).extend(vals)
    kernels = {}
    for name, ctr in merged.items():
        k = {"dispatches": max(len(v) for v in ctr.values()), "counters_mean": {c: sum(v) / len(v) for c, v in ctr.items()}}
        m = k["counters_mean"]
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            k["hbm_read_bytes_corrected"] = m.get("FETCH_SIZE", 0.0) * 1024 * 2
            k["hbm_write_bytes"] = m.get("WRITE_SIZE", 0.0) * 1024
            k["hbm_bytes_per_launch"] = k["hbm_read_bytes_corrected"] + k["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
            k["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / XCDS * CUS * SIMDS)
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
            k["mfma_flops_per_launch"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
        kernels[name] = k
    rec = {n: k for n, k in kernels.items() if "lstm_rec" in n and k["dispatches"] > 0}
    dom = {}
    if rec:
        for key in ("hbm_bytes_per_launch", "mfma_busy_frac", "mfma_flops_per_launch"):
            vals = [k[key] for k in rec.values() if key in k]
            if vals:
                dom[key] = sum(vals) / len(vals)
        dom["name"] = "persistent sub-band recurrent kernels (mean of the launches per step): " + ", ".join(sorted(rec))
    json.dump({"note": __doc__.strip().split("\n\n")[0], "build": source_stamp(), "kernels": kernels, "dominant_kernel": dom},
              open(out_json, "w"), indent=1)
    print(json.dumps({"dominant_kernel": dom, "kernels": {n: {c: v for c, v in k.items() if c != "counters_mean"} for n, k in kernels.items()}}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
