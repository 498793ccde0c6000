"""Static scan of the kernels INSIDE libfsn_hip.so for the gfx950 store-data hazard (fsn_common.h: fsn_hold_store_data;
measured by tools/probe_store_hazard.hip, profiles/r06_store_hazard.md): a 12 / 16-byte-per-lane store reads its data
registers lane quad by lane quad over the cycles after it issues; a vector instruction that writes them again

  * 1 or 2 issue slots behind a global / flat / scratch store or a buffer store with an IMMEDIATE soffset,
  * 1 issue slot behind a buffer store whose soffset is an SGPR

puts the NEW value into memory for lanes 8 - 15 / 12 - 15 of every 16.  hipcc (ROCm 7.2) keeps the two wait states of the first
case and NONE in the second (GCNHazardRecognizer exempts buffer stores with a register soffset - a rule of the first GCN parts
that gfx950 does not honour): that is what corrupted lstm2_g16_bwd_kernel's gate gradients in round 5.  LDS reads and memory
loads that land in the data registers were never wrong at any distance (their results return long after the store has read).

The RULE this file enforces (tests/test_host_cpu.py, library-wide): no VALU / MFMA write of a wide store's data registers
within UNSAFE_SLOTS = 2 issue slots (one slot of margin over the measured failure of the SGPR form).

usage: check_store_hazard.py [--window N] [so]    lists every wide store followed, within N instructions of straight-line code
(default 8; s_nop k counts as k + 1), by a write into its data registers, and says which of them break the rule."""
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(HERE, "fullsubnet_amd", "libfsn_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
UNSAFE_SLOTS = 2  # measured: wrong at D = 1 (SGPR soffset) / D <= 2 (immediate soffset, global); never at D >= 3
STORE = re.compile(r"^(buffer|global|flat|scratch)_store_dwordx[34]\b")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)\b")


def regs(tok):
    m = REG.fullmatch(tok.strip().rstrip(","))
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def store_data(op, args):
    """Data operand of a store: buffer_store: first operand; global / flat: second; scratch_store: second."""
    parts = [a.strip() for a in args.split(",")]
    if op.startswith("buffer_"):
        return regs(parts[0])
    return regs(parts[1]) if len(parts) > 1 else set()


def written(op, args):
    """Vector registers an instruction writes (destination = first operand of VALU / MFMA / loads / ds_read)."""
    if op.startswith(("s_", "buffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store", "v_cmp",
                      "v_nop", "buffer_wbl2", "buffer_inv", "global_atomic", "buffer_atomic", "ds_add", "ds_max", "ds_min")):
        if op.startswith("v_cmpx"):
            return set()
        return set()
    if op.startswith(("v_", "buffer_load", "global_load", "flat_load", "scratch_load", "ds_read", "ds_load", "ds_bpermute",
                      "ds_permute", "ds_swizzle")):
        if "_lds_" in op or op.endswith("_lds"):
            return set()
        first = args.split(",")[0]
        return regs(first)
    return set()


def code_objects(so):
    data = open(so, "rb").read()
    for k, m in enumerate(re.finditer(b"\x7fELF", data)):
        i = m.start()
        if i == 0:
            continue
        e_shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
        e_shentsize, e_shnum = struct.unpack_from("<HH", data, i + 0x3A)
        yield data[i:i + e_shoff + e_shentsize * e_shnum]


def scan(so=SO, window=8):
    """[(kernel, store line, distance in issue slots, overwriting line)]"""
    hits = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, blob in enumerate(code_objects(so)):
            path = os.path.join(tmp, f"co{k}.elf")
            with open(path, "wb") as f:
                f.write(blob)
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
            flat = []
            kernel = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    kernel = m.group(1)
                    continue
                m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(//.*)?$", line)
                if m and kernel is not None:
                    flat.append((m.group(1), m.group(2), kernel))
            for i, (op, args, kern) in enumerate(flat):
                if not STORE.match(op):
                    continue
                data = store_data(op, args)
                if not data:
                    continue
                slots = 0
                for op2, args2, kern2 in flat[i + 1:i + 1 + 4 * window]:
                    if kern2 != kern or op2.startswith(("s_branch", "s_endpgm", "s_barrier", "s_setpc")):  # a conditional branch: its fall-through
                        break
                    slots += (int(args2.split()[0]) + 1) if op2 == "s_nop" and args2 else 1
                    if slots > window:
                        break
                    if written(op2, args2) & data:
                        hits.append((kern, f"{op} {args}", slots, f"{op2} {args2}"))
                        break
    return hits


def unsafe(hits):
    """The hits that break the rule: a vector-ALU / matrix instruction writing the data registers within UNSAFE_SLOTS."""
    return [h for h in hits if h[2] <= UNSAFE_SLOTS and h[3].startswith("v_")]


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    window = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else 8
    if "--window" in sys.argv:
        a = [x for x in a if x != str(window)]
    hits = scan(a[0] if a else SO, window)
    for kern, st, d, ow in hits:
        print(f"{kern[-70:]}: `{st}` data written again {d} slot(s) later by `{ow}`")
    print(f"{len(hits)} store(s) whose data registers are written again within {window} issue slots; "
          f"{len(unsafe(hits))} of them by a vector instruction within {UNSAFE_SLOTS} (the measured unsafe distance)")
    sys.exit(1 if unsafe(hits) else 0)
