"""Register / LDS / scratch figures of the kernels INSIDE the built libfsn_hip.so (not of a fresh compile):
usage: so_kernel_resources.py [substring ...]   - what actually ships to the GPU box."""
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(HERE, "fullsubnet_amd", "libfsn_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernels(so=SO):
    data = open(so, "rb").read()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, m in enumerate(re.finditer(b"\x7fELF", data)):
            i = m.start()
            if i == 0:
                continue
            e_shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
            e_shentsize, e_shnum = struct.unpack_from("<HH", data, i + 0x3A)
            path = os.path.join(tmp, f"co{k}.elf")
            with open(path, "wb") as f:
                f.write(data[i:i + e_shoff + e_shentsize * e_shnum])
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                out[name] = {key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
                             for key in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size",
                                         "group_segment_fixed_size", "sgpr_count")}
    return out


if __name__ == "__main__":
    want = sys.argv[1:]
    for name, r in sorted(kernels().items()):
        if not want or any(w in name for w in want):
            print(name[-90:], r)
