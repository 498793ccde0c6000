"""Diagnosis builds: libfsn_hip.so with ONE translation unit recompiled under extra -D flags, linked against the shipped
objects into tools/bin/<name>.so (git-ignored, travels to the GPU box).  usage: build_variant.py NAME TU.hip -DX=1 ...
Load it with `fullsubnet_amd._lib.LIB_PATH = ...` before the first call (tools/diag_k32_bwd.py --lib)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from fullsubnet_amd import build as b  # noqa: E402


def main():
    name, tu, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    b.build(force=False, verbose=False)
    out = os.path.join(HERE, "tools", "bin")
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, f"{name}_{os.path.splitext(tu)[0]}.o")
    subprocess.run(["/opt/rocm/bin/hipcc"] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, tu), "-o", obj], check=True)
    objs = [obj if s == tu else b._obj(s) for s in b._sources()]
    so = os.path.join(out, f"{name}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", so], check=True)
    print(so)


if __name__ == "__main__":
    main()
