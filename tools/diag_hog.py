"""Diagnostic: the constant-rate counter's frequency as HIP reports it and as a timed hog kernel shows it, and whether
a hog that owns every CU's LDS really delays a kernel of another stream."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd as fsn  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
v = ctypes.c_int(0)
rc = hip.hipDeviceGetAttribute(ctypes.byref(v), 10017, 0)
print("hipDeviceAttributeWallClockRate rc", rc, "value", v.value, "kHz")
L = fsn._lib.lib()
sink = torch.zeros(1, device="cuda")
side = torch.cuda.Stream()


def hog(wgs, lds, heavy, ms, stream):
    fsn._lib.check(L.fsn_debug_hog(wgs, lds, heavy, float(ms), fsn._lib.dev_ptr(sink), stream.cuda_stream))


for wgs, lds, heavy in [(8, 1024, 0), (256, 160 * 1024, 0), (256, 64 * 1024, 1), (512, 32 * 1024, 1)]:
    for ms in (5.0, 20.0):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            a.record()
            hog(wgs, lds, heavy, ms, side)
            b.record()
        b.synchronize()
        print(f"hog {wgs} wgs x {lds // 1024} KB heavy={heavy} asked {ms} ms: measured {a.elapsed_time(b):.2f} ms")

# does it delay a kernel of another stream that needs LDS?  (an LDS-using torch kernel: softmax over rows)
x = torch.randn(4096, 4096, device="cuda")
torch.cuda.synchronize()
for label, launch_hog in (("no hog", False), ("hog 256 x 160 KB for 50 ms", True)):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if launch_hog:
        hog(256, 160 * 1024, 0, 50.0, side)
    a.record()
    y = torch.softmax(x, dim=1)
    z = torch.stft(x[0], 512, 256, return_complex=True) if False else None
    b.record()
    b.synchronize()
    side.synchronize()
    print(f"{label}: softmax on the default stream took {a.elapsed_time(b):.2f} ms")
