"""Accuracy of a split-precision (fp16 x 3 / bf16 x 3) replacement for the fp32 GEMMs of the path, emulated on
the CPU: every matmul of the oracle is replaced by a_hi b_hi + a_hi b_lo + a_lo b_hi with 16-bit halves and fp32
accumulation, and the compressed mask is compared with the reference's golden output.  Evidence for DESIGN 9 (split precision);
not part of the product or of the test suite (pytest does not collect it).  Run from the repo root:
python tests/experiments/emulate_f16x3.py"""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import fullsubnet_oracle as O
def split(x):
    x=np.asarray(x,np.float32)
    hi=x.astype(np.float16)
    lo=(x-hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)
def mm3(a,b):   # a [M,K], b [K,N] fp32 -> emulated fp16x3 with fp32 accumulation
    ah,al=split(a); bh,bl=split(b)
    t=lambda u,v: torch.from_numpy(np.ascontiguousarray(u)) @ torch.from_numpy(np.ascontiguousarray(v))
    return (t(ah,bh)+t(ah,bl)+t(al,bh)).numpy()
def mm_bf3(a,b):
    def sp(x):
        x=torch.from_numpy(np.ascontiguousarray(x,dtype=np.float32)); hi=x.bfloat16().float(); lo=(x-hi).bfloat16().float(); return hi,lo
    ah,al=sp(a); bh,bl=sp(b)
    return (ah@bh+ah@bl+al@bh).numpy()
z=np.load(os.path.join(ROOT, 'tests', 'golden', 'fsn_offline_b2.npz')); meta=ast.literal_eval(str(z['meta']))
params=O.make_params(seed=meta['seed_w'],gain=meta['gain'],mask_gain=meta['mask_gain'])
base=O.fullsubnet_forward(z['mag'][:,None],params)
print("fp32 oracle vs reference golden:", np.abs(base-z['crm']).max())
for name,fn in (("fp16x3",mm3),("bf16x3",mm_bf3)):
    O._matmul=fn
    got=O.fullsubnet_forward(z['mag'][:,None],params)
    print(name,"vs reference golden:", np.abs(got-z['crm']).max(), " vs fp32 oracle:", np.abs(got-base).max(), "mask range", np.abs(z['crm']).max())
