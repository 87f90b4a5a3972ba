"""Times the fused FILIP backward (bench shape) with each shared library given on the command
line (ablation builds of libpclip.so), one child process per library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, statistics, torch
sys.path.insert(0, %r)
import protein_clip_amd
from protein_clip_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from protein_clip_amd import kernels as K
B, L, D = 256, 256, 128
g = torch.Generator().manual_seed(0)
hA = torch.randn(B, L, D, generator=g).to(torch.bfloat16).float().cuda()
hB = torch.randn(B, L, D, generator=g).to(torch.bfloat16).float().cuda()
lA = torch.randint(128, L + 1, (B,), generator=g); lB = torch.randint(128, L + 1, (B,), generator=g)
ar = torch.arange(L)[None]
mA = (ar < lA[:, None]).to(torch.uint8).cuda(); mB = (ar < lB[:, None]).to(torch.uint8).cuda()
t = torch.tensor([1.0], device="cuda")
step = K.FilipStep(B, B, L, L, D, "cuda")
for _ in range(20): step.run(hA, hB, mA, mB, t)
lib = _lib.load()
from protein_clip_amd._lib import ptr, stream_of, check
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record()
    check(lib.pclip_filip_bwd_fused(ptr(step.hnA), ptr(step.mwA), B, L, ptr(step.hnB), ptr(step.mwB), B, L, ptr(t), ptr(step.simA), ptr(step.simB),
          ptr(step.argA), ptr(step.argB), ptr(step.gA), ptr(step.gB), ptr(hA), ptr(hB), D, 1, ptr(step.dhA), ptr(step.dhB), ptr(step.dt),
          ptr(step.bws), step.bws.numel(), stream_of(hA)), "bwd")
    b.record()
torch.cuda.synchronize()
print("%%-40s fused backward %%.3f ms" %% (sys.argv[1].split("/")[-1], statistics.median(a.elapsed_time(b) for a, b in ev)))
import ctypes
raw = ctypes.CDLL(sys.argv[1])
if hasattr(raw, "pclip_debug_read"):
    buf = (ctypes.c_ulonglong * 8)()
    raw.pclip_debug_read(buf)      # clear
    check(lib.pclip_filip_bwd_fused(ptr(step.hnA), ptr(step.mwA), B, L, ptr(step.hnB), ptr(step.mwB), B, L, ptr(t), ptr(step.simA), ptr(step.simB),
          ptr(step.argA), ptr(step.argB), ptr(step.gA), ptr(step.gB), ptr(hA), ptr(hB), D, 1, ptr(step.dhA), ptr(step.dhB), ptr(step.dt),
          ptr(step.bws), step.bws.numel(), stream_of(hA)), "bwd")
    torch.cuda.synchronize()
    raw.pclip_debug_read(buf)
    steps = buf[7]     # workgroup-steps (both sides)
    names = ["loop head", "dma issue", "static rows", "extra rows", "wait dma", "barrier"]
    print("   cycles per wave and step: " + ", ".join("%%s %%.0f" %% (n, buf[k] / (16.0 * steps)) for k, n in enumerate(names)))

''' % ROOT

for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr[-1500:], flush=True)
