"""Per-segment cycle counts of the FILIP score kernel's main loop (diagnostic build of the library with
-DSTREAM_STAMPS, see csrc/filip_stream.hip), benchmark shape: where a wave's time goes, by wave
position in the workgroup.  Usage: python3 tools/stream_stamps.py <library.so> [ragged|full]"""
import ctypes
import os
import sys

import torch

lib_path = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import _lib, kernels as K  # noqa: E402

_lib.LIB_PATH = lib_path
full = len(sys.argv) > 2 and sys.argv[2] == "full"
B, L, D = 256, 256, 128
g = torch.Generator().manual_seed(0)
torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
hA = torch.randn(B, L, D, generator=g).to(torch.bfloat16).float().cuda()
hB = torch.randn(B, L, D, generator=g).to(torch.bfloat16).float().cuda()
lA = torch.randint(128, L + 1, (B,), generator=g)
lB = torch.randint(128, L + 1, (B,), generator=g)
if full:
    lA[:] = L
    lB[:] = L
ar = torch.arange(L)[None]
mA = (ar < lA[:, None]).to(torch.uint8).cuda()
mB = (ar < lB[:, None]).to(torch.uint8).cuda()
t = torch.tensor([1.0], device="cuda")
hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
hnB, mwB = K.filip_prep(hB, mB, True)
lib = _lib.load()
slot = ctypes.c_void_p.in_dll(lib, "g_stream_dbg")
NB = 4096
dbg = torch.zeros(NB * 16 * 8, dtype=torch.int64, device="cuda")
for i in range(30):
    K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=True, unit_norm=True)
torch.cuda.synchronize()
slot.value = dbg.data_ptr()
K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=True, unit_norm=True)
torch.cuda.synchronize()
slot.value = None
v = dbg.cpu().view(NB, 16, 8).double()
used = v.sum((1, 2)) > 0
v = v[used]
names = ["copies issued", "tile records", "matrix chain", "keys, folds, atomic", "row epilogue", "waits before barrier", "barrier", "duties, emission"]
tot = v.sum(2)                      # [block][wave]
print("blocks %d; cycles per wave (median over blocks), share of each segment by wave position" % v.shape[0])
for w in (0, 3, 7, 11, 12, 15):
    sel = tot[:, w] > 0
    if not sel.any():
        continue
    share = (v[sel, w] / tot[sel, w, None]).median(0).values
    print("wave %2d (%5d blocks): total %8.0f  " % (w, int(sel.sum()), float(tot[sel, w].median())) + "  ".join("%s %4.1f%%" % (n, 100 * float(s)) for n, s in zip(names, share)))
allw = v.sum((0, 1))
print("all waves:", "  ".join("%s %4.1f%%" % (n, 100 * float(s / allw.sum())) for n, s in zip(names, allw)))
