"""Timeline of the FILIP score kernel's workgroups (diagnostic build of the library with
-DSTREAM_TIMELINE, see csrc/filip_stream.hip): for every CU the sequence of workgroups it ran, how
long their set-up and main loop took, how far apart the waves of a workgroup finish and how long the
CU sat between two workgroups.  Usage: python3 tools/stream_timeline.py <library.so> [ragged|full] [wgs]"""
import ctypes
import os
import sys
from collections import defaultdict

import torch

lib_path = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import _lib, kernels as K  # noqa: E402

_lib.LIB_PATH = lib_path   # the diagnostic build instead of the shipped library

full = len(sys.argv) > 2 and sys.argv[2] == "full"
wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
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
slot = ctypes.c_void_p.in_dll(lib, "g_stream_tl")
NB = 4096
tl = torch.zeros(NB * 32, dtype=torch.int64, device="cuda")
for i in range(40):   # warm: clocks settle
    K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=True, unit_norm=True, stream_wgs=wgs)
torch.cuda.synchronize()
slot.value = tl.data_ptr()
K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=True, unit_norm=True, stream_wgs=wgs)
torch.cuda.synchronize()
slot.value = None
v = tl.cpu().view(NB, 32)
v = v[v[:, 0] != 0]
n = v.shape[0]
t0 = int(v[:, 0].min())
us = lambda x: (x - t0) / 100.0   # 100 MHz counter
cus = defaultdict(list)
for b in range(n):
    r = v[b]
    ends = [int(x) for x in r[4:20] if x != 0]
    key = (int(r[22]) & 0xF, (int(r[21]) >> 8) & 0xFF)
    cus[key].append(dict(b=b, start=us(int(r[0])), pair=us(int(r[1])), setup=us(int(r[2])), loop=us(int(r[3])),
                         first_end=us(min(ends)), last_end=us(max(ends)), end=us(int(r[20]))))
print("workgroups", n, "CUs", len(cus), "kernel span %.1f us" % max(w["end"] for ws in cus.values() for w in ws))
import statistics as st
allw = [w for ws in cus.values() for w in ws]
print("set-up (entry -> first epoch landed): median %.1f us (pairing %.1f, stream + A tiles %.1f, first copies %.1f)" % (
    st.median(w["loop"] - w["start"] for w in allw), st.median(w["pair"] - w["start"] for w in allw),
    st.median(w["setup"] - w["pair"] for w in allw), st.median(w["loop"] - w["setup"] for w in allw)))
print("main loop (first wave out): median %.1f us; waves of a workgroup finish %.1f us apart (median), %.1f (max); tail after the last wave %.1f us" % (
    st.median(w["first_end"] - w["loop"] for w in allw), st.median(w["last_end"] - w["first_end"] for w in allw),
    max(w["last_end"] - w["first_end"] for w in allw), st.median(w["end"] - w["last_end"] for w in allw)))
gaps, idle_tail, per_cu = [], [], []
span = max(w["end"] for w in allw)
for key, ws in cus.items():
    ws.sort(key=lambda w: w["start"])
    for a, b in zip(ws, ws[1:]):
        gaps.append(b["start"] - a["end"])
    idle_tail.append(span - ws[-1]["end"])
    per_cu.append(len(ws))
print("workgroups per CU: min %d max %d; gap between two workgroups on a CU: median %.1f us, max %.1f" % (
    min(per_cu), max(per_cu), st.median(gaps) if gaps else 0.0, max(gaps) if gaps else 0.0))
print("first workgroup of a CU starts at: median %.1f us, max %.1f; idle at the end of the kernel: mean %.1f us, median %.1f, max %.1f" % (
    st.median(ws[0]["start"] for ws in cus.values()), max(ws[0]["start"] for ws in cus.values()),
    sum(idle_tail) / len(idle_tail), st.median(idle_tail), max(idle_tail)))
busy = sum(w["end"] - w["start"] for w in allw)
print("CU-time: %.1f %% inside workgroups, of which set-up %.1f %%, loop %.1f %%, wave skew + tail %.1f %%" % (
    100 * busy / (span * len(cus)), 100 * sum(w["loop"] - w["start"] for w in allw) / busy,
    100 * sum(w["first_end"] - w["loop"] for w in allw) / busy, 100 * sum(w["end"] - w["first_end"] for w in allw) / busy))
durs = sorted(w["end"] - w["start"] for w in allw)
print("workgroup durations (us): min %.0f p10 %.0f p25 %.0f median %.0f p75 %.0f p90 %.0f max %.0f, mean %.0f" % (
    durs[0], durs[len(durs) // 10], durs[len(durs) // 4], durs[len(durs) // 2], durs[3 * len(durs) // 4], durs[9 * len(durs) // 10], durs[-1],
    sum(durs) / len(durs)))
tot = sorted(sum(w["end"] - w["start"] for w in ws) for ws in cus.values())
print("busy time per CU (us): min %.0f median %.0f max %.0f, mean %.0f" % (tot[0], tot[len(tot) // 2], tot[-1], sum(tot) / len(tot)))
hist = defaultdict(int)
for d in durs:
    hist[int(d // 25) * 25] += 1
print("duration histogram (25-us bins):", dict(sorted(hist.items())))
if os.environ.get("TL_DUMP"):
    for key in sorted(cus)[:40]:
        print("xcc %d se/sh/cu %02x:" % key, "  ".join("b%d[%d..%d]" % (w["b"], w["start"], w["end"]) for w in cus[key]))
mhz = [(int(r[24]) - int(r[23])) / ((int(r[20]) - int(r[0])) / 100.0) for r in v if int(r[20]) > int(r[0]) and int(r[24]) > int(r[23])]
if mhz:
    print("shader clock while the kernel runs (cycle counter over the 100-MHz counter, per workgroup): median %.0f MHz, min %.0f, max %.0f" % (
        st.median(mhz), min(mhz), max(mhz)))
