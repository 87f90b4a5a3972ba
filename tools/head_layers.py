import sys, os, torch
sys.path.insert(0, "/root/repo")
from protein_clip_amd import kernels as K
def timed(fn, n=300):
    for _ in range(30): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, d = 256, 128
x = torch.randn(B, d, device="cuda"); t = torch.ones(1, device="cuda"); w = torch.randn(B, d, device="cuda")
for nblk in (0, 1, 2, 4):
    P = []
    for l in range(nblk):
        P += [torch.randn(d, d, device="cuda") / 11, torch.zeros(d, device="cuda"), torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")]
    P += [torch.randn(d, d, device="cuda") / 11, torch.zeros(d, device="cuda")]
    out, save = K.head_tail_fwd(x, P, None, 1.0, 1e-5, t)
    f = timed(lambda: K.head_tail_fwd(x, P, None, 1.0, 1e-5, t))
    b = timed(lambda: K.head_tail_bwd(x, P, None, 1.0, 1e-5, t, save, w))
    print(f"nblk={nblk}: fwd call {f:.1f} us, bwd call {b:.1f} us (host wrappers included)", flush=True)
