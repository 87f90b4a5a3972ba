import sys, torch
sys.path.insert(0, "/root/repo")
from protein_clip_amd import kernels as K
B, T, D = 256, 256, 128
h = torch.randn(B, T, D, device="cuda")
mask = (torch.arange(T, device="cuda")[None] < torch.randint(128, T + 1, (B, 1), device="cuda")).to(torch.uint8)
for _ in range(30):
    pooled, inv, m8 = K.masked_mean_fwd(h, mask)
    K.masked_mean_bwd(pooled, m8, inv, T)
torch.cuda.synchronize()
