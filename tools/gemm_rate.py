"""Rate of a plain bf16 GEMM (torch.matmul -> hipBLASLt) on this GPU, with random and with all-zero operands: the matrix
cores' sustained rate depends on the data (power), so this is the practical ceiling to set a kernel's rate against."""
import sys, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda"
for name, make in (("random", lambda: torch.randn(n, n, device=dev).to(torch.bfloat16)), ("zeros", lambda: torch.zeros(n, n, device=dev, dtype=torch.bfloat16))):
    a, b = make(), make()
    for _ in range(5): c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"bf16 GEMM {n}^3, {name} operands: {ms:.3f} ms, {2 * n ** 3 / ms * 1e-9:.0f} TFLOP/s")
