"""Print which library kernel torch.matmul dispatches for a bf16 NT GEMM (run under rocprofv3 --kernel-trace)."""
import sys
import torch
M, N, K = (int(x) for x in sys.argv[1:4])
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
for _ in range(3):
    torch.matmul(a, w.t())
torch.cuda.synchronize()
