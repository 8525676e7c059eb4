"""Developer script (GPU box, under rocprofv3 --kernel-trace): which hipBLASLt kernels torch.matmul picks for the encoder shapes
(the yard-stick's tile / wave geometry is in the kernel name; VGPR / LDS sizes are in the trace)."""
import torch
dev = torch.device("cuda:0")
for (M, N, K) in ((102000, 1280, 1280), (102000, 3840, 1280), (102000, 5120, 1280), (102000, 1280, 5120)):
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
