"""Which vendor-library kernels (Tensile names encode macro tile / pipeline) run the C2 GEMM shapes.  Diagnostic only."""
import torch, torch.nn.functional as F
shapes = [("c2 llm gate|up", 3361, 28672, 4096), ("c2 llm down", 3361, 4096, 14336), ("c2 llm qkv", 3361, 6144, 4096),
          ("hiera s3 fc1", 65536, 2304, 576), ("hiera s3 fc2", 65536, 576, 2304), ("clip fc2", 9232, 1024, 4096), ("iv2 fc2", 4100, 1408, 6144), ("square 8k", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): F.linear(a, w)
    torch.cuda.synchronize()
