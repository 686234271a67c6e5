"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the encoder's GEMM shapes -- B=256 bf16, or B=64 fp32 with
`fp32` as the argument: a yardstick for the kernels' targets, not part of the product.  usage: exp_gemm_ref.py [fp32]"""
import sys, torch, time
dev = torch.device("cuda", 0)
FP32 = len(sys.argv) > 1 and sys.argv[1] == "fp32"
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("layer2 3x3", 200704, 128, 1152), ("layer3 3x3", 50176, 256, 2304), ("layer4 3x3", 12544, 512, 4608),
          ("layer3 conv1", 50176, 256, 1024), ("layer3 conv3", 50176, 1024, 256), ("layer4 conv1", 12544, 512, 2048),
          ("layer4 conv3", 12544, 2048, 512), ("layer2 conv1", 200704, 128, 512), ("layer2 conv3", 200704, 512, 128)]
for name, M, N, K in shapes:
    if FP32:
        M //= 4                      # B = 64
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    else:
        a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(5): y = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): y = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{name:14s} M={M:6d} N={N:4d} K={K:4d}  {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF", flush=True)
