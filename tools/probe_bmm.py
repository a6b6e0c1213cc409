"""dev: what the library's batched fp32 GEMM reaches on the Winograd-domain shapes of the discriminator's stride-1 layers."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
for (P, M, K, N) in [(16, 4608, 128, 256), (16, 18432, 64, 128), (16, 1152, 256, 512), (16, 2304, 128, 256), (16, 4608, 256, 128), (16, 18432, 128, 64),
                     (1, 16 * 4608, 128, 256)]:
    a = torch.randn(P, M, K, device=dev)
    b = torch.randn(P, K, N, device=dev)
    for _ in range(3):
        c = torch.bmm(a, b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            c = torch.bmm(a, b)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 2.0 * P * M * K * N
    print(f"bmm P={P} M={M} K={K} N={N}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s   (direct conv = {2.25 * fl / 1e9:.2f} GFLOP)")
