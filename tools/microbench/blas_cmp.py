import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
M = 48000
for name, N, K in (("qkv", 3840, 1280), ("out", 1280, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120)):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): torch.nn.functional.linear(a, w, b)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"torch/hipBLASLt {name}: {us:.0f} us  {2*M*N*K/us/1e6:.0f} TF/s (bias only, bf16 out)")
