import torch, time
shapes = [(4096, 3840, 1280, 'sam qkv'), (4096, 1280, 1280, 'sam proj'), (4096, 5120, 1280, 'sam fc1'), (4096, 1280, 5120, 'sam fc2'),
          (320, 12288, 4096, 'llm qkv'), (320, 4096, 4096, 'llm o'), (320, 22016, 4096, 'llm gate/up'), (320, 4096, 11008, 'llm down'),
          (577, 3072, 1024, 'clip qkv'), (577, 4096, 1024, 'clip fc1'), (8192, 8192, 8192, 'square')]
for M, N, K, name in shapes:
    a = torch.randn(M, K, device='cuda', dtype=torch.bfloat16); w = torch.randn(N, K, device='cuda', dtype=torch.bfloat16)
    for _ in range(5): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): torch.nn.functional.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:12s} M={M:5d} N={N:6d} K={K:6d}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
