import torch, time
for mb in (64, 256, 604, 1208):
    n = mb * 1024 * 1024 // 4
    x = torch.rand(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"copy {mb} MB: {ms*1e3:.1f} us  -> {2*mb*1.048576/ms:.1f} GB/s (read+write)")
    e0.record()
    for _ in range(10): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"read {mb} MB: {ms*1e3:.1f} us  -> {mb*1.048576/ms:.1f} GB/s")
    e0.record()
    for _ in range(10): y.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"write {mb} MB: {ms*1e3:.1f} us  -> {mb*1.048576/ms:.1f} GB/s")
