"""Sustained HBM write / copy bandwidth of this GPU with plain torch kernels (the ceiling the full-output rollout is priced against)."""
import torch
dev = torch.device("cuda:0")
for gb in (1, 4):
    n = gb * (1 << 30) // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn, nbytes in (("fill", lambda: x.fill_(1.0), 4 * n), ("zero", lambda: x.zero_(), 4 * n),
                             ("copy", lambda: y.copy_(x), 8 * n)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name} {gb} GiB: {ms*1e3:.1f} us  {nbytes / ms / 1e9:.2f} TB/s", flush=True)
