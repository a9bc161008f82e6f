import torch, time
dev = torch.device("cuda", 0)
x = torch.empty(3_000_000_000, dtype=torch.float64, device=dev).fill_(1.0)   # 24 GB
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = timed(lambda: x.sum())
print("torch sum 24 GB read: %.3f ms  %.0f GB/s" % (ms, 24e9 / ms / 1e6))
y = torch.empty_like(x[:1_500_000_000])
ms = timed(lambda: y.copy_(x[:1_500_000_000]))
print("torch copy 12 GB -> 12 GB: %.3f ms  %.0f GB/s (read+write)" % (ms, 24e9 / ms / 1e6))
i8 = x.view(torch.int64)
ms = timed(lambda: (i8 < 0).any())
print("compare+any 24 GB read: %.3f ms  %.0f GB/s" % (ms, 24e9 / ms / 1e6))
