import torch, time
for mb in (224, 448, 1024, 4096):
    x = torch.empty(mb * 1024 * 1024 // 4, device="cuda", dtype=torch.float32)
    for _ in range(3): x.zero_()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): x.zero_()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("zero_ %5d MB: %.1f us  %.2f TB/s" % (mb, ms * 1e3, mb * 1.048576e6 / ms / 1e9))
    y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    a.record()
    for _ in range(20): y.copy_(x)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("copy  %5d MB: %.1f us  %.2f TB/s (read + write)" % (mb, ms * 1e3, 2 * mb * 1.048576e6 / ms / 1e9))
