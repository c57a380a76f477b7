import torch, time
x = torch.empty(512*1024*1024, dtype=torch.float32, device="cuda")  # 2 GB
y = torch.empty_like(x)
for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("read(sum)", lambda: x.sum())):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = x.numel() * 4 / 1e9 * (2 if name == "copy" else 1)
    print("%s: %.3f ms  %.2f TB/s" % (name, ms, gb / ms))
