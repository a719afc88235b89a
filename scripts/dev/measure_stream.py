"""Dev measurement (GPU): what plain streaming kernels reach on this chip (torch elementwise ops, fp32): copy, a+b, a+b+c"""
import torch
n = 64 * 1024 * 1024            # 256 MB per tensor
a, b, c, o = (torch.randn(n, device="cuda") for _ in range(4))


def t(fn, nbytes, name, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name:28s} {ms*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s")


t(lambda: o.copy_(a), 8 * n, "copy (1R 1W)")
t(lambda: torch.add(a, b, out=o), 12 * n, "a+b (2R 1W)")
t(lambda: torch.addcmul(a, b, c, out=o), 16 * n, "a+b*c (3R 1W)")
t(lambda: a.sum(), 4 * n, "sum (1R)")
t(lambda: o.fill_(1.0), 4 * n, "fill (1W)")
