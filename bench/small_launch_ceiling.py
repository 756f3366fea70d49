"""What a launch as small as the secondary kernels' can reach on this chip: library copy / fill / reduction over the same
number of bytes (config 3: Z = 0.417 GB), HIP events, median of 30.  Context for profiles/r02_secondary_kernels_config3.txt.
   python bench/small_launch_ceiling.py"""
import numpy as np
import torch

def t_ms(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

n = 65536 * 795
x = torch.randn(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
big = torch.empty(65536 * 11740, dtype=torch.float64, device="cuda")  # the Jacobian values of config 3: 6.2 GB
for name, fn, nbytes in (("copy 0.417 GB -> 0.417 GB (grad_f's traffic)", lambda: y.copy_(x), 16.0 * n),
                         ("sum over 0.417 GB (eval_f's traffic)", lambda: torch.sum(x), 8.0 * n),
                         ("fill 0.417 GB", lambda: y.fill_(1.0), 8.0 * n),
                         ("scale in place 0.417 GB (read + write)", lambda: x.mul_(1.0000001), 16.0 * n),
                         ("fill 6.16 GB (the hot launch's store volume)", lambda: big.fill_(1.0), 8.0 * big.numel())):
    ms = t_ms(fn)
    print(f"{name:55s} {ms:7.4f} ms  {nbytes / ms / 1e6:7.0f} GB/s = {nbytes / ms / 1e6 / 80:5.1f} % of 8 TB/s")
