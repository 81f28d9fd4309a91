"""Raw HBM rates of the box with torch's own kernels on a buffer the size of the layer-1 activation (0.956 GB): the
reference points for the HBM-bound kernels in DESIGN.md section 3 (measured: fill 6.87 TB/s, copy 5.22 TB/s read + write)."""
import torch, time
n = 256*3648*512
x = torch.empty(n, device="cuda", dtype=torch.bfloat16)
y = torch.empty(n, device="cuda", dtype=torch.bfloat16)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
gb = n * 2 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f"fill  {gb:.3f} GB  {ms:.4f} ms  {gb/ms:.2f} TB/s write")
ms = t(lambda: x.zero_()); print(f"zero  {gb:.3f} GB  {ms:.4f} ms  {gb/ms:.2f} TB/s write")
ms = t(lambda: y.copy_(x)); print(f"copy  {gb:.3f} GB  {ms:.4f} ms  {2*gb/ms:.2f} TB/s read+write")
ms = t(lambda: x.sum()); print(f"sum   {gb:.3f} GB  {ms:.4f} ms  {gb/ms:.2f} TB/s read")
