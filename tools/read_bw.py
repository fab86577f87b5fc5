"""dev tool: read-only HBM rate of plain torch reductions over a 1 GiB buffer (reference point for the pixel pass's streaming floor)"""
import torch
x = torch.randint(0, 255, (256, 2048, 2048), dtype=torch.uint8, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
gb = x.numel() / 1e9
for name, fn in (("f32 sum", lambda: x.view(torch.float32).sum()), ("i32 max", lambda: x.view(torch.int32).max()), ("i64 max", lambda: x.view(torch.int64).max()),
                 ("u8 max", lambda: x.max()), ("copy", lambda: x.clone())):
    us = t(fn)
    print(f"{name}: {us:.1f} us  {gb / us * 1e3 * (2 if name == 'copy' else 1):.2f} TB/s")
