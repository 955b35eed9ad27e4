"""Does the 256 MB Infinity Cache serve re-reads of data a previous kernel wrote?
Times (device events) a read-only pass, a write-only pass and a read-modify-write
pass over buffers of growing size, each repeated back to back on the same buffer."""
import torch, sys
dev = torch.device('cuda', 0)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
print('%8s %12s %12s %12s %14s' % ('MB', 'read GB/s', 'write GB/s', 'rmw GB/s(r+w)', 'copy GB/s(r+w)'))
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 8
    a = torch.randn(n, dtype=torch.float64, device=dev)
    b = torch.empty_like(a)
    rd = t(lambda: torch.sum(a))
    wr = t(lambda: a.fill_(1.5))
    rmw = t(lambda: a.mul_(1.0000001))
    cp = t(lambda: b.copy_(a))
    by = n * 8 / 1e9
    print('%8d %12.0f %12.0f %12.0f %14.0f' % (mb, by / rd, by / wr, 2 * by / rmw, 2 * by / cp), flush=True)
