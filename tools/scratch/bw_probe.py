import torch, time
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (67, 268, 1072):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device='cuda'); b = torch.randn(n, device='cuda')
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: a.copy_(b)); tr = t(lambda: b.sum())
    print(f'{mb} MB: fill {n*4/tf/1e12:.2f} TB/s ({tf*1e6:.0f} us)  copy {2*n*4/tc/1e12:.2f} TB/s rd+wr ({tc*1e6:.0f} us)  read(sum) {n*4/tr/1e12:.2f} TB/s')
