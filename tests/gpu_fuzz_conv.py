"""Randomised wrap-padded conv3d shapes against the oracle (script): python tests/gpu_fuzz_conv.py [n] [seed]"""
import sys
import torch
sys.path.insert(0, '.')
from oracle import cost_volume_oracle as O
from panogrf_amd import cost_reg as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g))
worst = 0.0
for it in range(n):
    cin = [4, 8, 16, 32, 64, 96][ri(0, 6)]
    cout = [4, 8, 16, 32, 64, 1][ri(0, 6)]
    stride = ri(1, 3)
    B, D, H, W = ri(1, 3), ri(1, 10), ri(1, 9), ri(2, 20)
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5, 'c.bias': torch.randn(cout, generator=g) * 0.1}
    conv = O._conv3(w, 'c', x, stride)
    use_res = bool(ri(0, 2)) and cout != 1
    res = torch.randn(conv.shape, generator=g) if use_res else None
    ref = torch.nn.functional.leaky_relu(conv, 0.01) + (res if use_res else 0)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    xin = R._Vol(R._to_ndhwc(x.cuda()), 0, cin)
    ns = R._small_shift(pc, xin, stride)
    Do, Ho, Wo = ref.shape[2:]
    resv = R._Vol(res.cuda().permute(0, 2, 3, 4, 1).contiguous(), 0, cout) if use_res else None
    out = R._conv(pc, xin, R._Vol.empty(B, Do, Ho, Wo, cout, 'cuda'), stride=stride, residual=resv)
    got = out.buf.permute(0, 4, 1, 2, 3).cpu()
    e = float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    worst = max(worst, e)
    print(f'{it:3d} cin={cin:3d} cout={cout:3d} s={stride} B={B} D={D} H={H} W={W:2d} res={int(use_res)} ns={ns}  err {e:.2e}' + ('' if e < 2e-5 else '  <-- CHECK'))
print('worst', worst)
# shapes the Winograd kernel takes (Cin % 32 = 0, Cout % 64 = 0, even D and H, W % 64 = 0, stride 1, no residual)
worst_w = 0.0
for it in range(max(4, n // 3)):
    cin = [32, 64, 96, 128][ri(0, 4)]
    cout = [64, 128][ri(0, 2)]
    B, D, H, W = ri(1, 3), 2 * ri(1, 5), 2 * ri(1, 5), 64 * ri(1, 3)
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = {'c.weight': torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5, 'c.bias': torch.randn(cout, generator=g) * 0.1}
    ref = torch.nn.functional.leaky_relu(O._conv3(w, 'c', x, 1), 0.01)
    pc = R._PackedConv(w['c.weight'], w['c.bias'], 'cuda')
    out = R._conv(pc, R._Vol(R._to_ndhwc(x.cuda()), 0, cin), R._Vol.empty(B, D, H, W, cout, 'cuda'), stride=1)
    e = float((out.buf.permute(0, 4, 1, 2, 3).cpu() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    worst_w = max(worst_w, e)
    print(f'wino {it:3d} cin={cin:3d} cout={cout:3d} B={B} D={D} H={H} W={W:3d}  err {e:.2e}' + ('' if e < 2e-5 else '  <-- CHECK'))
print('worst winograd', worst_w)
