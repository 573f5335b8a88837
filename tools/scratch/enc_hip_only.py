import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from panogrf_amd import encoders as EN
from panogrf_amd.encoders_hip import HipEncoders
dev = torch.device('cuda', 0); torch.manual_seed(0)
enc = EN.ResUNetLight().to(dev).eval(); vis = EN.DefaultVisEncoder().to(dev).eval()
imgs = torch.rand(2, 3, 512, 1024, device=dev); rf = torch.randn(2, 32, 64, 128, device=dev)
with torch.no_grad():
    he = HipEncoders(enc, vis, dev)
    for _ in range(6): he(imgs, rf)
torch.cuda.synchronize()
