import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
import numpy as np, torch
from mask_cyclegan_vc._hip import check, lib, ptr, stream
L = lib()
Cin, Cout, N, H, W = 128, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 1, 80, 64
spec = (Cin, Cout, 1, 3, 3, 2, 1, 1)
g = torch.Generator().manual_seed(11)
x = torch.randn(N, Cin, H, W, generator=g)
OH, OW = H // 2, W // 2
dy = torch.randn(N, Cout, OH, OW, generator=g)
scratch = torch.zeros(L.mcvc_layer_scratch_floats(N, H, W, *spec), device="cuda")
dw = torch.zeros(Cout, Cin, 3, 3, device="cuda")
xd, dyd = x.cuda(), dy.cuda()
check(L.mcvc_layer_wgrad(ptr(xd), ptr(dyd), ptr(dw), None, ptr(scratch), scratch.numel(), N, H, W, *spec, 5, stream()), "wgrad")
ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, stride=2, padding=1)
d = dw.cpu()
print("rel", float((d - ref).norm() / ref.norm()), "ratio of norms", float(d.norm() / ref.norm()))
for t in range(9):
    a, b = d[:, :, t // 3, t % 3], ref[:, :, t // 3, t % 3]
    print("tap", t, float((a - b).norm() / b.norm()), float((a * b).sum() / (b * b).sum()))
for cb in range(0, Cout, 64):
    a, b = d[cb:cb + 64], ref[cb:cb + 64]
    print("co block", cb, float((a - b).norm() / b.norm()))
