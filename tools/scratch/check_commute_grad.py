"""Isolate the 1e-3 gradient difference of the commuted (1x1 conv -> x2 bilinear) pair on odd grids: both orders through the HIP autograd
Functions (fp32 mode) against torch fp64 autograd of the reference order."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from uniception_amd import engine

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, H, W, C, crop) in [(2, 5, 7, 32, None), (2, 5, 7, 32, (9, 13)), (2, 10, 14, 32, None), (1, 3, 3, 64, None)]:
    conv = torch.nn.Conv2d(C, C, 1).to(dev)
    x0 = torch.randn(B, H, W, C, device=dev)
    Ho, Wo = 2 * H, 2 * W
    ch, cw = crop if crop else (Ho, Wo)
    wgt = torch.randn(B, ch, cw, C, device=dev)
    res = {}
    for order in ("ref", "commuted"):
        x = x0.clone().requires_grad_(True)
        conv.zero_grad()
        with engine.precision("fp32"):
            if order == "ref":
                y = engine.conv1x1(engine.bilinear(x, Ho, Wo, crop), conv)
            else:
                y = engine.bilinear(engine.conv1x1(x, conv), Ho, Wo, crop)
        (y * wgt).sum().backward()
        res[order] = (y.detach().double(), x.grad.double(), conv.weight.grad.double().clone(), conv.bias.grad.double().clone())
    xd = x0.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    wd = conv.weight.detach().double().clone().requires_grad_(True)
    bd = conv.bias.detach().double().clone().requires_grad_(True)
    up = F.interpolate(xd, size=(Ho, Wo), mode="bilinear", align_corners=True)[:, :, :ch, :cw]
    yd = F.conv2d(up, wd, bd)
    (yd * wgt.double().permute(0, 3, 1, 2)).sum().backward()
    want = (yd.detach().permute(0, 2, 3, 1), xd.grad.permute(0, 2, 3, 1), wd.grad, bd.grad)
    rl = lambda a, b: float((a - b.reshape(a.shape)).norm() / b.norm())
    for order in ("ref", "commuted"):
        print((B, H, W, C, crop), order, "y %.1e dx %.1e dW %.1e db %.1e" % tuple(rl(a, b) for a, b in zip(res[order], want)))
