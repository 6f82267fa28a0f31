"""Same-binary A/B of the eight-wave row kernel's FLAT form against the implicit-GEMM kernel on maps whose rows do not tile 512 pixels
(knob conv_rows_flat 1 / 0): the DINOv2-518 head's widths (148 / 296 / 592) and the 224 x 224 head's (56 / 112 / 224), fp16 operands
(the heads' default), plain / ReLU-on-load + bias + ReLU / residual epilogues.  TFLOP/s = 2 * M * 9 Cin * Cout / time."""
import sys
import time

import torch

sys.path.insert(0, ".")
from uniception_amd import ops  # noqa: E402


def run(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    dev = torch.device("cuda:0")
    shapes = [(16, 148, 148, 256, 256), (16, 296, 296, 256, 128), (16, 592, 592, 128, 128), (16, 74, 74, 256, 256),
              (128, 56, 56, 256, 256), (128, 112, 112, 256, 128), (64, 224, 224, 128, 128), (64, 64, 64, 256, 256),
              (16, 256, 256, 256, 128), (16, 512, 512, 128, 128)]
    for dt in (torch.float16, torch.bfloat16):
        for B, H, W, Cin, Cout in shapes:
            x = torch.randn(B, H, W, Cin, device=dev).to(dt)
            w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(dt)
            b = torch.randn(Cout, device=dev)
            r = torch.randn(B * H * W, Cout, device=dev).to(dt)
            fl = 2.0 * B * H * W * 9 * Cin * Cout
            row = [f"{str(dt)[6:]:9s} B{B} {H}x{W} {Cin}->{Cout}"]
            for name, kw in (("plain", {}), ("relu_a+relu", dict(relu_a=True, act="relu")), ("residual", dict(residual=r))):
                ts = []
                for flat in (0, 1):
                    with ops.tuning("conv_rows_flat", flat):
                        ts.append(run(lambda: ops.gemm(x, w, b, conv=(B, H, W, Cin, 1), **kw)))
                row.append(f"{name}: {fl / ts[0] / 1e12:6.0f} -> {fl / ts[1] / 1e12:6.0f} TF ({ts[0] / ts[1]:.3f}x)")
            print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
