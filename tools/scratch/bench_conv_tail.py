"""The regressor's conv3x3 (128 -> 128) + ReLU + fused 1x1 tail at 512 x 512 (the head's largest launch), fp16 and bf16 operands."""
import sys, time, torch
sys.path.insert(0, ".")
from uniception_amd import ops
dev = torch.device("cuda:0")
def run(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for dt in (torch.float16, torch.bfloat16):
    for (B, H, W, Cin) in [(32, 512, 512, 128), (32, 224, 224, 128)]:
        x = torch.randn(B, H, W, Cin, device=dev).to(dt)
        w = (torch.randn(128, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(dt)
        b = torch.randn(128, device=dev); w4 = torch.randn(4, 128, device=dev) / 11; b4 = torch.randn(4, device=dev)
        fl = 2.0 * B * H * W * 9 * Cin * 128
        t_tail = run(lambda: ops.gemm(x, w, b, act="relu", conv=(B, H, W, Cin, 1), tail=(w4, b4)))
        t_plain = run(lambda: ops.gemm(x, w, b, act="relu", conv=(B, H, W, Cin, 1)))
        print(f"{str(dt)[6:]:9s} B{B} {H}x{W} {Cin}->128  tail {t_tail*1e3:7.3f} ms {fl/t_tail/1e12:6.0f} TF   plain store {t_plain*1e3:7.3f} ms {fl/t_plain/1e12:6.0f} TF", flush=True)
