"""Anatomy of the bf16 attention forward (UC_ATTN_DBG, four-wave workgroups; results are wrong by construction): what remains
without the per-tile barrier (1), without the exp (2: P = S), without the PV MFMAs (4) and combinations."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import torch
    from uniception_amd import ops
    dev = torch.device("cuda:0")
    B, H, N = 64, 16, 1024
    q = torch.randn(B, N, H, 64, device=dev).bfloat16(); k = torch.randn(B, N, H, 64, device=dev).bfloat16()
    vt = ops.vt_pack(torch.randn(B, N, H, 64, device=dev).bfloat16())
    f = lambda: ops.attention(q, k, vt, 0.125, v_packed=True)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"UC_ATTN_DBG={os.environ.get('UC_ATTN_DBG', '0')}: {t*1e6:7.1f} us  {4.0*B*H*N*N*64/t/1e12:6.1f} TFLOP/s-equivalent")
else:
    for d in ("0", "1", "2", "3", "4", "5", "8", "16", "24", "0"):
        env = dict(os.environ, UC_ATTN_DBG=d, UC_ATTN_NW=os.environ.get("UC_ATTN_NW", "4"))
        subprocess.run([sys.executable, __file__, "run"], env=env)
