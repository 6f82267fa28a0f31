"""Does a row's result depend on where in the batch it sits?  Runs one encoder block on 24 images and on the last 12 of them and
compares every op's output rows bit by bit (found: FMA contraction differing between unrolled copies of the epilogue arithmetic)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dust3r_oracle as O
from tests.golden.cases import GAINS
from uniception_amd import engine, ops
from uniception_amd.models.factory import DUSt3R
from uniception_amd.models.encoders.base import ViTEncoderInput
gpu = torch.device("cuda:0")
model = DUSt3R(name="g", img_size=(64, 96), pred_head_type="dpt").eval()
O.fill_state_dict_(model.state_dict(), gain=1.0, gains=GAINS)
model = model.to(gpu)
model.encoder.enc_blocks = model.encoder.enc_blocks[:1]
g = torch.Generator().manual_seed(99)
B = 12
img = torch.randn(2 * B, 3, 64, 96, generator=g).to(gpu)
engine.CONCURRENT = False
log = []
names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and n in ("gemm", "attention", "attention_fwd", "layernorm", "patch_gather", "ln_stats_finalize")]
print("hooked:", names)
orig = {n: getattr(ops, n) for n in names}
def wrap(n):
    def f(*a, **k):
        out = orig[n](*a, **k)
        t = out[0] if isinstance(out, (tuple, list)) else out
        extra = {}
        side = getattr(t, "uc_ln", None)
        if side is not None: extra = {"twin": side.twin.clone(), "partial": side.partial.clone()}
        log.append((n, tuple(t.shape), t.detach().clone(), extra, {kk: (vv is not None) for kk, vv in k.items() if kk in ("ln","emit_ln","rope","vt","residual","act")}))
        return out
    return f
for n in names: setattr(ops, n, wrap(n))
def enc(x):
    log.clear()
    with torch.no_grad(), engine.precision("bf16"):
        y = model.encoder(ViTEncoderInput(image=x, data_norm_type="dust3r")).features.clone()
    return y, list(log)
yf, lf = enc(img)
yh, lh = enc(img[B:])
print(len(lf), len(lh))
for (n, s, t, ex, kw), (n2, s2, t2, ex2, kw2) in zip(lf, lh):
    rows = t2.shape[0]
    a = t.reshape(-1, t.shape[-1]) if t.dim() > 1 else t
    b = t2.reshape(-1, t2.shape[-1]) if t2.dim() > 1 else t2
    d = float((a[a.shape[0] - b.shape[0]:].float() - b.float()).abs().max()) if a.dim() == b.dim() and a.shape[1:] == b.shape[1:] else float("nan")
    msg = f"{n:10s} {str(s):22s} vs {str(s2):22s} {kw} max diff {d:.3e}"
    for k in ex:
        e1, e2 = ex[k], ex2[k]
        msg += f" | {k} {float((e1[e1.shape[0] - e2.shape[0]:].float() - e2.float()).abs().max()):.3e}"
    print(msg)
