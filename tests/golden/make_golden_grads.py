"""Generate golden GRADIENT vectors from the REAL reference (autograd over castacks/UniCeption's own modules on CPU).

Run like make_golden.py (same stubs / PYTHONPATH; build container only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo \
        python3 -B /root/repo/tests/golden/make_golden_grads.py

For each case: reference forward in fp32 with grad -> loss = sum over the two views of
mean(conf*|pts-gt|) - 0.2*mean(log conf) on seeded targets -> backward.  The oracle restatement is differentiated by
autograd too and must agree (rel-L2 < 5e-5 per parameter).  The fixture tests/golden/<case>__grads.npz holds the loss
and, per parameter, up to 512 evenly spaced gradient entries plus the gradient's L2 norm (data only).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dust3r_oracle as O  # noqa: E402
from tests.golden.cases import CASES, GAINS, GRAD_CASES, grad_targets, sample_indices  # noqa: E402
from tests.golden.make_golden import ComposedTwoView, rel_l2  # noqa: E402


def conf_loss(pts, conf, gt, alpha=0.2):
    r = (pts - gt).norm(dim=-1)
    c = conf[..., 0]
    return (c * r).mean() - alpha * c.log().mean()


def run(name):
    c = CASES[name]
    t0 = time.time()
    torch.manual_seed(0)
    if c.get("factory"):     # the reference's own factory model (BASELINE configs[2]: the ViT-L two-view model at full size)
        from uniception.models.factory import DUSt3R
        model = DUSt3R(name="g", img_size=tuple(c["img"]), pred_head_type=c["head"]).train()
    else:
        model = ComposedTwoView(c).train()
    O.fill_state_dict_(model.state_dict(), gain=1.0, gains=GAINS)
    img1, img2 = O.make_images(c["seed"], c["B"], *c["img"])
    gt1, gt2 = grad_targets(c)
    if c.get("factory"):
        v1 = {"img": img1, "instance": [str(i) for i in range(c["B"])], "data_norm_type": "dust3r"}
        v2 = {"img": img2, "instance": [str(100 + i) for i in range(c["B"])], "data_norm_type": "dust3r"}
        r1, r2 = model(v1, v2)
    else:
        r1, r2 = model(img1, img2, {})
    loss = conf_loss(r1["pts3d"], r1["conf"], gt1) + conf_loss(r2["pts3d_in_other_view"], r2["conf"], gt2)
    loss.backward()
    ref = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    t_ref = time.time() - t0

    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    o1, o2 = O.dust3r_forward(sd, img1, img2, head=c["head"], enc_depth=c["enc_depth"], enc_heads=c["enc_heads"],
                              dec_depth=c["dec_depth"], dec_heads=c["dec_heads"], patch_size=c["patch"],
                              indices=tuple(c["indices"]), collect={})
    lo = conf_loss(o1["pts3d"], o1["conf"], gt1) + conf_loss(o2["pts3d_in_other_view"], o2["conf"], gt2)
    lo.backward()
    worst = abs(float(lo.detach()) - float(loss.detach())) / abs(float(loss.detach()))
    # the reference registers some parameters under two names (dpt.py: scratch.layerK_rn is input_process.K.1);
    # the oracle's state-dict copy has them as separate leaves, so sum the gradients over each alias group
    alias = {}
    for k, v in model.state_dict().items():
        alias.setdefault(v.data_ptr(), []).append(k)
    groups = {ks[0]: ks for ks in alias.values()}
    groups = {k: next(ks for ks in alias.values() if k in ks) for k in ref}
    for k, g in ref.items():
        got = [sd[a].grad for a in groups[k] if sd[a].grad is not None]
        assert got, f"oracle produced no gradient for {k}"
        worst = max(worst, rel_l2(sum(got), g))
    assert worst < 5e-5, f"{name}: oracle autograd deviates from the reference by {worst}"

    save = {"loss": np.float64(float(loss.detach()))}
    for k, g in ref.items():
        idx = sample_indices(g.numel(), 512)
        save[k + "__samples"] = g.flatten()[idx].float().numpy()
        save[k + "__norm"] = np.float64(g.double().norm().item())
    np.savez_compressed(os.path.join(HERE, name + "__grads.npz"), **save)
    print(f"{name}: loss {float(loss.detach()):.6f}, {len(ref)} parameter gradients, reference fwd+bwd {t_ref:.1f}s, "
          f"oracle-vs-reference worst rel-L2 {worst:.2e}, total {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or GRAD_CASES):
        run(n)
