import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniception_amd import ops
_orig = ops.attention
def chk(q, k, v, scale, v_packed=False, **kw):
    o = _orig(q, k, v, scale, v_packed=v_packed, **kw)
    if not bool(torch.isfinite(o).all()):
        print("NONFINITE attention: q", tuple(q.shape), q.stride(), "k", tuple(k.shape), k.stride(), "v", tuple(v.shape), v.stride(),
              "q finite", bool(torch.isfinite(q).all()), "k finite", bool(torch.isfinite(k).all()), "v finite", bool(torch.isfinite(v.float()).all()),
              "k storage finite", bool(torch.isfinite(k._base if k._base is not None else k).all()))
        sys.stdout.flush()
    return o
ops.attention = chk
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "gpu", "tests/test_modules_gpu.py", "-k", "three_view or head_precision", "-s"]))
