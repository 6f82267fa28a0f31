"""Patch-grid positions (reference: utils/positional_encoding.py:8-23, libs/croco/patch_embed.py:19-31)."""
import torch


class PositionGetter(object):
    """Returns int64 (y, x) coordinates of a row-major h x w token grid, [b, h*w, 2].
    Cached per (h, w, device) — the reference caches per (h, w) only and so pins the first device seen
    (SURVEY.md Appendix C); keying on the device as well is the only behavioural difference."""

    def __init__(self):
        self.cache_positions = {}

    def __call__(self, b, h, w, device):
        device = torch.device(device)
        key = (h, w, device)
        if key not in self.cache_positions:
            ys = torch.arange(h, device=device).repeat_interleave(w)
            xs = torch.arange(w, device=device).repeat(h)
            self.cache_positions[key] = torch.stack((ys, xs), dim=-1)
        return self.cache_positions[key].view(1, h * w, 2).expand(b, -1, 2).clone()
