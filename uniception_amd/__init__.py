"""uniception_amd — MI355X-native (gfx950 / CDNA4) implementation of the UniCeption DUSt3R two-view
pointmap hot path behind the reference's nn.Module API.  See DESIGN.md / INTEGRATION.md.

Weight-cache contract (the one thing a drop-in user must know): modules keep compute-dtype / re-laid-out copies of their
parameters, keyed on (storage pointer, autograd version).  Ordinary updates (optimizer.step(), load_state_dict(), in-place
ops on the parameter) are seen; writes through ``param.data`` or raw pointers are NOT — call
``uniception_amd.invalidate_prepared(module)`` (or ``bump_weight_epoch()``) after them.
"""

__version__ = "0.2.0"


def bump_weight_epoch() -> None:
    """Invalidate every prepared weight copy (see the module docstring)."""
    from . import engine
    engine.bump_weight_epoch()


def invalidate_prepared(module=None) -> None:
    """Invalidate the prepared weight copies of `module` and its children (None: all)."""
    from . import engine
    engine.invalidate_prepared(module)
