from .dust3r import DUSt3R  # noqa: F401
