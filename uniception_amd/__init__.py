"""uniception_amd — MI355X-native (gfx950 / CDNA4) implementation of the UniCeption DUSt3R two-view
pointmap hot path behind the reference's nn.Module API.  See DESIGN.md / INTEGRATION.md."""

__version__ = "0.1.0"
