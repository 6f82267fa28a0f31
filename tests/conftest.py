import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


ATTN_RS_DEFAULT = int(os.environ.get("UC_ATTN_RS", "0"))


@pytest.fixture(autouse=True)
def _automatic_gemm_variant():
    """Tests that force a tile variant of the bf16 GEMM (ops.tuning_set) leave the library on its automatic choice."""
    yield
    import torch
    if torch.cuda.is_available():
        from uniception_amd import ops
        ops.tuning_set("gemm_variant", -3)
        ops.tuning_set("gemm_stagger", -1)
        ops.tuning_set("attn_role_split", ATTN_RS_DEFAULT)
        ops.tuning_set("conv_rows", 1)
        ops.tuning_set("conv_rows_flat", 1)
        ops.tuning_set("small_m_split", 2048)
