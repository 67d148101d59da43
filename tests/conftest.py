import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHTS = os.path.join(ROOT, "feartracker_amd", "weights", "fear_xs_noembs.fearw")
WEIGHTS_DEMO = os.path.join(ROOT, "feartracker_amd", "weights", "fear_xs_demo.fearw")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_net():
    from oracle.fear_oracle import OracleNet
    return OracleNet(WEIGHTS)


@pytest.fixture(scope="session", params=["throughput_plan", "small_batch_plan"])
def hip_net(request):
    """The engine on both of its launch plans: passes of <= FEAR_OPT_SMALL_PASS crops normally take the small-batch plan
    (split-K 16x16 kernels, two-stream head); with the option at 0 every pass takes the throughput plan (chain kernel, fused
    correlation / prediction epilogues) whatever its size."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the HIP path has no CPU fallback)")
    from feartracker_amd import FEARNetHIP
    net = FEARNetHIP(WEIGHTS, device=0, max_batch=64)
    if request.param == "throughput_plan":
        net.set_small_pass(0)
    return net
