import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# every recorded region of the suite (and of the rank processes it spawns) runs under the replay contract check: a torch op that
# moves device data outside a C-ABI launch / replay.step raises instead of going stale on replays (lkgd_amd/replay.py)
os.environ.setdefault("LKGD_REPLAY_STRICT", "1")

# the host side of the tests (the fp32 oracle) runs on the CPUs this process really has: a GPU box shows 256 logical CPUs behind a
# cgroup quota of 16, and torch's default of 128 intra-op threads there is up to 30 x slower than 16 (tools/hostcpus.py).  Set
# before torch starts its pool; rank processes the tests spawn inherit it and divide it among themselves (rank_threads()).
from tools.hostcpus import host_cpus   # noqa: E402

HOST_CPUS = host_cpus()
os.environ.setdefault("OMP_NUM_THREADS", str(HOST_CPUS))
os.environ.setdefault("MKL_NUM_THREADS", str(HOST_CPUS))
os.environ["LKGD_TEST_HOST_CPUS"] = str(HOST_CPUS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: duplicates coverage at a multiple of the cost; runs only with LKGD_SLOW=1")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """a bare `pytest` on a box without a GPU skips the `gpu` tests instead of failing them (the driver selects with -m)"""
    import torch
    if not os.environ.get("LKGD_SLOW"):
        slow = pytest.mark.skip(reason="slow duplicate of a default test (set LKGD_SLOW=1 to run it)")
        for it in items:
            if "slow" in it.keywords:
                it.add_marker(slow)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


C1_SEED = 31   # tests/golden/make_goldens.py: weights of the real-width configs[0] fixtures


@pytest.fixture(scope="session")
def c1_oracle_model():
    """the fp32 oracle UNet at the REAL width with the weights the reference ran with for the c1 fixtures (regenerated from
    the seed, rounded to fp16); 6 GB of host memory, shared by every test that needs it"""
    import torch
    from oracle import unet as ou
    o = ou.UNetSpatioTemporalConditionControlNetModel(ou.SVD_CONFIG)
    ou.init_weights_(o, C1_SEED)
    with torch.no_grad():
        for p in o.parameters():
            p.copy_(p.half().float())
    return o


@pytest.fixture(scope="session")
def c1_hip_model(c1_oracle_model):
    import torch
    from lkgd_amd import unet as pu
    with torch.device("meta"):
        m = pu.UNetSpatioTemporalConditionControlNetModel(pu.UNetConfig())
    m = m.to_empty(device="cpu")
    m.load_state_dict(c1_oracle_model.state_dict(), strict=True)
    return m.half().to("cuda:0")
