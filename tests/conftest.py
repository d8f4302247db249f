import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def note(line):
    """Measured values behind a tolerance, kept with the run's other outputs (gpurun_out/ is merged back)."""
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "e2e_measured.txt"), "a") as f:
            f.write(line + "\n")
    print(line)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def r18_blob():
    from failure_aware_vision_amd import weights
    return weights.make_synthetic("resnet18_cifar", seed=1)


@pytest.fixture(scope="session")
def r50_blob():
    from failure_aware_vision_amd import weights
    return weights.make_synthetic("resnet50", seed=1)


ENSEMBLE_SEEDS = (1, 2, 3, 4, 5)


@pytest.fixture(scope="session")
def r50_members():
    """BASELINE configs[3]: five independently seeded ResNet-50 checkpoints -> [(blob, info)] (member 0 = r50_blob's)."""
    from failure_aware_vision_amd import weights
    return [weights.make_synthetic("resnet50", seed=s) for s in ENSEMBLE_SEEDS]
