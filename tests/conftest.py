import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kd-6d-pose-adlp_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order under the driver's `pytest -x`: the cheap, deterministic element-wise comparisons report first, the
# whole-step statistical comparisons at the benchmark size last -- one late failure must not hide the kernel tests.
_ORDER = ["test_kernels_gpu", "test_losses_gpu", "test_step_gpu", "test_train_entry_gpu", "test_dzi_gpu",
          "test_eval_gpu", "test_comm_gpu", "test_bench_gpu", "test_fullsize_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(mod) if mod in _ORDER else -1       # CPU-side modules keep their place in front
    items.sort(key=rank)            # stable: the order inside a module is untouched


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
