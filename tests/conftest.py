import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is many small CPU tensor ops.  A GPU box shows all of the host's hardware threads
    # but grants a share of them: torch's default intra-op pool (one thread per visible core)
    # then spends its time on synchronisation.  A small pool is faster and steadier.
    try:
        import torch
        torch.set_num_threads(min(8, os.cpu_count() or 1))
    except Exception:
        pass


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
