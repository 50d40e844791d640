import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """On the GPU box: bring torch's HIP runtime up BEFORE the first test loads libhophip.so.  Some GPU tests hand torch tensors' device pointers to the library; torch's
    wheel carries its own HIP runtime, and when the library's (the image's /opt/rocm) has initialised the device first, torch's later initialisation can fail with
    "No HIP GPUs are available".  Initialised in this order the two share the process (bench.py does the same)."""
    if not any("gpu" in it.keywords for it in items):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
