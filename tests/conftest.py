import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "sh-assembly_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")

# torch bundles its own HIP runtime; load it before libshk.so pulls in /opt/rocm's, so that one
# process ends up with a single runtime (tests that hand torch tensors to the C ABI need both)
try:
    import torch  # noqa: F401,E402
except Exception:  # pragma: no cover
    torch = None
