import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # see zkp-ecdsa_amd/csrc/api.hip: zk_ctx_create

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)


try:   # tests that hand torch tensors to the engine need ONE HIP runtime in the process: the wheel bundles its own libamdhip64 (same soname as
    import torch   # /opt/rocm's), and whichever is loaded first serves both.  Loaded here, every subset of the suite runs like the whole suite.
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
