import json
import os
import sys

import pytest

# Several ranks of a decomposed run live in ONE process on ONE GPU in tests/test_gpu_slab.py, and in overlap mode 3 a rank's kernel
# polls for stores of its neighbour's kernel: with the runtime's default of four hardware queues two such kernels can share a
# queue and the poller then sits in front of the kernel it waits for until its bounded wait gives up (seen once in a full-suite
# run).  More hardware queues (read by the runtime when it initialises, so set here before anything touches the GPU) keep the
# streams apart; with one process per GPU -- every real run -- the situation cannot arise.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_vectors():
    with open(os.path.join(GOLDEN, "reference_unit_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import wafer_oracle
    wafer_oracle.build()
    return wafer_oracle
