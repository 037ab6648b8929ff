import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cpus() -> int:
    """affinity mask capped by the cgroup CPU quota (a GPU box shows 256 logical CPUs and grants 16: the oracle's OpenMP default
    of one thread per logical CPU runs several times slower there than one thread per granted CPU)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return n


os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cpus()))      # before the oracle's library is loaded


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    d = os.path.join(ROOT, "tests", "golden")
    return {n[:-5]: json.load(open(os.path.join(d, n))) for n in os.listdir(d) if n.endswith(".json")}


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu
    cpu.build()
    return cpu
