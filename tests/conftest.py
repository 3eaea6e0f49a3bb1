import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def reflib():
    import oracle_py
    if not oracle_py.ref_available():
        pytest.skip("oracle/_ref/libdarwin_ref.so not built (reference not mounted)")
    return oracle_py.RefLib()


@pytest.fixture(scope="session")
def hip_lib_path():
    from gact_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    return engine.LIB_PATH


_BLOCKS = {}


def workload_block(name):
    """a named workload's reads and candidates (gact_amd/workload.py), built once per session: simulating the reads and
    running the D-SOFT restatement takes seconds per block, and several files use the same ones"""
    from gact_amd import workload
    if name not in _BLOCKS:
        _BLOCKS[name] = workload.make_block(name, candidates="dsoft")
    return _BLOCKS[name]
