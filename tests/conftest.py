import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cref():
    """The plain-C CPU restatement (oracle/cfnmpc_ref.c), built on demand."""
    import cref as _cref
    _cref.build()
    return _cref


@pytest.fixture(scope="session")
def oracle():
    import cfnmpc_oracle
    return cfnmpc_oracle
