"""pytest configuration: markers, paths, and builders for the test-only oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _build_oracle():
    # `make` is a no-op when everything is up to date; the _ref target only
    # (re)builds when /root/reference exists (it does not on the GPU box).
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)


@pytest.fixture(scope="session")
def oracle_cli():
    _build_oracle()
    path = os.path.join(ORACLE_DIR, "ipxo_cli")
    assert os.path.exists(path)
    return path


@pytest.fixture(scope="session")
def oracle_lib():
    _build_oracle()
    from tests import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def golden():
    return GOLDEN
