import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = os.path.join(GOLDEN, "ref_fixtures")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding

    return binding.load("oracle")


@pytest.fixture(scope="session")
def ref_lib():
    """The real reference, built from /root/reference by oracle/Makefile (present in
    the build container; on the GPU box only if the prebuilt .so travelled)."""
    from oracle import binding

    try:
        return binding.load("ref")
    except (FileNotFoundError, OSError, Exception) as exc:  # noqa: BLE001
        pytest.skip("reference build oracle/_ref/libedm_ref.so unavailable: %s" % exc)


@pytest.fixture()
def workdir(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    return tmp_path
