"""Runs the reference's unit tests restated against the source-compatible C++ API
(tests/cpp/edm_api_test.cpp -> include/edm/*.h -> libedm.so -> libedm_hip.so) on the GPU."""
import os
import subprocess

import pytest

from conftest import FIXTURES, GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_reference_unit_tests_through_cpp_api(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "electronic-dance-music_amd", "host")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    # (working directory = the scratch directory: read_test.edm finds its target grid "2.grid.test" there)
    res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "edm_api_test"), FIXTURES, str(tmp_path), GOLDEN],
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "0 failed" in res.stdout
