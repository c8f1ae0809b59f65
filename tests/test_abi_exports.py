"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and
exports every symbol include/edm_hip.h declares; the product package never imports
the oracle."""
import os
import re

import edm_amd.hip as H

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    dll = H.lib()
    declared = H.exported_symbols()
    assert len(declared) >= 55
    missing = [s for s in declared if not hasattr(dll, s)]
    assert not missing, missing
    assert sorted(H._PROTOS) == declared  # the ctypes mirror binds exactly the declared ABI
    assert b"gfx950" in dll.edm_hip_version()


def test_header_cites_reference_interfaces():
    text = open(H.HEADER).read()
    for cite in ("gaussian_grid.h:176-372", "edm_bias.cpp:276-295", "grid.h:52-139", "edm_bias.cpp:401-411",
                 "fix_edm_pair.cpp:215-217"):
        assert cite in text


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "electronic-dance-music_amd")
    offenders = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                body = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"liboracle|edm_oracle\.h|from oracle|import oracle|oracle/", body):
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders


def test_no_gpu_means_loud_failure():
    if H.device_count() > 0:
        return
    try:
        H.require_gpu()
    except H.EdmHipError:
        pass
    else:
        raise AssertionError("require_gpu must raise without a GPU")
    try:
        H.Gauss.create([0.0], [1.0], [0.1], [0], 1, [0.1])
    except H.EdmHipError:
        pass
    else:
        raise AssertionError("creating a device grid without a GPU must fail loudly")
