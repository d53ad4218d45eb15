"""The C-ABI shared library must load on a machine without a GPU and export every symbol that
include/ppg.h declares.  No compute entry point is called here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib_path():
    import __graft_entry__ as g
    return g.build_hip()


def test_hip_library_exports_every_declared_symbol(hip_lib_path):
    header = open(os.path.join(ROOT, "include", "ppg.h")).read()
    declared = set(re.findall(r"\b(ppg_[a-z0-9_]+)\s*\(", header))
    from predpreygrass_amd import _abi
    assert declared == set(_abi.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(hip_lib_path)
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_host_only_entry_points(hip_lib_path):
    from predpreygrass_amd import _abi
    lib = _abi.bind(ctypes.CDLL(hip_lib_path))
    assert lib.ppg_abi_version() == _abi.ABI_VERSION
    assert lib.ppg_lexkey(10) < lib.ppg_lexkey(2) < lib.ppg_lexkey(20)


def test_gfx950_code_object_is_embedded(hip_lib_path):
    blob = open(hip_lib_path, "rb").read()
    assert b"gfx950" in blob and b"ppg_step_q2" in blob


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from predpreygrass_amd.batched import BatchedPredPreyGrass
    with pytest.raises(RuntimeError, match="needs a ROCm GPU"):
        BatchedPredPreyGrass({}, batch_size=1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "predpreygrass_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "wave_emu" not in text.replace(
                    "tests/wave_emu", ""), f
