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


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The ctypes mirrors in predpreygrass_amd/_abi.py must have exactly the layout gcc gives include/ppg.h."""
    import subprocess
    from predpreygrass_amd import _abi
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "ppg.h"\n'
        'int main(void) {\n'
        '  printf("%zu %zu %zu\\n", sizeof(ppg_config), sizeof(ppg_config_gen2), sizeof(ppg_buffers));\n'
        '  printf("%zu %zu %zu %zu\\n", offsetof(ppg_config, reward_mode), offsetof(ppg_config, n_drive),\n'
        '         offsetof(ppg_config, grass_opportunity_normalizer), offsetof(ppg_config, season_high_multiplier));\n'
        '  printf("%zu %zu %zu %zu\\n", offsetof(ppg_config_gen2, n_grass), offsetof(ppg_config_gen2, reward_predator_catch_prey),\n'
        '         offsetof(ppg_config_gen2, mutation_rate_prey), offsetof(ppg_config_gen2, mask_observation_with_visibility));\n'
        '  printf("%zu %zu\\n", offsetof(ppg_buffers, row_lastrep), offsetof(ppg_buffers, row_info));\n'
        '  printf("%d %d\\n", PPG_ABI_VERSION, PPG_ENV_WORDS);\n'
        '  printf("%zu %zu %zu %zu %zu\\n", sizeof(ppg_policy_spec), offsetof(ppg_policy_spec, conv_out), offsetof(ppg_policy_spec, n_fc),\n'
        '         offsetof(ppg_policy_spec, conv_w), offsetof(ppg_policy_spec, fc_b));\n'
        '  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    C1, C2, B = _abi.PpgConfig, _abi.PpgConfigGen2, _abi.PpgBuffers
    assert lines[0].split() == [str(ctypes.sizeof(C1)), str(ctypes.sizeof(C2)), str(ctypes.sizeof(B))]
    assert lines[1].split() == [str(getattr(C1, n).offset) for n in
                                ("reward_mode", "n_drive", "grass_opportunity_normalizer", "season_high_multiplier")]
    assert lines[2].split() == [str(getattr(C2, n).offset) for n in
                                ("n_grass", "reward_predator_catch_prey", "mutation_rate_prey", "mask_observation_with_visibility")]
    assert lines[3].split() == [str(B.row_lastrep.offset), str(B.row_info.offset)]
    assert lines[4].split() == [str(_abi.ABI_VERSION), str(_abi.ENV_WORDS)]
    PS = _abi.PpgPolicySpec
    assert lines[5].split() == [str(ctypes.sizeof(PS))] + [str(getattr(PS, n).offset) for n in ("conv_out", "n_fc", "conv_w", "fc_b")]
