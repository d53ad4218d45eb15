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
        '  printf("%zu %zu %zu %zu\\n", sizeof(ppg_fetch_header), offsetof(ppg_fetch_header, record_bytes), offsetof(ppg_fetch_header, overflow),\n'
        '         offsetof(ppg_fetch_header, bytes_used));\n'
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
    FH = _abi.PpgFetchHeader
    assert lines[6].split() == [str(ctypes.sizeof(FH))] + [str(getattr(FH, n).offset) for n in ("record_bytes", "overflow", "bytes_used")]
    assert ctypes.sizeof(FH) == 64


def test_policy_kernel_choice_and_lds_layout_for_every_shape(hip_lib_path):
    """ppg_policy_describe (host arithmetic only, no device): for every observation shape and network depth ppg_policy_create_spec
    accepts, the kernels' LDS layout fits a CU (160 KB), a sub-group fits the staging threads, and the two-role pipeline's areas
    (table, partial sums, row area, images) follow each other without overlap."""
    from predpreygrass_amd import _abi
    lib = _abi.bind(ctypes.CDLL(hip_lib_path))
    LDS = 160 * 1024
    seen = {0: 0, 1: 0, 2: 0, 3: 0}
    for layout in (_abi.POLICY_LAYOUT_CHW, _abi.POLICY_LAYOUT_HWC):
        for C in range(4, 9):
            for R in range(1, 16):
                for n_conv in range(1, 7):
                    for n_fc, n_actions in ((1, 5), (1, 9), (1, 16), (1, 17), (1, 25), (2, 9), (3, 9)):
                        if n_fc > 1 and n_conv != 3:
                            continue
                        sp = _abi.PpgPolicySpec()
                        sp.obs_channels, sp.obs_range, sp.n_actions, sp.layout, sp.flatten = C, R, n_actions, layout, _abi.POLICY_FLATTEN_NHWC
                        sp.n_conv, sp.n_fc = n_conv, n_fc
                        for l, c in enumerate(([16, 32, 64] + [64] * 3)[:n_conv]):
                            sp.conv_out[l] = c
                        for l in range(n_fc):
                            sp.fc_out[l] = 256 if l + 1 < n_fc else n_actions
                        out = (ctypes.c_int32 * 12)()
                        rc = lib.ppg_policy_describe(ctypes.byref(sp), out, 12)
                        P = (C if layout == _abi.POLICY_LAYOUT_HWC else R) * R
                        if n_fc > 1 and layout == _abi.POLICY_LAYOUT_CHW and C != 4:
                            assert rc != 0   # (ppg_policy_create_spec refuses it: the description does too)
                            continue
                        if rc != 0:
                            assert n_fc == 1 and P * 64 * 6 * 2 > 100 * 1024, (layout, C, R, n_conv, n_fc, n_actions)   # only images too large for one sample
                            continue
                        fam, st, lds, threads = out[0], out[1], out[2], out[3]
                        seen[fam] += 1
                        assert 1 <= st <= 16 and 0 < lds <= LDS, (fam, st, lds, C, R, n_conv)
                        assert fam == (0 if n_fc > 1 else 2 if n_conv > 3 else fam) or fam in (1, 3)
                        if fam == 3:
                            assert n_conv == 3 and n_actions <= 16 and threads == 512
                            table, region, ni, slots, red, raw, img, raw_bytes = out[4:12]
                            assert st * P <= 256 and table >= 100 and table % st == 0 and region % 16 == 0
                            assert red == table * 16 and raw == red + 2 * 4096 + 64 + 2048 and img == raw + raw_bytes + 1024
                            assert img + st * region + 18 * 32 * 2 == lds
                            row = C * R * R * 2
                            if ni:
                                assert row % 8 == 0 and 1 <= ni <= 3 and slots * (row // 8) <= 256 and slots * ni >= st and raw_bytes >= st * row
                            else:
                                assert raw_bytes == 0
                        else:
                            assert threads == 256 and st * P <= 512
    assert seen[0] > 0 and seen[1] > 0 and seen[2] > 0 and seen[3] > 100
    # a description exists exactly for the networks ppg_policy_create_spec can build: the same shape rules refuse the rest
    def spec(**kw):
        sp = _abi.PpgPolicySpec()
        sp.obs_channels, sp.obs_range, sp.n_actions, sp.layout, sp.flatten, sp.n_conv, sp.n_fc = 4, 9, 9, _abi.POLICY_LAYOUT_HWC, _abi.POLICY_FLATTEN_NHWC, 3, 1
        sp.conv_out[0], sp.conv_out[1], sp.conv_out[2], sp.fc_out[0] = 16, 32, 64, 9
        for k, v in kw.items():
            if isinstance(v, tuple):
                getattr(sp, k)[v[0]] = v[1]
            else:
                setattr(sp, k, v)
        return sp
    out = (ctypes.c_int32 * 12)()
    assert lib.ppg_policy_describe(ctypes.byref(spec()), out, 12) == 0
    for bad in (dict(conv_out=(0, 17)), dict(conv_out=(2, 65)), dict(conv_out=(1, 0)), dict(fc_out=(0, 8)), dict(flatten=7), dict(layout=3),
                dict(n_fc=2, fc_out=(0, 300)), dict(n_fc=2, n_conv=2, fc_out=(1, 9)), dict(n_actions=33, fc_out=(0, 33)), dict(obs_range=16)):
        assert lib.ppg_policy_describe(ctypes.byref(spec(**bad)), out, 12) == -1, bad
    # the reference's shapes: 7x7 predators, 9x9 prey, three convolutions, nine actions -> the pipeline with 9 / 7 samples per sub-group
    for R, st in ((7, 9), (9, 7)):
        sp = _abi.PpgPolicySpec()
        sp.obs_channels, sp.obs_range, sp.n_actions, sp.layout, sp.flatten, sp.n_conv, sp.n_fc = 4, R, 9, _abi.POLICY_LAYOUT_HWC, _abi.POLICY_FLATTEN_NHWC, 3, 1
        sp.conv_out[0], sp.conv_out[1], sp.conv_out[2], sp.fc_out[0] = 16, 32, 64, 9
        out = (ctypes.c_int32 * 12)()
        assert lib.ppg_policy_describe(ctypes.byref(sp), out, 12) == 0
        assert (out[0], out[1], out[3]) == (3, st, 512) and out[4] >= 518 and out[6] > 0
