"""The C-ABI library loads on a GPU-less host and exports every symbol include/lanefront.h
declares; creating a handle without a device fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lanefront.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from lane_slam_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "liblanefront.so lacks %s" % n
    assert set(names) == set(_lib.EXPORTS), (set(names) ^ set(_lib.EXPORTS))
    assert lib.lf_abi_version() == 5


def test_config_struct_matches_header_field_order():
    from lane_slam_amd.config import LfConfig
    src = open(os.path.join(ROOT, "include", "lanefront.h")).read()
    body = re.search(r"typedef struct lf_config \{(.*?)\} lf_config;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1]
        fields += [re.sub(r"\[.*", "", n).strip() for n in names.split(",")]
    assert fields == [f[0] for f in LfConfig._fields_]
    # same layout as the oracle's mirror (both sides are filled from one dict in the tests)
    from oracle.oracle import LfoConfig
    assert ctypes.sizeof(LfConfig) == ctypes.sizeof(LfoConfig)
    assert [f[0] for f in LfConfig._fields_] == [f[0] for f in LfoConfig._fields_]


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lane_slam_amd import FrontEnd, LanefrontError, LineDetectorHIP
    from lane_slam_amd.config import DEFAULT_DETECTOR_CONFIGURATION
    with pytest.raises(LanefrontError) as e:
        FrontEnd()
    assert "no CPU fallback" in str(e.value)
    det = LineDetectorHIP(dict(DEFAULT_DETECTOR_CONFIGURATION))     # construction is lazy, like the reference plugin
    import numpy as np
    with pytest.raises(LanefrontError):
        det.setImage(np.zeros((80, 160, 3), np.uint8))
    with pytest.raises(ValueError):
        LineDetectorHIP({"bogus": 1})


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "lane_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liblforacle" not in txt, f
                assert not [l for l in txt.splitlines() if l.lstrip().startswith("#include") and "oracle" in l], f


def test_no_fused_pack_shift_instruction_in_device_code():
    """ROCm 7.2 hipcc may fuse `(x >> 16)` + clamp-to-u8 + pack into v_ashr_pk_u8_i32 and then treat the
    upper half of its result as zero, which the MI355X does not honour (wrong byte in k_lbd_grad, found
    by the GPU parity tests).  The kernels keep the shift and the clamp apart; make sure no translation
    unit gets the instruction back."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "lane_slam_amd", "csrc", "k_lbd.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k_lbd.s")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                               "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
        asm = open(out).read()
    assert "v_ashr_pk_u8_i32" not in asm
    assert "k_lbd_grad" in asm
