"""-m "not gpu": libgenima_hip.so builds, loads and exports every symbol include/genima_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from genima_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "genima_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gn_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"libgenima_hip.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_loader_binds_and_reports_version():
    lib = _lib.load()
    assert lib.gn_version() == 101 == _lib.ABI_VERSION
    # 8 pointers, 8 int64, 20 int32 + float, batch/batch_inner (+pad to 8), 8 int64 batch strides, accumulate + fp8, 2 scale pointers,
    # out2 + ldo2 + split_n + ln_eps, ln_c1, out_row_width (+pad) + ldo_hi, up_phases (+tail pad)
    base = 8 * 8 + 8 * 8 + 24 * 4 + 8 * 8 + 8 + 2 * 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8  # ... k_append in up_phases' tail pad, a3, C3 (+pad), lda2
    assert ctypes.sizeof(_lib.StatsSink) == 32 and ctypes.sizeof(_lib.NormIn) == 56  # pointer + 6 int32; 3 pointers + float + 6 int32 (+ tail pad)
    assert ctypes.sizeof(_lib.NormOut) == 40  # 3 pointers + float + 3 int32
    assert ctypes.sizeof(_lib.GemmDesc) == base + 32 + 56 + 40  # the GroupNorm bridge's sink + norm_in, the reduce-side norm_out
    # ... and the compiled structs agree with the binding's (the loader refuses a mismatch)
    for which, cls in enumerate((_lib.GemmDesc, _lib.AttnDesc, _lib.GroupNormDesc, _lib.TBlockDesc, _lib.ConvGnDesc, _lib.StatsSink, _lib.NormIn, _lib.NormOut)):
        assert int(lib.gn_desc_sizeof(which)) == ctypes.sizeof(cls), cls.__name__


def test_product_fails_loudly_without_gpu():
    import torch

    from genima_amd import configs
    from genima_amd.engine import Engine
    from genima_amd.host import UNet2DConditionModel
    from genima_amd._lib import GenimaHipError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(GenimaHipError):
        Engine("cuda:0")
    with pytest.raises(GenimaHipError):
        Engine("cpu")
    m = UNet2DConditionModel.from_config(configs.TINY_UNET)
    with pytest.raises(GenimaHipError):
        m(torch.zeros(1, 4, 16, 16), 999, torch.zeros(1, 77, 128))
    with pytest.raises(GenimaHipError):
        m.to("cuda")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "genima_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"


def test_driver_entry_point_builds():
    """__graft_entry__.build() is what the driver runs on CPU each round: it must build (a no-op when the library is current), load the library,
    accept its version and import the package + the oracle modules."""
    import importlib
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    entry = importlib.import_module("__graft_entry__")
    entry.build()
    assert callable(entry.smoke)
