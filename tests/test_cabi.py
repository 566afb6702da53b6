"""The C-ABI library builds, loads without a GPU and exports every symbol include/padne_hip.h declares."""
import ctypes
import os
import re

import pytest

from padne_amd import _hip, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="padne_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(padne_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    lib = build.build(verbose=False)
    assert os.path.exists(lib) and lib.startswith(ROOT)


def test_every_declared_symbol_is_exported_and_bound():
    build.build(verbose=False)
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in padne_hip.h but not exported"
        assert name in _hip.SIGNATURES, f"{name} has no ctypes prototype in padne_amd/_hip.py"
    for name in _hip.SIGNATURES:
        assert name in names, f"{name} bound in _hip.py but not declared in the header"
    # the in-process team (several ranks on one GPU) is test scaffolding: its own header, not the drop-in boundary
    test_names = declared_symbols("padne_hip_test.h")
    # (+ one introspection call the multi-rank tests use to see that the product split is really in use)
    assert test_names and all(n.startswith(("padne_team_", "padne_ctx_join_team", "padne_csr_split_tiles", "padne_ctx_lockstep_groups",
                                                   "padne_asm_second_path_count", "padne_ctx_halo_exchange_time", "padne_ctx_reload_options")) for n in test_names)
    assert not set(test_names) & set(names)
    for name in test_names:
        assert hasattr(lib, name) and name in _hip.TEST_SIGNATURES
    assert sorted(_hip.TEST_SIGNATURES) == test_names


def test_abi_version_and_error_string():
    lib = _hip.load_library()
    assert lib.padne_abi_version() == 1
    assert isinstance(lib.padne_last_error(), bytes)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_product_path_fails_loudly_without_gpu():
    with pytest.raises(_hip.HipUnavailableError):
        _hip.Context(0)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(_hip.HipUnavailableError):
        _hip.load_library(str(tmp_path / "libpadne_hip.so"))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "padne_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "__never__", f"{f} mentions the oracle"
