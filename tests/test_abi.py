"""CPU checks of the C-ABI boundary: the library loads without a GPU and exports every symbol
include/ymk.h declares; the ctypes table in yomitoku_amd/_lib.py covers the same set."""
import os
import re

from yomitoku_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ymk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ymk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in ymk.h but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _header_symbols()


def test_version_and_error_slot():
    lib = _lib.load()
    assert lib.ymk_version() >= 100
    # an unknown model kind fails cleanly (no GPU needed: hipSetDevice fails first or kind check)
    h = lib.ymk_model_create(b"no-such-model", 0)
    assert not h
    assert lib.ymk_last_error()


def test_debug_options_exist_and_their_documented_defaults_agree():
    """ymk_debug_option keys the tools and tests use are known to the library (no GPU needed: they only set process-wide
    switches), an unknown key is refused, and the default of "conv_fast" is the same number in the kernel source, the
    header comment and the Python constant the tests reset it to."""
    lib = _lib.load()
    assert lib.ymk_debug_option(b"no_such_option", 1) != 0
    for key, default in (("conv_fast", _lib.CONV_FAST_DEFAULT), ("conv_variant", 0), ("no_splitk", 0), ("splitk_force", -1),
                         ("conv_split", -1), ("conv_split_tile", 0), ("prof_dump", 0), ("dec_rows", 0), ("parseq_no_rowmax", 0), ("amax_check", 0), ("rowmax_tile", 0), ("astat", 1), ("act_planes", 1), ("ar_publish", 1)):
        assert lib.ymk_debug_option(key.encode(), default) == 0, key
    src = open(os.path.join(ROOT, "yomitoku_amd", "csrc", "ymk_conv.hip")).read()
    assert int(re.search(r"g_conv_fast\{(\d+)\}", src).group(1)) == _lib.CONV_FAST_DEFAULT
    header = open(os.path.join(ROOT, "include", "ymk.h")).read()
    assert int(re.search(r'"conv_fast" \((\d+)\)', header).group(1)) == _lib.CONV_FAST_DEFAULT


def test_library_loads_with_roctx_ranges_asked_for():
    """YMK_ROCTX=1: the marker library is looked up at run time (dlopen) at the first range - asking for ranges must never be
    the reason the library does not load or a host-only entry point fails, with or without a marker library on the box."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from yomitoku_amd import _lib; l = _lib.load(); assert l.ymk_version() >= 100; print(l.ymk_device_count() >= -1)"
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, YMK_ROCTX="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("True"), r.stderr[-500:]
