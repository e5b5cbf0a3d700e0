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
                         ("conv_split", -1), ("conv_split_tile", 0), ("prof_dump", 0), ("dec_rows", 0), ("parseq_no_rowmax", 0), ("amax_check", 0)):
        assert lib.ymk_debug_option(key.encode(), default) == 0, key
    src = open(os.path.join(ROOT, "yomitoku_amd", "csrc", "ymk_conv.hip")).read()
    assert int(re.search(r"g_conv_fast\{(\d+)\}", src).group(1)) == _lib.CONV_FAST_DEFAULT
    header = open(os.path.join(ROOT, "include", "ymk.h")).read()
    assert int(re.search(r'"conv_fast" \((\d+)\)', header).group(1)) == _lib.CONV_FAST_DEFAULT
