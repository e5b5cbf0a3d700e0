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
