"""CPU: the C-ABI library loads and exports every symbol include/srganst.h declares, and the ctypes
signature table agrees with the header (names and argument counts).  No compute calls (no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def header_decls():
    src = open(os.path.join(ROOT, "include", "srganst.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(sst_\w+)\s*\(([^)]*)\)\s*;", src):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        decls[m.group(1)] = n
    return decls


def test_header_matches_ctypes_table_and_exports():
    from srganst import _abi
    decls = header_decls()
    assert len(decls) >= 35
    assert set(decls) == set(_abi.SIGNATURES), set(decls) ^ set(_abi.SIGNATURES)
    lib = ctypes.CDLL(_abi.LIB_PATH)
    for name, nargs in decls.items():
        assert hasattr(lib, name), f"{name} not exported by libsrganst.so"
        assert len(_abi.SIGNATURES[name][1]) == nargs, name


def test_library_identity_and_error_channel():
    from srganst import _abi
    lib = _abi.lib()
    assert lib.sst_arch() == b"gfx950" and lib.sst_version() >= 100
    # argument validation happens on the host before any launch: safe to exercise without a GPU
    rc = lib.sst_conv_fwd(None, None, None, None, None, None, None, None, 0.0, 0, None, None, None, 0, 1, 8, 8, 8, 8, 3, 1, None)
    assert rc != 0 and b"null pointer" in lib.sst_last_error()
    n = ctypes.c_int64()
    assert lib.sst_st_loss_workspace(16, 96, 96, ctypes.byref(n)) == 0 and n.value == 16 * 9
    assert lib.sst_conv_mtiles(16, 24, 24) == 288
    assert lib.sst_conv_packed_floats(64, 64, 3) == 2 * 1 * 9 * 8 * 256 + 2048 + 64 * 64 * 9   # + band-kernel copy
    assert lib.sst_conv_packed_floats(128, 128, 3) == 4 * 2 * 9 * 8 * 256 + 2048
    assert lib.sst_conv_stat_tiles(16, 24, 24, 64, 64, 3, 1) == 192 and lib.sst_conv_stat_tiles(16, 24, 24, 64, 64, 3, 2) == 96
