"""CPU checks of the C-ABI library: it loads without a GPU, exports every symbol that
include/mvmc.h declares, and its host-side MT19937 table equals NumPy's RandomState(0)."""
import ctypes
import os
import re

import pytest

import numpy as np

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mvmc.h")).read()
    return sorted(set(re.findall(r"\b(mvmc_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    from multiview_motion_capture_amd import _cabi
    lib = _cabi.load()
    declared = _declared_symbols()
    assert set(declared) == set(_cabi.SYMBOLS), (declared, _cabi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    header = open(os.path.join(ROOT, "include", "mvmc.h")).read()
    assert lib.mvmc_abi_version() == int(re.search(r"#define\s+MVMC_ABI_VERSION\s+(\d+)", header).group(1)) == _cabi.MVMC_ABI == 6


def test_seed_table_is_numpy_randomstate0():
    from multiview_motion_capture_amd import _cabi
    lib = _cabi.load()
    n = 64 * 64
    buf = (ctypes.c_double * n)()
    assert lib.mvmc_als_seed_table(ctypes.cast(buf, ctypes.c_void_p), n) == 0
    assert np.array_equal(np.frombuffer(buf, dtype=np.float64), np.random.RandomState(0).rand(n))
    # rand(n, r) is the row-major prefix of the same stream (mv_association.py:271)
    assert np.array_equal(np.frombuffer(buf, dtype=np.float64)[:20 * 8].reshape(20, 8),
                          np.random.RandomState(0).rand(20, 8))


def test_bad_arguments_are_reported_not_raised_in_c():
    from multiview_motion_capture_amd import _cabi
    lib = _cabi.load()
    assert lib.mvmc_fmats(None, None, 5, None, None) == 1
    assert lib.mvmc_status_string(1).decode() == "invalid argument"


def test_chain_buffers_struct_matches_header():
    """mvmcChainBuffers is passed by pointer: the ctypes mirror must list the header's fields in the header's order."""
    from multiview_motion_capture_amd import _cabi
    text = open(os.path.join(ROOT, "include", "mvmc.h")).read()
    body = text[text.index("typedef struct mvmcChainBuffers {"):text.index("} mvmcChainBuffers;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    ints, ptrs = [], []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        if "*" in decl:
            ptrs.append(decl.split("*")[-1].strip())
        else:
            assert decl.startswith("int32_t"), decl
            ints += [n.strip() for n in decl[len("int32_t"):].split(",")]
    assert tuple(ints) == _cabi.MvmcChainBuffers._INTS
    assert tuple(ptrs) == _cabi.MvmcChainBuffers._PTRS
    # 14 int32 fields (56 bytes): the first pointer starts there
    assert len(ints) == 14 and _cabi.MvmcChainBuffers.kps17.offset == 56
    assert ctypes.sizeof(_cabi.MvmcChainBuffers) == 56 + 8 * len(ptrs)


def test_a_library_of_another_abi_version_is_refused(tmp_path):
    """load() declares the argument types of ONE ABI version: a library that reports another (an old build behind MVMC_LIB_PATH for an
    A/B run) must be refused with a clear error instead of being called with shifted arguments (ADVICE, round 4)."""
    import subprocess, sys, textwrap
    src = tmp_path / "stub.c"
    src.write_text("int mvmc_abi_version(void) { return 3; }\n")
    so = tmp_path / "libmvmc_stub.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        from multiview_motion_capture_amd import _cabi
        try:
            _cabi.load()
        except _cabi.MvmcError as e:
            assert "ABI version 3" in str(e) and "version %%d" %% _cabi.MVMC_ABI in str(e), str(e)
            print("refused")
    """ % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MVMC_LIB_PATH=str(so)), capture_output=True, text=True)
    assert out.returncode == 0 and "refused" in out.stdout, out.stderr


def test_a_library_built_from_other_kernel_sources_is_refused(tmp_path, monkeypatch):
    """lib/BUILD_INFO.json (written by csrc/Makefile after linking) carries the hash of the kernel sources the library was built from;
    _cabi.load() refuses a library whose hash is not the tree's -- a stale .so that travelled to the GPU box -- or that has no record."""
    import json
    from multiview_motion_capture_amd import _buildinfo, _cabi
    info = _buildinfo.read()
    assert info is not None and info["kernel_sources_sha"] == _buildinfo.sources_sha(), "build() leaves a record of this tree"
    assert _cabi.build_info()["kernel_sources_sha"] == info["kernel_sources_sha"]
    monkeypatch.delenv("MVMC_LIB_PATH", raising=False)
    monkeypatch.setattr(_cabi, "_lib", None)
    stale = tmp_path / "BUILD_INFO.json"
    stale.write_text(json.dumps(dict(info, kernel_sources_sha="0" * 16)))
    monkeypatch.setattr(_buildinfo, "INFO_PATH", str(stale))
    with pytest.raises(_cabi.MvmcError, match="stale"):
        _cabi.load()
    monkeypatch.setattr(_buildinfo, "INFO_PATH", str(tmp_path / "missing.json"))
    with pytest.raises(_cabi.MvmcError, match="no BUILD_INFO.json"):
        _cabi.load()
