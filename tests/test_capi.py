"""CPU tests: the C-ABI library loads and exports every symbol include/dvpari.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dvpari.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(dvp, nat):
    lib = ctypes.CDLL(nat.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # and the python binding knows each of them
    assert set(syms) <= set(nat._SIGS), sorted(set(syms) - set(nat._SIGS))


def test_status_strings_and_version(dvp):
    assert dvp.lib.dvp_version() >= 100
    assert b"ok" == dvp.lib.dvp_strerror(0)
    assert b"invalid" in dvp.lib.dvp_strerror(-2)


def test_no_cpu_fallback_in_product_path():
    """The product package must not import or link anything under oracle/."""
    pkg = os.path.join(ROOT, "dv-pari_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import pyref" not in src and "c_oracle" not in src and "dvp_oracle" not in src, f
