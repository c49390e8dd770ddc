"""CPU tests: the C-ABI library loads and exports every symbol include/*.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="dvpari.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(dvp, nat):
    lib = ctypes.CDLL(nat.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    internal = declared_symbols("dvpari_internal.h")
    # test / sweep / measurement entries live in the internal header only: the boundary header stays what a host binds
    assert not [s for s in syms if s.startswith(("dvp_debug_", "dvp_ubench_", "dvp_profile_", "dvp_tune_")) or "_debug_" in s]
    assert not set(syms) & set(internal)
    syms = syms + internal
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


def test_cpp_example_builds_against_the_header(tmp_path):
    """examples/dvp_prove_cli.cpp sees only include/dvpari.h: it must compile with plain g++ and link against the
    library; without a GPU it reports that (exit 3), with one it reports the missing cache_dir (exit 1)"""
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    libdir = os.path.join(ROOT, "dv-pari_amd")
    exe = tmp_path / "dvp_prove_cli"
    subprocess.check_call([gxx, "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "dvp_prove_cli.cpp"), "-L" + libdir, "-ldvpari_hip", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe), str(tmp_path / "nowhere"), "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode in (1, 3), (out.returncode, out.stderr)
