"""Shared fixtures: package loader, library handles, markers."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "visual-inertial-odometry_amd")
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def load_package():
    """The package directory name carries hyphens, so it is imported under the alias `vio_amd`."""
    if "vio_amd" in sys.modules:
        return sys.modules["vio_amd"]
    spec = importlib.util.spec_from_file_location(
        "vio_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["vio_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference; only where /root/reference exists)")


@pytest.fixture(scope="session")
def vio():
    return load_package()


@pytest.fixture(scope="session")
def oracle_lib(vio):
    """CPU oracle (test infrastructure).  Built on demand with gcc."""
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "vio_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return vio.VioLib(so, "vioo_")


@pytest.fixture(scope="session")
def ref_lib(vio):
    """The compiled reference backend (oracle/_ref).  Skips where it cannot exist."""
    so = os.path.join(ORACLE_DIR, "_ref", "libvio_ref.so")
    if not os.path.exists(so):
        if not os.path.isdir("/root/reference/workspace"):
            pytest.skip("oracle/_ref is only buildable where /root/reference is mounted")
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "ref"])
    return vio.VioLib(so, "vior_")


@pytest.fixture(scope="session")
def hip_lib(vio):
    """The product library.  No fallback: a missing build is a failure, not a skip."""
    return vio.load_hip()


@pytest.fixture(scope="session")
def hip_debug_lib(vio):
    """The product's sources built with -DVIO_DEBUG_ENTRY_POINTS (vio_debug_chain_solve): not the product library."""
    # (build_hip is a no-op unless a source is newer than the library: a stale debug build must not test an old kernel against new sources;
    #  on a box without hipcc the library that travelled with the snapshot is used as it is)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    try:
        g.build_hip()
    except RuntimeError as exc:
        if "hipcc not found" not in str(exc):
            raise
    return vio.load_hip_debug()
