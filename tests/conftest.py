import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built():
    """Everything native is built once per session (a no-op when the .so files travelled with the snapshot)."""
    need = [os.path.join(ROOT, "regtools_amd", "libregtools_amd.so"), os.path.join(ROOT, "regtools_amd", "libregtools_synth.so"),
            os.path.join(ROOT, "oracle", "oracle_cli"), os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"), os.path.join(ROOT, "bin", "regtools-amd")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def oracle_cli():
    return os.path.join(ROOT, "oracle", "oracle_cli")


def run_oracle(args):
    r = subprocess.run([os.path.join(ROOT, "oracle", "oracle_cli"), "extract"] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout, r.stderr


@pytest.fixture(scope="session")
def gpu_ctx():
    """The device context of the -m gpu tests.  On a machine without a gfx950 device the product refuses to start (RGX_ERR_NO_DEVICE: there is
    no CPU fall-back) and the tests that need it are SKIPPED, not errors."""
    import regtools_amd
    try:
        ctx = regtools_amd.Context(0)
    except regtools_amd.RegtoolsError as e:
        if e.code == 4:                                   # RGX_ERR_NO_DEVICE
            pytest.skip("no MI355X here: %s" % str(e).strip())
        raise
    yield ctx
    ctx.close()
