"""
The dask front-ends under a REAL dask, threaded scheduler.  dask is not installed for the interpreter the GPU tests
run under (/usr/bin/python3), but the image also carries /opt/conda/bin/python3.9 with dask 2021.10 and numpy 1.26,
and the host path of the package needs neither torch nor anything newer: this test runs tests/dask_cases.py in that
interpreter as a child process (skipped, loudly, only where no such interpreter exists).  The no-dask counterpart,
which always runs, is tests/test_gpu_blocks.py.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANDIDATES = ("/opt/conda/bin/python3.9", "/opt/conda/bin/python3", "/opt/conda/bin/python")


def _python_with_dask():
    try:
        import dask  # noqa: F401
        return sys.executable
    except ImportError:
        pass
    for exe in CANDIDATES:
        if os.path.exists(exe):
            rc = subprocess.call([exe, "-c", "import dask.array, numpy"], stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL)
            if rc == 0:
                return exe
    return None


def test_dask_front_ends_with_real_dask_threaded():
    exe = _python_with_dask()
    if exe is None:
        pytest.skip("no interpreter with dask on this box (tests/test_gpu_blocks.py covers the block contract)")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "dask_cases.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0 and "DASK_CASES_OK" in out, out[-4000:]
