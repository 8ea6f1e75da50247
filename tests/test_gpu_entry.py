"""The driver's hooks in ONE process, in the order build() -> smoke(): the package is imported (and the C-ABI library
loaded) before anything has touched torch.cuda — the order that once left two HIP runtimes in the process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_then_smoke_in_one_process():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout
