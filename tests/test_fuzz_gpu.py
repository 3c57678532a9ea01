"""The differential fuzzers of tools/ as part of the GPU suite: a fixed seed and a short time budget each
(tools/fuzz_msm.py: MSM entry points vs the oracle; tools/fuzz_ckzg.py: c-kzg surface and NTT vs the oracle)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,seed", [("fuzz_msm.py", 11), ("fuzz_ckzg.py", 12)])
def test_differential_fuzz(tool, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "25", str(seed)], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900, cwd=ROOT)
    out = p.stdout.decode()
    assert p.returncode == 0 and "fuzz ok" in out, out[-3000:]
    cases = int(out.split("fuzz ok:")[1].split()[0])
    assert cases >= 10, out[-500:]
