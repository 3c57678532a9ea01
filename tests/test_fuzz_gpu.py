"""The differential fuzzers of tools/ as part of the GPU suite: a fixed seed and a time budget each
(tools/fuzz_msm.py: MSM entry points vs the oracle; tools/fuzz_ckzg.py: c-kzg surface and NTT vs the oracle;
tools/fuzz_g1.py: the G1 transforms under every stage form and the FK20 cell proofs), against BOTH builds of the
library (conftest.py: the product library and the forced-rare-path one; KZGAMD_LIB carries the choice to the tool).
The only wrong result this code base ever shipped (round 3's four-wave block sum, once per 1e5 small batches) was found
by fuzz_ckzg.py, so every run of the suite fuzzes both builds; longer sessions are logged under profiles/."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

# seconds per tool.  The whole -m gpu suite (both builds) is sized to stay under nine minutes on one MI355X; the fuzzers
# get 2.5 of them by default.  KZGAMD_FUZZ_SCALE multiplies every budget: 2.7 gives the 120 / 120 / 40 s per tool asked for
# after round 4 (profiles/r05_gpu_suite_both_flavours_long_fuzz.log is that run: 361 passed in 738 s), and
# tools/fuzz_session.sh runs sessions of any length outside the suite (profiles/r05_long_fuzz.log: 32 k cases).
BUDGET = {"product": {"fuzz_msm.py": 45, "fuzz_ckzg.py": 45, "fuzz_g1.py": 15},
          "exact": {"fuzz_msm.py": 10, "fuzz_ckzg.py": 10, "fuzz_g1.py": 8}}
SCALE = float(os.environ.get("KZGAMD_FUZZ_SCALE", "1"))


@pytest.mark.parametrize("tool,seed", [("fuzz_msm.py", 11), ("fuzz_ckzg.py", 12), ("fuzz_g1.py", 13)])
def test_differential_fuzz(kzg, tool, seed):
    flavour = "exact" if kzg.LIB_PATH.endswith("_exact.so") else "product"
    assert os.environ.get("KZGAMD_LIB") == kzg.LIB_PATH  # the tool loads the same build
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(int(BUDGET[flavour][tool] * SCALE)), str(seed)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200, cwd=ROOT)
    out = p.stdout.decode()
    assert p.returncode == 0 and "fuzz ok" in out, out[-3000:]
    assert os.path.basename(kzg.LIB_PATH) in out, out[-500:]  # the tool reports which library it ran
    cases = int(out.split("fuzz ok:")[1].split()[0])
    assert cases >= 10, out[-500:]
