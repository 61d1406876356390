"""A short run of the randomised parity sweeps (tools/fuzz_parity.py) with fixed seeds: shapes nobody wrote down by hand.
The long runs are recorded in profiles/r03_fuzz_parity.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode,cases,seed", [("seq", 60, 11), ("stack", 40, 12)])
def test_random_shapes_against_the_oracle(mode, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), str(cases), str(seed), mode],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " 0 FAILED" in r.stdout
