"""Short, fixed-seed runs of the randomized soaks under tools/ (round 4: the long runs are recorded in profiles/r04_soak.txt). Each script exits non-zero on the
first case that leaves its contract: parity mode bit-equal to the oracle (framebuffer + every counter) over random scenes x render knobs x shard splits x replica
groups x reference-RNG mode x device film; the production trees under extreme scales, offsets, shapes and awkward rays (nothing lost, nothing farther);
the analytic primitives against the oracle and the independent float64 closed form."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args", [("soak_parity.py", ["14", "31"]), ("soak_wide_rays.py", ["30", "32"]), ("soak_prims.py", ["12", "33"])])
def test_randomized_soak_short_run(script, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args, capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-6:])
    assert r.returncode == 0, tail
    assert " cases" in r.stdout.splitlines()[-1], tail
