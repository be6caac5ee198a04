"""Static checks on the generated code of the fused smoother (no GPU needed: hipcc cross-compiles).

The kernel prefetches with loads the compiler cannot see; tools/check_hidden_loads.py verifies on
the assembly that no compiler-generated instruction copies a register while such a load is pending
inside the time loop and that no variant spills to scratch memory."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_fused_smoother_has_no_copies_of_inflight_registers():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_hidden_loads.py")],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert " 0 findings" in p.stdout
