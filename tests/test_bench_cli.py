"""bench.py's handling of several ranks, as far as a machine without a GPU can show it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ranks_without_a_gpu_walk_the_attempts_and_give_up():
    """python bench.py --gpus 2 starts its own ranks; every rank's worker stops for lack of a GPU, the
    watching parent goes through its list of attempts once and leaves with a non-zero status: no result
    line, no endless retry, nothing computed on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode != 0
    assert not [line for line in p.stdout.splitlines() if line.startswith("{")]
    # the launcher ends the other rank as soon as one has given up, so the last line of one of them may be missing
    for rank in (0, 1):
        assert 3 <= p.stderr.count("bench.py rank %d:" % rank) <= 4, p.stderr[-3000:]
    assert "trying: RCCL, direct launches" in p.stderr and "trying: host callbacks over gloo, the line's own workload only" in p.stderr
    assert "giving up" in p.stderr and "no GPU visible" in p.stderr
