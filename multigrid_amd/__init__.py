"""multigrid_amd -- MI355X-native geometric multigrid V-cycle engine.

csrc/   hand-written gfx950 HIP kernels and the C ABI (include/mgrt.h) -> libmgrt.so
host/   C++ classes with the reference's public interface (Buffer2D/3D, MultigridSolver2D/3D,
        ProgramState) on top of the C ABI -> multigrid2D / multigrid3D executables
_lib.py ctypes binding of the C ABI over torch device memory, for tests and bench.py
"""
from . import _lib  # noqa: F401
