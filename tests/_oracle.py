"""ctypes bindings of the CPU oracle libraries (test infrastructure only).

`load("mgo")` -> oracle/libmg_oracle.so, the build's own C restatement.
`load("mgr")` -> oracle/_ref/libmg_ref.so, the reference's own kernels compiled for x86-64; present
only where it was built from /root/reference (`make -C oracle ref`), otherwise None.

Both expose the interface of oracle/mg_oracle.h through the same Python wrapper class.
"""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATHS = {
    "mgo": os.path.join(ROOT, "oracle", "libmg_oracle.so"),
    "mgr": os.path.join(ROOT, "oracle", "_ref", "libmg_ref.so"),
}
INNER, OUTER, DIRICHLET, NEUMANN = 0, 1, 2, 3
CELLTYPE_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, prefix, path):
        self.prefix = prefix
        self.lib = C.CDLL(path)
        self._cb = None
        i, p, l = C.c_int, C.c_void_p, C.c_long
        for S, r in (("_f32", C.c_float), ("_f64", C.c_double)):
            self._sig("smooth_pass" + S, None, [i, p, p, p, r, i, i, i, i, i])
            self._sig("residual" + S, None, [i, p, p, p, p, i, i, i])
            self._sig("restrict" + S, None, [i, p, p, p, i, i, i, i, i, i])
            self._sig("correct" + S, None, [i, p, p, p, p, i, i, i])
            self._sig("prolong" + S, None, [i, p, p, p, i, i, i])
            self._sig("zero_out" + S, None, [i, p, p, i, i, i])
            self._sig("l2norm_tree" + S, r, [p, l])
            self._sig("linfnorm_tree" + S, r, [p, l])
            self._sig("diff" + S, None, [p, p, p, l])
            self._sig("scale" + S, None, [p, p, C.c_float, l])
            self._sig("smoother_iterate" + S, None, [i, p, p, r, i, i, i, i, i])
            self._sig("iterate" + S, None, [i, p, p, r, i, i, i, i, i, i, i])
            self._sig("mg" + S, None, [i, p, p, r, i, i, i, i, i, i, i, i])
            self._sig("fmg" + S, None, [i, p, p, r, i, i, i, i, i, i, i, i])
            self._sig("discretize" + S, None, [i, p, p, i, i, i])
            self._sig("l2norm_exact" + S, C.c_double, [p, l])
        self._sig("box_cells", None, [i, p, i, i, i, C.POINTER(C.c_int)])
        self._sig("set_celltype", None, [C.c_void_p])
        self._sig("set_builtin_celltype", None, [i])
        self._sig("copy_level_cells", None, [i, p, i, i, i])
        self._sig("set_box_faces", None, [p])
        self._sig("num_threads", i, [])
        self._sig("set_num_threads", None, [i])

    def _sig(self, name, res, args):
        f = getattr(self.lib, self.prefix + "_" + name)
        f.restype = res
        f.argtypes = args
        setattr(self, "_" + name, f)

    @staticmethod
    def _S(a):
        return {np.dtype(np.float32): "_f32", np.dtype(np.float64): "_f64"}[a.dtype]

    @staticmethod
    def _dims(shape):
        """numpy shape (nz, ny, nx) or (ny, nx) -> dim, nx, ny, nz"""
        if len(shape) == 3:
            return 3, shape[2], shape[1], shape[0]
        return 2, shape[1], shape[0], 1

    # ---- kernels (arrays are C-contiguous numpy arrays indexed [z, y, x] / [y, x]) -------------
    def smooth_pass(self, cell, u, f, omega, odd, zcount=None):
        dim, nx, ny, nz = self._dims(u.shape)
        if zcount is None:
            zcount = nz
        getattr(self, "_smooth_pass" + self._S(u))(dim, _ptr(cell), _ptr(u), _ptr(f), omega, nx, ny, nz, odd, zcount)

    def residual(self, cell, r, u, f):
        dim, nx, ny, nz = self._dims(u.shape)
        getattr(self, "_residual" + self._S(u))(dim, _ptr(cell), _ptr(r), _ptr(u), _ptr(f), nx, ny, nz)

    def restrict(self, cell, coarse, fine):
        dim, nx, ny, nz = self._dims(fine.shape)
        _, cx, cy, cz = self._dims(coarse.shape)
        getattr(self, "_restrict" + self._S(fine))(dim, _ptr(cell), _ptr(coarse), _ptr(fine), nx, ny, nz, cx, cy, cz)

    def correct(self, cell, out, u, e):
        dim, nx, ny, nz = self._dims(u.shape)
        getattr(self, "_correct" + self._S(u))(dim, _ptr(cell), _ptr(out), _ptr(u), _ptr(e), nx, ny, nz)

    def prolong(self, cell, out, coarse):
        dim, nx, ny, nz = self._dims(out.shape)
        getattr(self, "_prolong" + self._S(out))(dim, _ptr(cell), _ptr(out), _ptr(coarse), nx, ny, nz)

    def zero_out(self, cell, u):
        dim, nx, ny, nz = self._dims(u.shape)
        getattr(self, "_zero_out" + self._S(u))(dim, _ptr(cell), _ptr(u), nx, ny, nz)

    def l2norm_tree(self, x):
        return getattr(self, "_l2norm_tree" + self._S(x))(_ptr(x), x.size)

    def linfnorm_tree(self, x):
        return getattr(self, "_linfnorm_tree" + self._S(x))(_ptr(x), x.size)

    def l2norm_exact(self, x):
        return getattr(self, "_l2norm_exact" + self._S(x))(_ptr(x), x.size)

    def diff(self, a, b):
        out = np.empty_like(a)
        getattr(self, "_diff" + self._S(a))(_ptr(out), _ptr(a), _ptr(b), a.size)
        return out

    def scale(self, a, m):
        out = np.empty_like(a)
        getattr(self, "_scale" + self._S(a))(_ptr(out), _ptr(a), m, a.size)
        return out

    # ---- host drivers --------------------------------------------------------------------------
    def smoother_iterate(self, u, f, omega, a, zfull=False):
        dim, nx, ny, nz = self._dims(u.shape)
        getattr(self, "_smoother_iterate" + self._S(u))(dim, _ptr(u), _ptr(f), omega, a, nx, ny, nz, int(zfull))

    def iterate(self, u, f, omega, a1, a2, v, zfull=False):
        dim, nx, ny, nz = self._dims(u.shape)
        getattr(self, "_iterate" + self._S(u))(dim, _ptr(u), _ptr(f), omega, a1, a2, v, nx, ny, nz, int(zfull))

    def mg(self, u, f, omega, a1, a2, v, iters, zfull=False):
        dim, nx, ny, nz = self._dims(u.shape)
        getattr(self, "_mg" + self._S(u))(dim, _ptr(u), _ptr(f), omega, a1, a2, v, iters, nx, ny, nz, int(zfull))

    def fmg(self, f, omega, a1, a2, v, iters, zfull=False):
        dim, nx, ny, nz = self._dims(f.shape)
        u = np.zeros_like(f)
        getattr(self, "_fmg" + self._S(f))(dim, _ptr(u), _ptr(f), omega, a1, a2, v, iters, nx, ny, nz, int(zfull))
        return u

    # ---- setup ---------------------------------------------------------------------------------
    def box_cells(self, shape, faces=(DIRICHLET,) * 6):
        dim, nx, ny, nz = self._dims(shape)
        cell = np.empty(shape, dtype=np.uint8)
        self._box_cells(dim, _ptr(cell), nx, ny, nz, (C.c_int * 6)(*faces))
        return cell

    def set_celltype(self, fn):
        """fn(x, y, z, nx, ny, nz) -> cell type, or None for the all-Dirichlet box."""
        self._cb = CELLTYPE_FN(fn) if fn is not None else None
        self._pyfn = fn
        self._set_celltype(C.cast(self._cb, C.c_void_p) if self._cb else None)

    def set_builtin_celltype(self, kind):
        """kind 1: the ball rule compiled into the oracle (no callback per point); 0: back to the box."""
        self._cb = self._pyfn = None
        self._set_builtin_celltype(kind)

    def set_box_faces(self, faces):
        """The box with these six face types on every level (None: back to the all-Dirichlet box)."""
        self._cb = self._pyfn = None
        self._set_box_faces((C.c_int * 6)(*faces) if faces is not None else None)

    def level_cells(self, shape):
        """The byte map of a grid under the active cell rule, as the cycle drivers see it."""
        dim, nx, ny, nz = self._dims(shape)
        cell = np.empty(shape, dtype=np.uint8)
        self._copy_level_cells(dim, _ptr(cell), nx, ny, nz)
        return cell

    def discretize(self, problem, shape, dtype):
        """problem 0: 3D tridimF1/tridimS1; 1: 2D sinfunc2; 2: 2D sinfunc1 -> (func, sol)"""
        dim, nx, ny, nz = self._dims(shape)
        func = np.zeros(shape, dtype=dtype)
        sol = np.zeros(shape, dtype=dtype)
        getattr(self, "_discretize" + self._S(func))(problem, _ptr(func), _ptr(sol), nx, ny, nz)
        return func, sol

    def num_threads(self):
        return self._num_threads()

    def set_num_threads(self, n):
        self._set_num_threads(n)


_cache = {}


def load(prefix="mgo"):
    if prefix not in _cache:
        path = _PATHS[prefix]
        _cache[prefix] = Oracle(prefix, path) if os.path.exists(path) else None
    return _cache[prefix]
