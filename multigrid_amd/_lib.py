"""ctypes binding of libmgrt.so (the C ABI declared in include/mgrt.h).

This is plumbing for tests and bench.py: device memory comes from torch tensors, whose data
pointers and stream handle are passed straight through the C ABI.  There is no fallback: when the
shared library is missing, or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGRT_LIB") or os.path.join(_HERE, "libmgrt.so")  # MGRT_LIB: development, A/B builds

INNER, OUTER, DIRICHLET, NEUMANN = 0, 1, 2, 3
ALL_DIRICHLET = (DIRICHLET,) * 6


class MgrtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("mgrt status %d: %s" % (status, message))
        self.status = status


class Grid(C.Structure):
    """mgrt_grid"""
    _fields_ = [
        ("dim", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("pitch", C.c_int), ("slab", C.c_longlong),
        ("y0", C.c_int), ("gny", C.c_int),
        ("cells", C.c_void_p), ("face", C.c_int * 6), ("neumann", C.c_int),
    ]


class PlanDesc(C.Structure):
    """mgrt_plan_desc"""
    _fields_ = [
        ("dim", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("elem_size", C.c_int), ("face", C.c_int * 6), ("zfull", C.c_int), ("use_graph", C.c_int),
        ("no_fuse", C.c_int), ("no_bottom", C.c_int), ("neumann", C.c_int), ("separate_restrict", C.c_int),
    ]


SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_size_t))
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
TRANSPORT_NONE, TRANSPORT_RCCL, TRANSPORT_CALLBACK = 0, 1, 2


class Transport(C.Structure):
    """mgrt_transport"""
    _fields_ = [
        ("kind", C.c_int), ("rank", C.c_int), ("nranks", C.c_int), ("rccl_id", C.c_ubyte * 128),
        ("ctx", C.c_void_p), ("sendrecv", SENDRECV_FN), ("allgather", ALLGATHER_FN), ("allreduce_sum", ALLREDUCE_FN),
    ]


class SlabDesc(C.Structure):
    """mgrt_slab_desc"""
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("elem_size", C.c_int), ("face", C.c_int * 6),
        ("zfull", C.c_int), ("use_graph", C.c_int), ("crossover", C.c_longlong),
    ]


# every symbol include/mgrt.h declares; checked by the CPU test-suite against the header
_KERNELS = ["smooth_pass", "smooth", "smooth_sweep", "smooth_sweep_residual", "smooth_sweep_residual_restrict", "correct_smooth_sweep", "residual", "restrict", "correct", "prolong", "zero_outside",
            "diff", "scale", "sumsq", "maxabs", "l2norm_reftree", "linfnorm_reftree", "sum_reftree"]
_RUNTIME = ["init", "shutdown", "last_error", "device_name", "alloc", "free", "upload", "download",
            "stream_create", "stream_destroy", "sync", "zero", "box_cells_host",
            "plan_create", "plan_destroy", "plan_levels", "plan_level_dims", "plan_set_cells",
            "plan_vcycle", "plan_fmg", "correct_smooth_sweep_supported", "smooth_sweep_residual_restrict_supported",
            "sumsq_noalloc_f32", "sweep_parts", "last_sweep_launches",
            "rccl_unique_id", "rccl_version", "rccl_selftest", "slab_plan_create", "slab_plan_destroy", "slab_plan_distributed_levels",
            "slab_plan_uses_graph", "slab_plan_rows", "slab_plan_vcycle", "slab_plan_fmg"]
SYMBOLS = ["mgrt_" + n for n in _RUNTIME] + ["mgrt_%s_%s" % (k, s) for k in _KERNELS for s in ("f32", "f64")]

_lib = None


def load():
    """Load libmgrt.so and declare prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                          % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    i, p, ll, d = C.c_int, C.c_void_p, C.c_longlong, C.c_double
    G = C.POINTER(Grid)
    lib.mgrt_last_error.restype = C.c_char_p
    lib.mgrt_last_error.argtypes = []
    sig = {
        "init": [i], "shutdown": [], "device_name": [C.c_char_p, i],
        "alloc": [C.POINTER(p), C.c_size_t], "free": [p],
        "upload": [p, p, C.c_size_t, p], "download": [p, p, C.c_size_t, p],
        "stream_create": [C.POINTER(p)], "stream_destroy": [p], "sync": [p], "zero": [p, C.c_size_t, p],
        "box_cells_host": [i, i, i, i, C.POINTER(i), p],
        "plan_create": [C.POINTER(p), C.POINTER(PlanDesc), p], "plan_destroy": [p],
        "plan_level_dims": [p, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)],
        "plan_set_cells": [p, i, p],
        "plan_vcycle": [p, p, p, d, i, i, i, i, C.POINTER(d)],
        "plan_fmg": [p, p, p, d, i, i, i, i],
        "sumsq_noalloc_f32": [G, p, p, i, i, p],
        "rccl_unique_id": [p],
        "rccl_selftest": [p],
        "slab_plan_create": [C.POINTER(p), C.POINTER(SlabDesc), C.POINTER(Transport), p],
        "slab_plan_destroy": [p],
        "slab_plan_rows": [p, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)],
        "slab_plan_vcycle": [p, p, p, d, i, i, i, i, C.POINTER(d)],
        "slab_plan_fmg": [p, p, p, d, i, i, i, i],
    }
    for name, args in sig.items():
        f = getattr(lib, "mgrt_" + name)
        f.restype = i
        f.argtypes = args
    lib.mgrt_plan_levels.restype = i
    lib.mgrt_plan_levels.argtypes = [p]
    for name in ("slab_plan_distributed_levels", "slab_plan_uses_graph"):
        getattr(lib, "mgrt_" + name).restype = i
        getattr(lib, "mgrt_" + name).argtypes = [p]
    lib.mgrt_sweep_parts.restype = i
    lib.mgrt_sweep_parts.argtypes = [i]
    lib.mgrt_last_sweep_launches.restype = i
    lib.mgrt_last_sweep_launches.argtypes = []
    lib.mgrt_correct_smooth_sweep_supported.restype = i
    lib.mgrt_correct_smooth_sweep_supported.argtypes = [G, i, i]
    lib.mgrt_smooth_sweep_residual_restrict_supported.restype = i
    lib.mgrt_smooth_sweep_residual_restrict_supported.argtypes = [G, i, i]
    for S, r in (("f32", C.c_float), ("f64", C.c_double)):
        ksig = {
            "smooth_pass": [G, p, p, r, i, i, i, i, p],
            "smooth": [G, p, p, r, i, i, p],
            "smooth_sweep": [G, p, p, p, r, i, i, p],
            "smooth_sweep_residual": [G, p, p, p, p, r, i, i, p],
            "correct_smooth_sweep": [G, G, p, p, p, p, r, i, i, p],
            "smooth_sweep_residual_restrict": [G, G, p, p, p, p, p, r, i, i, p],
            "residual": [G, p, p, p, i, p, i, i, p],
            "restrict": [G, G, p, p, i, i, p],
            "correct": [G, G, p, p, p, i, i, p],
            "prolong": [G, G, p, p, i, i, p],
            "zero_outside": [G, p, p],
            "diff": [G, p, p, p, p],
            "scale": [G, p, p, C.c_float, p],
            "sumsq": [G, p, p, i, i, p],
            "maxabs": [G, p, p, i, i, p],
            "l2norm_reftree": [p, ll, C.POINTER(r), p],
            "linfnorm_reftree": [p, ll, C.POINTER(r), p],
            "sum_reftree": [p, ll, C.POINTER(r), p],
        }
        for name, args in ksig.items():
            f = getattr(lib, "mgrt_%s_%s" % (name, S))
            f.restype = i
            f.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise MgrtError(status, load().mgrt_last_error().decode())


_initialised = None


def init(device=0):
    """mgrt_init on the given device (idempotent for the same device)."""
    global _initialised
    check(load().mgrt_init(device))
    _initialised = device
    if os.environ.get("MGRT_ONE_GRID"):  # development: A/B of the band-wise launches of large planes
        load().mgrt_sweep_parts(3 | 4)


def grid(shape, cells=None, faces=ALL_DIRICHLET, y0=0, gny=None, pitch=None, slab=None, neumann=0):
    """Build an mgrt_grid for arrays of numpy/torch shape (nz, ny, nx) or (ny, nx).

    `cells` is a device tensor (uint8, same shape) or None for the analytic box with `faces`.
    Keeps a reference to `cells` on the returned struct so the memory outlives the call.
    """
    g = Grid()
    if len(shape) == 3:
        g.dim, g.nz, g.ny, g.nx = 3, shape[0], shape[1], shape[2]
    else:
        g.dim, g.nz, g.ny, g.nx = 2, 1, shape[0], shape[1]
    g.pitch = g.nx if pitch is None else pitch
    g.slab = g.pitch * g.ny if slab is None else slab
    g.y0 = y0
    g.gny = g.ny if gny is None else gny
    g.cells = None if cells is None else cells.data_ptr()
    g._keep = cells
    g.face = (C.c_int * 6)(*faces)
    g.neumann = int(neumann)
    return g


def _suffix(t):
    import torch
    return {torch.float32: "f32", torch.float64: "f64"}[t.dtype]


def _stream(stream):
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    return C.c_void_p(stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _fn(name, t):
    return getattr(load(), "mgrt_%s_%s" % (name, _suffix(t)))


# ---- kernel wrappers over torch tensors --------------------------------------------------------
def smooth_pass(g, u, f, omega, odd, zcount=None, ylo=0, yhi=None, stream=None):
    check(_fn("smooth_pass", u)(C.byref(g), _p(u), _p(f), omega, odd, g.nz if zcount is None else zcount,
                                ylo, g.ny if yhi is None else yhi, _stream(stream)))


def smooth(g, u, f, omega, iters, zcount=None, stream=None):
    check(_fn("smooth", u)(C.byref(g), _p(u), _p(f), omega, iters, g.nz if zcount is None else zcount,
                           _stream(stream)))


def smooth_sweep(g, out, inp, f, omega, iters, zcount=None, stream=None):
    check(_fn("smooth_sweep", f)(C.byref(g), _p(out), _p(inp), _p(f), omega, iters,
                                   g.nz if zcount is None else zcount, _stream(stream)))


def correct_sweep_supported(fine, dtype, iters):
    """Whether correct_smooth_sweep has a kernel for this fine grid, torch dtype and iteration count."""
    import torch
    return bool(load().mgrt_correct_smooth_sweep_supported(C.byref(fine), 4 if dtype == torch.float32 else 8, iters))


def correct_smooth_sweep(fine, coarse, out, u, e, f, omega, iters, zcount=None, stream=None):
    check(_fn("correct_smooth_sweep", f)(C.byref(fine), C.byref(coarse), _p(out), _p(u), _p(e), _p(f), omega, iters,
                                           fine.nz if zcount is None else zcount, _stream(stream)))


def restrict_sweep_supported(fine, dtype, iters):
    """Whether smooth_sweep_residual_restrict has a kernel for this fine grid, torch dtype and iteration count."""
    import torch
    return bool(load().mgrt_smooth_sweep_residual_restrict_supported(C.byref(fine), 4 if dtype == torch.float32 else 8, iters))


def smooth_sweep_residual_restrict(fine, coarse, out, rc, inp, f, omega, iters, zcount=None, stream=None, r=None):
    """r: 3D only, a work array of the fine grid's size (the fine residual next to tile seams passes through it)."""
    check(_fn("smooth_sweep_residual_restrict", f)(C.byref(fine), C.byref(coarse), _p(out), _p(rc), _p(r), _p(inp), _p(f),
                                                     omega, iters, fine.nz if zcount is None else zcount, _stream(stream)))


def sweep_parts(mask):
    """mgrt_sweep_parts: which launches of the two-launch 3D sweep with restriction are issued (bit 0 the sweep,
    bit 1 the seam restriction; bit 2: large planes as one grid instead of bands); returns the previous mask.  For timing
    a kernel alone."""
    return load().mgrt_sweep_parts(mask)


def last_sweep_launches():
    """mgrt_last_sweep_launches: how many launches (bands of tile rows) the last fused 3D sweep kernel was issued as."""
    return load().mgrt_last_sweep_launches()


def smooth_sweep_residual(g, out, r, inp, f, omega, iters, zcount=None, stream=None):
    check(_fn("smooth_sweep_residual", f)(C.byref(g), _p(out), _p(r), _p(inp), _p(f), omega, iters,
                                            g.nz if zcount is None else zcount, _stream(stream)))


def residual(g, r, u, f, zero_unwritten=False, norm2=None, ylo=0, yhi=None, stream=None):
    check(_fn("residual", u)(C.byref(g), _p(r), _p(u), _p(f), int(zero_unwritten), _p(norm2), ylo,
                             g.ny if yhi is None else yhi, _stream(stream)))


def restrict(gf, gc, rc, rf, cylo=0, cyhi=None, stream=None):
    check(_fn("restrict", rf)(C.byref(gf), C.byref(gc), _p(rc), _p(rf), cylo, gc.ny if cyhi is None else cyhi,
                              _stream(stream)))


def correct(gf, gc, out, u, e, ylo=0, yhi=None, stream=None):
    check(_fn("correct", u)(C.byref(gf), C.byref(gc), _p(out), _p(u), _p(e), ylo, gf.ny if yhi is None else yhi,
                            _stream(stream)))


def prolong(gf, gc, out, c, ylo=0, yhi=None, stream=None):
    check(_fn("prolong", out)(C.byref(gf), C.byref(gc), _p(out), _p(c), ylo, gf.ny if yhi is None else yhi,
                              _stream(stream)))


def zero_outside(g, u, stream=None):
    check(_fn("zero_outside", u)(C.byref(g), _p(u), _stream(stream)))


def diff(g, out, a, b, stream=None):
    check(_fn("diff", a)(C.byref(g), _p(out), _p(a), _p(b), _stream(stream)))


def scale(g, out, x, m, stream=None):
    check(_fn("scale", x)(C.byref(g), _p(out), _p(x), m, _stream(stream)))


def sumsq(g, x, out, ylo=0, yhi=None, stream=None):
    check(_fn("sumsq", x)(C.byref(g), _p(x), _p(out), ylo, g.ny if yhi is None else yhi, _stream(stream)))


def sumsq_noalloc(g, x, out, ylo=0, yhi=None, stream=None):
    """mgrt_sumsq_noalloc_f32: sumsq with its scratch in a fixed array of the library (float only)."""
    check(load().mgrt_sumsq_noalloc_f32(C.byref(g), _p(x), _p(out), ylo, g.ny if yhi is None else yhi, _stream(stream)))


def maxabs(g, x, out, ylo=0, yhi=None, stream=None):
    check(_fn("maxabs", x)(C.byref(g), _p(x), _p(out), ylo, g.ny if yhi is None else yhi, _stream(stream)))


def l2norm_reftree(x, stream=None):
    r = (C.c_float if _suffix(x) == "f32" else C.c_double)()
    check(_fn("l2norm_reftree", x)(_p(x), x.numel(), C.byref(r), _stream(stream)))
    return r.value


def sum_reftree(x, stream=None):
    r = (C.c_float if _suffix(x) == "f32" else C.c_double)()
    check(_fn("sum_reftree", x)(_p(x), x.numel(), C.byref(r), _stream(stream)))
    return r.value


def linfnorm_reftree(x, stream=None):
    r = (C.c_float if _suffix(x) == "f32" else C.c_double)()
    check(_fn("linfnorm_reftree", x)(_p(x), x.numel(), C.byref(r), _stream(stream)))
    return r.value


def box_cells_host(shape, faces=ALL_DIRICHLET):
    import numpy as np
    cells = np.empty(shape, dtype=np.uint8)
    if len(shape) == 3:
        dim, nz, ny, nx = 3, *shape
    else:
        dim, nz, (ny, nx) = 2, 1, shape
    check(load().mgrt_box_cells_host(dim, nx, ny, nz, (C.c_int * 6)(*faces), cells.ctypes.data_as(C.c_void_p)))
    return cells


class Plan:
    """mgrt_plan: pre-allocated hierarchy; iterate/mg/fmg of MultigridSolver{2,3}D on device arrays."""

    def __init__(self, shape, dtype, faces=ALL_DIRICHLET, zfull=False, use_graph=False, no_fuse=False, no_bottom=False, neumann=0,
                 stream=None, separate_restrict=False):
        import torch
        d = PlanDesc()
        if len(shape) == 3:
            d.dim, d.nz, d.ny, d.nx = 3, shape[0], shape[1], shape[2]
        else:
            d.dim, d.nz, d.ny, d.nx = 2, 1, shape[0], shape[1]
        d.elem_size = {torch.float32: 4, torch.float64: 8}[dtype]
        d.face = (C.c_int * 6)(*faces)
        d.zfull = int(zfull)
        d.use_graph = int(use_graph)
        d.no_fuse = int(no_fuse)
        d.no_bottom = int(no_bottom)
        d.neumann = int(neumann)
        d.separate_restrict = int(separate_restrict)
        self.shape, self.dtype = tuple(shape), dtype
        self._h = C.c_void_p()
        check(load().mgrt_plan_create(C.byref(self._h), C.byref(d), _stream(stream)))

    def close(self):
        if self._h:
            check(load().mgrt_plan_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def levels(self):
        return load().mgrt_plan_levels(self._h)

    def level_shape(self, level):
        nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
        check(load().mgrt_plan_level_dims(self._h, level, C.byref(nx), C.byref(ny), C.byref(nz)))
        return (nz.value, ny.value, nx.value) if len(self.shape) == 3 else (ny.value, nx.value)

    def set_cells(self, level, host_cells):
        import numpy as np
        c = np.ascontiguousarray(host_cells, dtype=np.uint8)
        assert c.shape == self.level_shape(level)
        check(load().mgrt_plan_set_cells(self._h, level, c.ctypes.data_as(C.c_void_p)))

    def vcycle(self, u, f, omega=1.0, a1=2, a2=2, v=1, ncycles=1, history=False):
        """ncycles x iterate(u, f, ...).  Returns the residual history (ncycles+1 floats) or None."""
        assert u.is_contiguous() and f.is_contiguous() and u.dtype == self.dtype and f.dtype == self.dtype
        assert tuple(u.shape) == self.shape and tuple(f.shape) == self.shape
        hist = (C.c_double * (ncycles + 1))() if history else None
        check(load().mgrt_plan_vcycle(self._h, _p(u), _p(f), omega, a1, a2, v, ncycles, hist))
        return list(hist) if history else None

    def fmg(self, u_out, f, omega=1.0, a1=3, a2=3, v=2, iters=3):
        assert u_out.is_contiguous() and f.is_contiguous()
        assert tuple(u_out.shape) == self.shape and tuple(f.shape) == self.shape
        check(load().mgrt_plan_fmg(self._h, _p(u_out), _p(f), omega, a1, a2, v, iters))


def rccl_unique_id():
    """128 bytes of a fresh ncclUniqueId (made on one rank, handed to all)."""
    buf = (C.c_ubyte * 128)()
    check(load().mgrt_rccl_unique_id(buf))
    return bytes(buf)


def rccl_version():
    """ncclGetVersion of the RCCL the library bound, as 'major.minor.patch'; None where RCCL is absent."""
    v = C.c_int(0)
    if load().mgrt_rccl_version(C.byref(v)) != 0:
        return None
    return "%d.%d.%d" % (v.value // 10000, v.value // 100 % 100, v.value % 100)


def rccl_selftest(stream):
    """The RCCL operations of a slab plan on a one-rank communicator, direct and from a graph (raises on a finding)."""
    check(load().mgrt_rccl_selftest(C.c_void_p(stream.cuda_stream)))


class GlooTransport:
    """Blocking host callbacks over torch.distributed (any backend that moves CPU tensors): device
    buffers are staged through host memory.  For tests and rehearsals on a box with fewer GPUs than
    ranks; a measurement uses the RCCL transport."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch, self.group = dist, torch, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        lib = load()

        def down(ptr, nbytes):
            t = torch.empty(nbytes, dtype=torch.uint8)
            check(lib.mgrt_download(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), nbytes, None))
            return t

        def up(ptr, t):
            check(lib.mgrt_upload(C.c_void_p(ptr), C.c_void_p(t.data_ptr()), t.numel(), None))

        def sendrecv(ctx, n, peers, send, recv, nbytes):
            try:
                ops, landing = [], []
                for k in range(n):
                    peer = dist.get_global_rank(group, peers[k]) if group is not None else peers[k]
                    out = down(send[k], nbytes[k])
                    inc = torch.empty(nbytes[k], dtype=torch.uint8)
                    ops += [dist.P2POp(dist.isend, out, peer, group), dist.P2POp(dist.irecv, inc, peer, group)]
                    landing.append((recv[k], inc))
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                for ptr, t in landing:
                    up(ptr, t)
                return 0
            except Exception as err:  # an exception must not cross the C boundary
                print("GlooTransport.sendrecv: %r" % (err,), flush=True)
                return 1

        def allgather(ctx, send, recv, nbytes):
            try:
                src = down(send, nbytes)
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, src, group=group)
                up(recv, torch.cat(parts))
                return 0
            except Exception as err:
                print("GlooTransport.allgather: %r" % (err,), flush=True)
                return 1

        def allreduce(ctx, value):
            try:
                t = down(value, 8).view(torch.float64)
                dist.all_reduce(t, group=group)
                up(value, t.view(torch.uint8))
                return 0
            except Exception as err:
                print("GlooTransport.allreduce: %r" % (err,), flush=True)
                return 1

        self._keep = (SENDRECV_FN(sendrecv), ALLGATHER_FN(allgather), ALLREDUCE_FN(allreduce))

    def struct(self):
        t = Transport()
        t.kind, t.rank, t.nranks = TRANSPORT_CALLBACK, self.rank, self.world
        t.sendrecv, t.allgather, t.allreduce_sum = self._keep
        return t


class SlabPlan:
    """mgrt_slab_plan: the V-cycle on this rank's y-slab of a 3D grid (shape = whole grid, (nz, ny, nx)).

    transport: "rccl" (every rank passes the same `rccl_id` from rccl_unique_id()), a GlooTransport, or
    None with (rank, nranks) for the stand-alone share of one rank."""

    def __init__(self, shape, dtype, transport=None, rank=0, nranks=1, rccl_id=None, faces=ALL_DIRICHLET, zfull=True,
                 use_graph=True, crossover=129 * 129 * 257, stream=None):
        import torch
        d = SlabDesc()
        d.nz, d.ny, d.nx = shape
        d.elem_size = {torch.float32: 4, torch.float64: 8}[dtype]
        d.face = (C.c_int * 6)(*faces)
        d.zfull, d.use_graph, d.crossover = int(zfull), int(use_graph), int(crossover)
        if transport == "rccl":
            t = Transport()
            t.kind, t.rank, t.nranks = TRANSPORT_RCCL, rank, nranks
            t.rccl_id = (C.c_ubyte * 128)(*rccl_id)
        elif transport is None:
            t = Transport()
            t.kind, t.rank, t.nranks = TRANSPORT_NONE, rank, nranks
        else:
            t = transport.struct()
        self._transport = transport  # keeps the callbacks alive
        self.shape, self.dtype = tuple(shape), dtype
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self._h = C.c_void_p()
        check(load().mgrt_slab_plan_create(C.byref(self._h), C.byref(d), C.byref(t), C.c_void_p(self.stream.cuda_stream)))

    def close(self):
        if self._h:
            check(load().mgrt_slab_plan_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def distributed_levels(self):
        return load().mgrt_slab_plan_distributed_levels(self._h)

    @property
    def uses_graph(self):
        return bool(load().mgrt_slab_plan_uses_graph(self._h))

    def rows(self):
        """(lo, hi, own_lo, own_hi): global rows of level 0 stored and owned by this rank."""
        v = [C.c_int() for _ in range(4)]
        check(load().mgrt_slab_plan_rows(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def scatter(self, whole):
        lo, hi, _, _ = self.rows()
        return whole[:, lo:hi, :].contiguous()

    def owned(self, local):
        lo, _, a, b = self.rows()
        return local[:, a - lo:b - lo, :]

    def vcycle(self, u, f, omega=1.0, a1=2, a2=2, v=1, ncycles=1, history=False):
        """In place on this rank's rows (halo included); the work is queued on self.stream, which first
        waits for the current stream.  Returns the residual history over the whole grid, or None."""
        import torch
        assert u.is_contiguous() and f.is_contiguous() and u.dtype == self.dtype and f.dtype == self.dtype
        lo, hi, _, _ = self.rows()
        assert tuple(u.shape) == (self.shape[0], hi - lo, self.shape[2]) == tuple(f.shape)
        self.stream.wait_stream(torch.cuda.current_stream())
        hist = (C.c_double * (ncycles + 1))() if history else None
        check(load().mgrt_slab_plan_vcycle(self._h, _p(u), _p(f), omega, a1, a2, v, ncycles, hist))
        torch.cuda.current_stream().wait_stream(self.stream)
        return list(hist) if history else None

    def fmg(self, u_out, f, omega=1.0, a1=2, a2=2, v=1, iters=3):
        """Full multigrid from the right-hand side alone into this rank's rows of u_out (halo included)."""
        import torch
        assert u_out.is_contiguous() and f.is_contiguous() and u_out.dtype == self.dtype and f.dtype == self.dtype
        lo, hi, _, _ = self.rows()
        assert tuple(u_out.shape) == (self.shape[0], hi - lo, self.shape[2]) == tuple(f.shape)
        self.stream.wait_stream(torch.cuda.current_stream())
        check(load().mgrt_slab_plan_fmg(self._h, _p(u_out), _p(f), omega, a1, a2, v, iters))
        torch.cuda.current_stream().wait_stream(self.stream)
