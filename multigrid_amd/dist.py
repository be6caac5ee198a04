"""V-cycles on a 3D grid split into y-slabs, one slab per process (one process per GPU).

This is the Python statement of the schedule that mgrt_slab_plan (libmgrt.so, csrc/mgrt_slab.hip) runs
natively with RCCL and one graph per rank; bench.py uses that one.  This module stays as the form
that runs on CPU tensors with a replaceable kernel set, which is how the multi-process CPU tests check
the partitioning and the exchange schedule without a GPU.

The fine levels of the hierarchy of MultigridSolver3D::iterate (multigridsolver3D.cpp:41-80) are
cut along y: rank r owns the rows [B_r, B_r+1) of level 0 and the rows [B_r/2^l, B_r+1/2^l) of level
l, so the owner of coarse row j is the owner of fine row 2j and restriction and interpolation need
nothing from further away than the halo.  Every z column stays on one rank, so the z recurrence
inside a colour pass of the reference's 3D smoother (mg_1.cl:56-104) is untouched and every cell
goes through the same arithmetic as on one GPU: the results are bitwise those of the one-GPU plan.

Each rank stores its rows plus HALO rows of its neighbours on either side.  A fused smoother sweep
of two iterations (four colour passes) recomputes the halo rows it needs instead of exchanging
after every pass, so one exchange of HALO = 4 rows precedes each sweep.  Per level and V(2,2) cycle
that makes three exchanges (smoothed unknown, coarse right-hand side, coarse correction) plus one of
the unknown at the start of a cycle.  Below `crossover` points the restricted residual is
all-gathered and every rank runs the remaining levels on the whole coarse grid with the one-GPU
plan (the same work agglomeration onto one GPU would do, without the scatter back).

The data path is the C ABI of libmgrt.so (multigrid_amd._lib); torch.distributed moves the halos
(NCCL = RCCL over xGMI between GPUs).  `kernels` can be replaced by an object with the same
functions, which is how the CPU tests run this driver without a GPU.
"""
import torch
import torch.distributed as dist

HALO = 4  # rows: 2 per smoothing iteration fused into one sweep


def coarse_extent(n):
    return (n + 1) // 2


def plan_levels(shape, world, crossover, min_rows=2 * HALO):
    """How many levels are distributed, and the alignment of the slab boundaries.

    shape = (nz, ny, nx) of the finest grid.  A level is distributed while it recurses (every
    extent > 3), has more than `crossover` points and leaves every rank at least `min_rows` rows.
    Returns (ndist, align): boundaries are multiples of align = 2^ndist rows of level 0, so that
    they stay whole numbers down to the first level that is not distributed."""
    nz, ny, nx = shape
    ny0 = ny
    ndist = 0
    while world > 1 and nx > 3 and ny > 3 and nz > 3 and nx * ny * nz > crossover:
        q = (ny0 - 1) // (world * 2 ** (ndist + 1))
        if 2 * q < min_rows:  # rows per rank on the coarsest distributed level
            break
        ndist += 1
        nx, ny, nz = coarse_extent(nx), coarse_extent(ny), coarse_extent(nz)
    return ndist, 2 ** ndist


def boundaries(ny, world, align):
    """B_0 .. B_world: first rows of the slabs on level 0 (multiples of `align`), B_world = ny."""
    q = (ny - 1) // (world * align)
    return [r * q * align for r in range(world)] + [ny]


class _Level:
    """One distributed level on this rank."""

    def __init__(self, kernels, shape, y0, y1, dtype, device, with_unknown):
        nz, gny, nx = shape
        self.shape = shape
        self.y0, self.y1 = y0, y1                       # owned global rows
        self.lo, self.hi = max(0, y0 - HALO), min(gny, y1 + HALO)
        self.nyl = self.hi - self.lo
        self.own = (y0 - self.lo, y1 - self.lo)         # owned local rows
        self.local_shape = (nz, self.nyl, nx)
        self.grid = kernels.grid(self.local_shape, y0=self.lo, gny=gny)
        new = lambda: torch.zeros(self.local_shape, dtype=dtype, device=device)
        self.work, self.r = new(), new()
        self.rhs = new() if with_unknown else None
        self.e = new() if with_unknown else None
        self.sendbuf = {}


class SlabCycle:
    """Pre-planned distributed V/W-cycle for one 3D box problem (all faces Dirichlet by default).

    shape: (nz, ny, nx) of the whole finest grid.  Every rank of `group` constructs it with the same
    arguments.  Fields passed to vcycle() are this rank's rows including halo, see local_rows()."""

    def __init__(self, shape, dtype, group=None, kernels=None, device=None, zfull=True, crossover=129 ** 3,
                 faces=None, use_graph=True):
        if kernels is None:
            from . import _lib as kernels
        if len(shape) != 3:
            raise ValueError("SlabCycle handles 3D grids")
        self.k = kernels
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dtype, self.device, self.zfull = dtype, device, zfull
        self.shape = tuple(shape)
        self.faces = faces
        self.ndist, align = plan_levels(self.shape, self.world, crossover)
        self.bounds = boundaries(shape[1], self.world, align)
        self.levels = []
        s = self.shape
        for l in range(self.ndist):
            y0 = self.bounds[self.rank] >> l
            y1 = self.bounds[self.rank + 1] >> l if self.rank + 1 < self.world else s[1]
            self.levels.append(_Level(kernels, s, y0, y1, dtype, device, with_unknown=l > 0))
            s = tuple(coarse_extent(n) for n in s)
        # the first level every rank holds whole
        self.coarse_shape = s
        self.coarse_rows = [(self.bounds[r] >> self.ndist,
                             self.bounds[r + 1] >> self.ndist if r + 1 < self.world else s[1])
                            for r in range(self.world)]
        # all GPU work of this object runs on one side stream (graph capture needs a real stream)
        self.stream = torch.cuda.Stream(device) if device is not None and torch.device(device).type == "cuda" else None
        with self._on_stream():
            self.coarse_plan = kernels.Plan(s, dtype, zfull=zfull, use_graph=use_graph,
                                            **({"faces": faces} if faces is not None else {}))
        if self.ndist:
            self.c_rhs = torch.zeros(s, dtype=dtype, device=device)
            self.c_e = torch.zeros(s, dtype=dtype, device=device)
            self.c_grid = kernels.grid(s)
            maxrows = max(b - a for a, b in self.coarse_rows)
            self.gather_in = torch.zeros((s[0], maxrows, s[2]), dtype=dtype, device=device)
            self.gather_out = torch.zeros((self.world,) + tuple(self.gather_in.shape), dtype=dtype, device=device)
        self.norm = torch.zeros(1, dtype=torch.float64, device=device)
        # halo traffic goes through host memory when the backend cannot move device tensors
        self._stage = dist.is_initialized() and dist.get_backend(group) == "gloo" and device is not None \
            and torch.device(device).type == "cuda"

    def _on_stream(self):
        """Context: make the side stream current, ordered after and before the caller's stream."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self.stream is None:
                yield
                return
            outer = torch.cuda.current_stream()
            self.stream.wait_stream(outer)
            with torch.cuda.stream(self.stream):
                yield
            outer.wait_stream(self.stream)
        return ctx()

    # ---- layout ----------------------------------------------------------------------------------
    def local_rows(self):
        """(lo, hi, own_lo, own_hi): this rank stores global rows [lo, hi) of level 0 and owns
        [own_lo, own_hi).  With one rank, or a grid too small to split, that is the whole grid."""
        if not self.ndist:
            return 0, self.shape[1], 0, self.shape[1]
        L = self.levels[0]
        return L.lo, L.hi, L.y0, L.y1

    def scatter(self, whole):
        """This rank's rows (with halo) of a whole-grid array."""
        lo, hi, _, _ = self.local_rows()
        return whole[:, lo:hi, :].contiguous()

    def owned(self, local):
        lo, _, a, b = self.local_rows()
        return local[:, a - lo:b - lo, :]

    # ---- communication -----------------------------------------------------------------------------
    def _exchange(self, L, field):
        """Fill the halo rows of `field` on level L from the neighbours' owned rows."""
        if self.world == 1:
            return
        a, b = L.own
        ops, landing = [], []
        for nb, send_rows, recv_rows in ((self.rank + 1, (b - HALO, b), (b, b + HALO)),
                                         (self.rank - 1, (a, a + HALO), (a - HALO, a))):
            if nb < 0 or nb >= self.world:
                continue
            # one staging pair per level and neighbour: every field of a level has the same halo shape,
            # and an exchange has completed (wait below) before the next one reuses the pair
            if nb not in L.sendbuf:
                mk = lambda: torch.empty((L.local_shape[0], HALO, L.local_shape[2]), dtype=self.dtype,
                                         device="cpu" if self._stage else self.device)
                L.sendbuf[nb] = (mk(), mk())
            sb, rb = L.sendbuf[nb]
            sb.copy_(field[:, send_rows[0]:send_rows[1], :])
            peer = dist.get_global_rank(self.group, nb) if self.group is not None else nb
            ops.append(dist.P2POp(dist.isend, sb, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, rb, peer, self.group))
            landing.append((rb, recv_rows))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for rb, rows in landing:
            field[:, rows[0]:rows[1], :].copy_(rb)

    def _gather_coarse(self, cy0, cy1):
        """Every rank has written its own rows [cy0, cy1) of the first undistributed level's
        right-hand side into c_rhs; afterwards every rank holds all rows."""
        self.gather_in[:, :cy1 - cy0, :].copy_(self.c_rhs[:, cy0:cy1, :])
        if self._stage:
            src = self.gather_in.cpu()
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self.group)
        else:
            parts = list(self.gather_out.unbind(0))
            dist.all_gather(parts, self.gather_in, group=self.group)
        for r, (a, b) in enumerate(self.coarse_rows):
            if r != self.rank:
                self.c_rhs[:, a:b, :].copy_(parts[r][:, :b - a, :])

    # ---- the cycle ---------------------------------------------------------------------------------
    def _zcount(self, L):
        nz = L.shape[0]
        return nz if self.zfull else nz // 2 + nz % 2

    def _smooth(self, L, cur, alt, f, omega, a, halo_valid):
        """`a` iterations as sweeps of two (and one odd one); returns (cur, alt, halo_valid)."""
        left = a
        while left > 0:
            n = 2 if left >= 2 else 1
            if not halo_valid:
                self._exchange(L, cur)
            self.k.smooth_sweep(L.grid, alt, cur, f, omega, n, self._zcount(L))
            cur, alt = alt, cur
            halo_valid = False
            left -= n
        return cur, alt, halo_valid

    def _iterate(self, l, u, f, omega, a1, a2, v, halo_valid):
        """iterate() on distributed level l; u and f are local arrays.  Returns the array that holds
        the result (u or the level's work array) and whether its halo rows are valid."""
        k, L = self.k, self.levels[l]
        cur, alt = u, L.work
        cur, alt, halo_valid = self._smooth(L, cur, alt, f, omega, a1, halo_valid)
        last = l + 1 == self.ndist
        for it in range(v):
            if not halo_valid:
                self._exchange(L, cur)
                halo_valid = True
            a, b = L.own
            # residual on the owned rows and one more on each side: what the restriction reads
            rlo = a - 1 if L.y0 > 0 else a
            rhi = b + 1 if L.y1 < L.shape[1] else b
            k.residual(L.grid, L.r, cur, f, True, None, rlo, rhi)
            if last:
                cy0, cy1 = self.coarse_rows[self.rank]
                k.restrict(L.grid, self.c_grid, self.c_rhs, L.r, cy0, cy1)
                self._gather_coarse(cy0, cy1)
                self.c_e.zero_()
                self.coarse_plan.vcycle(self.c_e, self.c_rhs, omega, a1, a2, v, 1)
                e_local, egrid = self.c_e, self.c_grid
            else:
                C = self.levels[l + 1]
                k.restrict(L.grid, C.grid, C.rhs, L.r, C.own[0], C.own[1])
                self._exchange(C, C.rhs)
                C.e.zero_()
                res, ok = self._iterate(l + 1, C.e, C.rhs, omega, a1, a2, v, True)
                if not ok:
                    self._exchange(C, res)
                e_local, egrid = res, C.grid
            if it + 1 == v and a2 >= 2 and self._correct_in_sweep(L):
                # the last correction rides in the first post-smoothing sweep: same values, and the
                # corrected field is never written (halo rows of cur and e_local are valid here)
                k.correct_smooth_sweep(L.grid, egrid, alt, cur, e_local, f, omega, 2, self._zcount(L))
                cur, alt = alt, cur
                halo_valid = False
                a2 -= 2
            else:
                # every local row is corrected, halo rows included: they stay valid without an exchange
                k.correct(L.grid, egrid, cur, cur, e_local, 0, L.nyl)
        cur, alt, halo_valid = self._smooth(L, cur, alt, f, omega, a2, halo_valid)
        return cur, halo_valid

    def _correct_in_sweep(self, L):
        """Whether the kernel set has a sweep that applies the correction on this level's slab."""
        ask = getattr(self.k, "correct_sweep_supported", None)
        return bool(ask and ask(L.grid, self.dtype, 2))

    def vcycle(self, u, f, omega=1.0, a1=2, a2=2, v=1, ncycles=1, history=False):
        """ncycles x iterate(u, f, omega, a1, a2, v) in place on this rank's rows.  The halo rows
        of f must be filled (scatter() does); those of u are exchanged here.  With history=True
        returns ||f - A u||_2 over the whole grid before the first cycle and after each one."""
        with self._on_stream():
            if not self.ndist:
                return self.coarse_plan.vcycle(u, f, omega, a1, a2, v, ncycles, history)
            hist = []
            halo_valid = False
            if history:
                hist.append(self._residual_norm(u, f))
                halo_valid = True
            for _ in range(ncycles):
                res, halo_valid = self._iterate(0, u, f, omega, a1, a2, v, halo_valid)
                if res is not u:
                    u.copy_(res)
                if history:
                    hist.append(self._residual_norm(u, f))
                    halo_valid = True
            return hist if history else None

    def residual_norm(self, u, f):
        """||f - A u||_2 over the whole grid (exchanges the halo of u; all-reduce of one double)."""
        with self._on_stream():
            return self._residual_norm(u, f)

    def _residual_norm(self, u, f):
        if not self.ndist:
            return self.coarse_plan.vcycle(u, f, ncycles=0, history=True)[0]
        L = self.levels[0]
        self._exchange(L, u)
        a, b = L.own
        self.k.residual(L.grid, L.r, u, f, True, None, a, b)
        self.k.sumsq(L.grid, L.r, self.norm, a, b)
        if self.world > 1:
            if self._stage:
                t = self.norm.cpu()
                dist.all_reduce(t, group=self.group)
                return float(t.item()) ** 0.5
            dist.all_reduce(self.norm, group=self.group)
        return float(self.norm.item()) ** 0.5

    def describe(self):
        """The decomposition, for reports."""
        return {"slab_rows": [b - a for a, b in zip(self.bounds, self.bounds[1:])], "halo_rows": HALO,
                "distributed_levels": self.ndist, "replicated_from": "x".join(map(str, self.coarse_shape))}

    def close(self):
        self.coarse_plan.close()
