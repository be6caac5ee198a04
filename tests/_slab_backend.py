"""A stand-in for multigrid_amd._lib built on the oracle, so that the slab driver
(multigrid_amd/dist.py) can run on CPU tensors in the multi-process tests.

Every operation embeds the rank's rows into a zero-filled array of the WHOLE grid, runs the oracle's
whole-grid function and copies the requested rows back.  Rows a rank does not hold therefore read
as zeros: values computed from them are wrong in exactly the places where the real kernels' are
(the outermost halo rows), which is what the driver's exchange schedule has to cope with."""
import numpy as np
import torch

import _oracle


class _Grid:
    def __init__(self, shape, y0, gny):
        self.shape, self.y0, self.gny = tuple(shape), y0, gny
        self.whole = (shape[0], gny, shape[2])


class OracleKernels:
    def __init__(self):
        self.o = _oracle.load("mgo")
        assert self.o is not None, "oracle/libmg_oracle.so is not built"
        self.o.set_num_threads(1)
        self._cells = {}

    def grid(self, shape, y0=0, gny=None):
        return _Grid(shape, y0, shape[1] if gny is None else gny)

    def _cell(self, whole):
        if whole not in self._cells:
            self._cells[whole] = self.o.box_cells(whole)
        return self._cells[whole]

    @staticmethod
    def _embed(g, t):
        w = np.zeros(g.whole, dtype=t.numpy().dtype)
        w[:, g.y0:g.y0 + g.shape[1], :] = t.numpy()
        return w

    def smooth_sweep(self, g, out, inp, f, omega, iters, zcount=None):
        u, fw = self._embed(g, inp), self._embed(g, f)
        om = u.dtype.type(omega)
        for _ in range(iters):
            for odd in (0, 1):
                self.o.smooth_pass(self._cell(g.whole), u, fw, om, odd, zcount)
        out.numpy()[...] = u[:, g.y0:g.y0 + g.shape[1], :]

    def residual(self, g, r, u, f, zero_unwritten=False, norm2=None, ylo=0, yhi=None):
        uw, fw = self._embed(g, u), self._embed(g, f)
        rw = np.zeros_like(uw)
        self.o.residual(self._cell(g.whole), rw, uw, fw)
        yhi = g.shape[1] if yhi is None else yhi
        r.numpy()[:, ylo:yhi, :] = rw[:, g.y0 + ylo:g.y0 + yhi, :]

    def restrict(self, gf, gc, rc, rf, cylo=0, cyhi=None):
        fw = self._embed(gf, rf)
        cw = np.zeros(gc.whole, dtype=fw.dtype)
        self.o.restrict(self._cell(gf.whole), cw, fw)
        cyhi = gc.shape[1] if cyhi is None else cyhi
        rc.numpy()[:, cylo:cyhi, :] = cw[:, gc.y0 + cylo:gc.y0 + cyhi, :]

    def correct(self, gf, gc, out, u, e, ylo=0, yhi=None):
        uw, ew = self._embed(gf, u), self._embed(gc, e)
        ow = np.zeros_like(uw)
        self.o.correct(self._cell(gf.whole), ow, uw, ew)
        yhi = gf.shape[1] if yhi is None else yhi
        out.numpy()[:, ylo:yhi, :] = ow[:, gf.y0 + ylo:gf.y0 + yhi, :]

    def correct_sweep_supported(self, g, dtype, iters):
        return iters == 2  # so that the CPU tests walk the driver's fused branch as well

    def correct_smooth_sweep(self, gf, gc, out, u, e, f, omega, iters, zcount=None):
        tmp = torch.empty_like(u)
        self.correct(gf, gc, tmp, u, e)
        self.smooth_sweep(gf, out, tmp, f, omega, iters, zcount)

    def sumsq(self, g, x, out, ylo=0, yhi=None):
        yhi = g.shape[1] if yhi is None else yhi
        out[0] = float((x.numpy()[:, ylo:yhi, :].astype(np.float64) ** 2).sum())

    def Plan(self, shape, dtype, zfull=False, use_graph=False, faces=None):
        return _Plan(self.o, zfull)


class _Plan:
    def __init__(self, o, zfull):
        self.o, self.zfull = o, zfull

    def vcycle(self, u, f, omega=1.0, a1=2, a2=2, v=1, ncycles=1, history=False):
        un, fn = u.numpy(), f.numpy()
        hist = []

        def norm():
            r = np.zeros_like(un)
            self.o.residual(self.o.box_cells(un.shape), r, un, fn)
            return self.o.l2norm_exact(r)
        if history:
            hist.append(norm())
        for _ in range(ncycles):
            self.o.iterate(un, fn, un.dtype.type(omega), a1, a2, v, self.zfull)
            if history:
                hist.append(norm())
        return hist if history else None

    def close(self):
        pass
