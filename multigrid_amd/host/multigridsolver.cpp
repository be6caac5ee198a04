// multigridsolver.cpp -- MultigridSolver2D and MultigridSolver3D over the shared core.
//
// iterate / mg / fmg run from a pre-planned hierarchy (all work arrays allocated once, cycle
// replayed from a graph) whenever every level has odd extents; otherwise, and when the 2D solver
// prints its debug residuals, they recurse through the step methods exactly as
// multigridsolver3D.cpp:41-109 and multigridsolver2D.cpp:42-124 do.
#include "auxiliary.h"
#include "multigridsolver2D.h"
#include "multigridsolver3D.h"
#include <cassert>
#include <iostream>
#include <utility>

using mgrt::detail::Dims;
using mgrt::detail::SolverCore;

namespace {

Dims dimsOf(const Buffer2D &b)
{
    Dims d = {b.width(), b.height(), 1};
    return d;
}
Dims dimsOf(const Buffer3D &b)
{
    Dims d = {b.width(), b.height(), b.depth()};
    return d;
}
Buffer2D likeDims(const Buffer2D *, const Dims &n) { return Buffer2D(n.w, n.h); }
Buffer3D likeDims(const Buffer3D *, const Dims &n) { return Buffer3D(n.w, n.h, n.d); }
bool recurses(const Dims &n, int dim) { return n.w > 3 && n.h > 3 && (dim == 2 || n.d > 3); }

// The reference's recursion, one launch per step (multigridsolver3D.cpp:41-80).  `debug` prints
// the 2D solver's residual lines.
template <class Buf> void stepCycle(SolverCore &core, int dim, Buf &in, const Buf &func, real omega, int a1, int a2, int v,
                                    bool debug)
{
    const Dims n = dimsOf(in);
    core.smooth(in(), func(), n, omega, a1);
    if (recurses(n, dim)) {
        const Dims nc = SolverCore::coarser(n, dim);
        Buf tmp = likeDims((Buf *)0, n), residuals = likeDims((Buf *)0, nc), e = likeDims((Buf *)0, nc);
        // cells the residual kernel never writes (outer, Neumann) hold 0 instead of the
        // uninitialised memory of multigridsolver3D.cpp:53
        core.zeroMem(tmp(), n);
        for (int k = 0; k < v; ++k) {
            core.zeroMem(e(), nc);
            core.residual(tmp(), in(), func(), n);
            if (debug)
                std::cout << "DEBUG, printing residuals on grid: " << n.w << "x" << n.h << " k= " << k
                          << " BEFORE correction res= " << L2Norm(tmp, core.queue()) << std::endl;
            core.restrictTo(residuals(), nc, tmp(), n);
            stepCycle(core, dim, e, residuals, omega, a1, a2, v, debug);
            core.correct(tmp(), in(), e(), n, nc);
            if (debug) {
                Buf deb = likeDims((Buf *)0, n);
                core.zeroMem(deb(), n);
                core.residual(deb(), tmp(), func(), n);
                std::cout << "DEBUG, printing residuals on grid: " << n.w << "x" << n.h << " k= " << k
                          << " AFTER correction res= " << L2Norm(deb, core.queue()) << std::endl;
            }
            std::swap(in, tmp);
        }
    }
    core.smooth(in(), func(), n, omega, a2);
}

template <class Buf> Buf stepFmg(SolverCore &core, int dim, const Buf &func, real omega, int a1, int a2, int v, int iters,
                                 bool debug)
{
    const Dims n = dimsOf(func);
    Buf x0 = likeDims((Buf *)0, n);
    if (!recurses(n, dim)) {
        core.zeroMem(x0(), n);
    } else {
        const Dims nc = SolverCore::coarser(n, dim);
        Buf redFunc = likeDims((Buf *)0, nc);
        core.restrictTo(redFunc(), nc, func(), n);
        // the coarse problems take the default cycle count (multigridsolver3D.cpp:101)
        Buf guess = stepFmg(core, dim, redFunc, omega, a1, a2, v, 3, debug);
        core.prolong(x0(), guess(), n, nc);
    }
    for (int i = 0; i < iters; ++i)
        stepCycle(core, dim, x0, func, omega, a1, a2, v, debug);
    return x0;
}

template <class Buf> void sameDims(const Buf &a, const Buf &b)
{
    const Dims x = dimsOf(a), y = dimsOf(b);
    assert(x.w == y.w && x.h == y.h && x.d == y.d);
    (void)x;
    (void)y;
}

} // namespace

// ---- 3D -----------------------------------------------------------------------------------------

MultigridSolver3D::MultigridSolver3D(const char *, BorderHandler3D &handl)
    : m_Handl(handl), m_core(3, [&handl](mgrt::KernelArgs &a, const Dims &n) { handl.setarg(0, a, n.w, n.h, n.d); })
{
}

Buffer3D MultigridSolver3D::iterate(Buffer3D &in, const Buffer3D &func, real omega, int a1, int a2, int v)
{
    sameDims(in, func);
    if (!m_core.cycle(in(), func(), dimsOf(in), omega, a1, a2, v, 1))
        stepCycle(m_core, 3, in, func, omega, a1, a2, v, false);
    return in;
}

Buffer3D MultigridSolver3D::mg(Buffer3D &in, const Buffer3D &func, real omega, int a1, int a2, int v, int iters)
{
    sameDims(in, func);
    if (!m_core.cycle(in(), func(), dimsOf(in), omega, a1, a2, v, iters))
        for (int i = 0; i < iters; ++i)
            stepCycle(m_core, 3, in, func, omega, a1, a2, v, false);
    return in;
}

Buffer3D MultigridSolver3D::fmg(const Buffer3D &func, real omega, int a1, int a2, int v, int iters)
{
    Buffer3D out(func.width(), func.height(), func.depth());
    if (m_core.fmg(out(), func(), dimsOf(func), omega, a1, a2, v, iters))
        return out;
    return stepFmg(m_core, 3, func, omega, a1, a2, v, iters, false);
}

void MultigridSolver3D::smoother_iterate(Buffer3D &res, const Buffer3D &func, real omega, int a1)
{
    m_core.smooth(res(), func(), dimsOf(res), omega, a1);
}
void MultigridSolver3D::compute_residuals(Buffer3D &res, const Buffer3D &input, const Buffer3D &func)
{
    m_core.residual(res(), input(), func(), dimsOf(res));
}
void MultigridSolver3D::restrict(Buffer3D &res, const Buffer3D &input)
{
    m_core.restrictTo(res(), dimsOf(res), input(), dimsOf(input));
}
void MultigridSolver3D::correct_residual(Buffer3D &res, const Buffer3D &input, Buffer3D &residual)
{
    m_core.correct(res(), input(), residual(), dimsOf(res), dimsOf(residual));
}
void MultigridSolver3D::prolongate(Buffer3D &res, const Buffer3D &input)
{
    m_core.prolong(res(), input(), dimsOf(res), dimsOf(input));
}
void MultigridSolver3D::zero_mem(Buffer3D &res) { m_core.zeroMem(res(), dimsOf(res)); }
void MultigridSolver3D::zero_out(Buffer3D &res) { m_core.zeroOut(res(), dimsOf(res)); }

// ---- 2D -----------------------------------------------------------------------------------------

MultigridSolver2D::MultigridSolver2D(const char *, BorderHandler2D &handl)
    : m_debugPrintResiduals(false), m_Handl(handl),
      m_core(2, [&handl](mgrt::KernelArgs &a, const Dims &n) { handl.setarg(0, a, n.w, n.h); })
{
}

Buffer2D MultigridSolver2D::iterate(Buffer2D &in, const Buffer2D &func, real omega, int a1, int a2, int v)
{
    sameDims(in, func);
    if (m_debugPrintResiduals || !m_core.cycle(in(), func(), dimsOf(in), omega, a1, a2, v, 1))
        stepCycle(m_core, 2, in, func, omega, a1, a2, v, m_debugPrintResiduals);
    return in;
}

Buffer2D MultigridSolver2D::mg(Buffer2D &in, const Buffer2D &func, real omega, int a1, int a2, int v, int iters)
{
    sameDims(in, func);
    if (m_debugPrintResiduals || !m_core.cycle(in(), func(), dimsOf(in), omega, a1, a2, v, iters))
        for (int i = 0; i < iters; ++i)
            stepCycle(m_core, 2, in, func, omega, a1, a2, v, m_debugPrintResiduals);
    return in;
}

Buffer2D MultigridSolver2D::fmg(const Buffer2D &func, real omega, int a1, int a2, int v, int iters)
{
    if (!m_debugPrintResiduals) {
        Buffer2D out(func.width(), func.height());
        if (m_core.fmg(out(), func(), dimsOf(func), omega, a1, a2, v, iters))
            return out;
    }
    return stepFmg(m_core, 2, func, omega, a1, a2, v, iters, m_debugPrintResiduals);
}

void MultigridSolver2D::smoother_iterate(Buffer2D &res, const Buffer2D &func, real omega, int a1)
{
    m_core.smooth(res(), func(), dimsOf(res), omega, a1);
}
void MultigridSolver2D::compute_residuals(Buffer2D &res, const Buffer2D &input, const Buffer2D &func)
{
    m_core.residual(res(), input(), func(), dimsOf(res));
}
void MultigridSolver2D::restrict(Buffer2D &res, const Buffer2D &input)
{
    m_core.restrictTo(res(), dimsOf(res), input(), dimsOf(input));
}
void MultigridSolver2D::correct_residual(Buffer2D &res, const Buffer2D &input, Buffer2D &residual)
{
    m_core.correct(res(), input(), residual(), dimsOf(res), dimsOf(residual));
}
void MultigridSolver2D::prolongate(Buffer2D &res, const Buffer2D &input)
{
    m_core.prolong(res(), input(), dimsOf(res), dimsOf(input));
}
void MultigridSolver2D::zero_mem(Buffer2D &res) { m_core.zeroMem(res(), dimsOf(res)); }
void MultigridSolver2D::zero_out(Buffer2D &res) { m_core.zeroOut(res(), dimsOf(res)); }
