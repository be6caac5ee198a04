// multigridsolver2D.h -- BorderHandler2D and MultigridSolver2D with the public interface of
// multigridsolver2D.h:25-96 (OpenCL types replaced as in multigridsolver3D.h).
#ifndef MULTIGRIDSOLVER2D_H
#define MULTIGRIDSOLVER2D_H

#include "mgrt_solver_core.h"

class BorderHandler2D
{
public:
    enum CellType { CellInner, CellOuter, CellDirichlet, CellNeumann };
    virtual ~BorderHandler2D() {}
    virtual void setarg(int arg, mgrt::KernelArgs &ker, int dimx, int dimy) = 0;
    virtual CellType cellType(int x, int y, int dimx, int dimy) const = 0;
};

class MultigridSolver2D
{
public:
    MultigridSolver2D(const char *filename, BorderHandler2D &handl);

    Buffer2D iterate(Buffer2D &in, const Buffer2D &func, real omega = 2.0 / 3.0, int a1 = 4, int a2 = 4, int v = 1);
    Buffer2D mg(Buffer2D &in, const Buffer2D &func, real omega = 2.0 / 3.0, int a1 = 4, int a2 = 4, int v = 1,
                int iters = 3);
    Buffer2D fmg(const Buffer2D &func, real omega = 2.0 / 3.0, int a1 = 4, int a2 = 4, int v = 1, int iters = 3);

    void wait() { m_core.wait(); }
    mgrt::Stream &queue() { return m_core.queue(); }

    void smoother_iterate(Buffer2D &res, const Buffer2D &func, real omega, int a1);
    void compute_residuals(Buffer2D &res, const Buffer2D &input, const Buffer2D &func);
    void restrict(Buffer2D &res, const Buffer2D &input);
    void correct_residual(Buffer2D &res, const Buffer2D &input, Buffer2D &residual);
    void prolongate(Buffer2D &res, const Buffer2D &input);
    void zero_mem(Buffer2D &res);
    void zero_out(Buffer2D &res);

    // Print the residual norm on every level before and after each coarse-grid correction
    // (multigridsolver2D.cpp:64-68,79-85).  Cycles then run step by step instead of from a plan.
    bool m_debugPrintResiduals;

    void setUseGraph(bool on) { m_core.useGraph = on; }

private:
    Buffer2D iterateSteps(Buffer2D &in, const Buffer2D &func, real omega, int a1, int a2, int v);
    BorderHandler2D &m_Handl;
    mgrt::detail::SolverCore m_core;
};

#endif // MULTIGRIDSOLVER2D_H
