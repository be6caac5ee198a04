// multigridsolver3D.h -- BorderHandler3D and MultigridSolver3D with the public interface of
// multigridsolver3D.h:25-94.  The three OpenCL types of that interface are replaced in place:
// cl::Kernel& -> mgrt::KernelArgs&, cl::CommandQueue& -> mgrt::Stream&, cl_mem -> real*.
#ifndef MULTIGRIDSOLVER3D_H
#define MULTIGRIDSOLVER3D_H

#include "mgrt_solver_core.h"

class BorderHandler3D
{
public:
    enum CellType { CellInner, CellOuter, CellDirichlet, CellNeumann };
    virtual ~BorderHandler3D() {}

    // Describe the cells of a dimx*dimy*dimz grid in ker.grid: either the six face classes of a box
    // (ker.grid.face, ker.grid.cells left null) or a device byte map (ker.grid.cells) together with
    // its host copy (ker.host_cells).  `arg` is the kernel argument slot of the reference and is
    // not used.
    virtual void setarg(int arg, mgrt::KernelArgs &ker, int dimx, int dimy, int dimz) = 0;
    virtual CellType cellType(int x, int y, int z, int dimx, int dimy, int dimz) const = 0;
};

class MultigridSolver3D
{
public:
    // `filename` named the OpenCL source in the reference ("mg_1.cl", programstate.cpp:168); it is
    // accepted and ignored.
    MultigridSolver3D(const char *filename, BorderHandler3D &handl);

    // One cycle: a1 smoothing iterations, v coarse solves per level, a2 smoothing iterations
    // (multigridsolver3D.cpp:41-80).  The result is left in `in`, which is also returned.
    Buffer3D iterate(Buffer3D &in, const Buffer3D &func, real omega = 2.0 / 3.0, int a1 = 4, int a2 = 4, int v = 1);
    // `iters` cycles (multigridsolver3D.h:49-59)
    Buffer3D mg(Buffer3D &in, const Buffer3D &func, real omega = 2.0 / 3.0, int a1 = 4, int a2 = 4, int v = 1,
                int iters = 3);
    // full multigrid (multigridsolver3D.cpp:82-109)
    Buffer3D fmg(const Buffer3D &func, real omega = 2.0 / 3.0, int a1 = 4, int a2 = 4, int v = 1, int iters = 3);

    void wait() { m_core.wait(); }
    mgrt::Stream &queue() { return m_core.queue(); }

    void smoother_iterate(Buffer3D &res, const Buffer3D &func, real omega, int a1);
    void compute_residuals(Buffer3D &res, const Buffer3D &input, const Buffer3D &func);
    void restrict(Buffer3D &res, const Buffer3D &input);
    void correct_residual(Buffer3D &res, const Buffer3D &input, Buffer3D &residual);
    void prolongate(Buffer3D &res, const Buffer3D &input);
    void zero_mem(Buffer3D &res);
    void zero_out(Buffer3D &res);

    // Not in the reference: relax all z planes (see SolverCore::fullZRange) / graph replay.
    void setFullZRange(bool on) { m_core.fullZRange = on; }
    void setUseGraph(bool on) { m_core.useGraph = on; }
    // Not in the reference: make this solver one rank of several, one process per GPU (solver threads of one process all
    // run on the device mgrt_init selected: good for the callback transport on one GPU, not for RCCL).  iterate,
    // mg and fmg then take this rank's rows, slabRows(...) of them, of a box `globalHeight` rows high that is
    // cut along y; every rank calls them together.  Results are bitwise those of one GPU.
    void distribute(const mgrt_transport &t, int globalHeight, long long crossover = 129LL * 129 * 257)
    {
        m_core.distribute(t, globalHeight, crossover);
    }
    void slabRows(int width, int depth, int *lo, int *hi, int *ownLo, int *ownHi)
    {
        m_core.slabRows(width, depth, lo, hi, ownLo, ownHi);
    }

private:
    Buffer3D iterateSteps(Buffer3D &in, const Buffer3D &func, real omega, int a1, int a2, int v);
    BorderHandler3D &m_Handl;
    mgrt::detail::SolverCore m_core;
};

#endif // MULTIGRIDSOLVER3D_H
