// mgrt_solver_core.h -- the machinery under MultigridSolver2D and MultigridSolver3D: one stream,
// grid descriptors filled in by the border handler, the step kernels, and a cache of pre-planned
// level hierarchies for the cycle drivers.
#ifndef MGRT_SOLVER_CORE_H
#define MGRT_SOLVER_CORE_H

#include "buffer.h"
#include <functional>
#include <map>
#include <vector>

namespace mgrt { namespace detail {

struct Dims {
    int w, h, d; // d == 1 in 2D
    bool operator<(const Dims &o) const
    {
        if (w != o.w) return w < o.w;
        if (h != o.h) return h < o.h;
        return d < o.d;
    }
};

class SolverCore
{
public:
    // `bind` asks the border handler to describe a grid of the given extents.
    typedef std::function<void(KernelArgs &, const Dims &)> Binder;

    SolverCore(int dim, const Binder &bind);
    ~SolverCore();

    Stream &queue() { return m_queue; }
    void wait() { m_queue.finish(); }

    // step kernels; every argument is a dense device array of the stated extents
    void smooth(real *u, const real *f, const Dims &n, real omega, int iters);
    void residual(real *r, const real *u, const real *f, const Dims &n);
    void restrictTo(real *coarse, const Dims &nc, const real *fine, const Dims &nf);
    void correct(real *out, const real *u, const real *e, const Dims &nf, const Dims &nc);
    void prolong(real *out, const real *coarse, const Dims &nf, const Dims &nc);
    void zeroMem(real *p, const Dims &n);
    void zeroOut(real *p, const Dims &n);

    // cycle drivers.  `cycles` V/W cycles in place on u; false when no plan exists for these
    // extents (some level has an even extent) and the caller has to recurse step by step.
    bool cycle(real *u, const real *f, const Dims &n, real omega, int a1, int a2, int v, int cycles);
    bool fmg(real *out, const real *f, const Dims &n, real omega, int a1, int a2, int v, int iters);

    // Not in the reference (one device, one queue): from here on this solver is one rank of `t`: one process
    // per GPU (the runtime under it is a process-wide singleton bound to one device by mgrt_init; several
    // solver threads of one process share that device and are for the callback transport only) and cycle() / fmg() take this rank's rows of a 3D box `globalHeight` rows high, cut
    // along y (mgrt_slab_plan_* of libmgrt.so).  Levels of at most `crossover` points are solved whole on
    // every rank.  Box domains only.  The step kernels above keep working on whole arrays.
    void distribute(const mgrt_transport &t, int globalHeight, long long crossover);
    // the rows [lo, hi) this rank stores of a width x globalHeight x depth grid and the rows it owns
    void slabRows(int width, int depth, int *lo, int *hi, int *ownLo, int *ownHi);

    // 3D only: relax every z plane instead of the reference's first ceil(nz/2)
    // (multigridsolver3D.cpp:121).  Off by default: results then match the reference's.
    bool fullZRange;
    // replay cycles from a captured graph
    bool useGraph;

    static Dims coarser(const Dims &n, int dim)
    {
        Dims c = {(n.w + 1) / 2, (n.h + 1) / 2, dim == 3 ? (n.d + 1) / 2 : 1};
        return c;
    }

private:
    SolverCore(const SolverCore &);
    mgrt_grid describe(const Dims &n);
    int zcount(const Dims &n) const;
    struct PlanEntry {
        mgrt_plan *plan;
        bool zfull, graph;
    };
    mgrt_plan *planFor(const Dims &n);
    struct SlabEntry {
        mgrt_slab_plan *plan;
        bool zfull, graph;
    };
    mgrt_slab_plan *slabFor(const Dims &whole);
    mgrt_slab_plan *slabForRows(const Dims &local); // checks that `local` is this rank's share

    int m_dim;
    Binder m_bind;
    Stream m_queue;
    std::map<Dims, PlanEntry> m_plans;
    bool m_distributed;
    mgrt_transport m_transport;
    int m_globalHeight;
    long long m_crossover;
    std::map<Dims, SlabEntry> m_slabs; // keyed by the extents of the whole grid
};

}} // namespace mgrt::detail

#endif // MGRT_SOLVER_CORE_H
