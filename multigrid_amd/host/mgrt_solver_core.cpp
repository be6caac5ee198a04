// mgrt_solver_core.cpp -- step kernels and plan cache shared by the 2D and 3D solver classes.
#include "mgrt_solver_core.h"

namespace mgrt { namespace detail {

SolverCore::SolverCore(int dim, const Binder &bind)
    : fullZRange(false), useGraph(true), m_dim(dim), m_bind(bind), m_distributed(false), m_transport(), m_globalHeight(0),
      m_crossover(0)
{
}

SolverCore::~SolverCore()
{
    for (std::map<Dims, PlanEntry>::iterator it = m_plans.begin(); it != m_plans.end(); ++it)
        (void)mgrt_plan_destroy(it->second.plan);
    for (std::map<Dims, SlabEntry>::iterator it = m_slabs.begin(); it != m_slabs.end(); ++it)
        (void)mgrt_slab_plan_destroy(it->second.plan);
}

void SolverCore::distribute(const mgrt_transport &t, int globalHeight, long long crossover)
{
    if (m_dim != 3)
        throw Error("distribute: 3D grids only", MGRT_E_UNSUPPORTED);
    for (std::map<Dims, SlabEntry>::iterator it = m_slabs.begin(); it != m_slabs.end(); ++it)
        (void)mgrt_slab_plan_destroy(it->second.plan);
    m_slabs.clear();
    m_distributed = true;
    m_transport = t;
    m_globalHeight = globalHeight;
    m_crossover = crossover;
}

mgrt_slab_plan *SolverCore::slabFor(const Dims &whole)
{
    std::map<Dims, SlabEntry>::iterator it = m_slabs.find(whole);
    if (it != m_slabs.end()) {
        if (it->second.zfull == fullZRange && it->second.graph == useGraph)
            return it->second.plan;
        (void)mgrt_slab_plan_destroy(it->second.plan);
        m_slabs.erase(it);
    }
    const mgrt_grid top = describe(whole);
    if (top.cells)
        throw Error("distribute: box domains only (the border handler describes this grid by a cell map)", MGRT_E_UNSUPPORTED);
    mgrt_slab_desc d;
    d.nx = whole.w;
    d.ny = whole.h;
    d.nz = whole.d;
    d.elem_size = (int)sizeof(real);
    for (int i = 0; i < 6; ++i)
        d.face[i] = top.face[i];
    d.zfull = fullZRange ? 1 : 0;
    d.use_graph = useGraph ? 1 : 0;
    d.crossover = m_crossover;
    SlabEntry e = {0, fullZRange, useGraph};
    MGRT_CHECK(mgrt_slab_plan_create(&e.plan, &d, &m_transport, m_queue.get()));
    m_slabs[whole] = e;
    return e.plan;
}

void SolverCore::slabRows(int width, int depth, int *lo, int *hi, int *ownLo, int *ownHi)
{
    if (!m_distributed)
        throw Error("slabRows: call distribute first", MGRT_E_INVALID);
    const Dims whole = {width, m_globalHeight, depth};
    MGRT_CHECK(mgrt_slab_plan_rows(slabFor(whole), lo, hi, ownLo, ownHi));
}

mgrt_slab_plan *SolverCore::slabForRows(const Dims &local)
{
    const Dims whole = {local.w, m_globalHeight, local.d};
    mgrt_slab_plan *p = slabFor(whole);
    int lo = 0, hi = 0;
    MGRT_CHECK(mgrt_slab_plan_rows(p, &lo, &hi, 0, 0));
    if (hi - lo != local.h)
        throw Error("a distributed solver takes the rows slabRows names: this array has a different height", MGRT_E_INVALID);
    return p;
}

mgrt_grid SolverCore::describe(const Dims &n)
{
    KernelArgs a;
    a.host_cells = 0;
    mgrt_grid &g = a.grid;
    g.dim = m_dim;
    g.nx = n.w;
    g.ny = n.h;
    g.nz = n.d;
    g.pitch = n.w;
    g.slab = (long long)n.w * n.h;
    g.y0 = 0;
    g.gny = n.h;
    g.cells = 0;
    g.neumann = 0; // Neumann cells as the reference treats them
    for (int i = 0; i < 6; ++i)
        g.face[i] = MGRT_CELL_DIRICHLET;
    m_bind(a, n);
    return a.grid;
}

int SolverCore::zcount(const Dims &n) const
{
    if (m_dim != 3)
        return 1;
    return fullZRange ? n.d : n.d / 2 + n.d % 2; // multigridsolver3D.cpp:121
}

void SolverCore::smooth(real *u, const real *f, const Dims &n, real omega, int iters)
{
    const mgrt_grid g = describe(n);
    MGRT_CHECK(MGRT_K(smooth)(&g, u, f, omega, iters, zcount(n), m_queue.get()));
}

void SolverCore::residual(real *r, const real *u, const real *f, const Dims &n)
{
    const mgrt_grid g = describe(n);
    MGRT_CHECK(MGRT_K(residual)(&g, r, u, f, 0, 0, 0, g.ny, m_queue.get()));
}

void SolverCore::restrictTo(real *coarse, const Dims &nc, const real *fine, const Dims &nf)
{
    const mgrt_grid gf = describe(nf);
    mgrt_grid gc = describe(nc);
    MGRT_CHECK(MGRT_K(restrict)(&gf, &gc, coarse, fine, 0, gc.ny, m_queue.get()));
}

void SolverCore::correct(real *out, const real *u, const real *e, const Dims &nf, const Dims &nc)
{
    const mgrt_grid gf = describe(nf), gc = describe(nc);
    MGRT_CHECK(MGRT_K(correct)(&gf, &gc, out, u, e, 0, gf.ny, m_queue.get()));
}

void SolverCore::prolong(real *out, const real *coarse, const Dims &nf, const Dims &nc)
{
    const mgrt_grid gf = describe(nf), gc = describe(nc);
    MGRT_CHECK(MGRT_K(prolong)(&gf, &gc, out, coarse, 0, gf.ny, m_queue.get()));
}

void SolverCore::zeroMem(real *p, const Dims &n)
{
    MGRT_CHECK(mgrt_zero(p, sizeof(real) * (size_t)n.w * n.h * n.d, m_queue.get()));
}

void SolverCore::zeroOut(real *p, const Dims &n)
{
    const mgrt_grid g = describe(n);
    MGRT_CHECK(MGRT_K(zero_outside)(&g, p, m_queue.get()));
}

mgrt_plan *SolverCore::planFor(const Dims &n)
{
    std::map<Dims, PlanEntry>::iterator it = m_plans.find(n);
    if (it != m_plans.end()) {
        if (it->second.zfull == fullZRange && it->second.graph == useGraph)
            return it->second.plan;
        if (it->second.plan)
            (void)mgrt_plan_destroy(it->second.plan);
        m_plans.erase(it);
    }
    PlanEntry e = {0, fullZRange, useGraph};
    mgrt_plan_desc d;
    d.dim = m_dim;
    d.nx = n.w;
    d.ny = n.h;
    d.nz = n.d;
    d.elem_size = (int)sizeof(real);
    d.zfull = fullZRange ? 1 : 0;
    d.use_graph = useGraph ? 1 : 0;
    d.no_fuse = 0;
    d.no_bottom = 0;
    d.neumann = 0;
    d.separate_restrict = 0;
    const mgrt_grid top = describe(n);
    for (int i = 0; i < 6; ++i)
        d.face[i] = top.face[i];
    const mgrt_status st = mgrt_plan_create(&e.plan, &d, m_queue.get());
    if (st == MGRT_E_UNSUPPORTED) {
        e.plan = 0; // remembered: these extents recurse step by step
    } else {
        check(st, "mgrt_plan_create");
        // levels whose handler supplies a byte map get it installed
        const int levels = mgrt_plan_levels(e.plan);
        for (int l = 0; l < levels; ++l) {
            Dims m;
            MGRT_CHECK(mgrt_plan_level_dims(e.plan, l, &m.w, &m.h, &m.d));
            KernelArgs a;
            a.host_cells = 0;
            a.grid = top;
            a.grid.cells = 0;
            m_bind(a, m);
            if (a.host_cells)
                MGRT_CHECK(mgrt_plan_set_cells(e.plan, l, a.host_cells));
        }
    }
    m_plans.insert(std::make_pair(n, e));
    return e.plan;
}

bool SolverCore::cycle(real *u, const real *f, const Dims &n, real omega, int a1, int a2, int v, int cycles)
{
    if (m_distributed) {
        MGRT_CHECK(mgrt_slab_plan_vcycle(slabForRows(n), u, f, (double)omega, a1, a2, v, cycles, 0));
        return true;
    }
    mgrt_plan *p = planFor(n);
    if (!p)
        return false;
    MGRT_CHECK(mgrt_plan_vcycle(p, u, f, (double)omega, a1, a2, v, cycles, 0));
    return true;
}

bool SolverCore::fmg(real *out, const real *f, const Dims &n, real omega, int a1, int a2, int v, int iters)
{
    if (m_distributed) {
        MGRT_CHECK(mgrt_slab_plan_fmg(slabForRows(n), out, f, (double)omega, a1, a2, v, iters));
        return true;
    }
    mgrt_plan *p = planFor(n);
    if (!p)
        return false;
    MGRT_CHECK(mgrt_plan_fmg(p, out, f, (double)omega, a1, a2, v, iters));
    return true;
}

}} // namespace mgrt::detail
