// mgrt_plan.hip -- pre-planned level hierarchy and cycle drivers.
//
// Host-side restatement of MultigridSolver3D::iterate / mg / fmg (multigridsolver3D.cpp:41-109,
// multigridsolver3D.h:49-59; 2D twin multigridsolver2D.cpp:42-124) over the C-ABI kernels, with
// every work array allocated once at plan creation instead of three cl::Buffers per level per call
// (multigridsolver3D.cpp:53-55), and the whole cycle optionally replayed from a hipGraph.
#include "mgrt_internal.h"
#include <vector>

using namespace mgrt;

struct mgrt_plan {
    mgrt_plan_desc desc;
    hipStream_t stream = nullptr;
    struct Level {
        mgrt_grid g;
        void *r = nullptr;       // residual on this level (fine side of the restriction)
        void *work = nullptr;    // ping-pong partner of the unknown for the fused smoother sweeps
        void *rhs = nullptr;     // right-hand side of this level (level 0: the caller's f)
        void *e = nullptr;       // unknown of this level (level 0: the caller's u)
        void *fmg_f = nullptr;   // FMG: restricted right-hand side, allocated on first use
        void *fmg_u = nullptr;   // FMG: initial guess / solution on this level
        unsigned char *cells = nullptr;
        size_t n = 0;
    };
    std::vector<Level> lv;
    double *d_norms = nullptr; // device: residual norms of a run
    int norms_cap = 0;
    // cached graph of one iterate() call
    hipGraphExec_t exec = nullptr;
    int bottom = 1 << 30; // first level of the one-launch bottom, set at the start of every run
    struct Key {
        void *u;
        const void *f;
        double omega;
        int a1, a2, v;
    } key{};
};

namespace {

bool recurses(const mgrt_grid &g)
{
    return g.dim == 3 ? (g.nx > 3 && g.ny > 3 && g.nz > 3) : (g.nx > 3 && g.ny > 3);
}

int zcount_of(const mgrt_plan *p, const mgrt_grid &g)
{
    if (g.dim != 3)
        return 1;
    return p->desc.zfull ? g.nz : g.nz / 2 + g.nz % 2; // multigridsolver3D.cpp:121
}

void drop_graph(mgrt_plan *p)
{
    if (p->exec) {
        (void)hipGraphExecDestroy(p->exec);
        p->exec = nullptr;
    }
}

template <class R> struct K;
template <> struct K<float> {
    static constexpr auto smooth = mgrt_smooth_f32;
    static constexpr auto sweep = mgrt_smooth_sweep_f32;
    static constexpr auto sweep_res = mgrt_smooth_sweep_residual_f32;
    static constexpr auto correct_sweep = mgrt_correct_smooth_sweep_f32;
    static constexpr auto sweep_res_restrict = mgrt_smooth_sweep_residual_restrict_f32;
    static constexpr auto residual = mgrt_residual_f32;
    static constexpr auto restrict_ = mgrt_restrict_f32;
    static constexpr auto correct = mgrt_correct_f32;
    static constexpr auto prolong = mgrt_prolong_f32;
};
template <> struct K<double> {
    static constexpr auto smooth = mgrt_smooth_f64;
    static constexpr auto sweep = mgrt_smooth_sweep_f64;
    static constexpr auto sweep_res = mgrt_smooth_sweep_residual_f64;
    static constexpr auto correct_sweep = mgrt_correct_smooth_sweep_f64;
    static constexpr auto sweep_res_restrict = mgrt_smooth_sweep_residual_restrict_f64;
    static constexpr auto residual = mgrt_residual_f64;
    static constexpr auto restrict_ = mgrt_restrict_f64;
    static constexpr auto correct = mgrt_correct_f64;
    static constexpr auto prolong = mgrt_prolong_f64;
};

// First level from which the rest of the hierarchy runs as one launch, or the level count when
// none does: no byte maps from there down, and all arrays of those levels fit in LDS.
int bottom_from(const mgrt_plan *p)
{
    const int n = (int)p->lv.size();
    if (p->desc.no_bottom || p->desc.no_fuse || neumann_extension(&p->lv[0].g))
        return n;
    int first = n;
    for (int l = n - 1; l >= 0; --l) {
        if (p->lv[l].cells || n - l > bottom_cycle_max_levels())
            break;
        int nx[16], ny[16], nz[16];
        for (int m = l; m < n; ++m)
            nx[m - l] = p->lv[m].g.nx, ny[m - l] = p->lv[m].g.ny, nz[m - l] = p->lv[m].g.nz;
        if (bottom_cycle_lds_bytes(p->desc.elem_size, n - l, nx, ny, nz) > bottom_cycle_lds_budget())
            break;
        first = l;
    }
    return first;
}

template <class R>
mgrt_status run_bottom(mgrt_plan *p, int l, R *u, const R *u_in, const R *f, R omega, int a1, int a2, int v)
{
    const int n = (int)p->lv.size();
    int nx[16], ny[16], nz[16];
    for (int m = l; m < n; ++m)
        nx[m - l] = p->lv[m].g.nx, ny[m - l] = p->lv[m].g.ny, nz[m - l] = p->lv[m].g.nz;
    return bottom_cycle(p->desc.dim, n - l, nx, ny, nz, p->lv[l].g.face, p->desc.zfull, u, u_in, f, omega, a1, a2, v, p->stream);
}

// The fused 3D sweep covers analytic boxes and byte maps alike (the variants that carry the residual or
// the correction along: boxes only); 2D levels with a byte map smooth in place with the single-colour
// kernel.
bool sweepable(const mgrt_plan *p, const mgrt_grid &g)
{
    if (p->desc.no_fuse || neumann_extension(&g))
        return false;
    if (g.dim == 2)
        return g.cells == nullptr; // the 2D tile kernels know the analytic box only
    return plane_offsets_fit(&g, (size_t)p->desc.elem_size);
}

// `a` smoothing iterations on *cur, as `nsweeps` out-of-place fused sweeps that ping-pong between
// *cur and *alt (nsweeps between ceil(a/2) and a), or in place when nsweeps == 0.  With `res`, the
// last sweep also leaves the residual of the result there (restricted to the coarse grid `cg` when
// that is given without `e`); with `e` (on the coarse grid `cg`), the first sweep starts from
// *cur + 4*P(e), the coarse-grid correction (all of these: nsweeps > 0 only).
template <class R>
mgrt_status smooth_steps(mgrt_plan *p, mgrt_plan::Level &L, R *&cur, R *&alt, const R *f, R omega, int a, int nsweeps,
                         R *res = nullptr, bool cur_is_zero = false, const mgrt_grid *cg = nullptr, const R *e = nullptr)
{
    mgrt_stream s = (mgrt_stream)p->stream;
    const int zc = zcount_of(p, L.g);
    if (nsweeps == 0)
        return K<R>::smooth(&L.g, cur, f, omega, a, zc, s);
    for (int i = 0; i < nsweeps; ++i) {
        // spread the iterations: the first (a - nsweeps) sweeps do two, the rest one
        const int n = i < a - nsweeps ? 2 : 1;
        // a field known to be all zeros is not read (and was not cleared): the kernel takes NULL
        const R *in = cur_is_zero && i == 0 ? nullptr : cur;
        if (mgrt_status st = e && i == 0              ? K<R>::correct_sweep(&L.g, cg, alt, cur, e, f, omega, n, zc, s)
                             : res && i + 1 == nsweeps
                                 ? (cg ? K<R>::sweep_res_restrict(&L.g, cg, alt, res, (R *)L.r, in, f, omega, n, zc, s)
                                       : K<R>::sweep_res(&L.g, alt, res, in, f, omega, n, zc, s))
                                                       : K<R>::sweep(&L.g, alt, in, f, omega, n, zc, s))
            return st;
        R *t = cur;
        cur = alt;
        alt = t;
    }
    return MGRT_OK;
}

// iterate(), multigridsolver3D.cpp:41-80, on level l with unknown u and right-hand side f.
// The result is left in u.
// u_is_zero: u stands for the cleared unknown of zero_mem (:59) but has not been cleared; it is
// either never read (first sweep out of place) or cleared here.
template <class R>
mgrt_status iterate(mgrt_plan *p, int l, R *u, const R *f, R omega, int a1, int a2, int v, bool u_is_zero = false)
{
    mgrt_plan::Level &L = p->lv[l];
    mgrt_stream s = (mgrt_stream)p->stream;
    if (l >= p->bottom)
        return run_bottom<R>(p, l, u, u_is_zero ? nullptr : u, f, omega, a1, a2, v);
    R *cur = u, *alt = (R *)L.work;
    // Number of out-of-place sweeps for the pre- and post-smoother: as few as possible, but an
    // even total so that the result lands in u without a copy.
    int n1 = 0, n2 = 0;
    if (sweepable(p, L.g)) {
        n1 = (a1 + 1) / 2;
        n2 = (a2 + 1) / 2;
        if ((n1 + n2) % 2) {
            if (n2 < a2) ++n2;
            else if (n1 < a1) ++n1;
        }
    }
    if (u_is_zero && n1 == 0) {
        if (mgrt_status st = mgrt_zero(u, L.n * sizeof(R), s))
            return st;
        u_is_zero = false;
    }
    // On bandwidth-bound levels the last pre-smoothing sweep carries the residual pass along
    // (4-byte elements: with 8-byte ones the narrower strips make it slower than the two kernels).
    // In 2D every fused level does.  The last correction goes into the first post-smoothing sweep
    // where a kernel for that exists and pays: every 2D level, and in 3D the bandwidth-bound levels when that
    // sweep does two iterations (double precision: only planes of more tiles than the device has CUs).
    const bool res_in_sweep =
        n1 > 0 && v > 0 && recurses(L.g) &&
        (L.g.dim == 2 || (sizeof(R) == 4 && L.g.dim == 3 && !latency_bound_level(L.g) && !L.g.cells));
    const bool correct_in_sweep =
        n2 > 0 && a2 > 0 && v > 0 && recurses(L.g) &&
        (L.g.dim == 2 || (correct_sweep_3d_supported(&L.g, sizeof(R), n2 < a2 ? 2 : 1) && !latency_bound_level(L.g)));
    // (On latency-bound levels the plain sweep takes two planes a time step, which the sweep on the corrected field
    // cannot, for registers: correction + sweep is the faster pair there, 257^3: 26 + 87 against 115 us, 129^3: 7 + 41
    // against 56.)
    // That sweep hands the coarse level its right-hand side directly where a kernel for it exists: every 2D
    // level, and in 3D the bandwidth-bound levels in single precision when it does two iterations (the
    // fine residual is then written next to the seams between rows of tiles only).
    const bool restrict_in_sweep =
        res_in_sweep && (L.g.dim == 2 || (!p->desc.separate_restrict &&
                                          mgrt_smooth_sweep_residual_restrict_supported(&L.g, (int)sizeof(R), n1 < a1 ? 2 : 1)));
    // Where no sweep carries them, residual and restriction run as one launch (3D boxes): a march of its
    // own on the bandwidth-bound levels, tile by tile through LDS on the others.
    const bool march_residual_restrict =
        !p->desc.separate_restrict && !p->desc.no_fuse && recurses(L.g) && L.g.dim == 3 &&
        mgrt_smooth_sweep_residual_restrict_supported(&L.g, (int)sizeof(R), 0);
    if (a1 > 0) {
        R *res = restrict_in_sweep ? (R *)p->lv[l + 1].rhs : res_in_sweep ? (R *)L.r : nullptr;
        if (mgrt_status st = smooth_steps<R>(p, L, cur, alt, f, omega, a1, n1, res, u_is_zero,
                                             restrict_in_sweep ? &p->lv[l + 1].g : nullptr))
            return st;
    }
    if (recurses(L.g)) {
        mgrt_plan::Level &C = p->lv[l + 1];
        for (int k = 0; k < v; ++k) {
            // outer/Neumann cells of the residual are defined as 0 (the reference reads
            // uninitialised memory there)
            const bool have_r = res_in_sweep && k == 0, have_rc = restrict_in_sweep && k == 0;
            if (!have_r && !have_rc && march_residual_restrict) {
                // residual and restriction as one march over u and f: the fine residual stays in registers
                if (mgrt_status st = K<R>::sweep_res_restrict(&L.g, &C.g, nullptr, (R *)C.rhs, (R *)L.r, cur, f, omega, 0,
                                                              zcount_of(p, L.g), s))
                    return st;
            } else {
                if (!have_r)
                    if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, 0, L.g.ny, s))
                        return st;
                if (!have_rc)
                    if (mgrt_status st = K<R>::restrict_(&L.g, &C.g, (R *)C.rhs, (const R *)L.r, 0, C.g.ny, s))
                        return st;
            }
            if (mgrt_status st = iterate<R>(p, l + 1, (R *)C.e, (const R *)C.rhs, omega, a1, a2, v, true)) // zero_mem(i), :59
                return st;
            // in place: same values as the out-of-place kernel followed by the handle swap (:70-72)
            if (!(correct_in_sweep && k + 1 == v))
                if (mgrt_status st = K<R>::correct(&L.g, &C.g, cur, cur, (const R *)C.e, 0, L.g.ny, s))
                    return st;
        }
    }
    if (a2 > 0) {
        const mgrt_grid *cg = correct_in_sweep ? &p->lv[l + 1].g : nullptr;
        if (mgrt_status st = smooth_steps<R>(p, L, cur, alt, f, omega, a2, n2, nullptr, false, cg,
                                             correct_in_sweep ? (const R *)p->lv[l + 1].e : nullptr))
            return st;
    }
    if (cur != u) // odd number of sweeps (only when a1 + a2 == 1): bring the result home
        MGRT_HIP(hipMemcpyAsync(u, cur, L.n * sizeof(R), hipMemcpyDeviceToDevice, p->stream));
    return MGRT_OK;
}

template <class R> mgrt_status run_cycle(mgrt_plan *p, R *u, const R *f, R omega, int a1, int a2, int v)
{
    p->bottom = bottom_from(p);
    if (!p->desc.use_graph)
        return iterate<R>(p, 0, u, f, omega, a1, a2, v);
    const mgrt_plan::Key key{u, f, (double)omega, a1, a2, v};
    const bool same = p->exec && p->key.u == key.u && p->key.f == key.f && p->key.omega == key.omega &&
                      p->key.a1 == key.a1 && p->key.a2 == key.a2 && p->key.v == key.v;
    if (!same) {
        drop_graph(p);
        hipGraph_t graph = nullptr;
        MGRT_HIP(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
        const mgrt_status st = iterate<R>(p, 0, u, f, omega, a1, a2, v);
        const hipError_t e = hipStreamEndCapture(p->stream, &graph);
        if (st != MGRT_OK) {
            if (graph)
                (void)hipGraphDestroy(graph);
            return st;
        }
        MGRT_HIP(e);
        const hipError_t ei = hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        MGRT_HIP(ei);
        p->key = key;
    }
    MGRT_HIP(hipGraphLaunch(p->exec, p->stream));
    return MGRT_OK;
}

template <class R>
mgrt_status vcycle(mgrt_plan *p, R *u, const R *f, R omega, int a1, int a2, int v, int ncycles, double *res_hist)
{
    mgrt_plan::Level &L = p->lv[0];
    mgrt_stream s = (mgrt_stream)p->stream;
    if (res_hist && p->norms_cap < ncycles + 1) {
        if (p->d_norms)
            MGRT_HIP(hipFree(p->d_norms));
        p->d_norms = nullptr;
        MGRT_HIP(hipMalloc((void **)&p->d_norms, sizeof(double) * (size_t)(ncycles + 1)));
        p->norms_cap = ncycles + 1;
    }
    auto record = [&](int i) -> mgrt_status {
        return K<R>::residual(&L.g, (R *)L.r, u, f, 1, p->d_norms + i, 0, L.g.ny, s);
    };
    if (res_hist)
        if (mgrt_status st = record(0))
            return st;
    for (int c = 0; c < ncycles; ++c) {
        if (mgrt_status st = run_cycle<R>(p, u, f, omega, a1, a2, v))
            return st;
        if (res_hist)
            if (mgrt_status st = record(c + 1))
                return st;
    }
    if (res_hist) {
        MGRT_HIP(hipMemcpyAsync(res_hist, p->d_norms, sizeof(double) * (size_t)(ncycles + 1), hipMemcpyDeviceToHost,
                                p->stream));
        MGRT_HIP(hipStreamSynchronize(p->stream));
        for (int i = 0; i <= ncycles; ++i)
            res_hist[i] = sqrt(res_hist[i]);
    }
    return MGRT_OK;
}

// fmg(), multigridsolver3D.cpp:82-109, producing the solution of level l into u_out
template <class R>
mgrt_status fmg(mgrt_plan *p, int l, R *u_out, const R *f, R omega, int a1, int a2, int v, int iters)
{
    mgrt_plan::Level &L = p->lv[l];
    mgrt_stream s = (mgrt_stream)p->stream;
    if (recurses(L.g)) {
        mgrt_plan::Level &C = p->lv[l + 1];
        if (!C.fmg_f) {
            MGRT_HIP(hipMalloc(&C.fmg_f, C.n * sizeof(R)));
            MGRT_HIP(hipMalloc(&C.fmg_u, C.n * sizeof(R)));
        }
        // the right-hand side itself is restricted, unscaled (:98-99)
        if (mgrt_status st = K<R>::restrict_(&L.g, &C.g, (R *)C.fmg_f, f, 0, C.g.ny, s))
            return st;
        // the recursive call omits iters, so coarse levels run the default of 3 (:101)
        if (mgrt_status st = fmg<R>(p, l + 1, (R *)C.fmg_u, (const R *)C.fmg_f, omega, a1, a2, v, 3))
            return st;
        if (mgrt_status st = K<R>::prolong(&L.g, &C.g, u_out, (const R *)C.fmg_u, 0, L.g.ny, s))
            return st;
    } else {
        if (mgrt_status st = mgrt_zero(u_out, L.n * sizeof(R), s)) // :92-93
            return st;
    }
    for (int i = 0; i < iters; ++i) // mg(), multigridsolver3D.h:49-59
        if (mgrt_status st = iterate<R>(p, l, u_out, f, omega, a1, a2, v))
            return st;
    return MGRT_OK;
}

} // namespace

extern "C" {

mgrt_status mgrt_plan_create(mgrt_plan **plan, const mgrt_plan_desc *d, mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(plan && d, "mgrt_plan_create: null argument");
    *plan = nullptr;
    MGRT_REQUIRE(d->dim == 2 || d->dim == 3, "mgrt_plan_create: dim must be 2 or 3");
    MGRT_REQUIRE(d->elem_size == 4 || d->elem_size == 8, "mgrt_plan_create: elem_size must be 4 or 8");
    const int nz0 = d->dim == 3 ? d->nz : 1;
    MGRT_REQUIRE(d->nx > 0 && d->ny > 0 && nz0 > 0, "mgrt_plan_create: extents must be positive");
    MGRT_REQUIRE(d->neumann == 0 || d->neumann == 1, "mgrt_plan_create: neumann must be 0 or 1");
    for (int i = 0; i < 6; ++i)
        MGRT_REQUIRE(d->face[i] >= 1 && d->face[i] <= 3,
                     "mgrt_plan_create: face[%d] is not a class a face can have (outer, Dirichlet, Neumann)", i);

    mgrt_plan *p = new mgrt_plan;
    p->desc = *d;
    p->stream = (hipStream_t)s;
    if (!p->stream)
        p->desc.use_graph = 0; // the default stream cannot be captured: launch directly
    int nx = d->nx, ny = d->ny, nz = nz0;
    for (;;) {
        mgrt_plan::Level L;
        L.g.dim = d->dim;
        L.g.nx = nx;
        L.g.ny = ny;
        L.g.nz = nz;
        L.g.pitch = nx;
        L.g.slab = (long long)nx * ny;
        L.g.y0 = 0;
        L.g.gny = ny;
        L.g.cells = nullptr;
        L.g.neumann = d->neumann;
        for (int i = 0; i < 6; ++i)
            L.g.face[i] = d->face[i];
        L.n = (size_t)nx * ny * nz;
        p->lv.push_back(L);
        if (!recurses(L.g))
            break;
        // Coarsening is only meaningful for odd extents: the reference's interpolation kernels
        // assume coarse = n/2+1 (mg_1.cl:207), which equals (n+1)/2 only then.
        if (nx % 2 == 0 || ny % 2 == 0 || (d->dim == 3 && nz % 2 == 0)) {
            delete p;
            return fail(MGRT_E_UNSUPPORTED,
                        "mgrt_plan_create: level extents (%d,%d,%d) are not all odd; the reference's "
                        "interpolation is undefined there",
                        nx, ny, nz);
        }
        nx = (nx + 1) / 2;
        ny = (ny + 1) / 2;
        if (d->dim == 3)
            nz = (nz + 1) / 2;
    }
    // the norm scratch of this plan's stream exists before any cycle is captured into a graph
    double *parts = nullptr;
    if (mgrt_status st = partials_for(p->stream, &parts)) {
        delete p;
        return st;
    }
    const size_t w = (size_t)d->elem_size;
    for (size_t l = 0; l < p->lv.size(); ++l) {
        mgrt_plan::Level &L = p->lv[l];
        hipError_t e = hipSuccess;
        if (l + 1 < p->lv.size() || l == 0) // level 0: also the target of the residual history
            e = hipMalloc(&L.r, L.n * w);
        if (e == hipSuccess && !d->no_fuse)
            e = hipMalloc(&L.work, L.n * w);
        if (e == hipSuccess && l > 0)
            e = hipMalloc(&L.rhs, L.n * w);
        if (e == hipSuccess && l > 0)
            e = hipMalloc(&L.e, L.n * w);
        if (e != hipSuccess) {
            mgrt_plan_destroy(p);
            return fail((mgrt_status)e, "mgrt_plan_create: %s", hipGetErrorString(e));
        }
    }
    *plan = p;
    return MGRT_OK;
}

mgrt_status mgrt_plan_destroy(mgrt_plan *p)
{
    if (!p)
        return MGRT_OK;
    (void)hipStreamSynchronize(p->stream);
    drop_graph(p);
    for (auto &L : p->lv) {
        (void)hipFree(L.r);
        (void)hipFree(L.work);
        (void)hipFree(L.rhs);
        (void)hipFree(L.e);
        (void)hipFree(L.fmg_f);
        (void)hipFree(L.fmg_u);
        (void)hipFree(L.cells);
    }
    (void)hipFree(p->d_norms);
    delete p;
    return MGRT_OK;
}

int mgrt_plan_levels(const mgrt_plan *p) { return p ? (int)p->lv.size() : 0; }

mgrt_status mgrt_plan_level_dims(const mgrt_plan *p, int level, int *nx, int *ny, int *nz)
{
    MGRT_REQUIRE(p && level >= 0 && level < (int)p->lv.size(), "mgrt_plan_level_dims: bad level");
    if (nx) *nx = p->lv[level].g.nx;
    if (ny) *ny = p->lv[level].g.ny;
    if (nz) *nz = p->lv[level].g.nz;
    return MGRT_OK;
}

mgrt_status mgrt_plan_set_cells(mgrt_plan *p, int level, const unsigned char *host_cells)
{
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(p && host_cells && level >= 0 && level < (int)p->lv.size(), "mgrt_plan_set_cells: bad argument");
    mgrt_plan::Level &L = p->lv[level];
    // inner cells on a face of the grid would make the stencils read outside the arrays
    const mgrt_grid &g = L.g;
    for (int z = 0; z < g.nz; ++z)
        for (int y = 0; y < g.ny; ++y)
            for (int x = 0; x < g.nx; ++x) {
                const unsigned char t = host_cells[x + (size_t)g.nx * y + (size_t)g.nx * g.ny * z];
                MGRT_REQUIRE(t <= 3, "mgrt_plan_set_cells: value %d at (%d,%d,%d) is not a cell class", t, x, y, z);
                const bool on_face = x == 0 || y == 0 || x == g.nx - 1 || y == g.ny - 1 ||
                                     (g.dim == 3 && (z == 0 || z == g.nz - 1));
                MGRT_REQUIRE(!(on_face && t == MGRT_CELL_INNER),
                             "mgrt_plan_set_cells: inner cell on a face of the grid at (%d,%d,%d)", x, y, z);
            }
    if (!L.cells)
        MGRT_HIP(hipMalloc((void **)&L.cells, L.n));
    if (p->desc.neumann)
        return fail(MGRT_E_UNSUPPORTED, "mgrt_plan_set_cells: the first-order Neumann condition needs the analytic box");
    MGRT_HIP(hipMemcpyAsync(L.cells, host_cells, L.n, hipMemcpyHostToDevice, p->stream));
    MGRT_HIP(hipStreamSynchronize(p->stream));
    L.g.cells = L.cells;
    drop_graph(p);
    return MGRT_OK;
}

mgrt_status mgrt_plan_vcycle(mgrt_plan *p, void *u, const void *f, double omega, int a1, int a2, int v, int ncycles,
                             double *res_hist)
{
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(p && u && f, "mgrt_plan_vcycle: null argument");
    MGRT_REQUIRE(u != f, "mgrt_plan_vcycle: u and f are the same array");
    MGRT_REQUIRE(a1 >= 0 && a2 >= 0 && v >= 0 && ncycles >= 0, "mgrt_plan_vcycle: negative count");
    if (p->desc.elem_size == 4)
        return vcycle<float>(p, (float *)u, (const float *)f, (float)omega, a1, a2, v, ncycles, res_hist);
    return vcycle<double>(p, (double *)u, (const double *)f, omega, a1, a2, v, ncycles, res_hist);
}

mgrt_status mgrt_plan_fmg(mgrt_plan *p, void *u_out, const void *f, double omega, int a1, int a2, int v, int iters)
{
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(p && u_out && f, "mgrt_plan_fmg: null argument");
    MGRT_REQUIRE(u_out != f, "mgrt_plan_fmg: u_out and f are the same array");
    MGRT_REQUIRE(a1 >= 0 && a2 >= 0 && v >= 0 && iters >= 0, "mgrt_plan_fmg: negative count");
    p->bottom = bottom_from(p);
    if (p->desc.elem_size == 4)
        return fmg<float>(p, 0, (float *)u_out, (const float *)f, (float)omega, a1, a2, v, iters);
    return fmg<double>(p, 0, (double *)u_out, (const double *)f, omega, a1, a2, v, iters);
}

} // extern "C"
