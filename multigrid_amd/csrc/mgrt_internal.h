// mgrt_internal.h -- shared declarations of the libmgrt.so translation units (not installed).
#ifndef MGRT_INTERNAL_H
#define MGRT_INTERNAL_H

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "mgrt.h"

namespace mgrt {

// Record a failure message for mgrt_last_error() and return `code`.
mgrt_status fail(mgrt_status code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define MGRT_HIP(expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::mgrt::fail((mgrt_status)e_, "%s: %s", #expr, hipGetErrorString(e_));          \
    } while (0)

#define MGRT_REQUIRE(cond, ...)                                                                    \
    do {                                                                                           \
        if (!(cond))                                                                               \
            return ::mgrt::fail(MGRT_E_INVALID, __VA_ARGS__);                                      \
    } while (0)

// Per-device state created by mgrt_init.
struct Runtime {
    bool ready = false;
    int device = -1;
    int num_cu = 0;
    double *static_partials = nullptr; // device address of the fixed scratch array of mgrt_sumsq_noalloc_f32
    int sweep_parts = 3;               // mgrt_sweep_parts: which launches of a two-launch sweep operation are issued
    bool sweep_bands = true;           // mgrt_sweep_parts bit 2 clears it: a large plane in one launch instead of bands
    int last_sweep_launches = 0;       // launches the last fused 3D sweep kernel call was split into
};
Runtime &runtime();
mgrt_status require_init();

// Reduction scratch (one double per block, kMaxPartials entries) of the norm kernels.  Every stream
// has its own array, allocated when the stream first asks, so that norms running on different
// streams never share partial sums; work on one stream is ordered and shares one array.
mgrt_status partials_for(hipStream_t s, double **partials);
void release_partials(hipStream_t s); // the stream is being destroyed
void release_all_partials();

// Argument validation shared by the kernel launchers.
mgrt_status check_grid(const mgrt_grid *g, const char *who);

constexpr int kMaxPartials = 1 << 20;

// Levels of at most about 300^2 columns are bound by the latency of a z step of the fused smoother,
// not by bandwidth: they get narrower strips, and carrying the residual pass along does not pay.
inline bool latency_bound_level(const mgrt_grid &g) { return (long long)g.nx * g.ny <= 300LL * 300LL; }

// Blocks are dealt to the 8 XCDs round-robin (blocks b and b+8 share an L2).  This bijective map
// gives the blocks of one XCD consecutive work-item ids, so that neighbouring tiles, which touch
// the same halo cells and cache lines, meet in one L2.  A speed matter only.
__device__ __forceinline__ int xcd_contiguous(int b, int nb)
{
    const int q = nb / 8, r = nb % 8, xcd = b % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
}

// The coarsest levels of a 2D or 3D box hierarchy as one launch with every level in LDS
// (mgrt_bottom.hip): iterate(u, f, omega, a1, a2, v) on dense arrays of nx[0]*ny[0]*nz[0] elements,
// level l having extents nx[l], ny[l], nz[l].  u_in == nullptr: start from zeros.  u may alias u_in.
size_t bottom_cycle_lds_bytes(int elem_size, int nlev, const int *nx, const int *ny, const int *nz);
size_t bottom_cycle_lds_budget();
int bottom_cycle_max_levels();
mgrt_status bottom_cycle(int dim, int nlev, const int *nx, const int *ny, const int *nz, const int face[6], int zfull,
                         float *u, const float *u_in, const float *f, float omega, int a1, int a2, int v, hipStream_t hs);
mgrt_status bottom_cycle(int dim, int nlev, const int *nx, const int *ny, const int *nz, const int face[6], int zfull,
                         double *u, const double *u_in, const double *f, double omega, int a1, int a2, int v,
                         hipStream_t hs);

// Entries of the fixed device array behind mgrt_sumsq_noalloc_f32 (512 KiB per GPU, held until the
// process ends), and the set-up step that looks up its device address (mgrt_kernels.hip; called by
// mgrt_init with the device current).
constexpr int kStaticPartials = 1 << 16;
mgrt_status static_partials_setup(double **device_address);

// Whether the grid asks for the first-order Neumann condition (an extension the fused kernels do not
// carry) on a face that is Neumann.
inline bool neumann_extension(const mgrt_grid *g)
{
    if (!g->neumann || g->cells)
        return false;
    for (int i = 0; i < (g->dim == 3 ? 6 : 4); ++i)
        if (g->face[i] == MGRT_CELL_NEUMANN)
            return true;
    return false;
}

// The fused 3D sweeps address a z plane with 32-bit byte offsets and use offset 2^31 as the one the
// hardware range check always drops, so a plane (rows x pitch) must stay below 2 GiB.
inline bool plane_offsets_fit(const mgrt_grid *g, size_t elem)
{
    return (long long)g->pitch * g->ny * (long long)elem < (1LL << 31);
}

// Whether mgrt_correct_smooth_sweep has a 3D kernel for this level: the plan asks before it drops the
// separate correction.
inline bool correct_sweep_3d_supported(const mgrt_grid *g, size_t elem, int iters)
{
    // from 100^2 columns up: below that the coarse loads lengthen the march by more than the correction's own
    // launch costs (65^3: 42 us against 35 + 6)
    // 8-byte elements: tiles of twelve two-row strips, and only where the plane has more four-row tiles (56 x 24 points
    // of their own) than the device has CUs.  There the sweeps run at the rate of memory and the correction they spare
    // is a whole read and write of the field (1025^3: 7.0 ms against 3.9 + 6.3); on smaller planes the sweep is bound
    // by what a SIMD issues, and the narrow tiles cost more than the correction (513^3: 1.31 against 0.46 + 0.75 ms).
    const bool wide8 = elem == 8 && (long long)((g->nx + 55) / 56) * ((g->ny + 23) / 24) > runtime().num_cu;
    return g->dim == 3 && (elem == 4 || wide8) && iters == 2 && !g->cells && g->y0 % 2 == 0 &&
           g->nx % 2 == 1 && g->gny % 2 == 1 && g->nz % 2 == 1 && g->nz >= 5 && (long long)g->nx * g->ny >= 100LL * 100LL &&
           plane_offsets_fit(g, elem) && !neumann_extension(g);
}

// Whether mgrt_smooth_sweep_residual_restrict has a 3D kernel for this level (the restriction carried in
// the residual-carrying sweep): whole analytic boxes with odd extents; two iterations on bandwidth-bound
// levels with 4-byte elements, or no iteration at all (residual and restriction alone) with either
// element size on any level.
inline bool restrict_sweep_3d_supported(const mgrt_grid *g, size_t elem, int iters)
{
    if (!(g->dim == 3 && !g->cells && g->y0 == 0 && g->ny == g->gny && g->nx % 2 == 1 && g->gny % 2 == 1 && g->nz % 2 == 1 &&
          g->nz >= 5 && plane_offsets_fit(g, elem) && !neumann_extension(g)))
        return false;
    return iters == 0 || (iters == 2 && elem == 4 && !latency_bound_level(*g));
}

// Full-weighting restriction (mgrt_restrict) of `nrows` coarse rows j0, j0 + jstep, ... only.
mgrt_status restrict_rows(const mgrt_grid *fine, const mgrt_grid *coarse, float *rc, const float *rf, int j0, int jstep,
                          int nrows, hipStream_t hs);
mgrt_status restrict_rows(const mgrt_grid *fine, const mgrt_grid *coarse, double *rc, const double *rf, int j0, int jstep,
                          int nrows, hipStream_t hs);

// Rows [cylo, cyhi) of the coarse array rc = Restrict(f - A in) as one march over in and f (mgrt_smooth_fused.hip),
// on whole grids or y-slabs at even offsets; the fine residual rows around tile seams pass through r, the rest of
// r is not written.  The fine rows those coarse rows read, and their neighbours, must be held and valid.
mgrt_status residual_restrict_march(const mgrt_grid *fine, const mgrt_grid *coarse, float *rc, float *r, const float *in,
                                    const float *f, int cylo, int cyhi, hipStream_t hs);
mgrt_status residual_restrict_march(const mgrt_grid *fine, const mgrt_grid *coarse, double *rc, double *r, const double *in,
                                    const double *f, int cylo, int cyhi, hipStream_t hs);

// out = S^2(in) and rows [cylo, cyhi) of the coarse array rc = Restrict(f - A out): the restriction-carrying sweep
// on a whole grid or a y-slab at an even offset (4-byte elements).  cylo == cyhi: the sweep alone.
mgrt_status sweep_residual_restrict_rows(const mgrt_grid *fine, const mgrt_grid *coarse, float *out, float *rc, float *r,
                                         const float *in, const float *f, float omega, int zcount, int cylo, int cyhi,
                                         hipStream_t hs);

// rc = Restrict(f - A u) on a whole 3D analytic box, tile by tile with the fine residual in LDS only.
mgrt_status residual_restrict_tiles(const mgrt_grid *fine, const mgrt_grid *coarse, float *rc, const float *u, const float *f,
                                    hipStream_t hs);
mgrt_status residual_restrict_tiles(const mgrt_grid *fine, const mgrt_grid *coarse, double *rc, const double *u, const double *f,
                                    hipStream_t hs);

// out = S^2(in) on a latency-bound 3D analytic box (whole or y-slab) as two stages of one iteration each, two
// planes a time step (mgrt_smooth_pair.hip; 4-byte elements); six_strips: tiles of 64 x 12 instead of 64 x 16.  in
// may be null (zeros).
mgrt_status smooth_sweep_pair(const mgrt_grid *g, float *out, const float *in, const float *f, float omega, int zcount,
                              bool six_strips, hipStream_t hs);

// out = S^iters(in) on a 2D analytic-box grid, iters 1 or 2 (mgrt_smooth2d.hip); in may be null.
// res (optional) receives the residual of out or, when cg is given without e, its restriction to
// the coarse grid cg; e with its coarse grid cg (optional) makes the input in + 4*P(e).
mgrt_status smooth_sweep_2d(const mgrt_grid *g, const mgrt_grid *cg, float *out, float *res, const float *in,
                            const float *e, const float *f, float omega, int iters, hipStream_t hs);
mgrt_status smooth_sweep_2d(const mgrt_grid *g, const mgrt_grid *cg, double *out, double *res, const double *in,
                            const double *e, const double *f, double omega, int iters, hipStream_t hs);

} // namespace mgrt

#endif
