// mgrt_kernels.hip -- gfx950 kernels of the multigrid V-cycle and their C-ABI launchers.
//
// Numerics: every expression below keeps the operand order and the mixed precision of the
// reference's OpenCL source (bare literals such as 1.0/6.0 are double, so the float build does
// those products in double and rounds once).  The file is compiled with -ffp-contract=off: a
// fused multiply-add anywhere here would change results in the last bit.
//
// Wavefronts are 64 wide; blocks are one or four wavefronts.  All kernels are HBM-bound stencils:
// no matrix cores.
#include "mgrt_internal.h"
#include "mgrt_device.h"

using namespace mgrt;

// Block partials of mgrt_sumsq_noalloc_f32, the one entry point that allocates nothing: a fixed array
// in the code object instead of memory from hipMalloc.  Host code must not pass this name to a kernel
// (on the host it is a stand-in, not a device address): static_partials_setup() asks the runtime for
// the address once, with the device current, and everything uses that pointer.
static __device__ __align__(256) double scratch[mgrt::kStaticPartials];

namespace mgrt {
mgrt_status static_partials_setup(double **device_address)
{
    MGRT_HIP(hipGetSymbolAddress((void **)device_address, HIP_SYMBOL(scratch)));
    return MGRT_OK;
}
} // namespace mgrt

namespace {

constexpr int kWave = 64;

template <class R>
mgrt_status restrict_rows_impl(const mgrt_grid *fine, const mgrt_grid *coarse, R *rc, const R *rf, int j0, int jstep, int nrows,
                               hipStream_t hs);
template <class R>
mgrt_status residual_restrict_tiles_impl(const mgrt_grid *fine, const mgrt_grid *coarse, R *rc, const R *u, const R *f,
                                         hipStream_t hs);

} // namespace

namespace mgrt {
mgrt_status restrict_rows(const mgrt_grid *fine, const mgrt_grid *coarse, float *rc, const float *rf, int j0, int jstep,
                          int nrows, hipStream_t hs)
{
    return restrict_rows_impl<float>(fine, coarse, rc, rf, j0, jstep, nrows, hs);
}
mgrt_status restrict_rows(const mgrt_grid *fine, const mgrt_grid *coarse, double *rc, const double *rf, int j0, int jstep,
                          int nrows, hipStream_t hs)
{
    return restrict_rows_impl<double>(fine, coarse, rc, rf, j0, jstep, nrows, hs);
}
mgrt_status residual_restrict_tiles(const mgrt_grid *fine, const mgrt_grid *coarse, float *rc, const float *u, const float *f,
                                    hipStream_t hs)
{
    return residual_restrict_tiles_impl<float>(fine, coarse, rc, u, f, hs);
}
mgrt_status residual_restrict_tiles(const mgrt_grid *fine, const mgrt_grid *coarse, double *rc, const double *u, const double *f,
                                    hipStream_t hs)
{
    return residual_restrict_tiles_impl<double>(fine, coarse, rc, u, f, hs);
}
} // namespace mgrt

namespace {

// ---- cell classes ---------------------------------------------------------------------------
// Replaces getCellType/isBorder on a NaN-encoded Cell (mg_1.cl:14-26).  With a byte map the class
// is one load; for a box it is index arithmetic and costs no memory traffic.
template <int DIM, bool MAP>
__device__ __forceinline__ int cell_class(const mgrt_grid &g, int x, int y, int z)
{
    if (MAP)
        return g.cells[x + (long long)g.pitch * y + g.slab * z];
    const int yg = y + g.y0;
    if (DIM == 3) {
        if (yg == 0) return g.face[2];
        if (yg == g.gny - 1) return g.face[3];
        if (x == 0) return g.face[0];
        if (x == g.nx - 1) return g.face[1];
        if (z == 0) return g.face[4];
        if (z == g.nz - 1) return g.face[5];
    } else {
        if (x == 0) return g.face[0];
        if (x == g.nx - 1) return g.face[1];
        if (yg == 0) return g.face[2];
        if (yg == g.gny - 1) return g.face[3];
    }
    return MGRT_CELL_INNER;
}

__device__ __forceinline__ bool is_border(int t)
{
    return t == MGRT_CELL_DIRICHLET || t == MGRT_CELL_NEUMANN;
}

// Extension (mgrt_grid.neumann == 1, analytic box): element offset from a face cell to its neighbour
// one step inside along the normal of the face the cell is counted to (same priority as cell_class).
template <int DIM> __device__ __forceinline__ long long inward_offset(const mgrt_grid &g, int x, int y, int z)
{
    const int yg = y + g.y0;
    if (DIM == 3) {
        if (yg == 0) return g.pitch;
        if (yg == g.gny - 1) return -(long long)g.pitch;
        if (x == 0) return 1;
        if (x == g.nx - 1) return -1;
        if (z == 0) return g.slab;
        return -g.slab;
    }
    if (x == 0) return 1;
    if (x == g.nx - 1) return -1;
    if (yg == 0) return g.pitch;
    return -(long long)g.pitch;
}

// ---- smoother, 3D: one colour pass ------------------------------------------------------------
// iteration_kernel + do_rbgauss + jacobi_iteration, mg_1.cl:29-43,56-104.  One thread owns one
// (x,y) column of the active colour and marches it in ascending z: u_new[z-1] is carried in a
// register, u_old[z+1] comes from the prefetch ring.  The four in-plane neighbours have the other
// colour and are not written by this pass.  PF planes of loads are kept in flight per thread.
template <class R> struct PlaneVals {
    R xm, xp, ym, yp, own, f;
    int t;
};

// NEU: the first-order Neumann extension (mgrt_grid.neumann) is compiled in; the reference's kernels
// are the instantiations without it.
template <class R, bool MAP, int PF, bool NEU = false>
__global__ __launch_bounds__(kWave) void k_smooth3d_pass(mgrt_grid g, R *u, const R *__restrict__ f,
                                                         R omega, int odd, int zcount, int ylo, int yhi)
{
    const int hx = (g.nx + 1) / 2;
    const long long idx = (long long)blockIdx.x * kWave + threadIdx.x;
    if (idx >= (long long)hx * (yhi - ylo))
        return;
    const int y = ylo + (int)(idx / hx);
    const int x = 2 * (int)(idx % hx) + ((odd + y + g.y0) & 1);
    if (x >= g.nx)
        return;
    const long long p = g.pitch, s = g.slab;
    const long long col = x + p * y;
    // planes whose `own` value is needed: z < zcount, plus plane zcount as u_old[z+1] of the last
    const int zload = zcount < g.nz ? zcount + 1 : g.nz;

    auto load_plane = [&](int z) {
        PlaneVals<R> v;
        v.t = z < zcount ? cell_class<3, MAP>(g, x, y, z) : MGRT_CELL_OUTER;
        const long long b = col + s * z;
        v.own = u[b];
        v.xm = v.xp = v.ym = v.yp = v.f = 0;
        if (v.t == MGRT_CELL_INNER) {
            v.xp = u[b + 1];
            v.xm = u[b - 1];
            v.yp = u[b + p];
            v.ym = u[b - p];
            v.f = f[b];
        } else if (v.t == MGRT_CELL_DIRICHLET) {
            v.f = f[b];
        } else if (NEU && v.t == MGRT_CELL_NEUMANN) {
            // first-order condition: the neighbour inside.  Along z it is the plane above (its value
            // before this pass) or below (after), which the march has in hand; in the plane it has the
            // other colour and is read here.
            v.f = f[b];
            const long long in = inward_offset<3>(g, x, y, z);
            if (in != s && in != -s)
                v.xp = u[b + in];
        }
        return v;
    };

    PlaneVals<R> ring[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k)
        if (k < zload)
            ring[k] = load_plane(k);

    R below = 0; // value of plane z-1 as this pass leaves it
    for (int z0 = 0; z0 < zcount; z0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int z = z0 + k;
            if (z < zcount) {
                const PlaneVals<R> c = ring[k];
                if (z + PF < zload)
                    ring[k] = load_plane(z + PF);
                const R above = z + 1 < zload ? ring[(k + 1) % PF].own : (R)0;
                R next = c.own;
                if (c.t == MGRT_CELL_INNER) {
                    const R val = (1.0 / 6.0) * (c.xp + c.xm + c.yp + c.ym + above + below - c.f);
                    next = val * omega + (1.0 - omega) * c.own;
                    u[col + s * z] = next;
                } else if (c.t == MGRT_CELL_DIRICHLET) {
                    next = c.f;
                    u[col + s * z] = next;
                } else if (NEU && c.t == MGRT_CELL_NEUMANN) {
                    const long long in = inward_offset<3>(g, x, y, z);
                    const R inside = in == s ? above : (in == -s ? below : c.xp);
                    next = inside + c.f;
                    u[col + s * z] = next;
                }
                below = next;
            }
        }
    }
}

// ---- smoother, 2D: one colour pass (true red-black, mg_0.cl:29-40,52-98) ----------------------
template <class R, bool MAP, bool NEU = false>
__global__ __launch_bounds__(256) void k_smooth2d_pass(mgrt_grid g, R *u, const R *__restrict__ f, R omega,
                                                       int odd, int ylo, int yhi)
{
    const int hx = (g.nx + 1) / 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)hx * (yhi - ylo))
        return;
    const int y = ylo + (int)(idx / hx);
    const int x = 2 * (int)(idx % hx) + ((odd + y + g.y0) & 1);
    if (x >= g.nx)
        return;
    const long long p = g.pitch, b = x + p * y;
    const int t = cell_class<2, MAP>(g, x, y, 0);
    if (t == MGRT_CELL_INNER) {
        const R val = 0.25 * (u[b + 1] + u[b - 1] + u[b + p] + u[b - p] - f[b]);
        u[b] = val * omega + (1.0 - omega) * u[b];
    } else if (t == MGRT_CELL_DIRICHLET) {
        u[b] = f[b];
    } else if (NEU && t == MGRT_CELL_NEUMANN) {
        u[b] = u[b + inward_offset<2>(g, x, y, 0)] + f[b];
    }
}

// ---- block reduction of doubles into one partial per block (fixed order: deterministic) -------
template <bool MAX> __device__ __forceinline__ double combine(double a, double b)
{
    return MAX ? (a < b ? b : a) : a + b;
}

template <bool MAX> __device__ __forceinline__ double wave_reduce(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1)
        v = combine<MAX>(v, __shfl_down(v, off, kWave));
    return v;
}

template <bool MAX, int NWAVES> __device__ __forceinline__ void block_reduce_store(double v, double *partial)
{
    __shared__ double sh[NWAVES];
    v = wave_reduce<MAX>(v);
    const int lane = threadIdx.x % kWave, w = threadIdx.x / kWave;
    if (lane == 0)
        sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = sh[0];
#pragma unroll
        for (int i = 1; i < NWAVES; ++i)
            a = combine<MAX>(a, sh[i]);
        *partial = a;
    }
}

// second stage: one block folds the partials in a fixed order
template <bool MAX> __global__ __launch_bounds__(256) void k_fold_partials(const double *partials, int n, double *out)
{
    double v = 0;
    for (int i = threadIdx.x; i < n; i += 256)
        v = combine<MAX>(v, partials[i]);
    block_reduce_store<MAX, 4>(v, out);
}

// ---- residual (mg_1.cl:46-53,111-133, mg_0.cl:43-49,105-126) -----------------------------------
// One thread per (x,y), marching a chunk of ZC planes with the three z values of its column held
// in registers.  NORM adds sum(r*r) in double, one partial per block.
template <class R, int DIM, bool MAP, bool NORM, int ZC, bool NEU = false>
__global__ __launch_bounds__(256) void k_residual(mgrt_grid g, R *r, const R *__restrict__ u,
                                                  const R *__restrict__ f, int zero_unwritten,
                                                  double *partials, int ylo, int yhi, unsigned nbxy)
{
    // 1D launch, z chunk slowest, handed out XCD-contiguously: the rows y-1/y+1 and the chunk's
    // two halo planes are then found in the L2 of the XCD that fetched them for a neighbour block.
    const unsigned bid = (unsigned)xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const unsigned bxy = bid % nbxy, bz = bid / nbxy;
    const unsigned idx = bxy * 256u + threadIdx.x;
    const bool live = idx < (unsigned)g.nx * (unsigned)(yhi - ylo);
    double acc = 0;
    if (live) {
        const int y = ylo + (int)(idx / (unsigned)g.nx), x = (int)(idx % (unsigned)g.nx);
        const long long p = g.pitch, s = g.slab;
        const int zb = DIM == 3 ? (int)bz * ZC : 0;
        const int ze = DIM == 3 ? (zb + ZC < g.nz ? zb + ZC : g.nz) : 1;
        long long b = x + p * y + s * zb;
        R um = (DIM == 3 && zb > 0) ? u[b - s] : (R)0;
        R uc = u[b];
        for (int z = zb; z < ze; ++z, b += s) {
            const R up = (DIM == 3 && z + 1 < g.nz) ? u[b + s] : (R)0;
            const int t = cell_class<DIM, MAP>(g, x, y, z);
            if (t == MGRT_CELL_INNER) {
                R v;
                if (DIM == 3)
                    v = f[b] - (u[b + 1] + u[b - 1] + u[b + p] + u[b - p] + up + um - 6 * uc);
                else
                    v = f[b] - (u[b + 1] + u[b - 1] + u[b + p] + u[b - p] - 4 * uc);
                r[b] = v;
                if (NORM)
                    acc += (double)v * (double)v;
            } else if (t == MGRT_CELL_DIRICHLET) {
                const R v = f[b] - uc;
                r[b] = v;
                if (NORM)
                    acc += (double)v * (double)v;
            } else if (NEU && t == MGRT_CELL_NEUMANN) {
                const R v = f[b] - (uc - u[b + inward_offset<DIM>(g, x, y, z)]);
                r[b] = v;
                if (NORM)
                    acc += (double)v * (double)v;
            } else if (zero_unwritten) {
                r[b] = 0;
            }
            um = uc;
            uc = up;
        }
    }
    if (NORM)
        block_reduce_store<false, 4>(acc, partials + bid);
}

// 3D residual with 16 bytes per lane: a thread owns VX consecutive x of one row (4 floats or 2
// doubles) and walks a chunk of planes.  A read path fed with 4-byte loads tops out near 4 TB/s on
// this chip and 8- or 16-byte loads reach 6.2-6.4 (rows need not be 16-byte aligned); per point
// this kernel issues 7/VX wide memory instructions instead of 8 narrow ones.  Same expression and
// operand order as k_residual.
template <class R, int VX> struct alignas(sizeof(R)) VecU {
    R v[VX];
};
template <class R, int VX> __device__ __forceinline__ void load_vec(R (&d)[VX], const R *p, int nval)
{
    if (nval == VX) {
        const VecU<R, VX> t = *reinterpret_cast<const VecU<R, VX> *>(p);
#pragma unroll
        for (int e = 0; e < VX; ++e) d[e] = t.v[e];
    } else {
#pragma unroll
        for (int e = 0; e < VX; ++e) d[e] = e < nval ? p[e] : (R)0;
    }
}

template <class R, bool MAP, bool NORM>
__global__ __launch_bounds__(256) void k_residual3d_v(mgrt_grid g, R *r, const R *__restrict__ u,
                                                      const R *__restrict__ f, int zero_unwritten,
                                                      double *partials, int ylo, int yhi, unsigned nbxy, int ZC)
{
    constexpr int VX = 16 / (int)sizeof(R);
    const unsigned bid = (unsigned)xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const unsigned bxy = bid % nbxy, bz = bid / nbxy;
    const unsigned nvx = ((unsigned)g.nx + VX - 1) / VX;
    const unsigned idx = bxy * 256u + threadIdx.x;
    const bool live = idx < nvx * (unsigned)(yhi - ylo);
    double acc = 0;
    if (live) {
        const int y = ylo + (int)(idx / nvx), x0 = (int)(idx % nvx) * VX;
        const int nval = g.nx - x0 < VX ? g.nx - x0 : VX;
        const long long p = g.pitch, s = g.slab;
        const int zb = (int)bz * ZC, ze = zb + ZC < g.nz ? zb + ZC : g.nz;
        const bool has_ym = y > 0, has_yp = y + 1 < g.ny, has_l = x0 > 0, has_r = x0 + VX < g.nx;
        long long b = x0 + p * y + s * zb;
        R um[VX], uc[VX], up[VX], fv[VX], ym[VX], yp[VX];
#pragma unroll
        for (int e = 0; e < VX; ++e) um[e] = 0;
        if (zb > 0) load_vec<R, VX>(um, u + b - s, nval);
        load_vec<R, VX>(uc, u + b, nval);
        for (int z = zb; z < ze; ++z, b += s) {
#pragma unroll
            for (int e = 0; e < VX; ++e) up[e] = ym[e] = yp[e] = 0;
            if (z + 1 < g.nz) load_vec<R, VX>(up, u + b + s, nval);
            load_vec<R, VX>(fv, f + b, nval);
            if (has_ym) load_vec<R, VX>(ym, u + b - p, nval);
            if (has_yp) load_vec<R, VX>(yp, u + b + p, nval);
            const R left = has_l ? u[b - 1] : (R)0, right = has_r ? u[b + VX] : (R)0;
            R out[VX];
            bool wr[VX];
            bool all = nval == VX;
#pragma unroll
            for (int e = 0; e < VX; ++e) {
                const R xm = e == 0 ? left : uc[e > 0 ? e - 1 : 0];
                const R xp = e == VX - 1 ? right : uc[e + 1 < VX ? e + 1 : 0];
                const int t = e < nval ? cell_class<3, MAP>(g, x0 + e, y, z) : MGRT_CELL_OUTER;
                const R vi = fv[e] - (xp + xm + yp[e] + ym[e] + up[e] + um[e] - 6 * uc[e]);
                const R vd = fv[e] - uc[e];
                wr[e] = t == MGRT_CELL_INNER || t == MGRT_CELL_DIRICHLET;
                out[e] = t == MGRT_CELL_INNER ? vi : (t == MGRT_CELL_DIRICHLET ? vd : (R)0);
                if (NORM && wr[e])
                    acc += (double)out[e] * (double)out[e];
                all = all && (wr[e] || zero_unwritten);
            }
            if (all) {
                VecU<R, VX> t;
#pragma unroll
                for (int e = 0; e < VX; ++e) t.v[e] = out[e];
                *reinterpret_cast<VecU<R, VX> *>(r + b) = t;
            } else {
#pragma unroll
                for (int e = 0; e < VX; ++e)
                    if (e < nval && (wr[e] || zero_unwritten))
                        r[b + e] = out[e];
            }
#pragma unroll
            for (int e = 0; e < VX; ++e) {
                um[e] = uc[e];
                uc[e] = up[e];
            }
        }
    }
    if (NORM)
        block_reduce_store<false, 4>(acc, partials + bid);
}

// ---- full-weighting restriction (mg_1.cl:146-190, mg_0.cl:139-157) ----------------------------
// One thread per coarse point.  `fine`/`coarse` may be y-slabs: coarse global row jg maps to fine
// global row 2*jg.  Coarse rows cylo, cylo + jstep, ... (nrows of them).
template <class R, int DIM, bool MAP, bool NEU = false>
__global__ __launch_bounds__(256) void k_restrict(mgrt_grid fg, mgrt_grid cg, R *rc, const R *__restrict__ rf,
                                                  int cylo, int nrows, unsigned nbxy, int jstep)
{
    const unsigned bid = (unsigned)xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const unsigned idx = (bid % nbxy) * 256u + threadIdx.x;
    if (idx >= (unsigned)cg.nx * (unsigned)nrows)
        return;
    const int j = cylo + (int)(idx / (unsigned)cg.nx) * jstep, i = (int)(idx % (unsigned)cg.nx);
    const int k = DIM == 3 ? (int)(bid / nbxy) : 0;
    const int x = 2 * i, y = 2 * (j + cg.y0) - fg.y0, z = 2 * k;
    const long long p = fg.pitch, s = fg.slab;
    const R *c = rf + x + p * y + s * z;
    const long long d = i + (long long)cg.pitch * j + cg.slab * k;

    // forward 2x2(x2) block, tested in the reference's short-circuit order; cells past the far
    // faces are never classified because the first test already hits the face.  A centre on a face
    // of the grid always injects: the reference would weight cells outside its arrays there if the
    // face carried a non-border class.
    const bool on_face = x == 0 || x == fg.nx - 1 || y + fg.y0 == 0 || y + fg.y0 == fg.gny - 1 ||
                         (DIM == 3 && (z == 0 || z == fg.nz - 1));
    bool border = on_face || is_border(cell_class<DIM, MAP>(fg, x, y, z));
    if (!border) {
        border = is_border(cell_class<DIM, MAP>(fg, x + 1, y, z)) ||
                 is_border(cell_class<DIM, MAP>(fg, x, y + 1, z)) ||
                 is_border(cell_class<DIM, MAP>(fg, x + 1, y + 1, z));
        if (DIM == 3 && !border)
            border = is_border(cell_class<DIM, MAP>(fg, x, y, z + 1)) ||
                     is_border(cell_class<DIM, MAP>(fg, x + 1, y, z + 1)) ||
                     is_border(cell_class<DIM, MAP>(fg, x, y + 1, z + 1)) ||
                     is_border(cell_class<DIM, MAP>(fg, x + 1, y + 1, z + 1));
    }
    if (border) {
        // extension: the coarse Neumann cell's equation spans twice the spacing and its correction is
        // applied fourfold, so it takes half the fine residual
        const bool halve = NEU && cell_class<DIM, false>(fg, x, y, z) == MGRT_CELL_NEUMANN;
        rc[d] = halve ? (R)(0.5 * c[0]) : c[0];
    } else if (DIM == 3) {
        rc[d] = 1.0 / 64.0 * (c[s + p + 1] + c[s + p - 1] + c[s - p + 1] + c[s - p - 1] + c[-s + p + 1] +
                              c[-s + p - 1] + c[-s - p + 1] + c[-s - p - 1]) +
                1.0 / 32.0 * (c[p + 1] + c[p - 1] + c[-p + 1] + c[-p - 1] + c[-s + p] + c[-s - p] + c[-s + 1] +
                              c[-s - 1] + c[s + p] + c[s - p] + c[s + 1] + c[s - 1]) +
                1.0 / 16.0 * (c[p] + c[-p] + c[1] + c[-1] + c[s] + c[-s]) + 1.0 / 8.0 * (c[0]);
    } else {
        rc[d] = 1.0 / 16.0 * (c[-p - 1] + c[-p + 1] + c[p - 1] + c[p + 1]) +
                1.0 / 8.0 * (c[-p] + c[1] + c[-1] + c[p]) + 1.0 / 4.0 * (c[0]);
    }
}

// ---- residual and restriction together, tile by tile -----------------------------------------------
// rc = Restrict(f - A u) on a whole 3D analytic box (residual_kernel mg_1.cl:111-133 with the cells it
// leaves unwritten counted as 0, then reduction_kernel :146-190) without the fine residual in memory: a
// block forms the residuals under its tile of CX x CY x CZ coarse points, (2CX+1)(2CY+1)(2CZ+1) of them,
// in LDS and sums them there, every expression as in k_residual3d_v and k_restrict.  For the levels on
// which a march over z costs its step latency whatever it computes.
template <class R>
__global__ __launch_bounds__(1024) void k_residual_restrict3d(mgrt_grid fg, mgrt_grid cg, R *rc, const R *__restrict__ u,
                                                              const R *__restrict__ f, int ntx, int nty)
{
    constexpr int CX = 31, CY = 4, CZ = 4, FX = 2 * CX + 1, FY = 2 * CY + 1, FZ = 2 * CZ + 1; // FX = 63: a row a wavefront
    constexpr int ROWS = FY * FZ, NW = 16, NIT = (ROWS + NW - 1) / NW;
    static_assert(FX <= 64, "a row of the tile is one wavefront wide");
    __shared__ R rs[FZ * FY * 64];
    const int b = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int i0 = (b % ntx) * CX, j0 = (b / ntx % nty) * CY, k0 = (b / (ntx * nty)) * CZ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p = fg.pitch, s = fg.slab;
    // residuals: row `row` of the tile (y fastest, then z) by wavefront row % 16, a lane a point.  All loads
    // of all rows are issued before the first is used: addresses are clamped into the grid's interior, and
    // what a clamped point computes is discarded.
    const int x = 2 * i0 - 1 + lane, xc = x < 1 ? 1 : (x > fg.nx - 2 ? fg.nx - 2 : x);
    R um[NIT], up[NIT], ym[NIT], yp[NIT], zm[NIT], zp[NIT], uc[NIT], fv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = wave + NW * it, y = 2 * j0 - 1 + row % FY, z = 2 * k0 - 1 + row / FY;
        const int yc = y < 1 ? 1 : (y > fg.ny - 2 ? fg.ny - 2 : y), zc = z < 1 ? 1 : (z > fg.nz - 2 ? fg.nz - 2 : z);
        const long long a = xc + p * yc + s * zc;
        up[it] = u[a + 1], um[it] = u[a - 1], yp[it] = u[a + p], ym[it] = u[a - p], zp[it] = u[a + s], zm[it] = u[a - s];
        // the point itself may lie on a face (a Dirichlet cell takes f - u there): its own address
        const bool in = x >= 0 && x < fg.nx && y >= 0 && y < fg.ny && z >= 0 && z < fg.nz;
        const long long own = in ? x + p * y + s * z : a;
        uc[it] = u[own], fv[it] = f[own];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = wave + NW * it, y = 2 * j0 - 1 + row % FY, z = 2 * k0 - 1 + row / FY;
        const bool in = x >= 0 && x < fg.nx && y >= 0 && y < fg.ny && z >= 0 && z < fg.nz;
        const int cls = in ? cell_class<3, false>(fg, x, y, z) : (int)MGRT_CELL_OUTER;
        const R vi = fv[it] - (up[it] + um[it] + yp[it] + ym[it] + zp[it] + zm[it] - 6 * uc[it]);
        const R vd = fv[it] - uc[it];
        if (row < ROWS)
            rs[row * 64 + lane] = cls == MGRT_CELL_INNER ? vi : (cls == MGRT_CELL_DIRICHLET ? vd : (R)0);
    }
    __syncthreads();
    const int ci = threadIdx.x % 32, cj = threadIdx.x / 32 % CY, ck = (int)threadIdx.x / (32 * CY);
    const int i = i0 + ci, j = j0 + cj, k = k0 + ck;
    if (ci >= CX || ck >= CZ || i >= cg.nx || j >= cg.ny || k >= cg.nz)
        return;
    const int xf = 2 * i, yf = 2 * j, zf = 2 * k;
    constexpr int P = 64, S = 64 * FY;
    const R *c = rs + (2 * ci + 1) + P * (2 * cj + 1) + S * (2 * ck + 1);
    // as k_restrict: a centre on a face of the grid injects, and so does one whose forward 2x2x2 block
    // holds a border cell
    const bool on_face = xf == 0 || xf == fg.nx - 1 || yf == 0 || yf == fg.gny - 1 || zf == 0 || zf == fg.nz - 1;
    bool border = on_face || is_border(cell_class<3, false>(fg, xf, yf, zf));
    if (!border)
        border = is_border(cell_class<3, false>(fg, xf + 1, yf, zf)) || is_border(cell_class<3, false>(fg, xf, yf + 1, zf)) ||
                 is_border(cell_class<3, false>(fg, xf + 1, yf + 1, zf)) || is_border(cell_class<3, false>(fg, xf, yf, zf + 1)) ||
                 is_border(cell_class<3, false>(fg, xf + 1, yf, zf + 1)) || is_border(cell_class<3, false>(fg, xf, yf + 1, zf + 1)) ||
                 is_border(cell_class<3, false>(fg, xf + 1, yf + 1, zf + 1));
    R val;
    if (border)
        val = c[0];
    else
        val = 1.0 / 64.0 * (c[S + P + 1] + c[S + P - 1] + c[S - P + 1] + c[S - P - 1] + c[-S + P + 1] + c[-S + P - 1] +
                            c[-S - P + 1] + c[-S - P - 1]) +
              1.0 / 32.0 * (c[P + 1] + c[P - 1] + c[-P + 1] + c[-P - 1] + c[-S + P] + c[-S - P] + c[-S + 1] + c[-S - 1] +
                            c[S + P] + c[S - P] + c[S + 1] + c[S - 1]) +
              1.0 / 16.0 * (c[P] + c[-P] + c[1] + c[-1] + c[S] + c[-S]) + 1.0 / 8.0 * (c[0]);
    rc[i + (long long)cg.pitch * j + cg.slab * k] = val;
}

// ---- interpolation (mg_1.cl:200-267, mg_0.cl:167-222) -----------------------------------------
// dot(real4, real4) evaluated left to right in R: ((a0*b0 + a1*b1) + a2*b2) + a3*b3.
template <class R> __device__ __forceinline__ R dot4(R a0, R a1, R a2, R a3, R b0, R b1, R b2, R b3)
{
    return a0 * b0 + a1 * b1 + a2 * b2 + a3 * b3;
}

// MODE 0: out = u + 4*P(e) (residual_correct_kernel); MODE 1: out = P(e) (prolongation_kernel).
// The weight of a coarse neighbour one step up in x/y/z is 0 when the fine index is even; such
// neighbours are still multiplied in, as the reference does, but their index is clamped to the
// coarse array so that no address outside it is formed.
template <class R, int DIM, bool MAP, int MODE>
__global__ __launch_bounds__(256) void k_interp(mgrt_grid fg, mgrt_grid cg, R *out, const R *u,
                                                const R *__restrict__ e, int ylo, int yhi)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)fg.nx * (yhi - ylo))
        return;
    const int y = ylo + (int)(idx / fg.nx), x = (int)(idx % fg.nx);
    const int z = DIM == 3 ? blockIdx.y : 0;
    const int yg = y + fg.y0;
    const long long d = x + (long long)fg.pitch * y + fg.slab * z;
    const int ci = x / 2, cj = yg / 2 - cg.y0, ck = z / 2;
    const int ci1 = ci + 1 < cg.nx ? ci + 1 : ci;
    const int cj1 = cj + 1 < cg.ny ? cj + 1 : cj;
    const long long cp = cg.pitch, cs = cg.slab;
    const R *e0 = e + cs * ck;
    R val;
    if (is_border(cell_class<DIM, MAP>(fg, x, y, z))) {
        val = e0[ci + cp * cj];
    } else if (DIM == 2 && MODE == 1) {
        // the 2D prolongation kernel spells the four parity cases out (mg_0.cl:210-218)
        if (x % 2 == 0 && yg % 2 == 0)
            val = e0[ci + cp * cj];
        else if (x % 2 == 0)
            val = 0.5 * (e0[ci + cp * cj] + e0[ci + cp * cj1]);
        else if (yg % 2 == 0)
            val = 0.5 * (e0[ci + cp * cj] + e0[ci1 + cp * cj]);
        else
            val = 0.25 * (e0[ci + cp * cj] + e0[ci + cp * cj1] + e0[ci1 + cp * cj1] + e0[ci1 + cp * cj]);
    } else {
        const R wu = 0.5 * (x % 2), ww = 0.5 * (yg % 2);
        if (DIM == 3) {
            const R wv = 0.5 * (z % 2);
            const int ck1 = ck + 1 < cg.nz ? ck + 1 : ck;
            const R *e1 = e + cs * ck1;
            const R n10 = (1.0 - wu) * (1.0 - ww) * (1.0 - wv), n11 = (1.0 - ww) * wu * (1.0 - wv),
                    n12 = (1.0 - wu) * ww * (1.0 - wv), n13 = ww * wu * (1.0 - wv);
            const R n20 = (1.0 - wu) * (1.0 - ww) * wv, n21 = (1.0 - ww) * wu * wv, n22 = (1.0 - wu) * ww * wv,
                    n23 = ww * wu * wv;
            val = dot4<R>(e0[ci + cp * cj], e0[ci1 + cp * cj], e0[ci + cp * cj1], e0[ci1 + cp * cj1], n10, n11, n12,
                          n13) +
                  dot4<R>(e1[ci + cp * cj], e1[ci1 + cp * cj], e1[ci + cp * cj1], e1[ci1 + cp * cj1], n20, n21, n22,
                          n23);
        } else {
            const R n0 = (1.0 - wu) * (1.0 - ww), n1 = (1.0 - ww) * wu, n2 = (1.0 - wu) * ww, n3 = ww * wu;
            val = dot4<R>(e0[ci + cp * cj], e0[ci1 + cp * cj], e0[ci + cp * cj1], e0[ci1 + cp * cj1], n0, n1, n2, n3);
        }
    }
    out[d] = MODE == 0 ? u[d] + val * 4 : val;
}

// 3D interpolation as a z-march with 16 bytes per lane on the fine side: a thread owns VX
// consecutive fine x (4 floats / 2 doubles, starting at an even x) of the fine row pair (2j, 2j+1),
// walks a chunk of coarse planes and carries the coarse values of plane k+1 over as plane k of the
// next step; it keeps the VX/2+1 coarse values per coarse row and plane that its points
// interpolate from.  Per 4*VX fine points and coarse plane: 2*(VX/2+1) coarse loads, 4 wide loads
// of u, 4 wide stores.  `zchunk` coarse planes per thread.  Same expressions, same operand order as
// k_interp.
template <class R, bool MAP, int MODE>
__global__ __launch_bounds__(256) void k_interp3d_v(mgrt_grid fg, mgrt_grid cg, R *out, const R *u,
                                                    const R *__restrict__ e, int ylo, int yhi, int ncrows,
                                                    unsigned nbxy, int zchunk)
{
    constexpr int VX = 16 / (int)sizeof(R), NC = VX / 2 + 1;
    const unsigned bid = (unsigned)xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const unsigned bxy = bid % nbxy, zc = bid / nbxy;
    const unsigned nvx = ((unsigned)fg.nx + VX - 1) / VX;
    const unsigned idx = bxy * 256u + threadIdx.x;
    if (idx >= nvx * (unsigned)ncrows)
        return;
    const int x0 = (int)(idx % nvx) * VX;
    const int nval = fg.nx - x0 < VX ? fg.nx - x0 : VX;
    // global coarse row handled by this thread; its fine rows are 2*cjg and 2*cjg+1
    const int cjg = (ylo + fg.y0) / 2 + (int)(idx / nvx);
    const int cj = cjg - cg.y0;
    const int cjc = cj < cg.ny ? cj : cg.ny - 1; // rows past the range are masked below
    const int cj1 = cjc + 1 < cg.ny ? cjc + 1 : cjc;
    const long long cp = cg.pitch, cs = cg.slab;
    long long o0[NC], o1[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int ci = x0 / 2 + k < cg.nx ? x0 / 2 + k : cg.nx - 1;
        o0[k] = ci + cp * cjc;
        o1[k] = ci + cp * cj1;
    }
    int yl[2];
    bool rowok[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        yl[b] = 2 * cjg + b - fg.y0;
        rowok[b] = yl[b] >= ylo && yl[b] < yhi;
    }
    const int kb = (int)zc * zchunk;
    const int ke = kb + zchunk < cg.nz ? kb + zchunk : cg.nz;
    // [plane][row][k]
    R lo[2][NC], hi[2][NC];
    {
        const R *pl = e + cs * kb;
#pragma unroll
        for (int k = 0; k < NC; ++k) hi[0][k] = pl[o0[k]], hi[1][k] = pl[o1[k]];
    }
    for (int ck = kb; ck < ke; ++ck) {
#pragma unroll
        for (int k = 0; k < NC; ++k) lo[0][k] = hi[0][k], lo[1][k] = hi[1][k];
        {
            const R *pl = e + cs * (ck + 1 < cg.nz ? ck + 1 : ck);
#pragma unroll
            for (int k = 0; k < NC; ++k) hi[0][k] = pl[o0[k]], hi[1][k] = pl[o1[k]];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int z = 2 * ck + c;
            if (z >= fg.nz)
                break;
            const R wv = 0.5 * c;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (!rowok[b])
                    continue;
                const R ww = 0.5 * b;
                const long long d = x0 + (long long)fg.pitch * yl[b] + fg.slab * z;
                R uv[VX], res[VX];
                if (MODE == 0)
                    load_vec<R, VX>(uv, u + d, nval);
#pragma unroll
                for (int el = 0; el < VX; ++el) {
                    const R wu = 0.5 * (el % 2);
                    const int k = el / 2;
                    R val;
                    if (el >= nval || is_border(cell_class<3, MAP>(fg, x0 + el, yl[b], z))) {
                        val = lo[0][k];
                    } else {
                        const R n10 = (1.0 - wu) * (1.0 - ww) * (1.0 - wv), n11 = (1.0 - ww) * wu * (1.0 - wv),
                                n12 = (1.0 - wu) * ww * (1.0 - wv), n13 = ww * wu * (1.0 - wv);
                        const R n20 = (1.0 - wu) * (1.0 - ww) * wv, n21 = (1.0 - ww) * wu * wv,
                                n22 = (1.0 - wu) * ww * wv, n23 = ww * wu * wv;
                        val = dot4<R>(lo[0][k], lo[0][k + 1], lo[1][k], lo[1][k + 1], n10, n11, n12, n13) +
                              dot4<R>(hi[0][k], hi[0][k + 1], hi[1][k], hi[1][k + 1], n20, n21, n22, n23);
                    }
                    res[el] = MODE == 0 ? uv[el] + val * 4 : val;
                }
                if (nval == VX) {
                    VecU<R, VX> t;
#pragma unroll
                    for (int el = 0; el < VX; ++el) t.v[el] = res[el];
                    *reinterpret_cast<VecU<R, VX> *>(out + d) = t;
                } else {
#pragma unroll
                    for (int el = 0; el < VX; ++el)
                        if (el < nval)
                            out[d + el] = res[el];
                }
            }
        }
    }
}

// ---- element-wise helpers -----------------------------------------------------------------------
template <class R, int DIM, bool MAP> __global__ __launch_bounds__(256) void k_zero_outside(mgrt_grid g, R *u)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)g.nx * g.ny)
        return;
    const int y = (int)(idx / g.nx), x = (int)(idx % g.nx), z = DIM == 3 ? blockIdx.y : 0;
    if (cell_class<DIM, MAP>(g, x, y, z) == MGRT_CELL_OUTER)
        u[x + (long long)g.pitch * y + g.slab * z] = 0;
}

template <class R> __global__ __launch_bounds__(256) void k_diff(mgrt_grid g, R *out, const R *a, const R *b)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)g.nx * g.ny)
        return;
    const int y = (int)(idx / g.nx), x = (int)(idx % g.nx);
    const long long i = x + (long long)g.pitch * y + g.slab * blockIdx.y;
    out[i] = a[i] - b[i];
}

// Mult, common_kernels.cl:65-72: the factor is a float whatever `real` is
template <class R> __global__ __launch_bounds__(256) void k_scale(mgrt_grid g, R *out, const R *in, float m)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)g.nx * g.ny)
        return;
    const int y = (int)(idx / g.nx), x = (int)(idx % g.nx);
    const long long i = x + (long long)g.pitch * y + g.slab * blockIdx.y;
    out[i] = in[i] * m;
}

// exact sum of squares / max |x|, double accumulation, one partial per block
template <class R, bool MAX>
__global__ __launch_bounds__(256) void k_norm_partials(mgrt_grid g, const R *__restrict__ x, double *partials,
                                                       int ylo, int yhi, int zc)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    if (idx < (long long)g.nx * (yhi - ylo)) {
        const int y = ylo + (int)(idx / g.nx), xx = (int)(idx % g.nx);
        const int zb = blockIdx.y * zc, ze = zb + zc < g.nz ? zb + zc : g.nz;
        long long b = xx + (long long)g.pitch * y + g.slab * zb;
        for (int z = zb; z < ze; ++z, b += g.slab) {
            const double v = (double)x[b];
            acc = MAX ? combine<true>(acc, v < 0 ? -v : v) : acc + v * v;
        }
    }
    block_reduce_store<MAX, 4>(acc, partials + blockIdx.x + (size_t)gridDim.x * blockIdx.y);
}

// ---- the reference's lossy reduction tree (auxiliary.cpp:119-135, common_kernels.cl:13-53) -----
// Level sizes go n -> max(1, n/4); output i folds inputs [4i, min(n, 4i+4)) left to right in R
// starting from 0.  SQUARE applies the element-wise square of the L2Norm kernel on the way in;
// MAX is the LInfNorm fold max(res, fabs(x)).
template <class R, bool MAX, bool SQUARE>
__global__ __launch_bounds__(256) void k_reftree_level(const R *in, R *out, long long n, long long newn)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= newn)
        return;
    const long long end = n < 4 * i + 4 ? n : 4 * i + 4;
    R res = 0.0;
    for (long long b = 4 * i; b < end; ++b) {
        R v = in[b];
        if (SQUARE)
            v = v * v;
        if (MAX) {
            const R a = v < 0 ? -v : v;
            res = res < a ? a : res;
        } else {
            res += v;
        }
    }
    out[i] = res;
}

inline unsigned blocks_for(long long n, int per) { return (unsigned)((n + per - 1) / per); }

// Planes per thread for the z-marching streaming kernels: `most` on large grids (register reuse
// along z), fewer on small ones so that the launch still has a few blocks per CU.
inline int z_chunk(unsigned blocks_xy, int nz, int most)
{
    int zc = most;
    while (zc > 1 && (unsigned long long)blocks_xy * ((nz + zc - 1) / zc) < 1024)
        zc /= 2;
    return zc;
}

// ---- launchers ---------------------------------------------------------------------------------
template <class R>
mgrt_status smooth_pass(const mgrt_grid *g, R *u, const R *f, R omega, int odd, int zcount, int ylo, int yhi,
                        mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(g, "mgrt_smooth_pass")) return st;
    MGRT_REQUIRE(u && f, "mgrt_smooth_pass: null field");
    MGRT_REQUIRE(odd == 0 || odd == 1, "mgrt_smooth_pass: odd must be 0 or 1");
    MGRT_REQUIRE(0 <= ylo && ylo <= yhi && yhi <= g->ny, "mgrt_smooth_pass: rows [%d,%d) outside [0,%d)", ylo, yhi,
                 g->ny);
    if (g->dim == 2)
        zcount = 1;
    MGRT_REQUIRE(zcount >= 0 && zcount <= g->nz, "mgrt_smooth_pass: zcount %d outside [0,%d]", zcount, g->nz);
    const long long n = (long long)((g->nx + 1) / 2) * (yhi - ylo);
    if (n == 0 || zcount == 0)
        return MGRT_OK;
    hipStream_t hs = (hipStream_t)s;
    if (g->dim == 3) {
        constexpr int PF = 8;
        if (neumann_extension(g))
            k_smooth3d_pass<R, false, PF, true><<<blocks_for(n, kWave), kWave, 0, hs>>>(*g, u, f, omega, odd, zcount, ylo, yhi);
        else if (g->cells)
            k_smooth3d_pass<R, true, PF><<<blocks_for(n, kWave), kWave, 0, hs>>>(*g, u, f, omega, odd, zcount, ylo, yhi);
        else
            k_smooth3d_pass<R, false, PF><<<blocks_for(n, kWave), kWave, 0, hs>>>(*g, u, f, omega, odd, zcount, ylo, yhi);
    } else {
        if (neumann_extension(g))
            k_smooth2d_pass<R, false, true><<<blocks_for(n, 256), 256, 0, hs>>>(*g, u, f, omega, odd, ylo, yhi);
        else if (g->cells)
            k_smooth2d_pass<R, true><<<blocks_for(n, 256), 256, 0, hs>>>(*g, u, f, omega, odd, ylo, yhi);
        else
            k_smooth2d_pass<R, false><<<blocks_for(n, 256), 256, 0, hs>>>(*g, u, f, omega, odd, ylo, yhi);
    }
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R>
mgrt_status smooth(const mgrt_grid *g, R *u, const R *f, R omega, int iters, int zcount, mgrt_stream s)
{
    MGRT_REQUIRE(iters >= 0, "mgrt_smooth: iters < 0");
    MGRT_REQUIRE(g != nullptr, "mgrt_smooth: grid is null");
    for (int i = 0; i < iters; ++i)
        for (int odd = 0; odd < 2; ++odd)
            if (mgrt_status st = smooth_pass<R>(g, u, f, omega, odd, zcount, 0, g->ny, s))
                return st;
    return MGRT_OK;
}

template <bool MAX> mgrt_status fold(const double *parts, int nparts, double *out, hipStream_t hs)
{
    k_fold_partials<MAX><<<1, 256, 0, hs>>>(parts, nparts, out);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R>
mgrt_status residual(const mgrt_grid *g, R *r, const R *u, const R *f, int zero_unwritten, double *norm2, int ylo,
                     int yhi, mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(g, "mgrt_residual")) return st;
    MGRT_REQUIRE(r && u && f, "mgrt_residual: null field");
    MGRT_REQUIRE(r != u && r != f, "mgrt_residual: r must not alias u or f");
    MGRT_REQUIRE(0 <= ylo && ylo <= yhi && yhi <= g->ny, "mgrt_residual: rows [%d,%d) outside [0,%d)", ylo, yhi, g->ny);
    hipStream_t hs = (hipStream_t)s;
    const long long n = (long long)g->nx * (yhi - ylo);
    if (n == 0) {
        if (norm2)
            MGRT_HIP(hipMemsetAsync(norm2, 0, sizeof(double), hs));
        return MGRT_OK;
    }
    constexpr int ZC = 16;
    const bool neu = neumann_extension(g);
    const bool vec = g->dim == 3 && !neu; // the 16-byte kernel; otherwise one thread per column, chunks of ZC planes
    const long long nthreads = vec ? (long long)((g->nx + 16 / (int)sizeof(R) - 1) / (16 / (int)sizeof(R))) * (yhi - ylo) : n;
    const unsigned nbxy = blocks_for(nthreads, 256);
    int zc3 = vec ? z_chunk(nbxy, g->nz, ZC) : (g->dim == 3 ? ZC : 1);
    // with the norm every block leaves a partial sum: longer chunks keep their number within the scratch
    while (norm2 && vec && zc3 < g->nz && (size_t)nbxy * blocks_for(g->nz, zc3) > (size_t)kMaxPartials)
        zc3 *= 2;
    dim3 grid(nbxy * (g->dim == 3 ? blocks_for(g->nz, zc3) : 1));
    const int nparts = (int)grid.x;
    MGRT_REQUIRE(!norm2 || (size_t)nparts <= (size_t)kMaxPartials, "mgrt_residual: grid too large for the fused norm");
    double *parts = nullptr;
    if (norm2)
        if (mgrt_status st = partials_for(hs, &parts)) return st;
#define MGRT_LAUNCH_RES(DIM, MAP, NORM)                                                                      \
    k_residual<R, DIM, MAP, NORM, ZC><<<grid, 256, 0, hs>>>(*g, r, u, f, zero_unwritten, parts, ylo, yhi, nbxy)
#define MGRT_LAUNCH_RES_NEU(DIM, NORM)                                                                        \
    k_residual<R, DIM, false, NORM, ZC, true><<<grid, 256, 0, hs>>>(*g, r, u, f, zero_unwritten, parts, ylo, yhi, nbxy)
#define MGRT_LAUNCH_RES3(MAP, NORM)                                                                           \
    k_residual3d_v<R, MAP, NORM><<<grid, 256, 0, hs>>>(*g, r, u, f, zero_unwritten, parts, ylo, yhi, nbxy, zc3)
    const bool map = g->cells != nullptr, nrm = norm2 != nullptr;
    if (neu) { // the extension lives in the plain kernel
        if (g->dim == 3) { if (nrm) MGRT_LAUNCH_RES_NEU(3, true); else MGRT_LAUNCH_RES_NEU(3, false); }
        else             { if (nrm) MGRT_LAUNCH_RES_NEU(2, true); else MGRT_LAUNCH_RES_NEU(2, false); }
    } else if (g->dim == 3) {
        if (map) { if (nrm) MGRT_LAUNCH_RES3(true, true); else MGRT_LAUNCH_RES3(true, false); }
        else     { if (nrm) MGRT_LAUNCH_RES3(false, true); else MGRT_LAUNCH_RES3(false, false); }
    } else {
        if (map) { if (nrm) MGRT_LAUNCH_RES(2, true, true); else MGRT_LAUNCH_RES(2, true, false); }
        else     { if (nrm) MGRT_LAUNCH_RES(2, false, true); else MGRT_LAUNCH_RES(2, false, false); }
    }
#undef MGRT_LAUNCH_RES
#undef MGRT_LAUNCH_RES3
#undef MGRT_LAUNCH_RES_NEU
    MGRT_HIP(hipGetLastError());
    if (nrm)
        return fold<false>(parts, nparts, norm2, hs);
    return MGRT_OK;
}

mgrt_status check_pair(const mgrt_grid *fine, const mgrt_grid *coarse, const char *who)
{
    if (mgrt_status st = check_grid(fine, who)) return st;
    if (mgrt_status st = check_grid(coarse, who)) return st;
    MGRT_REQUIRE(fine->dim == coarse->dim, "%s: fine and coarse dim differ", who);
    MGRT_REQUIRE(coarse->nx == (fine->nx + 1) / 2 && coarse->gny == (fine->gny + 1) / 2 &&
                     coarse->nz == (fine->nz + 1) / 2,
                 "%s: coarse extents (%d,%d,%d) are not (n+1)/2 of the fine (%d,%d,%d)", who, coarse->nx,
                 coarse->gny, coarse->nz, fine->nx, fine->gny, fine->nz);
    return MGRT_OK;
}

template <class R>
mgrt_status restrict_fw(const mgrt_grid *fine, const mgrt_grid *coarse, R *rc, const R *rf, int cylo, int cyhi,
                        mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_pair(fine, coarse, "mgrt_restrict")) return st;
    MGRT_REQUIRE(rc && rf, "mgrt_restrict: null field");
    MGRT_REQUIRE(0 <= cylo && cylo <= cyhi && cyhi <= coarse->ny, "mgrt_restrict: coarse rows [%d,%d) outside [0,%d)",
                 cylo, cyhi, coarse->ny);
    if (cylo < cyhi) {
        // fine rows 2*jg-1 .. 2*jg+1 must be held (the outermost only where they exist globally)
        const int flo = 2 * (cylo + coarse->y0) - fine->y0, fhi = 2 * (cyhi - 1 + coarse->y0) - fine->y0;
        MGRT_REQUIRE(flo >= 0 && fhi < fine->ny, "mgrt_restrict: fine slab does not hold rows %d..%d", flo + fine->y0,
                     fhi + fine->y0);
        MGRT_REQUIRE((flo >= 1 || flo + fine->y0 == 0) && (fhi + 1 < fine->ny || fhi + fine->y0 == fine->gny - 1),
                     "mgrt_restrict: fine slab lacks the halo row of the requested coarse rows");
    }
    const long long n = (long long)coarse->nx * (cyhi - cylo);
    if (n == 0)
        return MGRT_OK;
    const unsigned nbxy = blocks_for(n, 256);
    dim3 grid(nbxy * (unsigned)coarse->nz);
    hipStream_t hs = (hipStream_t)s;
    const bool map = fine->cells != nullptr;
    if (neumann_extension(fine)) {
        if (fine->dim == 3) k_restrict<R, 3, false, true><<<grid, 256, 0, hs>>>(*fine, *coarse, rc, rf, cylo, cyhi - cylo, nbxy, 1);
        else                k_restrict<R, 2, false, true><<<grid, 256, 0, hs>>>(*fine, *coarse, rc, rf, cylo, cyhi - cylo, nbxy, 1);
    } else if (fine->dim == 3) {
        if (map) k_restrict<R, 3, true><<<grid, 256, 0, hs>>>(*fine, *coarse, rc, rf, cylo, cyhi - cylo, nbxy, 1);
        else     k_restrict<R, 3, false><<<grid, 256, 0, hs>>>(*fine, *coarse, rc, rf, cylo, cyhi - cylo, nbxy, 1);
    } else {
        if (map) k_restrict<R, 2, true><<<grid, 256, 0, hs>>>(*fine, *coarse, rc, rf, cylo, cyhi - cylo, nbxy, 1);
        else     k_restrict<R, 2, false><<<grid, 256, 0, hs>>>(*fine, *coarse, rc, rf, cylo, cyhi - cylo, nbxy, 1);
    }
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R>
mgrt_status restrict_rows_impl(const mgrt_grid *fine, const mgrt_grid *coarse, R *rc, const R *rf, int j0, int jstep, int nrows,
                               hipStream_t hs)
{
    MGRT_REQUIRE(fine->dim == 3 && !fine->cells && !neumann_extension(fine), "restrict_rows: 3D analytic boxes only");
    if (nrows <= 0)
        return MGRT_OK;
    MGRT_REQUIRE(j0 >= 0 && jstep >= 1 && j0 + (long long)(nrows - 1) * jstep < coarse->ny, "restrict_rows: rows outside the grid");
    const unsigned nbxy = blocks_for((long long)coarse->nx * nrows, 256);
    k_restrict<R, 3, false><<<dim3(nbxy * (unsigned)coarse->nz), 256, 0, hs>>>(*fine, *coarse, rc, rf, j0, nrows, nbxy, jstep);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R>
mgrt_status residual_restrict_tiles_impl(const mgrt_grid *fine, const mgrt_grid *coarse, R *rc, const R *u, const R *f,
                                         hipStream_t hs)
{
    MGRT_REQUIRE(fine->dim == 3 && !fine->cells && !neumann_extension(fine) && fine->y0 == 0 && fine->ny == fine->gny &&
                     coarse->y0 == 0 && coarse->ny == coarse->gny,
                 "residual_restrict_tiles: whole 3D analytic boxes only");
    MGRT_REQUIRE(fine->nx >= 3 && fine->ny >= 3 && fine->nz >= 3, "residual_restrict_tiles: extents below 3");
    const int ntx = (coarse->nx + 30) / 31, nty = (coarse->ny + 3) / 4, ntz = (coarse->nz + 3) / 4;
    k_residual_restrict3d<R><<<dim3((unsigned)(ntx * nty * ntz)), 1024, 0, hs>>>(*fine, *coarse, rc, u, f, ntx, nty);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R, int MODE>
mgrt_status interp(const mgrt_grid *fine, const mgrt_grid *coarse, R *out, const R *u, const R *e, int ylo, int yhi,
                   mgrt_stream s)
{
    const char *who = MODE == 0 ? "mgrt_correct" : "mgrt_prolong";
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_pair(fine, coarse, who)) return st;
    MGRT_REQUIRE(out && e && (MODE == 1 || u), "%s: null field", who);
    MGRT_REQUIRE(0 <= ylo && ylo <= yhi && yhi <= fine->ny, "%s: rows [%d,%d) outside [0,%d)", who, ylo, yhi, fine->ny);
    if (ylo < yhi) {
        const int clo = (ylo + fine->y0) / 2 - coarse->y0, chi = (yhi - 1 + fine->y0 + 1) / 2 - coarse->y0;
        MGRT_REQUIRE(clo >= 0 && chi < coarse->ny, "%s: coarse slab does not hold rows %d..%d", who, clo + coarse->y0,
                     chi + coarse->y0);
    }
    const long long n = (long long)fine->nx * (yhi - ylo);
    if (n == 0)
        return MGRT_OK;
    dim3 grid(blocks_for(n, 256), fine->nz);
    hipStream_t hs = (hipStream_t)s;
    const bool map = fine->cells != nullptr;
    if (fine->dim == 3) {
        constexpr int VX = 16 / (int)sizeof(R);
        const int cj_lo = (ylo + fine->y0) / 2, cj_hi = (yhi - 1 + fine->y0) / 2;
        const int ncrows = cj_hi - cj_lo + 1;
        const unsigned nbxy = blocks_for((long long)((fine->nx + VX - 1) / VX) * ncrows, 256);
        const int zchunk = z_chunk(nbxy, coarse->nz, 8);
        const dim3 g1(nbxy * blocks_for(coarse->nz, zchunk));
        if (map) k_interp3d_v<R, true, MODE><<<g1, 256, 0, hs>>>(*fine, *coarse, out, u, e, ylo, yhi, ncrows, nbxy, zchunk);
        else     k_interp3d_v<R, false, MODE><<<g1, 256, 0, hs>>>(*fine, *coarse, out, u, e, ylo, yhi, ncrows, nbxy, zchunk);
    } else {
        if (map) k_interp<R, 2, true, MODE><<<grid, 256, 0, hs>>>(*fine, *coarse, out, u, e, ylo, yhi);
        else     k_interp<R, 2, false, MODE><<<grid, 256, 0, hs>>>(*fine, *coarse, out, u, e, ylo, yhi);
    }
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R> mgrt_status zero_outside(const mgrt_grid *g, R *u, mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(g, "mgrt_zero_outside")) return st;
    MGRT_REQUIRE(u, "mgrt_zero_outside: null field");
    bool any_outer = g->cells != nullptr;
    for (int i = 0; i < 6 && !any_outer; ++i)
        any_outer = g->face[i] == MGRT_CELL_OUTER;
    if (!any_outer)
        return MGRT_OK; // a box without outer faces has nothing to clear
    dim3 grid(blocks_for((long long)g->nx * g->ny, 256), g->nz);
    hipStream_t hs = (hipStream_t)s;
    if (g->dim == 3) {
        if (g->cells) k_zero_outside<R, 3, true><<<grid, 256, 0, hs>>>(*g, u);
        else          k_zero_outside<R, 3, false><<<grid, 256, 0, hs>>>(*g, u);
    } else {
        if (g->cells) k_zero_outside<R, 2, true><<<grid, 256, 0, hs>>>(*g, u);
        else          k_zero_outside<R, 2, false><<<grid, 256, 0, hs>>>(*g, u);
    }
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R> mgrt_status diff(const mgrt_grid *g, R *out, const R *a, const R *b, mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(g, "mgrt_diff")) return st;
    MGRT_REQUIRE(out && a && b, "mgrt_diff: null field");
    dim3 grid(blocks_for((long long)g->nx * g->ny, 256), g->nz);
    k_diff<R><<<grid, 256, 0, (hipStream_t)s>>>(*g, out, a, b);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R> mgrt_status scale(const mgrt_grid *g, R *out, const R *in, float m, mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(g, "mgrt_scale")) return st;
    MGRT_REQUIRE(out && in, "mgrt_scale: null field");
    dim3 grid(blocks_for((long long)g->nx * g->ny, 256), g->nz);
    k_scale<R><<<grid, 256, 0, (hipStream_t)s>>>(*g, out, in, m);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R, bool MAX>
mgrt_status norm(const mgrt_grid *g, const R *x, double *out, int ylo, int yhi, mgrt_stream s)
{
    const char *who = MAX ? "mgrt_maxabs" : "mgrt_sumsq";
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(g, who)) return st;
    MGRT_REQUIRE(x && out, "%s: null pointer", who);
    MGRT_REQUIRE(0 <= ylo && ylo <= yhi && yhi <= g->ny, "%s: rows [%d,%d) outside [0,%d)", who, ylo, yhi, g->ny);
    hipStream_t hs = (hipStream_t)s;
    const long long n = (long long)g->nx * (yhi - ylo);
    if (n == 0) {
        MGRT_HIP(hipMemsetAsync(out, 0, sizeof(double), hs));
        return MGRT_OK;
    }
    const unsigned gx = blocks_for(n, 256);
    // keep the number of partials bounded: deepen the z chunk as needed
    int zc = 16;
    while ((long long)gx * blocks_for(g->nz, zc) > (long long)kMaxPartials)
        zc *= 2;
    dim3 grid(gx, blocks_for(g->nz, zc));
    MGRT_REQUIRE((size_t)grid.x * grid.y <= (size_t)kMaxPartials, "%s: grid too large", who);
    double *parts = nullptr;
    if (mgrt_status st = partials_for(hs, &parts)) return st;
    k_norm_partials<R, MAX><<<grid, 256, 0, hs>>>(*g, x, parts, ylo, yhi, zc);
    MGRT_HIP(hipGetLastError());
    return fold<MAX>(parts, (int)(grid.x * grid.y), out, hs);
}

// mgrt_sumsq for callers that cannot allocate: the same two kernels, the block partials in the fixed
// device array above.  No hipMalloc, hipFree or other allocating call on this path.
mgrt_status sumsq_noalloc(const mgrt_grid *g, const float *x, double *out, int ylo, int yhi, mgrt_stream s)
{
    const char *who = "mgrt_sumsq_noalloc_f32";
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(g, who)) return st;
    MGRT_REQUIRE(x && out, "%s: null pointer", who);
    MGRT_REQUIRE(0 <= ylo && ylo <= yhi && yhi <= g->ny, "%s: rows [%d,%d) outside [0,%d)", who, ylo, yhi, g->ny);
    double *parts = runtime().static_partials;
    MGRT_REQUIRE(parts != nullptr, "%s: the scratch array was not set up", who);
    hipStream_t hs = (hipStream_t)s;
    const long long n = (long long)g->nx * (yhi - ylo);
    if (n == 0) {
        MGRT_HIP(hipMemsetAsync(out, 0, sizeof(double), hs));
        return MGRT_OK;
    }
    const unsigned gx = blocks_for(n, 256);
    // one partial per block: the rows times columns of one plane alone must fit the array, deeper z
    // chunks take care of the planes
    if ((long long)gx > (long long)kStaticPartials)
        return fail(MGRT_E_INVALID, "%s: %lld points per plane need %u block partials, the fixed scratch array holds %d "
                                    "(at most %lld points per plane); use mgrt_sumsq_f32",
                    who, n, gx, kStaticPartials, (long long)kStaticPartials * 256);
    int zc = 16;
    while ((long long)gx * blocks_for(g->nz, zc) > (long long)kStaticPartials)
        zc *= 2;
    dim3 grid(gx, blocks_for(g->nz, zc));
    k_norm_partials<float, false><<<grid, 256, 0, hs>>>(*g, x, parts, ylo, yhi, zc);
    MGRT_HIP(hipGetLastError());
    return fold<false>(parts, (int)(grid.x * grid.y), out, hs);
}

// Blocking: allocates the tree's intermediate arrays like the reference does (auxiliary.cpp:124).
// MODE 0: L2 norm (square, sum, sqrt); 1: L-infinity norm; 2: plain sum (Average, auxiliary.h:84-94).
template <class R, int MODE> mgrt_status reftree(const R *x, long long n, R *result, mgrt_stream s)
{
    constexpr bool MAX = MODE == 1;
    const char *who = MODE == 1 ? "mgrt_linfnorm_reftree" : MODE == 0 ? "mgrt_l2norm_reftree" : "mgrt_sum_reftree";
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(x && result && n > 0, "%s: bad argument", who);
    hipStream_t hs = (hipStream_t)s;
    R host = 0;
    if (n == 1) {
        // performReduction returns its input untouched when size == 1 (auxiliary.cpp:121); the L2
        // path has squared it first
        MGRT_HIP(hipMemcpyAsync(&host, x, sizeof(R), hipMemcpyDeviceToHost, hs));
        MGRT_HIP(hipStreamSynchronize(hs));
        *result = MODE == 0 ? (R)sqrt((double)(host * host)) : host;
        return MGRT_OK;
    }
    R *buf[2] = {nullptr, nullptr};
    const long long n1 = n / 4 > 1 ? n / 4 : 1;
    MGRT_HIP(hipMalloc((void **)&buf[0], sizeof(R) * (size_t)n1));
    if (hipMalloc((void **)&buf[1], sizeof(R) * (size_t)(n1 / 4 > 1 ? n1 / 4 : 1)) != hipSuccess) {
        (void)hipFree(buf[0]);
        return fail((mgrt_status)hipErrorOutOfMemory, "%s: out of device memory", who);
    }
    const R *in = x;
    long long size = n;
    int which = 0;
    bool first = true;
    while (size != 1) {
        const long long newn = size / 4 > 1 ? size / 4 : 1;
        if (first && MODE == 0)
            k_reftree_level<R, false, true><<<blocks_for(newn, 256), 256, 0, hs>>>(in, buf[which], size, newn);
        else
            k_reftree_level<R, MAX, false><<<blocks_for(newn, 256), 256, 0, hs>>>(in, buf[which], size, newn);
        first = false;
        in = buf[which];
        which ^= 1;
        size = newn;
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(&host, in, sizeof(R), hipMemcpyDeviceToHost, hs);
    if (e == hipSuccess)
        e = hipStreamSynchronize(hs);
    (void)hipFree(buf[0]);
    (void)hipFree(buf[1]);
    if (e != hipSuccess)
        return fail((mgrt_status)e, "%s: %s", who, hipGetErrorString(e));
    *result = MODE == 0 ? (R)sqrt((double)host) : host;
    return MGRT_OK;
}

} // namespace

// ---- C ABI ---------------------------------------------------------------------------------------
#define MGRT_DEFINE_KERNELS(R, S)                                                                              \
    extern "C" mgrt_status mgrt_smooth_pass##S(const mgrt_grid *g, R *u, const R *f, R omega, int odd, int zcount, \
                                               int ylo, int yhi, mgrt_stream s)                                \
    { return smooth_pass<R>(g, u, f, omega, odd, zcount, ylo, yhi, s); }                                       \
    extern "C" mgrt_status mgrt_smooth##S(const mgrt_grid *g, R *u, const R *f, R omega, int iters, int zcount, \
                                          mgrt_stream s)                                                       \
    { return smooth<R>(g, u, f, omega, iters, zcount, s); }                                                    \
    extern "C" mgrt_status mgrt_residual##S(const mgrt_grid *g, R *r, const R *u, const R *f, int zero_unwritten, \
                                            double *norm2, int ylo, int yhi, mgrt_stream s)                    \
    { return residual<R>(g, r, u, f, zero_unwritten, norm2, ylo, yhi, s); }                                    \
    extern "C" mgrt_status mgrt_restrict##S(const mgrt_grid *fine, const mgrt_grid *coarse, R *rc, const R *rf, \
                                            int cylo, int cyhi, mgrt_stream s)                                 \
    { return restrict_fw<R>(fine, coarse, rc, rf, cylo, cyhi, s); }                                            \
    extern "C" mgrt_status mgrt_correct##S(const mgrt_grid *fine, const mgrt_grid *coarse, R *out, const R *u,  \
                                           const R *e, int ylo, int yhi, mgrt_stream s)                        \
    { return interp<R, 0>(fine, coarse, out, u, e, ylo, yhi, s); }                                             \
    extern "C" mgrt_status mgrt_prolong##S(const mgrt_grid *fine, const mgrt_grid *coarse, R *out, const R *c,  \
                                           int ylo, int yhi, mgrt_stream s)                                    \
    { return interp<R, 1>(fine, coarse, out, nullptr, c, ylo, yhi, s); }                                       \
    extern "C" mgrt_status mgrt_zero_outside##S(const mgrt_grid *g, R *u, mgrt_stream s)                       \
    { return zero_outside<R>(g, u, s); }                                                                       \
    extern "C" mgrt_status mgrt_diff##S(const mgrt_grid *g, R *out, const R *a, const R *b, mgrt_stream s)     \
    { return diff<R>(g, out, a, b, s); }                                                                       \
    extern "C" mgrt_status mgrt_scale##S(const mgrt_grid *g, R *out, const R *in, float m, mgrt_stream s)      \
    { return scale<R>(g, out, in, m, s); }                                                                     \
    extern "C" mgrt_status mgrt_sumsq##S(const mgrt_grid *g, const R *x, double *out, int ylo, int yhi,        \
                                         mgrt_stream s)                                                        \
    { return norm<R, false>(g, x, out, ylo, yhi, s); }                                                         \
    extern "C" mgrt_status mgrt_maxabs##S(const mgrt_grid *g, const R *x, double *out, int ylo, int yhi,       \
                                          mgrt_stream s)                                                       \
    { return norm<R, true>(g, x, out, ylo, yhi, s); }                                                          \
    extern "C" mgrt_status mgrt_l2norm_reftree##S(const R *x, long long n, R *result, mgrt_stream s)           \
    { return reftree<R, 0>(x, n, result, s); }                                                                 \
    extern "C" mgrt_status mgrt_linfnorm_reftree##S(const R *x, long long n, R *result, mgrt_stream s)         \
    { return reftree<R, 1>(x, n, result, s); }                                                                 \
    extern "C" mgrt_status mgrt_sum_reftree##S(const R *x, long long n, R *result, mgrt_stream s)              \
    { return reftree<R, 2>(x, n, result, s); }

MGRT_DEFINE_KERNELS(float, _f32)
MGRT_DEFINE_KERNELS(double, _f64)

extern "C" mgrt_status mgrt_sumsq_noalloc_f32(const mgrt_grid *g, const float *x, double *out, int ylo, int yhi,
                                              mgrt_stream s)
{
    return sumsq_noalloc(g, x, out, ylo, yhi, s);
}
