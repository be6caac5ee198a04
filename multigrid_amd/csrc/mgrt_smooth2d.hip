// mgrt_smooth2d.hip -- the 2D smoother as register-resident tiles that perform whole red-black
// iterations per sweep: out = S^iters(in), every element of `in` and `f` read once (plus halo
// re-reads served by the caches), `out` written once.
//
// The reference's 2D pass is a true red-black pass (colour (x+y)%2, mg_0.cl:52-98): a point of one
// colour reads only points of the other.  One wave owns a tile, lane = x, and holds the tile's RT rows
// of `u` and `f` in registers: y neighbours are the lane's own registers, x neighbours come from the
// adjacent lanes by DPP, so a colour pass touches neither LDS nor a barrier, and because a pass
// leaves the other colour alone the in-place register update sees exactly the values of the serial
// schedule.  Of the 64 x RT points loaded, the (64 - 2H) x (RT - 2H) interior is stored, H = 2*iters:
// halo rings go stale one ring per pass from the outside in and never reach the interior.  Every
// point is computed by the reference's expression, so the result equals 2*iters launches of the
// single-colour kernel bit for bit.  Analytic box domains (and y-slabs of them) only.
//
// Short tiles win: a wave's work is one dependent chain per pass, so the sweep is bound by how many
// waves run side by side rather than by the rows each of them re-reads from the caches.
#include "mgrt_internal.h"
#include "mgrt_device.h"
#include <type_traits>

using namespace mgrt;

namespace {

constexpr int kWavesPerBlock = 4;

// waves per SIMD the register file admits with `values` field values per lane and some room to work in
template <class R, int values> constexpr int cols_waves()
{
    constexpr int need = values * (int)(sizeof(R) / 4) + 48, w = 512 / need;
    return w < 1 ? 1 : w > 8 ? 8 : w;
}

// Scalars that are read inside the pass loop go through this first: the compiler must treat them as
// changed in every pass, so it derives the per-row lane masks where they are used instead of
// computing all of them ahead of the loop, which would take more scalar registers than there are.
__device__ __forceinline__ int per_pass(int v)
{
    asm volatile("" : "+s"(v));
    return v;
}

// Where the tile's input comes from: a field, zeros, or a field plus the interpolated coarse-grid
// correction (the input of the post-smoother).
enum Input { FIELD, ZEROS, CORRECTED };

// RES = 1: also r = residual of `out`, which takes one more halo ring (the residual of a point reads
// the final values of its four neighbours).
// RES = 2: the residual stays in registers and `res` receives its full-weighting restriction on the
// coarse grid `cg` (reduction_kernel, mg_0.cl:139-157): one more ring again, since a coarse point
// weighs the residuals around it.  Whole grids with odd extents: the tile interiors then start on
// even rows and columns, so every coarse point lies in exactly one of them, and a coarse point
// injects exactly when it lies on a face.

template <class R, int ITERS, int RT, bool OMEGA1, Input IN, int RES>
__global__ __launch_bounds__(64 * kWavesPerBlock, (cols_waves<R, 2 * RT + (IN == CORRECTED ? RT + 8 : 0) + (RES == 2 ? 12 : 0)>())) void k_smooth2d_cols(
    mgrt_grid g, mgrt_grid cg, const R *__restrict__ in, R *__restrict__ out, const R *__restrict__ f,
    const R *__restrict__ e, R *__restrict__ res, R omega, int ntx, int ntiles)
{
    constexpr int H = 2 * ITERS + RES, OX = 64 - 2 * H, OY = RT - 2 * H;
    static_assert(OY > 0, "tile too short for its halo");
    static_assert(RES != 2 || (H % 2 == 0 && OY % 2 == 0), "tile interiors must start on even rows");
    const int lane = threadIdx.x & 63;
    // the wave index through readfirstlane: everything derived from the tile stays in scalar registers
    const int tile = xcd_contiguous(blockIdx.x, gridDim.x) * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (tile >= ntiles)
        return; // whole waves leave; the kernel has no barrier
    const int x = (tile % ntx) * OX - H + lane, y0 = (tile / ntx) * OY - H;
    const int nx = g.nx, ny = g.ny;
    const long long pitch = g.pitch;
    // rows of the tile that lie in the grid: first <= j <= last
    const int first = max(0, -y0), last = min(RT - 1, ny - 1 - y0);

    // Cells outside the grid are never relaxed and never read by a cell that is, so their loads only
    // have to be safe: the coordinates are clamped into the grid instead of predicated.
    const int xc = min(max(x, 0), nx - 1);
    const long long at = xc + pitch * (long long)(y0 + first);
    const R *pf = f + at;
    const R *pu = (IN == ZEROS ? f : in) + at;
    R u[RT], b[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        u[j] = IN == ZEROS ? (R)0 : *pu;
        b[j] = *pf;
        const long long step = j >= first && j < last ? pitch : 0; // next row, clamped
        pf += step;
        pu += step;
    }

    // classes in the order of the 2D branch of cell_class: the x faces win the corners
    const bool x_inner = x > 0 && x < nx - 1;
    const bool x_dirichlet = (x == 0 && g.face[0] == MGRT_CELL_DIRICHLET) || (x == nx - 1 && g.face[1] == MGRT_CELL_DIRICHLET);
    // a point (lane, row j) belongs to the colour of pass `odd` when (x + y + y_offset + odd) is even
    const int yg0 = y0 + g.y0; // global row of j = 0
    const bool even0 = ((x + yg0) & 1) == 0;
    // rows strictly inside the global grid: jlo <= j < jhi; the face rows are jlo - 1 and jhi
    const int jlo = 1 - yg0, jhi = g.gny - 1 - yg0;
    // -1 where the face is not Dirichlet: no row matches
    const int jdlo = g.face[2] == MGRT_CELL_DIRICHLET ? jlo - 1 : -1, jdhi = g.face[3] == MGRT_CELL_DIRICHLET ? jhi : -1;

    if constexpr (IN == CORRECTED) {
        // u += 4 * P(e) with the expressions of the interpolation kernel (bilinear weights, every
        // product formed, zero weights included; injection on Dirichlet and Neumann cells).  The
        // tile's fine rows read RT/2 + 2 coarse rows, loaded once per lane at x/2 and x/2 + 1.
        constexpr int NC = RT / 2 + 2;
        const int cbase = (yg0 >> 1) - cg.y0; // coarse row under j = 0 (floor division)
        const int ci = xc >> 1, ci1 = min(ci + 1, cg.nx - 1);
        R c0[NC], c1[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const long long row = cg.pitch * (long long)min(max(cbase + k, 0), cg.ny - 1);
            c0[k] = e[ci + row];
            c1[k] = e[ci1 + row];
        }
        const bool x_face = x == 0 || x == nx - 1;
        const bool x_border = (x == 0 && (g.face[0] == MGRT_CELL_DIRICHLET || g.face[0] == MGRT_CELL_NEUMANN)) ||
                              (x == nx - 1 && (g.face[1] == MGRT_CELL_DIRICHLET || g.face[1] == MGRT_CELL_NEUMANN));
        const bool lo_border = g.face[2] == MGRT_CELL_DIRICHLET || g.face[2] == MGRT_CELL_NEUMANN;
        const bool hi_border = g.face[3] == MGRT_CELL_DIRICHLET || g.face[3] == MGRT_CELL_NEUMANN;
        const R wu = 0.5 * (xc % 2);
        auto correct = [&](auto parity) {
            constexpr int P = decltype(parity)::value; // parity of yg0
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                const int k = (j + P) >> 1; // a constant once the loop is unrolled
                const R ww = 0.5 * ((j + P) % 2);
                const bool border = x_face ? x_border : (j == jlo - 1 && lo_border) || (j == jhi && hi_border);
                const R n0 = (1.0 - wu) * (1.0 - ww), n1 = (1.0 - ww) * wu, n2 = (1.0 - wu) * ww, n3 = ww * wu;
                R interp = c0[k] * n0 + c1[k] * n1 + c0[k + 1] * n2 + c1[k + 1] * n3;
                if constexpr (sizeof(R) == 4)
                    asm("" : "+v"(interp)); // formed for every row: a branch on `border` per row costs more than the sum
                const R val = border ? c0[k] : interp;
                u[j] = u[j] + val * 4;
            }
        };
        if (yg0 & 1)
            correct(std::integral_constant<int, 1>{});
        else
            correct(std::integral_constant<int, 0>{});
    }

    // Tiles whose rows 1 .. RT-2 all lie strictly inside the grid (nearly all of them) need no
    // per-row class: four lane masks serve every row of every pass.
    const bool rows_inner = jlo <= 1 && jhi >= RT - 1 && first == 0 && last == RT - 1;
    auto passes = [&](auto inner_rows) {
#pragma unroll 1
        for (int pass = 0; pass < 2 * ITERS; ++pass) {
            const bool mine_even = even0 == ((pass & 1) == 0); // colour membership of the even rows j
            const int lo = per_pass(max(jlo, first)), hi = per_pass(min(jhi, last + 1));
            const int dlo = per_pass(jdlo >= first ? jdlo : -1), dhi = per_pass(jdhi <= last ? jdhi : -1);
            const int fst = per_pass(first), lst = per_pass(last);
#pragma unroll
            for (int j = 1; j < RT - 1; ++j) {
                // every pass makes one more ring of the tile's halo worthless: pass number p (from 0) is only owed
                // rows p + 1 .. RT - 2 - p, the others cannot reach a row that is stored (a uniform branch a row)
                if (j < 1 + pass || j > RT - 2 - pass)
                    continue;
                const bool mine = (j & 1) ? !mine_even : mine_even;
                bool relax, fixed;
                if constexpr (decltype(inner_rows)::value) {
                    relax = mine && x_inner;
                    fixed = mine && x_dirichlet;
                } else {
                    relax = mine && x_inner && j >= lo && j < hi;
                    fixed = mine && ((x_dirichlet && j >= fst && j <= lst) || (x_inner && (j == dlo || j == dhi)));
                }
                // mg_0.cl:38: 0.25 is a double literal, the assignment rounds to R.  A quarter of a float is exact in double, so
                // for 4-byte elements this is the exact quotient rounded once, which is what the single-precision product is
                // for every input (tools/div6_exhaustive.c, column Q; the compiler makes that product of it by itself).
                const R val = 0.25 * (from_upper_lane(u[j]) + from_lower_lane(u[j]) + u[j + 1] + u[j - 1] - b[j]);
                // for omega == 1 the double expression reduces exactly to val + 0*u evaluated in R
                // (the product keeps the sign of a zero result and a non-finite u as the reference does)
                R gs;
                if constexpr (OMEGA1)
                    gs = sizeof(R) == 4 ? fma_r((R)0, u[j], val) // one rounding, of the sum: the product is exact
                                        : val + (R)0 * u[j];     // (8-byte elements: the fused form makes the variants spill)
                else
                    gs = val * omega + (1.0 - omega) * u[j];
                // Keep the update unconditional (4-byte elements; with 8-byte ones the registers it takes spill): left
                // alone, the compiler sinks it under branches on relax, and the exec-mask blocks that result cost more
                // than the updates they skip (4097^2: 88 -> 75 us a sweep with residual and restriction).
                if constexpr (sizeof(R) == 4)
                    asm("" : "+v"(gs));
                u[j] = relax ? gs : fixed ? b[j] : u[j];
            }
        }
    };
    if (rows_inner)
        passes(std::true_type{});
    else
        passes(std::false_type{});

    const bool stores = lane >= H && lane < 64 - H && x < nx;
    if (stores) {
        R *po = out + x + pitch * (long long)(y0 + H); // y0 + H >= 0 for every tile
#pragma unroll
        for (int j = H; j < RT - H; ++j) {
            if (j <= last)
                *po = u[j];
            po += pitch;
        }
    }
    if constexpr (RES != 0) {
        // the residual kernel's expressions; cells that are neither inner nor Dirichlet get 0
        const int lo = max(jlo, first), hi = min(jhi, last + 1);
        const int dlo = jdlo >= first ? jdlo : -1, dhi = jdhi <= last ? jdhi : -1;
        constexpr int JA = RES == 2 ? H - 1 : H, JB = RES == 2 ? RT - H + 1 : RT - H;
        R *pr = res + x + pitch * (long long)(y0 + H);
#pragma unroll
        for (int j = JA; j < JB; ++j) {
            const R xp = from_upper_lane(u[j]), xm = from_lower_lane(u[j]); // all lanes take part
            const bool inner = x_inner && j >= lo && j < hi;
            const bool dirichlet = (x_dirichlet && j >= first && j <= last) || (x_inner && (j == dlo || j == dhi));
            R vin = b[j] - (xp + xm + u[j + 1] + u[j - 1] - 4 * u[j]);
            if constexpr (sizeof(R) == 4)
                asm("" : "+v"(vin)); // formed for every row, like the update
            const R v = inner ? vin : dirichlet ? b[j] - u[j] : (R)0;
            if constexpr (RES == 1) {
                if (stores && j <= last)
                    *pr = v;
                pr += pitch;
            } else {
                b[j] = v; // the right-hand side has done its part
            }
        }
        if constexpr (RES == 2) {
            const bool x_face = x == 0 || x == nx - 1;
            const bool coarse_lane = stores && (x & 1) == 0;
            R *pc = res + (x >> 1);
#pragma unroll
            for (int j = H; j < RT - H; j += 2) { // y0 + j is even: y0 + H is
                const R am = from_lower_lane(b[j - 1]), ap = from_upper_lane(b[j - 1]);
                const R bm = from_lower_lane(b[j + 1]), bp = from_upper_lane(b[j + 1]);
                const R cm = from_lower_lane(b[j]), cp = from_upper_lane(b[j]);
                // mg_0.cl:152-156 with c = the fine point under the coarse one
                // (4-byte elements: a float times a power of two is exact in double, so fused multiply-adds round what the
                // reference's separate multiplies and adds round, in three double-precision instructions instead of five)
                R w;
                if constexpr (sizeof(R) == 4) {
                    double t = 1.0 / 16.0 * (double)(am + ap + bm + bp);
                    t = __builtin_fma((double)(b[j - 1] + cp + cm + b[j + 1]), 1.0 / 8.0, t);
                    w = (R)__builtin_fma((double)b[j], 1.0 / 4.0, t);
                    asm("" : "+v"(w));
                } else {
                    w = 1.0 / 16.0 * (am + ap + bm + bp) + 1.0 / 8.0 * (b[j - 1] + cp + cm + b[j + 1]) + 1.0 / 4.0 * (b[j]);
                }
                const bool face = x_face || j == jlo - 1 || j == jhi;
                if (coarse_lane && j <= last)
                    pc[cg.pitch * (long long)((y0 + j) >> 1)] = face ? b[j] : w;
            }
        }
    }
}

template <class R, int ITERS, int RT, Input IN, int RES>
mgrt_status launch2d_cols(const mgrt_grid *g, const mgrt_grid *cg, R *out, R *res, const R *in, const R *e, const R *f,
                          R omega, hipStream_t hs)
{
    constexpr int H = 2 * ITERS + RES;
    const int ntx = (g->nx + 64 - 2 * H - 1) / (64 - 2 * H), nty = (g->ny + RT - 2 * H - 1) / (RT - 2 * H);
    const int ntiles = ntx * nty, blocks = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid(blocks), block(64 * kWavesPerBlock);
    if (omega == (R)1)
        k_smooth2d_cols<R, ITERS, RT, true, IN, RES><<<grid, block, 0, hs>>>(*g, cg ? *cg : *g, in, out, f, e, res, omega, ntx, ntiles);
    else
        k_smooth2d_cols<R, ITERS, RT, false, IN, RES><<<grid, block, 0, hs>>>(*g, cg ? *cg : *g, in, out, f, e, res, omega, ntx, ntiles);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

// Tile heights measured on V(2,2) cycles: 20 rows from 2049^2 up; below, 12 rows, or 14 with the
// wider halo of the residual variant; 24 and 16 with the restriction.
template <class R, int ITERS, Input IN, int RES>
mgrt_status sweep2d(const mgrt_grid *g, const mgrt_grid *cg, R *out, R *res, const R *in, const R *e, const R *f, R omega,
                    hipStream_t hs)
{
    const bool large = (long long)g->nx * g->ny >= 1500LL * 1500LL;
    if constexpr (RES == 2) {
        if (!large) return launch2d_cols<R, ITERS, 16, IN, RES>(g, cg, out, res, in, e, f, omega, hs);
        return launch2d_cols<R, ITERS, 24, IN, RES>(g, cg, out, res, in, e, f, omega, hs);
    } else if constexpr (RES == 1) {
        if (!large) return launch2d_cols<R, ITERS, 14, IN, RES>(g, cg, out, res, in, e, f, omega, hs);
        return launch2d_cols<R, ITERS, 20, IN, RES>(g, cg, out, res, in, e, f, omega, hs);
    } else {
        if (!large) return launch2d_cols<R, ITERS, 12, IN, RES>(g, cg, out, res, in, e, f, omega, hs);
        return launch2d_cols<R, ITERS, 20, IN, RES>(g, cg, out, res, in, e, f, omega, hs);
    }
}

template <class R>
mgrt_status sweep2d_any(const mgrt_grid *g, const mgrt_grid *cg, R *out, R *res, const R *in, const R *e, const R *f,
                        R omega, int iters, hipStream_t hs)
{
#define MGRT_2D(IN, RES) \
    (iters == 1 ? sweep2d<R, 1, IN, RES>(g, cg, out, res, in, e, f, omega, hs) : sweep2d<R, 2, IN, RES>(g, cg, out, res, in, e, f, omega, hs))
    if (e)
        return MGRT_2D(CORRECTED, 0);
    if (res && cg) // res lives on the coarse grid
        return in ? MGRT_2D(FIELD, 2) : MGRT_2D(ZEROS, 2);
    if (res)
        return in ? MGRT_2D(FIELD, 1) : MGRT_2D(ZEROS, 1);
    return in ? MGRT_2D(FIELD, 0) : MGRT_2D(ZEROS, 0);
#undef MGRT_2D
}

} // namespace

namespace mgrt {

// out = S^iters(in); with res != nullptr also res = residual of out, or, when cg is given without e,
// its restriction to the coarse grid cg; with e != nullptr (and cg) the input is in + 4*P(e) instead
// of in.  in == nullptr: a field of zeros.
mgrt_status smooth_sweep_2d(const mgrt_grid *g, const mgrt_grid *cg, float *out, float *res, const float *in,
                            const float *e, const float *f, float omega, int iters, hipStream_t hs)
{
    return sweep2d_any<float>(g, cg, out, res, in, e, f, omega, iters, hs);
}
mgrt_status smooth_sweep_2d(const mgrt_grid *g, const mgrt_grid *cg, double *out, double *res, const double *in,
                            const double *e, const double *f, double omega, int iters, hipStream_t hs)
{
    return sweep2d_any<double>(g, cg, out, res, in, e, f, omega, iters, hs);
}

} // namespace mgrt
