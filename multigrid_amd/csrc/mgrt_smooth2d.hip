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

// waves per SIMD the register file admits with 2*RT values per lane and some room to work in
template <class R, int RT> constexpr int cols_waves()
{
    constexpr int need = 2 * RT * (int)(sizeof(R) / 4) + 40, w = 512 / need;
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

template <class R, int ITERS, int RT, bool OMEGA1, bool ZERO_IN>
__global__ __launch_bounds__(64 * kWavesPerBlock, (cols_waves<R, RT>())) void k_smooth2d_cols(
    mgrt_grid g, const R *__restrict__ in, R *__restrict__ out, const R *__restrict__ f, R omega, int ntx, int ntiles)
{
    constexpr int H = 2 * ITERS, OX = 64 - 2 * H, OY = RT - 2 * H;
    static_assert(OY > 0, "tile too short for its halo");
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
    const long long at = min(max(x, 0), nx - 1) + pitch * (long long)(y0 + first);
    const R *pf = f + at;
    const R *pu = (ZERO_IN ? f : in) + at;
    R u[RT], b[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        u[j] = ZERO_IN ? (R)0 : *pu;
        b[j] = *pf;
        const long long step = j >= first && j < last ? pitch : 0; // next row, clamped
        pf += step;
        pu += step;
    }

    // classes in the order of the 2D branch of cell_class: the x faces win the corners
    const bool x_inner = x > 0 && x < nx - 1;
    const bool x_dirichlet = (x == 0 && g.face[0] == MGRT_CELL_DIRICHLET) || (x == nx - 1 && g.face[1] == MGRT_CELL_DIRICHLET);
    // a point (lane, row j) belongs to the colour of pass `odd` when (x + y + y_offset + odd) is even
    const bool even0 = ((x + y0 + g.y0) & 1) == 0;
    // rows strictly inside the global grid: jlo <= j < jhi; the face rows are jlo - 1 and jhi
    const int jlo = 1 - (y0 + g.y0), jhi = g.gny - 1 - (y0 + g.y0);
    // -1 where the face is not Dirichlet: no row matches
    const int jdlo = g.face[2] == MGRT_CELL_DIRICHLET ? jlo - 1 : -1, jdhi = g.face[3] == MGRT_CELL_DIRICHLET ? jhi : -1;

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
                const bool mine = (j & 1) ? !mine_even : mine_even;
                bool relax, fixed;
                if constexpr (decltype(inner_rows)::value) {
                    relax = mine && x_inner;
                    fixed = mine && x_dirichlet;
                } else {
                    relax = mine && x_inner && j >= lo && j < hi;
                    fixed = mine && ((x_dirichlet && j >= fst && j <= lst) || (x_inner && (j == dlo || j == dhi)));
                }
                const R val = 0.25 * (from_upper_lane(u[j]) + from_lower_lane(u[j]) + u[j + 1] + u[j - 1] - b[j]);
                R gs;
                if constexpr (OMEGA1)
                    gs = val;
                else
                    gs = val * omega + (1.0 - omega) * u[j];
                u[j] = relax ? gs : fixed ? b[j] : u[j];
            }
        }
    };
    if (rows_inner)
        passes(std::true_type{});
    else
        passes(std::false_type{});

    if (lane >= H && lane < 64 - H && x < nx) {
        R *po = out + x + pitch * (long long)(y0 + H); // y0 + H >= 0 for every tile
#pragma unroll
        for (int j = H; j < RT - H; ++j) {
            if (j <= last)
                *po = u[j];
            po += pitch;
        }
    }
}

template <class R, int ITERS, int RT>
mgrt_status launch2d_cols(const mgrt_grid *g, R *out, const R *in, const R *f, R omega, hipStream_t hs)
{
    constexpr int H = 2 * ITERS;
    const int ntx = (g->nx + 64 - 2 * H - 1) / (64 - 2 * H), nty = (g->ny + RT - 2 * H - 1) / (RT - 2 * H);
    const int ntiles = ntx * nty, blocks = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid(blocks), block(64 * kWavesPerBlock);
    const bool one = omega == (R)1;
    if (one && !in)
        k_smooth2d_cols<R, ITERS, RT, true, true><<<grid, block, 0, hs>>>(*g, in, out, f, omega, ntx, ntiles);
    else if (one)
        k_smooth2d_cols<R, ITERS, RT, true, false><<<grid, block, 0, hs>>>(*g, in, out, f, omega, ntx, ntiles);
    else if (!in)
        k_smooth2d_cols<R, ITERS, RT, false, true><<<grid, block, 0, hs>>>(*g, in, out, f, omega, ntx, ntiles);
    else
        k_smooth2d_cols<R, ITERS, RT, false, false><<<grid, block, 0, hs>>>(*g, in, out, f, omega, ntx, ntiles);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R, int ITERS>
mgrt_status sweep2d(const mgrt_grid *g, R *out, const R *in, const R *f, R omega, hipStream_t hs)
{
    // measured on V(2,2) cycles: 16 rows from 2049^2 up, 12 rows below
    if ((long long)g->nx * g->ny >= 1500LL * 1500LL)
        return launch2d_cols<R, ITERS, 16>(g, out, in, f, omega, hs);
    return launch2d_cols<R, ITERS, 12>(g, out, in, f, omega, hs);
}

} // namespace

namespace mgrt {

mgrt_status smooth_sweep_2d(const mgrt_grid *g, float *out, const float *in, const float *f, float omega, int iters,
                            hipStream_t hs)
{
    return iters == 1 ? sweep2d<float, 1>(g, out, in, f, omega, hs) : sweep2d<float, 2>(g, out, in, f, omega, hs);
}
mgrt_status smooth_sweep_2d(const mgrt_grid *g, double *out, const double *in, const double *f, double omega, int iters,
                            hipStream_t hs)
{
    return iters == 1 ? sweep2d<double, 1>(g, out, in, f, omega, hs) : sweep2d<double, 2>(g, out, in, f, omega, hs);
}

} // namespace mgrt
