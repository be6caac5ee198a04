// mgrt_smooth2d.hip -- the 2D smoother as tiles in LDS that perform whole red-black iterations per
// sweep: out = S^iters(in), every element of `in` and `f` read once, `out` written once.
//
// The reference's 2D pass is a true red-black pass (colour (x+y)%2, mg_0.cl:52-98): a point of one
// colour reads only points of the other.  A workgroup loads a tile plus a halo of 2*iters rings into
// LDS, runs the 2*iters colour passes there in place with a barrier between passes, and stores the
// tile's interior.  Halo rings go stale one ring per pass from the outside in and never reach the
// interior.  Every point is computed by the reference's expression from the values the serial
// schedule would feed it, so the result equals 2*iters launches of the single-colour kernel bit
// for bit.  Analytic box domains (and y-slabs of them) only.
#include "mgrt_internal.h"

using namespace mgrt;

namespace {

constexpr int kTX = 64, kTY = 32, kThreads = 256;

__device__ __forceinline__ int class2d(const mgrt_grid &g, int x, int y)
{
    // same order as the 2D branch of cell_class
    const int yg = y + g.y0;
    if (x == 0) return g.face[0];
    if (x == g.nx - 1) return g.face[1];
    if (yg == 0) return g.face[2];
    if (yg == g.gny - 1) return g.face[3];
    return MGRT_CELL_INNER;
}

template <class R, int ITERS>
__global__ __launch_bounds__(kThreads) void k_smooth2d_fused(mgrt_grid g, const R *__restrict__ in, R *__restrict__ out,
                                                            const R *__restrict__ f, R omega, int ntx)
{
    constexpr int H = 2 * ITERS, W = kTX + 2 * H, HT = kTY + 2 * H, P = W + 1; // P: padded row pitch
    __shared__ R su[HT][P], sf[HT][P];
    const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int x0 = (tile % ntx) * kTX - H, y0 = (tile / ntx) * kTY - H;
    const long long p = g.pitch;

    // all loads of the tile are issued before the first one is consumed
    constexpr int NL = (W * HT + kThreads - 1) / kThreads;
    R ru[NL], rf[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = threadIdx.x + k * kThreads;
        const int lx = i % W, ly = i / W, x = x0 + lx, y = y0 + ly;
        const bool live = i < W * HT && x >= 0 && x < g.nx && y >= 0 && y < g.ny;
        const long long b = x + p * y;
        ru[k] = live && in ? in[b] : (R)0; // in == nullptr stands for a field of zeros
        rf[k] = live ? f[b] : (R)0;
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = threadIdx.x + k * kThreads;
        if (i < W * HT) {
            su[i / W][i % W] = ru[k];
            sf[i / W][i % W] = rf[k];
        }
    }
    __syncthreads();

    constexpr int HW = (W - 2 + 1) / 2; // points of one colour per row, rows 1 .. HT-2, columns 1 .. W-2
#pragma unroll
    for (int pass = 0; pass < 2 * ITERS; ++pass) {
        const int odd = pass & 1;
        for (int i = threadIdx.x; i < HW * (HT - 2); i += kThreads) {
            const int ly = 1 + i / HW;
            const int y = y0 + ly;
            // colour (x + y + y_offset + odd) % 2 == 0 as in k_smooth2d_pass: x = 2*k + ((odd + yg) & 1)
            const int par = (odd + y + g.y0 + x0 + 1) & 1; // parity of lx - 1 that puts x on the colour
            const int lx = 1 + 2 * (i % HW) + par;
            if (lx > W - 2)
                continue;
            const int x = x0 + lx;
            if (x < 0 || x >= g.nx || y < 0 || y >= g.ny)
                continue;
            const int t = class2d(g, x, y);
            if (t == MGRT_CELL_INNER) {
                const R val = 0.25 * (su[ly][lx + 1] + su[ly][lx - 1] + su[ly + 1][lx] + su[ly - 1][lx] - sf[ly][lx]);
                su[ly][lx] = val * omega + (1.0 - omega) * su[ly][lx];
            } else if (t == MGRT_CELL_DIRICHLET) {
                su[ly][lx] = sf[ly][lx];
            }
        }
        __syncthreads();
    }

    for (int i = threadIdx.x; i < kTX * kTY; i += kThreads) {
        const int lx = H + i % kTX, ly = H + i / kTX, x = x0 + lx, y = y0 + ly;
        if (x < g.nx && y < g.ny)
            out[x + p * y] = su[ly][lx];
    }
}

template <class R, int ITERS>
mgrt_status launch2d(const mgrt_grid *g, R *out, const R *in, const R *f, R omega, hipStream_t hs)
{
    const int ntx = (g->nx + kTX - 1) / kTX, nty = (g->ny + kTY - 1) / kTY;
    k_smooth2d_fused<R, ITERS><<<dim3(ntx * nty), kThreads, 0, hs>>>(*g, in, out, f, omega, ntx);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

} // namespace

namespace mgrt {

mgrt_status smooth_sweep_2d(const mgrt_grid *g, float *out, const float *in, const float *f, float omega, int iters,
                            hipStream_t hs)
{
    return iters == 1 ? launch2d<float, 1>(g, out, in, f, omega, hs) : launch2d<float, 2>(g, out, in, f, omega, hs);
}
mgrt_status smooth_sweep_2d(const mgrt_grid *g, double *out, const double *in, const double *f, double omega, int iters,
                            hipStream_t hs)
{
    return iters == 1 ? launch2d<double, 1>(g, out, in, f, omega, hs) : launch2d<double, 2>(g, out, in, f, omega, hs);
}

} // namespace mgrt
