// mgrt_smooth_pair.hip -- the two-stage sweep of the latency-bound 3D levels, two planes a time step.
//
// Same schedule as the two-stage form of k_smooth3d_fused (mgrt_smooth_fused.hip): `out = S^2(in)`, a
// block of 2*NW wavefronts owns a tile of 64 x 2*NW columns with a recomputed halo of four rings, a
// wavefront a strip of two rows (one row pair, split by colour), the first NW wavefronts perform the
// first red-black iteration and hand every finished plane and its right-hand side to the other NW
// through LDS, which perform the second and store.  On these levels a time step costs what the
// wavefronts of a SIMD issue in it, and a good third of that is not arithmetic: the barrier, the
// counted waits, the LDS hand-over, the address bookkeeping.  Here a step takes two planes, so all of
// that is shared by twice the updates.
//
// Why two planes can share a step.  Colour pass d of a stage (d = 1, 2) works on the plane pair two
// pairs below the one pass d-1 works on: with planes 2k, 2k+1 entering at step k, pass 1 updates planes
// 2k-2, 2k-1 and pass 2 planes 2k-4, 2k-3, the lower plane of a pair first.  A point still sees exactly
// what the serial NDRange schedule of the reference feeds it (mg_1.cl:29-43,56-104: colour ignores z, a
// column is a recurrence in ascending z): its in-plane neighbours, of the other colour, were finished
// by the pass before in the previous step (and published by their strips at the start of this one), its
// own column's z-1 value is new (the previous step, or the lower plane of this pair), its z+1 value old
// (the pass has not reached it).  The passes of a step touch different planes or colours and need
// nothing from each other, so one barrier a step is enough.  Every point is computed by the
// reference's expression in the reference's order: bit for bit the single-colour kernel.
//
// Measured against the one-plane form: 257^3 107 -> 95 us, 129^3 50 -> 46 us a sweep, nothing below that
// (the instructions a plane costs fall by a tenth: 57 and 51 against 61 for the two stages); 4-byte elements
// only: with 8-byte ones the rings of two plane pairs do not fit the registers 12 or 16 wavefronts leave.
// A variant that applies the coarse-grid correction on the way in was written, gave the same bits and the
// same time as the one-plane form (129^3: 54.7 against 55.4 us) and was dropped.
#include "mgrt_internal.h"
#include "mgrt_device.h"
#include <type_traits>

using namespace mgrt;

namespace {

constexpr int kWave = 64;

template <class R, int NW, int PD, bool OMEGA1>
__global__ __launch_bounds__(NW * 2 * kWave) void k_smooth3d_fused_pair(mgrt_grid g, const R *__restrict__ in,
                                                                      R *__restrict__ out, const R *__restrict__ f, R omega,
                                                                      int zcount, int ntx)
{
    constexpr int H = 4;                 // halo rings: two colour passes a stage, two stages
    constexpr int ND = 8, UN = 4;        // ring period in planes, time steps unrolled (8 planes)
    constexpr int LAG = 3;               // steps between a plane pair leaving the first stage and entering the second
    constexpr int OX = kWave - 2 * H, OY = 2 * NW - 2 * H;
    static_assert(UN % PD == 0 && PD >= 1, "prefetch depth must divide the unrolled loop");
    // edge rows of the strips for the two passes and the two planes of a pair: [step parity][pass][plane][strip][lane]
    extern __shared__ __align__(16) unsigned char smem_raw[];
    R(*bnd)[2][2][2 * NW][kWave] = reinterpret_cast<R(*)[2][2][2 * NW][kWave]>(smem_raw);
    // hand-over between the stages: [step parity][strip][plane][u | f][colour][lane]
    R(*hand)[NW][2][2][2][kWave] = reinterpret_cast<R(*)[NW][2][2][2][kWave]>(smem_raw + sizeof(R) * 2 * 2 * 2 * 2 * NW * kWave);

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
    // Which strip a wavefront takes.  The passes a strip sits out (below) leave the edge strips with less to do, and a
    // time step lasts as long as the busiest SIMD; wavefront i of a block runs on SIMD i % 4, so the strips are dealt
    // such that every SIMD gets the same number of colour passes a step (NW = 8: 6 of 8, NW = 6: 4 of 6).  Four bits a
    // wavefront; the first NW wavefronts are the first stage.  A speed matter only: any bijection gives the same values.
    static_assert(NW == 6 || NW == 8, "strip tables");
    constexpr unsigned long long kStrips = NW == 8 ? 0x6543'1702'6427'5310ull  // 0 1 3 5 7 2 4 6 | 2 0 7 1 3 4 5 6
                                                   : 0x0000'3250'4142'5031ull; // 1 3 0 5 2 4 | 1 4 0 5 2 3
    const int stage = wave / NW, w = (int)((kStrips >> (4 * wave)) & 15u);
    const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx = tile % ntx, ty = tile / ntx;
    const int x = tx * OX - H + lane;
    const int y0 = ty * OY - H + 2 * w; // first row of the strip
    const bool xin = x >= 0 && x < g.nx;
    const int s0 = (x + y0 + g.y0) & 1; // row y0 + s0 holds this lane's colour-0 point

    constexpr unsigned kDrop = 0x80000000u;
    const unsigned plane_bytes = (unsigned)(g.pitch * (long long)g.ny * (long long)sizeof(R));
    const R *inb = in ? in : f;
    const unsigned in_bytes = in ? plane_bytes : 0u; // in == nullptr: a field of zeros
    unsigned off[2], soff[2];
    int rowcls[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int y = y0 + r, yg = y + g.y0;
        const bool live = xin && y >= 0 && y < g.ny;
        const bool keep = live && lane >= H && lane < kWave - H && 2 * w + r >= H && 2 * w + r < 2 * NW - H;
        const unsigned o = ((unsigned)x + (unsigned)g.pitch * (unsigned)y) * (unsigned)sizeof(R);
        off[r] = live ? o : kDrop;
        soff[r] = keep ? o : kDrop;
        int c = -1;
        if (!live) c = MGRT_CELL_OUTER;
        else if (yg == 0) c = g.face[2];
        else if (yg == g.gny - 1) c = g.face[3];
        else if (x == 0) c = g.face[0];
        else if (x == g.nx - 1) c = g.face[1];
        rowcls[r] = c;
    }
    bool interior[2], latdir[2], latfix[2];
    {
        const int c0 = s0 ? rowcls[1] : rowcls[0], c1 = s0 ? rowcls[0] : rowcls[1];
        interior[0] = c0 < 0, interior[1] = c1 < 0;
        latdir[0] = c0 == MGRT_CELL_DIRICHLET, latdir[1] = c1 == MGRT_CELL_DIRICHLET;
        latfix[0] = latdir[0] && !interior[0], latfix[1] = latdir[1] && !interior[1];
    }
    const long long slab = g.slab;
    const int zlast = g.nz - 1;
    const int zlim = zcount < g.nz ? zcount : g.nz;
    const int wu = w > 0 ? w - 1 : 0, wd = w + 1 < NW ? w + 1 : NW - 1;
    // stage-local steps: plane 2k+1 enters at step k and is final (depth 5) at step k+2; all planes are
    // final once step ceil((nz + 3) / 2) has run
    const int nsteps = (g.nz + 3 + 1) / 2 + 1;
    const int nsteps_all = nsteps + LAG;

    auto march = [&](auto stage_tag) {
        constexpr int STG = decltype(stage_tag)::value;
        constexpr bool FIRST = STG == 0, LAST = STG == 1;
        constexpr int lag = STG * LAG, WB = STG * NW;
        constexpr int kLoads = FIRST ? 8 : 0;       // vector-memory operations a fast step issues
        constexpr int kYounger = (PD - 1) * kLoads;             // ... after the loads of the slot being waited for
        static_assert(kYounger < 64, "too many vector-memory operations in flight for a counted wait");
        bool pass_needed[2];
#pragma unroll
        for (int d = 1; d <= 2; ++d) {
            const int P = 2 * STG + d;
            pass_needed[d - 1] = 2 * w + 1 >= P && 2 * w <= 2 * NW - 1 - P;
        }
        R C[2][ND], F[2][ND];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < ND; ++d) C[c][d] = F[c][d] = 0;
        R pu[PD][2][2], pf[PD][2][2]; // [slot][plane of the pair][row]

        if constexpr (FIRST) {
#pragma unroll
            for (int k = 0; k < PD; ++k)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int z = 2 * k + q;
                    const long long zo = slab * (z < zlast ? z : zlast);
                    const desc_t ru = plane_desc(inb + zo, in_bytes), rf = plane_desc(f + zo, plane_bytes);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        buf_load_hidden(pu[k][q][r], ru, off[r]);
                        buf_load_hidden(pf[k][q][r], rf, off[r]);
                    }
                }
        }
        auto drain = [&]() { // every hidden load has landed and no later use can move above this point
            if constexpr (FIRST) {
#pragma unroll
                for (int k = 0; k < PD; ++k)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        wait_slot2<0>(pu[k][q][0], pu[k][q][1], pf[k][q][0], pf[k][q][1]);
            }
        };
        drain();

        auto group = [&](auto fast_tag, const int g0) {
            constexpr bool FAST = decltype(fast_tag)::value;
            const int i0 = g0 - lag; // first stage-local step of the group: planes 2*i0, 2*i0+1 enter
            desc_t d_in = plane_desc(inb + slab * (2 * (i0 + PD)), in_bytes), d_f = plane_desc(f + slab * (2 * (i0 + PD)), plane_bytes);
            desc_t d_out = plane_desc(out + slab * (2 * i0 - 4), plane_bytes);
            const unsigned long long slab_bytes = (unsigned long long)slab * sizeof(R);
            auto advance = [&](desc_t &d, unsigned long long by) {
                const unsigned long long a = (((unsigned long long)(unsigned)d.y << 32) | (unsigned)d.x) + by;
                d.x = (int)(unsigned)a;
                d.y = (int)(unsigned)(a >> 32); // addresses stay below 2^48: the upper half of word 1 stays 0
            };
            auto plus = [&](desc_t d, unsigned long long by) {
                advance(d, by);
                return d;
            };
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int k = i0 + u; // stage-local step; g0 is a multiple of UN
                if (FAST || g0 + u < nsteps_all) {
#define SL(d) ((2 * (u) + 1 - (d)) & (ND - 1)) /* ring slot of depth d: plane 2k+1-d */
                    const int ps = u % PD, par = u & 1;
                    // ---- planes 2k, 2k+1: prefetch registers -> ring (depths 1 and 0) ------------------
                    if constexpr (FIRST) {
                        if (FAST) {
                            wait_slot2<kYounger>(pu[ps][0][0], pu[ps][0][1], pf[ps][0][0], pf[ps][0][1]);
                            wait_slot2<kYounger>(pu[ps][1][0], pu[ps][1][1], pf[ps][1][0], pf[ps][1][1]);
                        } else {
                            wait_slot2<0>(pu[ps][0][0], pu[ps][0][1], pf[ps][0][0], pf[ps][0][1]);
                            wait_slot2<0>(pu[ps][1][0], pu[ps][1][1], pf[ps][1][0], pf[ps][1][1]);
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const R a0 = pu[ps][q][0], a1 = pu[ps][q][1];
                            const R b0 = pf[ps][q][0], b1 = pf[ps][q][1];
                            C[0][SL(1 - q)] = s0 ? a1 : a0;
                            C[1][SL(1 - q)] = s0 ? a0 : a1;
                            F[0][SL(1 - q)] = s0 ? b1 : b0;
                            F[1][SL(1 - q)] = s0 ? b0 : b1;
                            // the slot's registers are dead before they are requested again below
                            asm volatile("" : "+v"(C[0][SL(1 - q)]), "+v"(C[1][SL(1 - q)]), "+v"(F[0][SL(1 - q)]), "+v"(F[1][SL(1 - q)]));
                        }
                        // ---- planes 2(k+PD), 2(k+PD)+1 -> prefetch slot ps ------------------------------
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int zn = 2 * (k + PD) + q;
                            const long long zo = slab * (zn < zlast ? zn : zlast);
                            const desc_t ru = FAST ? plus(d_in, q * slab_bytes) : plane_desc(inb + zo, in_bytes);
                            const desc_t rf = FAST ? plus(d_f, q * slab_bytes) : plane_desc(f + zo, plane_bytes);
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                buf_load_hidden(pu[ps][q][r], ru, off[r]);
                                buf_load_hidden(pf[ps][q][r], rf, off[r]);
                            }
                        }
                    }
                    // ---- publish the strip's pair of the colour each pass reads, for both planes ---------
                    // pass d (colour d-1) works on depths 2d+1 (first) and 2d and reads the other colour there
#pragma unroll
                    for (int d = 1; d <= 2; ++d)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            bnd[par][d - 1][q][WB + w][lane] = C[(d - 1) ^ 1][SL(2 * d + 1 - q)];
                    __syncthreads();
                    if constexpr (!FIRST) {
                        // the pair the first stage finished in the previous step, already split by colour
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                C[c][SL(1 - q)] = hand[par ^ 1][w][q][0][c][lane];
                                F[c][SL(1 - q)] = hand[par ^ 1][w][q][1][c][lane];
                            }
                    }
                    // ---- the passes ----------------------------------------------------------------------
#pragma unroll
                    for (int d = 1; d <= 2; ++d) {
                        const int c = d - 1;
                        const bool sc = (s0 ^ c) != 0; // the lower row of the pair holds the colour-c point
                        // Every colour pass makes one more ring of the tile's halo worthless: pass number P of the four
                        // (two a stage) is owed rows P .. 2*NW - 1 - P only, nothing that is stored can depend on the
                        // others.  A strip with neither of its rows in that range sits the pass out (wave-uniform).
                        if (!pass_needed[d - 1])
                            continue;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int dep = 2 * d + 1 - q, z = 2 * k + 1 - dep;
                            if (FAST || (z >= 0 && z < zlim)) {
                                const R up = bnd[par][d - 1][q][WB + wu][lane];
                                const R dn = bnd[par][d - 1][q][WB + wd][lane];
                                const int zcls = z == 0 ? g.face[4] : (z == zlast ? g.face[5] : MGRT_CELL_INNER);
                                const bool zin = zcls == MGRT_CELL_INNER, zdir = zcls == MGRT_CELL_DIRICHLET;
                                const R o = C[c ^ 1][SL(dep)];
                                const R xm = from_lower_lane(o), xp = from_upper_lane(o);
                                const R yp = sc ? dn : o; // u[y+1]
                                const R ym = sc ? o : up; // u[y-1]
                                const R own = C[c][SL(dep)];
                                const R fv = F[c][SL(dep)];
                                // mg_1.cl:35-42: neighbours summed in R in this order (z+1 old, z-1 new), times the
                                // double literal 1.0/6.0, rounded to R
                                const R val = (1.0 / 6.0) * (xp + xm + yp + ym + C[c][SL(dep - 1)] + C[c][SL(dep + 1)] - fv);
                                // mg_1.cl:69; for omega == 1 the double expression reduces exactly to val + 0*own in R: one fused multiply-add
                                R gs = OMEGA1 ? fma_r((R)0, own, val) : (R)(val * omega + (1.0 - omega) * own);
                                asm("" : "+v"(gs)); // keep the update unconditional (no exec-mask blocks)
                                const bool take_gs = FAST ? interior[c] : (interior[c] && zin);
                                const bool take_f = FAST ? latfix[c] : (interior[c] ? zdir : latdir[c]);
                                // second stage: a lateral Dirichlet point already holds its f, bit for bit (the first
                                // stage's pass put it there), so between the z faces one select is enough
                                C[c][SL(dep)] = take_gs ? gs : (FAST && STG == 1 ? own : (take_f ? fv : own));
                            }
                        }
                    }
                    // ---- planes 2k-4, 2k-3 (depths 5, 4) are final for this stage ---------------------------
                    if constexpr (!LAST) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                hand[par][w][q][0][c][lane] = C[c][SL(5 - q)];
                                hand[par][w][q][1][c][lane] = F[c][SL(5 - q)];
                            }
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int zf = 2 * k - 4 + q;
                            if (FAST || (zf >= 0 && zf <= zlast)) {
                                const desc_t ro = FAST ? plus(d_out, q * slab_bytes) : plane_desc(out + slab * zf, plane_bytes);
#pragma unroll
                                for (int r = 0; r < 2; ++r)
                                    buf_store_hidden((r ^ s0) ? C[1][SL(5 - q)] : C[0][SL(5 - q)], ro, soff[r]);
                            }
                        }
                    }
#undef SL
                    if (FAST) {
                        if (FIRST)
                            advance(d_in, 2 * slab_bytes), advance(d_f, 2 * slab_bytes);
                        if (LAST)
                            advance(d_out, 2 * slab_bytes);
                    }
                }
            }
        };

        // A fast group: every plane its passes touch lies strictly between the z faces and below zlim, the
        // planes it prefetches exist, the planes it stores exist.
        auto is_fast = [&](int i0) {
            return 2 * i0 - 6 >= 1 && 2 * (i0 + UN - 1) - 1 < zlim && 2 * (i0 + UN - 1) - 1 < zlast &&
                   2 * (i0 + UN - 1 + PD) + 1 <= zlast;
        };
        for (int g0 = 0; g0 < nsteps_all;) {
            if (is_fast(g0 - lag)) {
                do {
                    group(std::true_type{}, g0);
                    g0 += UN;
                } while (is_fast(g0 - lag));
                drain();
            } else {
                group(std::false_type{}, g0);
                g0 += UN;
                drain();
            }
        }
    };

    if (stage == 0)
        march(std::integral_constant<int, 0>{});
    else
        march(std::integral_constant<int, 1>{});
}

template <class R, int NW, int PD>
mgrt_status launch_pair(const mgrt_grid *g, R *out, const R *in, const R *f, R omega, int zcount, hipStream_t hs)
{
    constexpr int H = 4, OX = kWave - 2 * H, OY = 2 * NW - 2 * H;
    static_assert(OY > 0 && OY % 2 == 0 && NW * 2 * kWave <= 1024, "tile shape");
    const int ntx = (g->nx + OX - 1) / OX, nty = (g->ny + OY - 1) / OY;
    const size_t lds = sizeof(R) * kWave * (2 * 2 * 2 * 2 * NW + 2 * NW * 2 * 2 * 2);
    static bool raised = false; // per instantiation: 64 KB of dynamic LDS and more need the attribute
    if (!raised) {
        MGRT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth3d_fused_pair<R, NW, PD, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MGRT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth3d_fused_pair<R, NW, PD, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    const dim3 grid(ntx * nty);
    if (omega == (R)1)
        k_smooth3d_fused_pair<R, NW, PD, true><<<grid, NW * 2 * kWave, lds, hs>>>(*g, in, out, f, omega, zcount, ntx);
    else
        k_smooth3d_fused_pair<R, NW, PD, false><<<grid, NW * 2 * kWave, lds, hs>>>(*g, in, out, f, omega, zcount, ntx);
    runtime().last_sweep_launches = 1;
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

} // namespace

namespace mgrt {

mgrt_status smooth_sweep_pair(const mgrt_grid *g, float *out, const float *in, const float *f, float omega, int zcount,
                              bool six_strips, hipStream_t hs)
{
    return six_strips ? launch_pair<float, 6, 2>(g, out, in, f, omega, zcount, hs) : launch_pair<float, 8, 2>(g, out, in, f, omega, zcount, hs);
}

} // namespace mgrt
