// mgrt_bottom.hip -- the coarsest levels of a cycle in one launch.
//
// From about 17^3 points down a level is a handful of microseconds of work, and a V(2,2) cycle
// spends six launches on it: the cycle is then bound by launch latency.  This kernel runs
// iterate() (multigridsolver3D.cpp:41-80) for the bottom of the hierarchy as ONE workgroup with
// every array of every level in LDS: the colour passes of the smoother in place (mg_1.cl:29-43,
// 56-104; a thread owns a column and marches z, as k_smooth3d_pass does), residual (mg_1.cl:46-53,
// 111-133), restriction (mg_1.cl:146-190) and correction (mg_1.cl:200-233), separated by
// workgroup barriers.  Expressions and operand order are those of the per-step kernels in
// mgrt_kernels.hip, so the values are the same bit for bit.  The 2D twins (mg_0.cl) run as grids of
// one plane with their own expressions.  Analytic box domains only.
#include "mgrt_internal.h"

using namespace mgrt;

namespace {

constexpr int kMaxLev = 8, kThreads = 1024;

struct BottomArgs {
    int dim;
    int nlev;
    int nx[kMaxLev], ny[kMaxLev], nz[kMaxLev];
    int face[6];
    int zfull, a1, a2, v;
};

struct Lvl {
    int nx, ny, nz, p, s, n, zcount;
};

template <int DIM> __device__ __forceinline__ int box_class(const Lvl &L, const int (&face)[6], int x, int y, int z)
{
    // same order as cell_class<DIM, false>
    if (DIM == 2) {
        if (x == 0) return face[0];
        if (x == L.nx - 1) return face[1];
        if (y == 0) return face[2];
        if (y == L.ny - 1) return face[3];
        return MGRT_CELL_INNER;
    }
    if (y == 0) return face[2];
    if (y == L.ny - 1) return face[3];
    if (x == 0) return face[0];
    if (x == L.nx - 1) return face[1];
    if (z == 0) return face[4];
    if (z == L.nz - 1) return face[5];
    return MGRT_CELL_INNER;
}

__device__ __forceinline__ bool border_class(int t) { return t == MGRT_CELL_DIRICHLET || t == MGRT_CELL_NEUMANN; }

// floor(i / d) and i mod d for 0 <= i < 2^21, d >= 1 without the forty-instruction integer division: (i + 0.5) / d lies
// at least 0.5 / d away from every integer, the hardware reciprocal is good to one unit in the last place and the
// product rounds once more, which together displace it by less than (i / d) * 2^-22.
struct FastDiv {
    int d;
    float rcp;
    __device__ __forceinline__ explicit FastDiv(int d_) : d(d_), rcp(__builtin_amdgcn_rcpf((float)d_)) {}
    __device__ __forceinline__ int quot(int i) const { return (int)(((float)i + 0.5f) * rcp); }
    __device__ __forceinline__ void split(int i, int &q, int &r) const { q = quot(i), r = i - q * d; }
};

template <class R> __device__ __forceinline__ R dot4b(R a0, R a1, R a2, R a3, R b0, R b1, R b2, R b3)
{
    return a0 * b0 + a1 * b1 + a2 * b2 + a3 * b3;
}

template <class R, int DIM>
__global__ __launch_bounds__(kThreads) void k_bottom_cycle(BottomArgs A, R *u_out, const R *u_in, const R *f_in, R omega)
{
    extern __shared__ __align__(16) unsigned char smem[];
    R *const base = reinterpret_cast<R *>(smem);
    // Level geometry and array placement sit in LDS: indexing the kernel argument arrays (or local
    // arrays of pointers) with a run-time level makes the compiler move them to scratch memory,
    // which costs a memory round trip per phase.  Level m occupies 3 arrays of n_m elements
    // (u, f, r), the last one 2.
    __shared__ int s_geo[kMaxLev][4]; // nx, ny, nz, offset of u
    if (threadIdx.x == 0) {
        int off = 0;
#pragma unroll
        for (int m = 0; m < kMaxLev; ++m) {
            s_geo[m][0] = A.nx[m], s_geo[m][1] = A.ny[m], s_geo[m][2] = A.nz[m], s_geo[m][3] = off;
            off += m < A.nlev ? 3 * A.nx[m] * A.ny[m] * A.nz[m] : 0;
        }
    }
    __syncthreads();
    auto level = [&](int l) {
        Lvl L;
        L.nx = s_geo[l][0], L.ny = s_geo[l][1], L.nz = s_geo[l][2];
        L.p = L.nx, L.s = L.nx * L.ny, L.n = L.s * L.nz;
        L.zcount = DIM == 2 || A.zfull ? L.nz : L.nz / 2 + L.nz % 2; // multigridsolver3D.cpp:121
        return L;
    };
    auto u_of = [&](int l) { return base + s_geo[l][3]; };
    const int tid = threadIdx.x;
    const int(&face)[6] = A.face;
    // The classes of the two z faces as values of their own: selected by address inside a loop (face[z == 0 ? 4 : 5]) they
    // become a vector load from the argument segment, a trip to memory in the middle of a time step.
    const int face_z0 = A.face[4], face_z1 = A.face[5];

    {
        // loads in batches of 8 per thread: a plain loop would wait for each load in turn
        const int n0 = A.nx[0] * A.ny[0] * A.nz[0];
        for (int i0 = 0; i0 < n0; i0 += 8 * kThreads) {
            R ru[8], rf[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + tid + k * kThreads;
                ru[k] = i < n0 && u_in ? u_in[i] : (R)0;
                rf[k] = i < n0 ? f_in[i] : (R)0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + tid + k * kThreads;
                if (i < n0) {
                    base[i] = ru[k];
                    base[n0 + i] = rf[k];
                }
            }
        }
    }
    __syncthreads();

    // `iters` smoothing iterations on level l, in place (smoother_iterate, multigridsolver3D.cpp:111-132)
    const bool omega_is_one = omega == (R)1;
    auto smooth = [&](int l, int iters) {
        const Lvl L = level(l);
        R *u = u_of(l);
        const R *f = u + L.n;
        const int hx = (L.nx + 1) / 2;
        if (DIM == 3 && iters > 0 && 2 * iters * hx * L.ny <= kThreads) {
            // All colour passes at once, each one plane behind the pass before it: a thread owns one
            // column of one pass, and in time step t pass number p works on plane t - p.  The 3D
            // colouring ignores z, so a pass at plane z needs the pass before it only up to plane
            // z (done a step earlier) and must not yet see the pass after it there (which comes a
            // step later): one barrier per step, and zcount + passes - 1 steps instead of
            // passes * zcount.  Every point is computed by the same expression from the same values
            // as in the pass-by-pass schedule below.
            const int ncol = hx * L.ny, npass = 2 * iters;
            int p, idx, y, xh;
            FastDiv(ncol).split(tid, p, idx);
            FastDiv(hx).split(idx, y, xh);
            const int x = 2 * xh + ((p + y) & 1);
            const bool active = p < npass && x < L.nx;
            const int lat = !active ? (int)MGRT_CELL_OUTER
                                    : y == 0 ? face[2] : y == L.ny - 1 ? face[3] : x == 0 ? face[0] : x == L.nx - 1 ? face[1] : -1;
            const int col = active ? x + L.p * y : 0;
            const bool gather = lat < 0;
            const int oxp = gather ? 1 : 0, oxm = gather ? -1 : 0, oyp = gather ? L.p : 0, oym = gather ? -L.p : 0;
            const int ztop = L.nz - 1;
            // what a step costs is what the wavefronts of a SIMD issue in it: one pointer per operand stream, advanced by a
            // plane a step, and the column's class on its first, inner and last plane worked out beforehand
            const int cls_first = lat >= 0 ? lat : face_z0, cls_inner = lat >= 0 ? lat : (int)MGRT_CELL_INNER,
                      cls_last = lat >= 0 ? lat : face_z1;
            // Byte offsets into the LDS block, one per operand stream, each made opaque once so that the compiler keeps
            // seven finished addresses and advances them (it otherwise keeps partial sums and adds the level's base again
            // before every read).  o_own is plane t - p of the column: used only once that plane exists.
            const int plane_bytes = L.s * (int)sizeof(R);
            int o_own = (int)((u + col - L.s * p - base) * (long)sizeof(R));
            int o_xp = o_own + oxp * (int)sizeof(R), o_xm = o_own + oxm * (int)sizeof(R), o_yp = o_own + oyp * (int)sizeof(R),
                o_ym = o_own + oym * (int)sizeof(R), o_up = o_own + plane_bytes, o_f = o_own + L.n * (int)sizeof(R);
            asm volatile("" : "+v"(o_own), "+v"(o_xp), "+v"(o_xm), "+v"(o_yp), "+v"(o_ym), "+v"(o_up), "+v"(o_f));
            auto at = [&](int off) -> R & { return *reinterpret_cast<R *>(smem + off); };
            R below = 0;
            for (int t = 0; t < L.zcount + npass - 1; ++t) {
                const int z = t - p;
                if (active && z >= 0 && z < L.zcount) {
                    const R xp = at(o_xp), xm = at(o_xm), yp = at(o_yp), ym = at(o_ym);
                    // the plane above the last one is the last one itself (the array that follows u in LDS is read and dropped)
                    const R own = at(o_own), up = at(o_up), fz = at(o_f);
                    const R above = z < ztop ? up : own;
                    const int cls = z == 0 ? cls_first : z == ztop ? cls_last : cls_inner;
                    const R val = (1.0 / 6.0) * (xp + xm + yp + ym + above + below - fz);
                    R gs;
                    if (omega_is_one)
                        gs = val + (R)0 * own; // what the full expression reduces to, exactly
                    else
                        gs = val * omega + (1.0 - omega) * own;
                    const R next = cls == MGRT_CELL_INNER ? gs : (cls == MGRT_CELL_DIRICHLET ? fz : own);
                    at(o_own) = next;
                    below = next;
                }
                o_own += plane_bytes, o_xp += plane_bytes, o_xm += plane_bytes, o_yp += plane_bytes, o_ym += plane_bytes,
                    o_up += plane_bytes, o_f += plane_bytes;
                __syncthreads();
            }
            return;
        }
        for (int it = 0; it < 2 * iters; ++it) {
            const int odd = it & 1;
            const FastDiv by_hx(hx);
            for (int idx = tid; idx < hx * L.ny; idx += kThreads) {
                int y, xh;
                by_hx.split(idx, y, xh);
                const int x = 2 * xh + ((odd + y) & 1);
                if (x >= L.nx)
                    continue;
                if (DIM == 2) { // true red-black point update, mg_0.cl:29-40,52-98
                    const int b = x + L.p * y;
                    const int t = box_class<2>(L, face, x, y, 0);
                    if (t == MGRT_CELL_INNER) {
                        const R val = 0.25 * (u[b + 1] + u[b - 1] + u[b + L.p] + u[b - L.p] - f[b]);
                        u[b] = val * omega + (1.0 - omega) * u[b];
                    } else if (t == MGRT_CELL_DIRICHLET) {
                        u[b] = f[b];
                    }
                    continue;
                }
                // class of the column where it does not depend on z, else -1
                const int lat = y == 0 ? face[2] : y == L.ny - 1 ? face[3] : x == 0 ? face[0] : x == L.nx - 1 ? face[1] : -1;
                const int col = x + L.p * y;
                // The in-plane neighbours have the other colour and do not change during this pass,
                // and the column's own planes above z are still old: everything a step needs except
                // the value just computed below it is fetched one step ahead, so that the z
                // recurrence carries only add, subtract, the 1/6 product and the relaxation.
                // Columns on a lateral face never relax; their neighbour offsets collapse onto the
                // column itself so that no address outside the array is formed.
                const bool gather = lat < 0;
                const int oxp = gather ? 1 : 0, oxm = gather ? -1 : 0, oyp = gather ? L.p : 0, oym = gather ? -L.p : 0;
                const int ztop = L.nz - 1;
                struct Ahead { // raw values of the next step; summed only after this step's chain
                    R xp, xm, yp, ym, above, f;
                };
                auto fetch = [&](int z) { // plane z of the neighbours and of f, plane z + 1 of the column
                    const int b = col + L.s * (z < ztop ? z : ztop);
                    Ahead a;
                    a.xp = u[b + oxp], a.xm = u[b + oxm], a.yp = u[b + oyp], a.ym = u[b + oym];
                    a.above = u[col + L.s * (z + 1 < ztop ? z + 1 : ztop)];
                    a.f = f[b];
                    return a;
                };
                const Ahead first = fetch(0);
                R sum4 = first.xp + first.xm + first.yp + first.ym; // mg_1.cl:35-42: the first three additions
                R own = u[col], above = first.above, fz = first.f;
                R below = 0;
                for (int z = 0; z < L.zcount; ++z) {
                    const Ahead nx = fetch(z + 1);
                    const int t = lat >= 0 ? lat : (z == 0 ? face_z0 : z == ztop ? face_z1 : (int)MGRT_CELL_INNER);
                    const R val = (1.0 / 6.0) * (sum4 + above + below - fz);
                    R gs;
                    if (omega_is_one)
                        gs = val + (R)0 * own; // what the full expression reduces to, exactly
                    else
                        gs = val * omega + (1.0 - omega) * own;
                    const R next = t == MGRT_CELL_INNER ? gs : (t == MGRT_CELL_DIRICHLET ? fz : own);
                    u[col + L.s * z] = next; // frozen cells get their own value back
                    below = next;
                    // nothing of the next step is touched before this point: its loads have the
                    // whole chain above to land
                    __builtin_amdgcn_sched_barrier(0);
                    sum4 = nx.xp + nx.xm + nx.yp + nx.ym;
                    own = above, above = nx.above, fz = nx.f;
                }
            }
            __syncthreads();
        }
    };

    auto residual = [&](int l) {
        const Lvl L = level(l);
        const R *u = u_of(l), *f = u + L.n;
        R *r = u_of(l) + 2 * L.n;
        const FastDiv by_plane(L.s), by_row(L.nx);
        for (int i = tid; i < L.n; i += kThreads) {
            int x, y, z, inplane;
            by_plane.split(i, z, inplane);
            by_row.split(inplane, y, x);
            const int t = box_class<DIM>(L, face, x, y, z);
            R v = 0;
            if (t == MGRT_CELL_INNER)
                v = DIM == 3 ? f[i] - (u[i + 1] + u[i - 1] + u[i + L.p] + u[i - L.p] + u[i + L.s] + u[i - L.s] - 6 * u[i])
                             : f[i] - (u[i + 1] + u[i - 1] + u[i + L.p] + u[i - L.p] - 4 * u[i]);
            else if (t == MGRT_CELL_DIRICHLET)
                v = f[i] - u[i];
            r[i] = v;
        }
        __syncthreads();
    };

    // F[l+1] = Restrict(Rs[l]); U[l+1] = 0
    auto restrict_zero = [&](int l) {
        const Lvl Lf = level(l), Lc = level(l + 1);
        const R *rf = u_of(l) + 2 * Lf.n;
        R *e = u_of(l + 1), *rc = e + Lc.n;
        const int p = Lf.p, s = Lf.s;
        const FastDiv by_plane(Lc.s), by_row(Lc.nx);
        for (int d = tid; d < Lc.n; d += kThreads) {
            int i, j, k, inplane;
            by_plane.split(d, k, inplane);
            by_row.split(inplane, j, i);
            const int x = 2 * i, y = 2 * j, z = DIM == 3 ? 2 * k : 0;
            const R *c = rf + x + p * y + s * z;
            const bool on_face = x == 0 || x == Lf.nx - 1 || y == 0 || y == Lf.ny - 1 ||
                                 (DIM == 3 && (z == 0 || z == Lf.nz - 1));
            bool border = on_face || border_class(box_class<DIM>(Lf, face, x, y, z));
            if (!border) {
                border = border_class(box_class<DIM>(Lf, face, x + 1, y, z)) ||
                         border_class(box_class<DIM>(Lf, face, x, y + 1, z)) ||
                         border_class(box_class<DIM>(Lf, face, x + 1, y + 1, z));
                if (DIM == 3 && !border)
                    border = border_class(box_class<DIM>(Lf, face, x, y, z + 1)) ||
                             border_class(box_class<DIM>(Lf, face, x + 1, y, z + 1)) ||
                             border_class(box_class<DIM>(Lf, face, x, y + 1, z + 1)) ||
                             border_class(box_class<DIM>(Lf, face, x + 1, y + 1, z + 1));
            }
            R val;
            if (border) {
                val = c[0];
            } else if (DIM == 3) {
                val = 1.0 / 64.0 * (c[s + p + 1] + c[s + p - 1] + c[s - p + 1] + c[s - p - 1] + c[-s + p + 1] +
                                    c[-s + p - 1] + c[-s - p + 1] + c[-s - p - 1]) +
                      1.0 / 32.0 * (c[p + 1] + c[p - 1] + c[-p + 1] + c[-p - 1] + c[-s + p] + c[-s - p] + c[-s + 1] +
                                    c[-s - 1] + c[s + p] + c[s - p] + c[s + 1] + c[s - 1]) +
                      1.0 / 16.0 * (c[p] + c[-p] + c[1] + c[-1] + c[s] + c[-s]) + 1.0 / 8.0 * (c[0]);
            } else {
                val = 1.0 / 16.0 * (c[-p - 1] + c[-p + 1] + c[p - 1] + c[p + 1]) +
                      1.0 / 8.0 * (c[-p] + c[1] + c[-1] + c[p]) + 1.0 / 4.0 * (c[0]);
            }
            rc[d] = val;
            e[d] = 0;
        }
        __syncthreads();
    };

    // U[l] += 4 * P(U[l+1]) (residual_correct_kernel, in place)
    auto correct = [&](int l) {
        const Lvl Lf = level(l), Lc = level(l + 1);
        R *u = u_of(l);
        const R *e = u_of(l + 1);
        const int cp = Lc.p, cs = Lc.s;
        const FastDiv by_plane(Lf.s), by_row(Lf.nx);
        for (int d = tid; d < Lf.n; d += kThreads) {
            int x, y, z, inplane;
            by_plane.split(d, z, inplane);
            by_row.split(inplane, y, x);
            const int ci = x / 2, cj = y / 2, ck = z / 2;
            const int ci1 = ci + 1 < Lc.nx ? ci + 1 : ci;
            const int cj1 = cj + 1 < Lc.ny ? cj + 1 : cj;
            const int ck1 = ck + 1 < Lc.nz ? ck + 1 : ck;
            const R *e0 = e + cs * ck, *e1 = e + cs * ck1;
            R val;
            if (border_class(box_class<DIM>(Lf, face, x, y, z))) {
                val = e0[ci + cp * cj];
            } else if (DIM == 3) {
                const R wu = 0.5 * (x % 2), ww = 0.5 * (y % 2), wv = 0.5 * (z % 2);
                const R n10 = (1.0 - wu) * (1.0 - ww) * (1.0 - wv), n11 = (1.0 - ww) * wu * (1.0 - wv),
                        n12 = (1.0 - wu) * ww * (1.0 - wv), n13 = ww * wu * (1.0 - wv);
                const R n20 = (1.0 - wu) * (1.0 - ww) * wv, n21 = (1.0 - ww) * wu * wv, n22 = (1.0 - wu) * ww * wv,
                        n23 = ww * wu * wv;
                val = dot4b<R>(e0[ci + cp * cj], e0[ci1 + cp * cj], e0[ci + cp * cj1], e0[ci1 + cp * cj1], n10, n11, n12,
                               n13) +
                      dot4b<R>(e1[ci + cp * cj], e1[ci1 + cp * cj], e1[ci + cp * cj1], e1[ci1 + cp * cj1], n20, n21, n22,
                               n23);
            } else { // mg_0.cl:167-194
                const R wu = 0.5 * (x % 2), ww = 0.5 * (y % 2);
                const R n0 = (1.0 - wu) * (1.0 - ww), n1 = (1.0 - ww) * wu, n2 = (1.0 - wu) * ww, n3 = ww * wu;
                val = dot4b<R>(e0[ci + cp * cj], e0[ci1 + cp * cj], e0[ci + cp * cj1], e0[ci1 + cp * cj1], n0, n1, n2, n3);
            }
            u[d] = u[d] + val * 4;
        }
        __syncthreads();
    };

    // iterate() without recursion: walk down smoothing, walk up correcting; k[l] counts the coarse
    // solves done on level l.  Control flow is uniform over the workgroup.
    unsigned long long kpack = 0; // 8 bits per level
    auto kget = [&](int l) { return (int)((kpack >> (8 * l)) & 255u); };
    auto kset = [&](int l, int val) { kpack = (kpack & ~(255ull << (8 * l))) | ((unsigned long long)val << (8 * l)); };
    int l = 0;
    bool down = true;
    for (;;) {
        if (down) {
            smooth(l, A.a1);
            const bool rec = l + 1 < A.nlev && A.v > 0;
            if (rec) {
                kset(l, 0);
                residual(l);
                restrict_zero(l);
                ++l;
            } else {
                smooth(l, A.a2);
                down = false;
            }
        } else {
            if (l == 0)
                break;
            --l;
            correct(l);
            kset(l, kget(l) + 1);
            if (kget(l) < A.v) {
                residual(l);
                restrict_zero(l);
                ++l;
                down = true;
            } else {
                smooth(l, A.a2);
            }
        }
    }

    {
        const int n0 = A.nx[0] * A.ny[0] * A.nz[0];
        for (int i = tid; i < n0; i += kThreads)
            u_out[i] = base[i];
    }
}

template <class R> size_t lds_need(int nlev, const int *nx, const int *ny, const int *nz)
{
    size_t elems = 0;
    for (int l = 0; l < nlev; ++l)
        elems += (size_t)nx[l] * ny[l] * nz[l] * (l + 1 < nlev ? 3 : 2);
    return elems * sizeof(R);
}

constexpr size_t kLdsBudget = 150 * 1024; // of the 160 KB of a CU

template <class R>
mgrt_status bottom(int dim, int nlev, const int *nx, const int *ny, const int *nz, const int face[6], int zfull, R *u,
                   const R *u_in, const R *f, R omega, int a1, int a2, int v, hipStream_t hs)
{
    MGRT_REQUIRE(nlev >= 1 && nlev <= kMaxLev, "bottom cycle: %d levels", nlev);
    MGRT_REQUIRE(v < 256, "bottom cycle: %d coarse solves per level", v);
    const size_t need = lds_need<R>(nlev, nx, ny, nz);
    MGRT_REQUIRE(need <= kLdsBudget, "bottom cycle: %zu bytes of LDS needed", need);
    BottomArgs A = {};
    A.dim = dim;
    A.nlev = nlev;
    for (int l = 0; l < nlev; ++l)
        A.nx[l] = nx[l], A.ny[l] = ny[l], A.nz[l] = nz[l];
    for (int i = 0; i < 6; ++i)
        A.face[i] = face[i];
    A.zfull = zfull, A.a1 = a1, A.a2 = a2, A.v = v;
    static bool raised = false; // per process and element type: allow more than 64 KB of dynamic LDS
    if (!raised) {
        MGRT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bottom_cycle<R, 3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        MGRT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bottom_cycle<R, 2>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        raised = true;
    }
    if (dim == 3)
        k_bottom_cycle<R, 3><<<1, kThreads, need, hs>>>(A, u, u_in, f, omega);
    else
        k_bottom_cycle<R, 2><<<1, kThreads, need, hs>>>(A, u, u_in, f, omega);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

} // namespace

namespace mgrt {

size_t bottom_cycle_lds_bytes(int elem_size, int nlev, const int *nx, const int *ny, const int *nz)
{
    return elem_size == 4 ? lds_need<float>(nlev, nx, ny, nz) : lds_need<double>(nlev, nx, ny, nz);
}
size_t bottom_cycle_lds_budget() { return kLdsBudget; }
int bottom_cycle_max_levels() { return kMaxLev; }

mgrt_status bottom_cycle(int dim, int nlev, const int *nx, const int *ny, const int *nz, const int face[6], int zfull,
                         float *u, const float *u_in, const float *f, float omega, int a1, int a2, int v, hipStream_t hs)
{
    return bottom<float>(dim, nlev, nx, ny, nz, face, zfull, u, u_in, f, omega, a1, a2, v, hs);
}
mgrt_status bottom_cycle(int dim, int nlev, const int *nx, const int *ny, const int *nz, const int face[6], int zfull,
                         double *u, const double *u_in, const double *f, double omega, int a1, int a2, int v,
                         hipStream_t hs)
{
    return bottom<double>(dim, nlev, nx, ny, nz, face, zfull, u, u_in, f, omega, a1, a2, v, hs);
}

} // namespace mgrt
