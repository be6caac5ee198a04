// mgrt_slab.hip -- V-cycles on a 3D grid cut into y-slabs, one slab per process (one process per GPU).
//
// The reference runs on one device with one in-order queue (clcontextloader.cpp:57,
// multigridsolver3D.cpp:36) and has nothing of the kind; this is the cycle of
// MultigridSolver3D::iterate (multigridsolver3D.cpp:41-80) spread over several GPUs so that every
// cell still goes through exactly the arithmetic it goes through on one: the grid is cut along y
// only, every z column (and with it the z recurrence of a colour pass, mg_1.cl:56-104) stays on one
// rank, and the results are bitwise those of the one-GPU plan.
//
// Rank r owns rows [B_r, B_r+1) of level 0 and rows [B_r / 2^l, B_r+1 / 2^l) of level l (B_r a
// multiple of 2^ndist), so the owner of coarse row j owns fine row 2j and restriction and
// interpolation reach no further than the halo.  A rank stores its rows plus kHalo rows of each
// neighbour.  A fused sweep of two iterations consumes four halo rows instead of exchanging after
// every colour pass, so one exchange precedes each sweep: per level and V(2,2) cycle the smoothed
// unknown, the coarse right-hand side and the coarse correction travel once, plus the unknown at the
// start of a cycle.  Levels with at most `crossover` points are gathered and solved whole on every
// rank with the one-GPU plan (agglomeration without the scatter back).
//
// Halos and gathered rows travel by one of three transports: RCCL (ncclSend/ncclRecv pairs with the
// two neighbours, ncclAllGather, ncclAllReduce, all on the plan's stream, so that a whole cycle is
// captured in one hipGraph per rank and replayed); host callbacks (blocking; tests and other
// transports); or none, where the copies stay on the device: the share of one rank of a larger run,
// for timing on a single GPU.
#include "mgrt_internal.h"
#include <dlfcn.h>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

using namespace mgrt;

namespace {

constexpr int kHalo = 4; // rows: 2 per smoothing iteration fused into one sweep

// ---- RCCL, bound at run time so that the library loads where RCCL is absent -----------------------
struct mgrt_rccl_id_by_value { // ncclUniqueId, which ncclCommInitRank takes by value
    char internal[128];
};
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, mgrt_rccl_id_by_value, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr; // optional
};
constexpr int kNcclChar = 0, kNcclDouble = 8, kNcclSum = 0; // ncclDataType_t / ncclRedOp_t values of rccl.h

mgrt_status rccl(Rccl **out)
{
    static Rccl r;
    static std::once_flag once; // solver threads of one process may arrive together
    std::call_once(once, [] {
        // the copy a host framework has already loaded, when there is one
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib)
                break;
        }
        if (r.lib) {
#define MGRT_SYM(field, name) *(void **)(&r.field) = dlsym(r.lib, name)
            MGRT_SYM(GetUniqueId, "ncclGetUniqueId");
            MGRT_SYM(CommInitRank, "ncclCommInitRank");
            MGRT_SYM(CommDestroy, "ncclCommDestroy");
            MGRT_SYM(Send, "ncclSend");
            MGRT_SYM(Recv, "ncclRecv");
            MGRT_SYM(GroupStart, "ncclGroupStart");
            MGRT_SYM(GroupEnd, "ncclGroupEnd");
            MGRT_SYM(AllGather, "ncclAllGather");
            MGRT_SYM(AllReduce, "ncclAllReduce");
            MGRT_SYM(GetErrorString, "ncclGetErrorString");
            MGRT_SYM(GetVersion, "ncclGetVersion");
#undef MGRT_SYM
        }
    });
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd ||
        !r.AllGather || !r.AllReduce || !r.GetErrorString)
        return fail(MGRT_E_UNSUPPORTED, "RCCL (librccl.so.1) is not available: %s", r.lib ? "a symbol is missing" : dlerror());
    *out = &r;
    return MGRT_OK;
}

#define MGRT_NCCL(expr)                                                                            \
    do {                                                                                           \
        const int e_ = (expr);                                                                     \
        if (e_ != 0)                                                                               \
            return ::mgrt::fail(MGRT_E_TRANSPORT, "%s: %s", #expr, nc->GetErrorString(e_));        \
    } while (0)

int coarse_extent(int n) { return (n + 1) / 2; }

} // namespace

struct mgrt_slab_plan {
    mgrt_slab_desc desc;
    mgrt_transport tr;
    hipStream_t stream = nullptr;
    // halos that independent work can hide travel on a stream of their own, forked from and joined to
    // `stream` by events (inside a captured cycle these become edges of the graph)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    Rccl *nc = nullptr;
    void *comm = nullptr;
    int rank = 0, world = 1, ndist = 0;
    std::vector<int> bounds; // first rows of the slabs on level 0, then ny
    struct Level {
        mgrt_grid g;          // this rank's rows of the level, halo included
        int nz, gny, nx;      // extents of the whole level
        int y0, y1;           // owned global rows
        int lo, hi, nyl;      // stored global rows and their count
        int own0, own1;       // owned local rows
        size_t n;             // elements stored
        void *work = nullptr, *r = nullptr, *rhs = nullptr, *e = nullptr;
        void *fmg_f = nullptr, *fmg_u = nullptr; // FMG: restricted right-hand side and solution, allocated on first use
        void *sendbuf[2] = {nullptr, nullptr}, *recvbuf[2] = {nullptr, nullptr}; // [0] lower neighbour, [1] upper
        size_t halo_bytes = 0;
    };
    std::vector<Level> lv;
    // the first level every rank holds whole, and the one-GPU plan that takes it from there
    int cnx = 0, cny = 0, cnz = 0;
    std::vector<int> crow0, crow1; // rows of it each rank computes
    mgrt_grid cgrid;
    mgrt_plan *coarse = nullptr;
    void *c_rhs = nullptr, *c_e = nullptr, *gather_in = nullptr, *gather_out = nullptr;
    void *c_fmg_f = nullptr, *c_fmg_u = nullptr; // FMG on the first level held whole, allocated on first use
    int maxrows = 0;
    double *d_norms = nullptr;
    int norms_cap = 0;
    hipGraphExec_t exec = nullptr;
    bool graph_failed = false; // capture or instantiation was refused once: cycles are launched directly
    bool broken = false;       // a transport call failed: halos may be stale, the plan refuses further work
    bool warmed = false;       // one cycle has run outside a capture (RCCL sets its connections up on first use)
    struct Key {
        void *u;
        const void *f;
        double omega;
        int a1, a2, v;
    } key{};
};

namespace {

size_t elem(const mgrt_slab_plan *p) { return (size_t)p->desc.elem_size; }

bool capturable(const mgrt_slab_plan *p) { return p->tr.kind != MGRT_TRANSPORT_CALLBACK; }

// How many levels are cut into slabs: while the level recurses, has more than `crossover` points and
// leaves every rank at least two halos of rows.  Slab boundaries are multiples of 2^ndist rows.
int distributed_levels(int nx, int ny, int nz, int world, long long crossover)
{
    const int ny0 = ny;
    int nd = 0;
    while (world > 1 && nx > 3 && ny > 3 && nz > 3 && (long long)nx * ny * nz > crossover) {
        const int q = (ny0 - 1) / (world << (nd + 1));
        if (2 * q < 2 * kHalo)
            break;
        ++nd;
        nx = coarse_extent(nx), ny = coarse_extent(ny), nz = coarse_extent(nz);
    }
    return nd;
}

// Rows [r0, r0 + kHalo) of every plane of a field <-> a dense staging buffer, for both neighbours in one
// launch (two rectangular copies each way cost 4 x 5 us per exchange, most of what a rank of a grown grid
// loses against one GPU).  Everything is counted in 4-byte words; block (x, z, k): words of plane z for
// neighbour k.
struct HaloCopy {
    unsigned *field;
    unsigned *buf[2];        // staging buffer per neighbour, null where there is none
    long long row0[2];       // first word of the rows in plane 0
    long long slab_words;    // words per plane of the field
    unsigned seg_words;      // words per plane in a staging buffer: kHalo rows
};

template <bool UNPACK> __global__ __launch_bounds__(256) void k_halo_rows(HaloCopy c)
{
    const unsigned k = blockIdx.z, z = blockIdx.y;
    unsigned *buf = c.buf[k];
    if (!buf)
        return;
    unsigned *f = c.field + c.row0[k] + (long long)z * c.slab_words;
    unsigned *b = buf + (size_t)z * c.seg_words;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < c.seg_words; i += gridDim.x * 256u)
        if (UNPACK)
            f[i] = b[i];
        else
            b[i] = f[i];
}

template <bool UNPACK>
mgrt_status halo_rows(const mgrt_slab_plan *p, const mgrt_slab_plan::Level &L, void *field, void *const buf[2], const bool has[2],
                      const int row[2], hipStream_t on)
{
    const size_t wpe = elem(p) / 4; // words per element
    HaloCopy c;
    c.field = (unsigned *)field;
    for (int k = 0; k < 2; ++k) {
        c.buf[k] = has[k] ? (unsigned *)buf[k] : nullptr;
        c.row0[k] = (long long)row[k] * L.g.pitch * (long long)wpe;
    }
    c.slab_words = L.g.slab * (long long)wpe;
    c.seg_words = (unsigned)(kHalo * (size_t)L.g.pitch * wpe);
    MGRT_REQUIRE(L.nz <= 65535, "mgrt_slab_plan: %d planes are more than one launch of the halo kernel covers", L.nz);
    unsigned bx = (c.seg_words + 255u) / 256u;
    if (bx > 64u)
        bx = 64u;
    k_halo_rows<UNPACK><<<dim3(bx, (unsigned)L.nz, 2), 256, 0, on>>>(c);
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

// Fill the halo rows of `field` on level L from the neighbours' owned rows; the work is queued on `on`.
mgrt_status exchange_on(mgrt_slab_plan *p, mgrt_slab_plan::Level &L, void *field, hipStream_t on);

mgrt_status exchange(mgrt_slab_plan *p, mgrt_slab_plan::Level &L, void *field) { return exchange_on(p, L, field, p->stream); }

// The same on the side stream: exchange_begin returns at once, the halo rows are valid after exchange_end.
// What is queued on the main stream in between runs beside the transfer and must not touch halo rows.
// (With the callback transport the host blocks inside exchange_begin, so nothing overlaps; the path is
// the same, which is how the one-GPU tests cover it.)
bool overlaps(const mgrt_slab_plan *p) { return p->world > 1 && p->side; }

mgrt_status exchange_begin(mgrt_slab_plan *p, mgrt_slab_plan::Level &L, void *field)
{
    if (!overlaps(p))
        return exchange(p, L, field);
    MGRT_HIP(hipEventRecord(p->ev_fork, p->stream));
    MGRT_HIP(hipStreamWaitEvent(p->side, p->ev_fork, 0));
    if (mgrt_status st = exchange_on(p, L, field, p->side))
        return st;
    MGRT_HIP(hipEventRecord(p->ev_join, p->side));
    return MGRT_OK;
}

mgrt_status exchange_end(mgrt_slab_plan *p)
{
    if (overlaps(p))
        MGRT_HIP(hipStreamWaitEvent(p->stream, p->ev_join, 0));
    return MGRT_OK;
}

mgrt_status exchange_on(mgrt_slab_plan *p, mgrt_slab_plan::Level &L, void *field, hipStream_t on)
{
    if (p->world == 1)
        return MGRT_OK;
    const int peer[2] = {p->rank - 1, p->rank + 1};
    const int send_row[2] = {L.own0, L.own1 - kHalo}, recv_row[2] = {L.own0 - kHalo, L.own1};
    bool has[2] = {peer[0] >= 0, peer[1] < p->world};
    if (mgrt_status st = halo_rows<false>(p, L, field, L.sendbuf, has, send_row, on))
        return st;
    if (p->tr.kind == MGRT_TRANSPORT_RCCL) {
        // A group that was opened is always closed, whatever fails inside it: an error return between
        // the two would leave every later RCCL call of this thread nested in it and never issued.
        Rccl *nc = p->nc;
        MGRT_NCCL(nc->GroupStart());
        int err = 0;
        const char *what = "ncclSend";
        for (int k = 0; k < 2 && !err; ++k)
            if (has[k]) {
                err = nc->Send(L.sendbuf[k], L.halo_bytes, kNcclChar, peer[k], p->comm, on);
                if (!err)
                    what = "ncclRecv", err = nc->Recv(L.recvbuf[k], L.halo_bytes, kNcclChar, peer[k], p->comm, on);
            }
        const int end = nc->GroupEnd();
        if (!err && end)
            what = "ncclGroupEnd", err = end;
        if (err) {
            p->broken = true;
            return fail(MGRT_E_TRANSPORT, "mgrt_slab_plan: %s with rank %d or %d failed: %s", what, peer[0], peer[1],
                        nc->GetErrorString(err));
        }
    } else if (p->tr.kind == MGRT_TRANSPORT_CALLBACK) {
        int peers[2], n = 0;
        void *send[2], *recv[2];
        size_t bytes[2];
        for (int k = 0; k < 2; ++k)
            if (has[k])
                peers[n] = peer[k], send[n] = L.sendbuf[k], recv[n] = L.recvbuf[k], bytes[n] = L.halo_bytes, ++n;
        MGRT_HIP(hipStreamSynchronize(on));
        if (p->tr.sendrecv(p->tr.ctx, n, peers, send, recv, bytes) != 0)
            return fail(MGRT_E_TRANSPORT, "mgrt_slab_plan: the sendrecv callback failed");
    } else { // no wire: what would be received is what was packed (same bytes moved on the device)
        for (int k = 0; k < 2; ++k)
            if (has[k])
                MGRT_HIP(hipMemcpyAsync(L.recvbuf[k], L.sendbuf[k], L.halo_bytes, hipMemcpyDeviceToDevice, on));
    }
    return halo_rows<true>(p, L, field, L.recvbuf, has, recv_row, on);
}

// Every rank has written its own rows of an array of the first undistributed level (`whole`: c_rhs in a
// cycle); afterwards every rank holds all rows.
mgrt_status gather_coarse(mgrt_slab_plan *p, void *whole)
{
    const size_t w = elem(p), row = (size_t)p->cnx * w, plane = row * p->cny;
    const size_t part = (size_t)p->cnz * p->maxrows * row; // bytes per rank in the gather buffers
    const int a = p->crow0[p->rank], b = p->crow1[p->rank];
    MGRT_HIP(hipMemcpy2DAsync(p->gather_in, p->maxrows * row, (const char *)whole + a * row, plane, (b - a) * row, p->cnz,
                              hipMemcpyDeviceToDevice, p->stream));
    if (p->tr.kind == MGRT_TRANSPORT_RCCL) {
        Rccl *nc = p->nc;
        MGRT_NCCL(nc->AllGather(p->gather_in, p->gather_out, part, kNcclChar, p->comm, p->stream));
    } else if (p->tr.kind == MGRT_TRANSPORT_CALLBACK) {
        MGRT_HIP(hipStreamSynchronize(p->stream));
        if (p->tr.allgather(p->tr.ctx, p->gather_in, p->gather_out, part) != 0)
            return fail(MGRT_E_TRANSPORT, "mgrt_slab_plan: the allgather callback failed");
    } else {
        for (int r = 0; r < p->world; ++r)
            MGRT_HIP(hipMemcpyAsync((char *)p->gather_out + r * part, p->gather_in, part, hipMemcpyDeviceToDevice, p->stream));
    }
    for (int r = 0; r < p->world; ++r) {
        if (r == p->rank)
            continue;
        const int ra = p->crow0[r], rb = p->crow1[r];
        MGRT_HIP(hipMemcpy2DAsync((char *)whole + ra * row, plane, (const char *)p->gather_out + r * part, p->maxrows * row,
                                  (rb - ra) * row, p->cnz, hipMemcpyDeviceToDevice, p->stream));
    }
    return MGRT_OK;
}

template <class R> struct K;
template <> struct K<float> {
    static constexpr auto sweep = mgrt_smooth_sweep_f32;
    static constexpr auto sweep_res = mgrt_smooth_sweep_residual_f32;
    static constexpr auto correct_sweep = mgrt_correct_smooth_sweep_f32;
    static constexpr auto residual = mgrt_residual_f32;
    static constexpr auto restrict_ = mgrt_restrict_f32;
    static constexpr auto correct = mgrt_correct_f32;
    static constexpr auto prolong = mgrt_prolong_f32;
    static constexpr auto sumsq = mgrt_sumsq_f32;
};
template <> struct K<double> {
    static constexpr auto sweep = mgrt_smooth_sweep_f64;
    static constexpr auto sweep_res = mgrt_smooth_sweep_residual_f64;
    static constexpr auto correct_sweep = mgrt_correct_smooth_sweep_f64;
    static constexpr auto residual = mgrt_residual_f64;
    static constexpr auto restrict_ = mgrt_restrict_f64;
    static constexpr auto correct = mgrt_correct_f64;
    static constexpr auto prolong = mgrt_prolong_f64;
    static constexpr auto sumsq = mgrt_sumsq_f64;
};

int zcount_of(const mgrt_slab_plan *p, const mgrt_slab_plan::Level &L)
{
    return p->desc.zfull ? L.nz : L.nz / 2 + L.nz % 2; // multigridsolver3D.cpp:121
}

// `a` iterations as sweeps of two (and one odd one), each preceded by the exchange its halo needs.
// With `res` the last sweep also leaves f - A u there, good on the rows whose stencil stays inside the
// owned rows (the rows next to them were smoothed from halo rows that are stale by then).
// `zero`: the field is known to be zero everywhere and `cur` has not been cleared; the first sweep then
// reads no input at all (a null input stands for zeros), as in the one-GPU plan.
// With `cg` (and `res`) that sweep also restricts that residual: rows [c0, c1) of the coarse array crhs, the fine
// residual staying in registers (only the rows next to tile seams pass through res).
template <class R>
mgrt_status smooth(mgrt_slab_plan *p, mgrt_slab_plan::Level &L, R *&cur, R *&alt, const R *f, R omega, int a, bool &halo_valid,
                   R *res = nullptr, bool zero = false, const mgrt_grid *cg = nullptr, R *crhs = nullptr, int c0 = 0, int c1 = 0)
{
    mgrt_stream s = (mgrt_stream)p->stream;
    for (int left = a; left > 0;) {
        const int n = left >= 2 ? 2 : 1;
        const R *in = zero ? nullptr : cur;
        if (!halo_valid && !zero)
            if (mgrt_status st = exchange(p, L, cur))
                return st;
        zero = false;
        if (res && cg && left == n) {
            if constexpr (sizeof(R) == 4) {
                if (mgrt_status st = sweep_residual_restrict_rows(&L.g, cg, alt, crhs, res, in, f, omega, zcount_of(p, L), c0, c1,
                                                                  p->stream))
                    return st;
            } else {
                return fail(MGRT_E_UNSUPPORTED, "mgrt_slab_plan: no restriction-carrying sweep for this element size");
            }
        } else if (res && left == n) {
            if (mgrt_status st = K<R>::sweep_res(&L.g, alt, res, in, f, omega, n, zcount_of(p, L), s))
                return st;
        } else if (mgrt_status st = K<R>::sweep(&L.g, alt, in, f, omega, n, zcount_of(p, L), s))
            return st;
        R *t = cur;
        cur = alt, alt = t;
        halo_valid = false;
        left -= n;
    }
    return MGRT_OK;
}

// iterate() on distributed level l; u and f are this rank's rows.  *result is the array that holds
// the outcome (u or the level's work array).
template <class R>
mgrt_status iterate(mgrt_slab_plan *p, int l, R *u, const R *f, R omega, int a1, int a2, int v, bool &halo_valid, R **result,
                    bool u_is_zero = false)
{
    mgrt_slab_plan::Level &L = p->lv[l];
    mgrt_stream s = (mgrt_stream)p->stream;
    R *cur = u, *alt = (R *)L.work;
    if (u_is_zero && a1 == 0) { // nothing reads the zeros through a sweep first: write them
        if (mgrt_status st = mgrt_zero(u, L.n * sizeof(R), s))
            return st;
        u_is_zero = false;
    }
    // As on one GPU, the last pre-smoothing sweep of a bandwidth-bound level in single precision carries
    // the residual pass along (slabs too thin for strips of four rows: measured slower than the two kernels).
    const bool res_in_sweep = a1 > 0 && v > 0 && sizeof(R) == 4 && !latency_bound_level(L.g) && L.own1 - L.own0 > 2;
    const bool last = l + 1 == p->ndist;
    // where the restricted residual goes: this rank's rows [c0, c1) of the next level
    const mgrt_grid *cg = last ? &p->cgrid : &p->lv[l + 1].g;
    R *crhs = last ? (R *)p->c_rhs : (R *)p->lv[l + 1].rhs;
    const int c0 = last ? p->crow0[p->rank] : p->lv[l + 1].own0, c1 = last ? p->crow1[p->rank] : p->lv[l + 1].own1;
    // ... and the restriction of the coarse rows in between as well, where that sweep does two iterations: the
    // residual then stays in registers (the first and the last coarse row follow from residual rows of their own)
    const bool fusable = c1 - c0 > 2 && L.g.nx % 2 == 1 && L.g.gny % 2 == 1 && L.nz % 2 == 1 && L.nz >= 5 && L.g.y0 % 2 == 0 &&
                         plane_offsets_fit(&L.g, sizeof(R));
    const bool rst_in_sweep = res_in_sweep && fusable && a1 % 2 == 0;
    if (mgrt_status st = smooth<R>(p, L, cur, alt, f, omega, a1, halo_valid, res_in_sweep ? (R *)L.r : nullptr, u_is_zero,
                                   rst_in_sweep ? cg : nullptr, crhs, c0 + 1, c1 - 1))
        return st;
    for (int it = 0; it < v; ++it) {
        // residual on the owned rows and one more on each side: what the restriction reads.  The rows
        // whose stencil stays inside the owned rows are formed while the halo of the unknown travels,
        // the two rows at either end afterwards.
        const int rlo = L.y0 > 0 ? L.own0 - 1 : L.own0, rhi = L.y1 < L.gny ? L.own1 + 1 : L.own1;
        const int f0 = 2 * (c0 + cg->y0) - L.g.y0, f1 = 2 * (c1 - 1 + cg->y0) - L.g.y0; // fine rows of coarse rows c0, c1 - 1
        bool restricted = false;
        if (rst_in_sweep && it == 0) {
            // the sweep restricted the coarse rows in between; the first and the last need the neighbours' rows
            if (mgrt_status st = exchange(p, L, cur))
                return st;
            halo_valid = true;
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, f0 - 1 > rlo ? f0 - 1 : rlo, f0 + 2, s))
                return st;
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, f1 - 1, f1 + 2 < rhi ? f1 + 2 : rhi, s))
                return st;
            if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c0, c0 + 1, s))
                return st;
            if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c1 - 1, c1, s))
                return st;
            restricted = true;
        } else if (res_in_sweep && it == 0) {
            // The sweep left the rows in between; the two at either end need the neighbours' rows.  While
            // those travel, the coarse rows that only read what the sweep left are restricted.
            if (mgrt_status st = exchange_begin(p, L, cur))
                return st;
            const bool split = c1 - c0 > 2;
            if (split)
                if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c0 + 1, c1 - 1, s))
                    return st;
            if (mgrt_status st = exchange_end(p))
                return st;
            halo_valid = true;
            const int ilo = L.own0 + 1, ihi = L.own1 - 1;
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, rlo, ilo, s))
                return st;
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, ihi, rhi, s))
                return st;
            if (split) {
                if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c0, c0 + 1, s))
                    return st;
                if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c1 - 1, c1, s))
                    return st;
                restricted = true;
            }
        } else if (fusable && !latency_bound_level(L.g)) {
            // Bandwidth-bound level without a residual from the sweep (double precision): residual and
            // restriction as one march, the fine residual in registers.  The coarse rows in between read owned
            // rows only and are formed while the halo of the unknown travels; the first and the last follow
            // from a few residual rows of their own.
            if (!halo_valid)
                if (mgrt_status st = exchange_begin(p, L, cur))
                    return st;
            if (mgrt_status st = residual_restrict_march(&L.g, cg, crhs, (R *)L.r, cur, f, c0 + 1, c1 - 1, p->stream))
                return st;
            if (!halo_valid) {
                if (mgrt_status st = exchange_end(p))
                    return st;
                halo_valid = true;
            }
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, f0 - 1 > rlo ? f0 - 1 : rlo, f0 + 2, s))
                return st;
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, f1 - 1, f1 + 2 < rhi ? f1 + 2 : rhi, s))
                return st;
            if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c0, c0 + 1, s))
                return st;
            if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c1 - 1, c1, s))
                return st;
            restricted = true;
        } else if (!halo_valid && L.own1 - L.own0 > 2) {
            if (mgrt_status st = exchange_begin(p, L, cur))
                return st;
            const int ilo = L.own0 + 1, ihi = L.own1 - 1;
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, ilo, ihi, s))
                return st;
            if (mgrt_status st = exchange_end(p))
                return st;
            halo_valid = true;
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, rlo, ilo, s))
                return st;
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, ihi, rhi, s))
                return st;
        } else {
            if (!halo_valid) {
                if (mgrt_status st = exchange(p, L, cur))
                    return st;
                halo_valid = true;
            }
            if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, cur, f, 1, nullptr, rlo, rhi, s))
                return st;
        }
        const R *e_local;
        const mgrt_grid *egrid;
        if (last) {
            if (!restricted)
                if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c0, c1, s))
                    return st;
            if (mgrt_status st = gather_coarse(p, p->c_rhs))
                return st;
            if (mgrt_status st = mgrt_zero(p->c_e, (size_t)p->cnx * p->cny * p->cnz * sizeof(R), s))
                return st;
            if (mgrt_status st = mgrt_plan_vcycle(p->coarse, p->c_e, p->c_rhs, omega, a1, a2, v, 1, nullptr))
                return st;
            e_local = (const R *)p->c_e, egrid = &p->cgrid;
        } else {
            mgrt_slab_plan::Level &C = p->lv[l + 1];
            if (!restricted)
                if (mgrt_status st = K<R>::restrict_(&L.g, cg, crhs, (const R *)L.r, c0, c1, s))
                    return st;
            if (mgrt_status st = exchange(p, C, C.rhs))
                return st;
            // the coarse unknown starts from zero (zero_mem, multigridsolver3D.cpp:59) without being cleared
            bool ok = true;
            R *res = nullptr;
            if (mgrt_status st = iterate<R>(p, l + 1, (R *)C.e, (const R *)C.rhs, omega, a1, a2, v, ok, &res, true))
                return st;
            if (!ok)
                if (mgrt_status st = exchange(p, C, res))
                    return st;
            e_local = res, egrid = &C.g;
        }
        if (it + 1 == v && a2 >= 2 && mgrt_correct_smooth_sweep_supported(&L.g, (int)sizeof(R), 2)) {
            // the last correction rides in the first post-smoothing sweep: same values, and the
            // corrected field is never written (halo rows of cur and e_local are valid here)
            if (mgrt_status st = K<R>::correct_sweep(&L.g, egrid, alt, cur, e_local, f, omega, 2, zcount_of(p, L), s))
                return st;
            R *t = cur;
            cur = alt, alt = t;
            halo_valid = false;
            a2 -= 2;
        } else {
            // every local row is corrected, halo rows included: they stay valid without an exchange
            if (mgrt_status st = K<R>::correct(&L.g, egrid, cur, cur, e_local, 0, L.nyl, s))
                return st;
        }
    }
    if (mgrt_status st = smooth<R>(p, L, cur, alt, f, omega, a2, halo_valid))
        return st;
    *result = cur;
    return MGRT_OK;
}

template <class R> mgrt_status cycle_body(mgrt_slab_plan *p, R *u, const R *f, R omega, int a1, int a2, int v)
{
    bool halo_valid = false;
    R *res = nullptr;
    if (mgrt_status st = iterate<R>(p, 0, u, f, omega, a1, a2, v, halo_valid, &res))
        return st;
    if (res != u)
        MGRT_HIP(hipMemcpyAsync(u, res, p->lv[0].n * sizeof(R), hipMemcpyDeviceToDevice, p->stream));
    return MGRT_OK;
}

// fmg() of multigridsolver3D.cpp:82-109 on distributed level l: this rank's rows of the solution into u_out,
// f with valid halo rows.  Launched directly (it runs once per solve).
template <class R>
mgrt_status fmg_level(mgrt_slab_plan *p, int l, R *u_out, const R *f, R omega, int a1, int a2, int v, int iters)
{
    mgrt_slab_plan::Level &L = p->lv[l];
    mgrt_stream s = (mgrt_stream)p->stream;
    if (l + 1 == p->ndist) {
        const size_t cn = (size_t)p->cnx * p->cny * p->cnz * sizeof(R);
        if (!p->c_fmg_f) {
            MGRT_HIP(hipMalloc(&p->c_fmg_f, cn));
            MGRT_HIP(hipMalloc(&p->c_fmg_u, cn));
        }
        // the right-hand side itself is restricted, unscaled (:98-99); every rank its rows, then all of them
        if (mgrt_status st = K<R>::restrict_(&L.g, &p->cgrid, (R *)p->c_fmg_f, f, p->crow0[p->rank], p->crow1[p->rank], s))
            return st;
        if (mgrt_status st = gather_coarse(p, p->c_fmg_f))
            return st;
        // the recursive call omits iters, so coarse levels run the default of 3 (:101)
        if (mgrt_status st = mgrt_plan_fmg(p->coarse, p->c_fmg_u, p->c_fmg_f, omega, a1, a2, v, 3))
            return st;
        if (mgrt_status st = K<R>::prolong(&L.g, &p->cgrid, u_out, (const R *)p->c_fmg_u, 0, L.nyl, s))
            return st;
    } else {
        mgrt_slab_plan::Level &C = p->lv[l + 1];
        if (!C.fmg_f) {
            MGRT_HIP(hipMalloc(&C.fmg_f, C.n * sizeof(R)));
            MGRT_HIP(hipMalloc(&C.fmg_u, C.n * sizeof(R)));
            MGRT_HIP(hipMemsetAsync(C.fmg_f, 0, C.n * sizeof(R), p->stream)); // halo rows nobody fills start defined
        }
        if (mgrt_status st = K<R>::restrict_(&L.g, &C.g, (R *)C.fmg_f, f, C.own0, C.own1, s))
            return st;
        if (mgrt_status st = exchange(p, C, C.fmg_f))
            return st;
        if (mgrt_status st = fmg_level<R>(p, l + 1, (R *)C.fmg_u, (const R *)C.fmg_f, omega, a1, a2, v, 3))
            return st;
        if (mgrt_status st = exchange(p, C, C.fmg_u))
            return st;
        if (mgrt_status st = K<R>::prolong(&L.g, &C.g, u_out, (const R *)C.fmg_u, 0, L.nyl, s))
            return st;
    }
    // every local row was interpolated, halo rows included
    bool halo_valid = true;
    for (int i = 0; i < iters; ++i) { // mg(), multigridsolver3D.h:49-59
        R *res = nullptr;
        if (mgrt_status st = iterate<R>(p, l, u_out, f, omega, a1, a2, v, halo_valid, &res))
            return st;
        if (res != u_out)
            MGRT_HIP(hipMemcpyAsync(u_out, res, L.n * sizeof(R), hipMemcpyDeviceToDevice, p->stream));
    }
    return MGRT_OK;
}

void drop_graph(mgrt_slab_plan *p)
{
    if (p->exec) {
        (void)hipGraphExecDestroy(p->exec);
        p->exec = nullptr;
    }
}

// One cycle: replayed from a graph of this rank's kernels, copies and RCCL operations where the
// transport can be captured; launched directly otherwise, and for good once a capture was refused.
template <class R> mgrt_status run_cycle(mgrt_slab_plan *p, R *u, const R *f, R omega, int a1, int a2, int v)
{
    if (p->broken)
        return fail(MGRT_E_TRANSPORT, "mgrt_slab_plan: a transport call failed earlier; create a new plan");
    if (!p->desc.use_graph || !capturable(p) || p->graph_failed || !p->warmed) {
        p->warmed = true;
        return cycle_body<R>(p, u, f, omega, a1, a2, v);
    }
    const mgrt_slab_plan::Key key{u, f, (double)omega, a1, a2, v};
    const bool same = p->exec && p->key.u == key.u && p->key.f == key.f && p->key.omega == key.omega && p->key.a1 == key.a1 &&
                      p->key.a2 == key.a2 && p->key.v == key.v;
    if (!same) {
        drop_graph(p);
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            p->graph_failed = true;
            return cycle_body<R>(p, u, f, omega, a1, a2, v);
        }
        const mgrt_status st = cycle_body<R>(p, u, f, omega, a1, a2, v);
        const hipError_t e = hipStreamEndCapture(p->stream, &graph);
        hipError_t ei = e;
        if (st == MGRT_OK && e == hipSuccess)
            ei = hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0);
        if (graph)
            (void)hipGraphDestroy(graph);
        if (st != MGRT_OK || ei != hipSuccess) {
            // Nothing of the captured cycle has run.  A failed transport call ends the plan (its peers are
            // out of step); anything else: run the cycle directly, now and from here on, once the side
            // stream has left the abandoned capture as well.
            (void)hipGetLastError();
            p->exec = nullptr;
            p->graph_failed = true;
            if (p->broken || st == MGRT_E_TRANSPORT) {
                p->broken = true;
                return st != MGRT_OK ? st : fail(MGRT_E_TRANSPORT, "mgrt_slab_plan: a transport call failed inside a capture");
            }
            if (p->side)
                (void)hipStreamSynchronize(p->side);
            (void)hipStreamSynchronize(p->stream);
            (void)hipGetLastError();
            return cycle_body<R>(p, u, f, omega, a1, a2, v);
        }
        p->key = key;
    }
    MGRT_HIP(hipGraphLaunch(p->exec, p->stream));
    return MGRT_OK;
}

// ||f - A u||^2 over the whole grid into d_norms[i]
template <class R> mgrt_status record_norm(mgrt_slab_plan *p, R *u, const R *f, int i)
{
    mgrt_stream s = (mgrt_stream)p->stream;
    if (!p->ndist)
        return fail(MGRT_E_INVALID, "mgrt_slab_plan: internal: norm of an undistributed plan");
    mgrt_slab_plan::Level &L = p->lv[0];
    if (mgrt_status st = exchange(p, L, u))
        return st;
    if (mgrt_status st = K<R>::residual(&L.g, (R *)L.r, u, f, 1, nullptr, L.own0, L.own1, s))
        return st;
    if (mgrt_status st = K<R>::sumsq(&L.g, (const R *)L.r, p->d_norms + i, L.own0, L.own1, s))
        return st;
    if (p->world > 1) {
        if (p->tr.kind == MGRT_TRANSPORT_RCCL) {
            Rccl *nc = p->nc;
            MGRT_NCCL(nc->AllReduce(p->d_norms + i, p->d_norms + i, 1, kNcclDouble, kNcclSum, p->comm, p->stream));
        } else if (p->tr.kind == MGRT_TRANSPORT_CALLBACK) {
            MGRT_HIP(hipStreamSynchronize(p->stream));
            if (p->tr.allreduce_sum(p->tr.ctx, p->d_norms + i) != 0)
                return fail(MGRT_E_TRANSPORT, "mgrt_slab_plan: the allreduce callback failed");
        }
    }
    return MGRT_OK;
}

template <class R>
mgrt_status vcycle(mgrt_slab_plan *p, R *u, const R *f, R omega, int a1, int a2, int v, int ncycles, double *hist)
{
    if (!p->ndist) // too small to cut (or one rank): the whole grid with the one-GPU plan
        return mgrt_plan_vcycle(p->coarse, u, f, omega, a1, a2, v, ncycles, hist);
    if (hist && p->norms_cap < ncycles + 1) {
        (void)hipFree(p->d_norms);
        p->d_norms = nullptr;
        MGRT_HIP(hipMalloc((void **)&p->d_norms, sizeof(double) * (size_t)(ncycles + 1)));
        p->norms_cap = ncycles + 1;
    }
    if (hist)
        if (mgrt_status st = record_norm<R>(p, u, f, 0))
            return st;
    for (int c = 0; c < ncycles; ++c) {
        if (mgrt_status st = run_cycle<R>(p, u, f, omega, a1, a2, v))
            return st;
        if (hist)
            if (mgrt_status st = record_norm<R>(p, u, f, c + 1))
                return st;
    }
    if (hist) {
        MGRT_HIP(hipMemcpyAsync(hist, p->d_norms, sizeof(double) * (size_t)(ncycles + 1), hipMemcpyDeviceToHost, p->stream));
        MGRT_HIP(hipStreamSynchronize(p->stream));
        for (int i = 0; i <= ncycles; ++i)
            hist[i] = sqrt(hist[i]);
    }
    return MGRT_OK;
}

} // namespace

extern "C" {

mgrt_status mgrt_rccl_version(int *version)
{
    MGRT_REQUIRE(version != nullptr, "mgrt_rccl_version: null argument");
    Rccl *nc = nullptr;
    if (mgrt_status st = rccl(&nc))
        return st;
    if (!nc->GetVersion)
        return fail(MGRT_E_UNSUPPORTED, "mgrt_rccl_version: this RCCL has no ncclGetVersion");
    MGRT_NCCL(nc->GetVersion(version));
    return MGRT_OK;
}

mgrt_status mgrt_rccl_unique_id(unsigned char id[128])
{
    MGRT_REQUIRE(id != nullptr, "mgrt_rccl_unique_id: null argument");
    Rccl *nc = nullptr;
    if (mgrt_status st = rccl(&nc)) return st;
    MGRT_NCCL(nc->GetUniqueId(id));
    return MGRT_OK;
}

// The RCCL operations of a slab plan on a communicator of one rank: a send/receive pair with itself in
// one group, an all-gather and an all-reduce, queued directly and then replayed from a captured graph,
// each result checked.  What a single GPU can tell about the binding (symbols, the id passed by value,
// the enum values) and about capturing RCCL work on the installed runtime.
mgrt_status mgrt_rccl_selftest(mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(s != nullptr, "mgrt_rccl_selftest: a stream of its own is required (the default stream cannot be captured)");
    Rccl *nc = nullptr;
    if (mgrt_status st = rccl(&nc)) return st;
    hipStream_t hs = (hipStream_t)s;
    constexpr size_t kBytes = 4096;
    mgrt_rccl_id_by_value id;
    MGRT_NCCL(nc->GetUniqueId(&id));
    void *comm = nullptr;
    MGRT_NCCL(nc->CommInitRank(&comm, 1, id, 0));
    unsigned char *a = nullptr, *b = nullptr, *c = nullptr;
    double *d = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    auto queue = [&]() -> mgrt_status {
        MGRT_NCCL(nc->GroupStart());
        int err = nc->Send(a, kBytes, kNcclChar, 0, comm, hs);
        if (!err)
            err = nc->Recv(b, kBytes, kNcclChar, 0, comm, hs);
        const int end = nc->GroupEnd(); // always closed, see exchange_on
        if (err || end)
            return fail(MGRT_E_TRANSPORT, "mgrt_rccl_selftest: send/receive group: %s", nc->GetErrorString(err ? err : end));
        MGRT_NCCL(nc->AllGather(b, c, kBytes, kNcclChar, comm, hs));
        MGRT_NCCL(nc->AllReduce(d, d + 1, 1, kNcclDouble, kNcclSum, comm, hs));
        return MGRT_OK;
    };
    auto check = [&](int fill, double value, const char *how) -> mgrt_status {
        unsigned char back[kBytes];
        double sum = 0;
        MGRT_HIP(hipMemcpyAsync(back, c, kBytes, hipMemcpyDeviceToHost, hs));
        MGRT_HIP(hipMemcpyAsync(&sum, d + 1, sizeof sum, hipMemcpyDeviceToHost, hs));
        MGRT_HIP(hipStreamSynchronize(hs));
        for (size_t i = 0; i < kBytes; ++i)
            if (back[i] != (unsigned char)fill)
                return fail(MGRT_E_TRANSPORT, "mgrt_rccl_selftest (%s): byte %zu came back as %d, not %d", how, i, back[i], fill);
        if (sum != value)
            return fail(MGRT_E_TRANSPORT, "mgrt_rccl_selftest (%s): all-reduce gave %g, not %g", how, sum, value);
        return MGRT_OK;
    };
    auto fill = [&](int byte, double value) -> mgrt_status {
        MGRT_HIP(hipMemsetAsync(a, byte, kBytes, hs));
        MGRT_HIP(hipMemsetAsync(b, 0, kBytes, hs));
        MGRT_HIP(hipMemsetAsync(c, 0, kBytes, hs));
        MGRT_HIP(hipMemcpyAsync(d, &value, sizeof value, hipMemcpyHostToDevice, hs));
        MGRT_HIP(hipMemsetAsync(d + 1, 0, sizeof(double), hs));
        MGRT_HIP(hipStreamSynchronize(hs)); // `value` is on this frame
        return MGRT_OK;
    };
    auto run = [&]() -> mgrt_status {
        MGRT_HIP(hipMalloc((void **)&a, kBytes));
        MGRT_HIP(hipMalloc((void **)&b, kBytes));
        MGRT_HIP(hipMalloc((void **)&c, kBytes));
        MGRT_HIP(hipMalloc((void **)&d, 2 * sizeof(double)));
        if (mgrt_status st = fill(0x5a, 1.25)) return st;
        if (mgrt_status st = queue()) return st;
        if (mgrt_status st = check(0x5a, 1.25, "direct")) return st;
        if (mgrt_status st = fill(0xa5, -3.5)) return st;
        MGRT_HIP(hipStreamBeginCapture(hs, hipStreamCaptureModeThreadLocal));
        const mgrt_status st = queue();
        const hipError_t e = hipStreamEndCapture(hs, &graph);
        if (st != MGRT_OK) return st;
        MGRT_HIP(e);
        MGRT_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        MGRT_HIP(hipGraphLaunch(exec, hs));
        if (mgrt_status st2 = check(0xa5, -3.5, "replayed from a graph")) return st2;
        if (mgrt_status st2 = fill(0x3c, 7.0)) return st2;
        MGRT_HIP(hipGraphLaunch(exec, hs));
        return check(0x3c, 7.0, "replayed a second time");
    };
    const mgrt_status st = run();
    (void)hipStreamSynchronize(hs);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    for (void *q : {(void *)a, (void *)b, (void *)c, (void *)d})
        (void)hipFree(q);
    (void)nc->CommDestroy(comm);
    return st;
}

mgrt_status mgrt_slab_plan_create(mgrt_slab_plan **plan, const mgrt_slab_desc *d, const mgrt_transport *t, mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(plan && d && t, "mgrt_slab_plan_create: null argument");
    *plan = nullptr;
    MGRT_REQUIRE(d->elem_size == 4 || d->elem_size == 8, "mgrt_slab_plan_create: elem_size must be 4 or 8");
    MGRT_REQUIRE(d->nx > 0 && d->ny > 0 && d->nz > 0, "mgrt_slab_plan_create: extents must be positive");
    MGRT_REQUIRE(t->nranks >= 1 && t->rank >= 0 && t->rank < t->nranks, "mgrt_slab_plan_create: rank %d of %d", t->rank, t->nranks);
    MGRT_REQUIRE(t->kind == MGRT_TRANSPORT_NONE || t->kind == MGRT_TRANSPORT_RCCL || t->kind == MGRT_TRANSPORT_CALLBACK,
                 "mgrt_slab_plan_create: unknown transport %d", t->kind);
    MGRT_REQUIRE(t->kind != MGRT_TRANSPORT_CALLBACK || (t->sendrecv && t->allgather && t->allreduce_sum),
                 "mgrt_slab_plan_create: a callback is missing");
    MGRT_REQUIRE(s != nullptr, "mgrt_slab_plan_create: a stream of its own is required (the default stream cannot be captured)");

    mgrt_slab_plan *p = new mgrt_slab_plan;
    p->desc = *d;
    p->tr = *t;
    p->stream = (hipStream_t)s;
    p->rank = t->rank, p->world = t->nranks;
    auto bail = [&](mgrt_status st) {
        mgrt_slab_plan_destroy(p);
        return st;
    };
    if (t->kind == MGRT_TRANSPORT_RCCL && p->world > 1) {
        if (mgrt_status st = rccl(&p->nc)) return bail(st);
        mgrt_rccl_id_by_value id;
        memcpy(id.internal, t->rccl_id, sizeof id.internal);
        const int e = p->nc->CommInitRank(&p->comm, p->world, id, p->rank);
        if (e != 0)
            return bail(fail(MGRT_E_TRANSPORT, "ncclCommInitRank: %s", p->nc->GetErrorString(e)));
    }
    if (p->world > 1) {
        hipError_t es = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
        if (es == hipSuccess) es = hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming);
        if (es == hipSuccess) es = hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming);
        if (es != hipSuccess)
            return bail(fail((mgrt_status)es, "mgrt_slab_plan_create: %s", hipGetErrorString(es)));
    }
    p->ndist = distributed_levels(d->nx, d->ny, d->nz, p->world, d->crossover);
    const int align = 1 << p->ndist;
    const int q = (d->ny - 1) / (p->world * align);
    for (int r = 0; r < p->world; ++r)
        p->bounds.push_back(r * q * align);
    p->bounds.push_back(d->ny);

    const size_t w = (size_t)d->elem_size;
    int nx = d->nx, ny = d->ny, nz = d->nz;
    for (int l = 0; l < p->ndist; ++l) {
        mgrt_slab_plan::Level L;
        L.nx = nx, L.gny = ny, L.nz = nz;
        L.y0 = p->bounds[p->rank] >> l;
        L.y1 = p->rank + 1 < p->world ? p->bounds[p->rank + 1] >> l : ny;
        L.lo = L.y0 - kHalo > 0 ? L.y0 - kHalo : 0;
        L.hi = L.y1 + kHalo < ny ? L.y1 + kHalo : ny;
        L.nyl = L.hi - L.lo;
        L.own0 = L.y0 - L.lo, L.own1 = L.y1 - L.lo;
        L.g.dim = 3, L.g.nx = nx, L.g.ny = L.nyl, L.g.nz = nz;
        L.g.pitch = nx, L.g.slab = (long long)nx * L.nyl;
        L.g.y0 = L.lo, L.g.gny = ny, L.g.cells = nullptr, L.g.neumann = 0;
        for (int i = 0; i < 6; ++i)
            L.g.face[i] = d->face[i];
        L.n = (size_t)nx * L.nyl * nz;
        L.halo_bytes = (size_t)nz * kHalo * nx * w;
        p->lv.push_back(L);
        nx = coarse_extent(nx), ny = coarse_extent(ny), nz = coarse_extent(nz);
    }
    p->cnx = nx, p->cny = ny, p->cnz = nz;
    for (int r = 0; r < p->world; ++r) {
        p->crow0.push_back(p->bounds[r] >> p->ndist);
        p->crow1.push_back(r + 1 < p->world ? p->bounds[r + 1] >> p->ndist : ny);
        if (p->crow1[r] - p->crow0[r] > p->maxrows)
            p->maxrows = p->crow1[r] - p->crow0[r];
    }
    hipError_t e = hipSuccess;
    for (auto &L : p->lv) {
        const bool inner = &L != &p->lv[0];
        for (void **a : {&L.work, &L.r})
            if (e == hipSuccess) e = hipMalloc(a, L.n * w);
        if (inner)
            for (void **a : {&L.rhs, &L.e})
                if (e == hipSuccess) e = hipMalloc(a, L.n * w);
        for (int k = 0; k < 2; ++k) {
            if (e == hipSuccess) e = hipMalloc(&L.sendbuf[k], L.halo_bytes);
            if (e == hipSuccess) e = hipMalloc(&L.recvbuf[k], L.halo_bytes);
        }
        // halo rows nobody fills (beyond the first and last slab there are none) start defined
        if (e == hipSuccess) e = hipMemsetAsync(L.work, 0, L.n * w, p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(L.r, 0, L.n * w, p->stream);
        if (inner && e == hipSuccess) e = hipMemsetAsync(L.rhs, 0, L.n * w, p->stream);
    }
    if (p->ndist) {
        const size_t cn = (size_t)p->cnx * p->cny * p->cnz * w, part = (size_t)p->cnz * p->maxrows * p->cnx * w;
        if (e == hipSuccess) e = hipMalloc(&p->c_rhs, cn);
        if (e == hipSuccess) e = hipMalloc(&p->c_e, cn);
        if (e == hipSuccess) e = hipMalloc(&p->gather_in, part);
        if (e == hipSuccess) e = hipMalloc(&p->gather_out, part * p->world);
        if (e == hipSuccess) e = hipMemsetAsync(p->gather_in, 0, part, p->stream);
        mgrt_grid &c = p->cgrid;
        c.dim = 3, c.nx = p->cnx, c.ny = p->cny, c.nz = p->cnz, c.pitch = p->cnx, c.slab = (long long)p->cnx * p->cny;
        c.y0 = 0, c.gny = p->cny, c.cells = nullptr, c.neumann = 0;
        for (int i = 0; i < 6; ++i)
            c.face[i] = d->face[i];
    }
    if (e != hipSuccess)
        return bail(fail((mgrt_status)e, "mgrt_slab_plan_create: %s", hipGetErrorString(e)));
    mgrt_plan_desc cd = {};
    cd.dim = 3, cd.nx = p->cnx, cd.ny = p->cny, cd.nz = p->cnz, cd.elem_size = d->elem_size;
    for (int i = 0; i < 6; ++i)
        cd.face[i] = d->face[i];
    cd.zfull = d->zfull;
    // inside a captured distributed cycle the coarse levels are part of that capture
    cd.use_graph = p->ndist ? 0 : d->use_graph;
    if (mgrt_status st = mgrt_plan_create(&p->coarse, &cd, s))
        return bail(st);
    double *parts = nullptr;
    if (mgrt_status st = partials_for(p->stream, &parts))
        return bail(st);
    MGRT_HIP(hipStreamSynchronize(p->stream));
    *plan = p;
    return MGRT_OK;
}

mgrt_status mgrt_slab_plan_destroy(mgrt_slab_plan *p)
{
    if (!p)
        return MGRT_OK;
    (void)hipStreamSynchronize(p->stream);
    drop_graph(p);
    (void)mgrt_plan_destroy(p->coarse);
    for (auto &L : p->lv) {
        for (void *a : {L.work, L.r, L.rhs, L.e, L.fmg_f, L.fmg_u, L.sendbuf[0], L.sendbuf[1], L.recvbuf[0], L.recvbuf[1]})
            (void)hipFree(a);
    }
    for (void *a : {p->c_rhs, p->c_e, p->c_fmg_f, p->c_fmg_u, p->gather_in, p->gather_out, (void *)p->d_norms})
        (void)hipFree(a);
    if (p->comm && p->nc)
        (void)p->nc->CommDestroy(p->comm);
    if (p->side) {
        (void)hipStreamSynchronize(p->side);
        (void)hipStreamDestroy(p->side);
    }
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    delete p;
    return MGRT_OK;
}

int mgrt_slab_plan_distributed_levels(const mgrt_slab_plan *p) { return p ? p->ndist : 0; }

int mgrt_slab_plan_uses_graph(const mgrt_slab_plan *p)
{
    return p && p->ndist && p->desc.use_graph && capturable(p) && !p->graph_failed;
}

mgrt_status mgrt_slab_plan_rows(const mgrt_slab_plan *p, int *lo, int *hi, int *own_lo, int *own_hi)
{
    MGRT_REQUIRE(p != nullptr, "mgrt_slab_plan_rows: null plan");
    int a = 0, b = p->desc.ny, c = 0, d = p->desc.ny;
    if (p->ndist) {
        const mgrt_slab_plan::Level &L = p->lv[0];
        a = L.lo, b = L.hi, c = L.y0, d = L.y1;
    }
    if (lo) *lo = a;
    if (hi) *hi = b;
    if (own_lo) *own_lo = c;
    if (own_hi) *own_hi = d;
    return MGRT_OK;
}

mgrt_status mgrt_slab_plan_vcycle(mgrt_slab_plan *p, void *u, const void *f, double omega, int a1, int a2, int v, int ncycles,
                                  double *res_hist)
{
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(p && u && f, "mgrt_slab_plan_vcycle: null argument");
    MGRT_REQUIRE(u != f, "mgrt_slab_plan_vcycle: u and f are the same array");
    MGRT_REQUIRE(a1 >= 0 && a2 >= 0 && v >= 0 && ncycles >= 0, "mgrt_slab_plan_vcycle: negative count");
    if (p->broken)
        return fail(MGRT_E_TRANSPORT, "mgrt_slab_plan_vcycle: a transport call failed earlier; create a new plan");
    if (p->desc.elem_size == 4)
        return vcycle<float>(p, (float *)u, (const float *)f, (float)omega, a1, a2, v, ncycles, res_hist);
    return vcycle<double>(p, (double *)u, (const double *)f, omega, a1, a2, v, ncycles, res_hist);
}

mgrt_status mgrt_slab_plan_fmg(mgrt_slab_plan *p, void *u_out, const void *f, double omega, int a1, int a2, int v, int iters)
{
    if (mgrt_status st = require_init()) return st;
    MGRT_REQUIRE(p && u_out && f, "mgrt_slab_plan_fmg: null argument");
    MGRT_REQUIRE(u_out != f, "mgrt_slab_plan_fmg: u_out and f are the same array");
    MGRT_REQUIRE(a1 >= 0 && a2 >= 0 && v >= 0 && iters >= 0, "mgrt_slab_plan_fmg: negative count");
    if (p->broken)
        return fail(MGRT_E_TRANSPORT, "mgrt_slab_plan_fmg: a transport call failed earlier; create a new plan");
    if (!p->ndist) // too small to cut (or one rank): the whole grid with the one-GPU plan
        return mgrt_plan_fmg(p->coarse, u_out, f, omega, a1, a2, v, iters);
    if (p->desc.elem_size == 4)
        return fmg_level<float>(p, 0, (float *)u_out, (const float *)f, (float)omega, a1, a2, v, iters);
    return fmg_level<double>(p, 0, (double *)u_out, (const double *)f, omega, a1, a2, v, iters);
}

} // extern "C"
