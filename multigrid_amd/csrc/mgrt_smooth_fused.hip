// mgrt_smooth_fused.hip -- the 3D smoother as one z-march that performs whole red-black
// iterations per sweep: `out = S^iters(in)` with every element of `in` and `f` read once and every
// element of `out` written once (plus tile halos, which are L2 hits).
//
// Why this is legal.  The reference's 3D colouring ignores z (mg_1.cl:98), so inside one colour
// pass each (x,y) column is a recurrence in ascending z, and the four in-plane neighbours of a
// point always have the other colour.  Hence at plane z the first colour needs only the other
// colour's previous values of plane z, its own new value at z-1 and its own old value at z+1; the
// second colour needs the first colour's new values of plane z.  All passes can therefore be
// applied plane by plane in one march in which every colour pass runs one plane behind the pass
// before it: in a time step, pass number d (d = 1 .. 2*iters; colour (d-1)%2 of iteration
// (d+1)/2) updates the plane d below the one just loaded.  The passes of one step touch different
// planes and colours and do not depend on each other.  Every point is still computed by exactly
// the reference's expression from exactly the values the serial NDRange schedule would feed it, so
// the result is identical bit for bit to 2*iters launches of the single-colour kernel.
//
// Shape.  A block of NW wavefronts owns a tile of 64 x (NW*SR) columns including a halo of
// H = 2*iters columns on every side (each colour update consumes one ring of halo).  A wavefront
// owns a strip of SR consecutive rows, one lane per x; a lane keeps its SR columns in registers
// for all planes in flight:
//   C[c][d][j]  value of the colour-c point of row pair j (rows 2j, 2j+1) at ring depth d, where
//               depth d is plane i-d when plane i has just been loaded: d=0 is the raw plane,
//               d = 1 .. 2*iters the plane pass d updates, d+1 its z-1 carry, and d=2*iters the
//               plane that is finished and stored at the end of the step;
//   F[c][d][j]  the right-hand side of the same points;
//   prefetch registers for PD further planes of `in` and `f` (the HBM latency cover).
// x neighbours come from the adjacent lanes by DPP wave shifts (no memory), y neighbours inside the
// strip are the lane's own registers, and only the strip's first and last row are exchanged with
// the neighbouring wavefronts through LDS: once per time step for all passes together (one
// barrier per step, LDS double-buffered by step parity).
// Storing by colour instead of by row makes every lane update a point in every pass: which of the
// two rows of a pair holds the lane's colour-c point depends on the lane's parity only.
#include "mgrt_internal.h"
#include "mgrt_device.h"
#include <type_traits>

using namespace mgrt;

namespace {

constexpr int kWave = 64;

// Hooks of the diagnostic build (tools/diag/mgrt_stamps.hip defines them and includes this file: time a
// wavefront spends in each segment of a fast time step).  Empty in the product.
#ifndef MGRT_STAMP
#define MGRT_STAMP_DECL()
#define MGRT_STAMP_BEGIN()
#define MGRT_STAMP(k)
#define MGRT_STAMP_STEP()
#define MGRT_STAMP_END(wave, lane)
#endif

//
// RES adds one more pass, two planes behind the last colour pass, where every neighbour is final:
// res = f - A out on inner points, f - out on Dirichlet points, 0 elsewhere (residual_kernel,
// mg_1.cl:46-53,111-133, with the cells it leaves unwritten set to 0).  It costs one more ring of
// halo and two more planes in the rings, and saves a separate pass over out and f.
//
// COR makes the input in + 4*P(e) instead of in: the coarse-grid correction (residual_correct_kernel,
// mg_1.cl:200-233) applied to every plane on its way from the prefetch registers into the ring, so
// that the post-smoother needs no corrected field in memory.  A lane keeps the coarse values its SR
// rows interpolate from (SR/2+1 coarse rows at x/2 and x/2+1) for the two coarse planes around the
// fine plane; every second step one coarse plane is prefetched along with the fine one.  The tile's
// first row must be even (globally), so that which rows and planes sit between coarse points is
// known at compile time and the eight trilinear weights are per-lane constants.
struct CoarseField {
    const void *e; // COR: the coarse field to interpolate from
    void *rc;      // RST: the coarse right-hand side to write
    long long pitch, slab;
    int nx, ny, nz, y0;
    int ylo, yhi;  // RST: the coarse rows (of the coarse array) this launch writes
    int kchunk;    // RST without smoothing: coarse planes a block forms (blockIdx.y counts the chunks); 0: all of them
};

// dot(real4, real4) as the reference evaluates it: left to right in R
template <class R> __device__ __forceinline__ R dot4(R a0, R a1, R a2, R a3, R b0, R b1, R b2, R b3)
{
    return a0 * b0 + a1 * b1 + a2 * b2 + a3 * b3;
}

//
// STAGES = 2 splits the march between two groups of NW wavefronts that own the same strips: the first
// group performs the first ITERS iterations and hands every finished plane (and its right-hand side)
// to the second group through LDS instead of memory, which performs ITERS more and stores.  A
// wavefront then issues half the instructions per time step; on levels whose sweep time is the
// latency of a step rather than bandwidth that is what counts.  The second group runs
// LAG = 2*ITERS + 1 steps behind the first; both share the one barrier per step.
//
// RST (with RES) carries the full-weighting restriction (reduction_kernel, mg_1.cl:146-190) of that
// residual along and writes it straight to the coarse right-hand side; the fine residual then never
// reaches memory except next to the seams between tile rows (below).  A lane keeps the residuals of
// its SR rows for the last three planes, split by colour as the rings are.  Coarse points sit at even
// x, y, z: colour 0, on the upper row of a row pair (tiles start on odd rows), so an even lane holds a
// coarse point's centre as its own colour-0 value, the rows above and below as its own colour-1 values
// of this pair and the next one, and finds the six remaining in-plane values in its two neighbour
// lanes (DPP).  The next pair of a strip's last pair comes from the strip below through LDS, a step
// later, so coarse plane K is formed one step after the residual of fine plane 2K + 1.  The 27 terms
// are summed in the reference's order; coarse points on a face of the grid take the fine value at
// their place (mg_1.cl:154-160: for odd extents the forward 2x2x2 block of a coarse point touches a
// border cell exactly when the point lies on a face).  A coarse point needs residuals one ring further
// out than the tile stores values: in x the tiles are cut one column narrower on either side for it;
// in y that would take more tiles than a 513^2 plane leaves CUs, so the coarse row on the seam between
// two tile rows is left out here, the three fine residual rows around it are written to r, and a
// restriction launch for those coarse rows alone follows.
//
// MAP: the domain is described by one byte per point (g.cells) instead of six face classes.  The
// bytes of a plane are prefetched beside its values (SR more loads per lane and step), packed two bits
// a row into one register per plane in flight, and looked up where the box variant uses per-lane
// constants; nothing else changes, so any domain a BorderHandler can describe takes the fused sweep.
template <class R, int ITERS, int SR, int NW, int PD, bool OMEGA1, bool RES, bool COR_ = false, int STAGES = 1,
          bool MAP = false, bool RST = false>
__global__ __launch_bounds__(NW *STAGES *kWave) void k_smooth3d_fused(mgrt_grid g, const R *__restrict__ in,
                                                              R *__restrict__ out, const R *__restrict__ f,
                                                              R *__restrict__ res, R omega, int zcount, int ntx,
                                                              int nty, int ty0, CoarseField cf)
{
    static_assert(SR % 2 == 0, "rows are handled in pairs");
    static_assert(!COR_ || (PD == 2 && !RES), "the corrected input is built for a prefetch depth of two planes");
    static_assert(STAGES == 1 || (STAGES == 2 && !RES), "stages hand on whole smoothed planes only");
    static_assert(!MAP || (STAGES == 1 && !RES && !COR_ && SR <= 16), "the byte-map variant is the plain sweep");
    static_assert(!RST || (RES && !COR_ && STAGES == 1 && !MAP), "the restriction rides on the residual pass");
    constexpr int NP = 2 * ITERS, J = SR / 2;    // colour passes per stage
    constexpr int H = NP * STAGES + (RES ? 1 : 0); // halo rings consumed by all stages together
    constexpr int HX = H + (RST ? 1 : 0);        // in x the restriction takes one more
    constexpr int LAG = NP + 1;                  // steps between a plane leaving one stage and entering the next
    constexpr int DR = NP + 2;             // ring depth of the residual pass
    constexpr int ND = RES ? NP + 4 : NP + 2, NSET = RES ? NP + 2 : NP;
    // The plane rings (period ND) and the prefetch ring (period PD) are addressed with indices that
    // are compile-time constants inside a time loop unrolled UN times, UN a multiple of both: plane
    // i lives in ring slot i % ND and prefetch slot i % PD, nothing is ever shifted.
    static_assert(ND % PD == 0 || PD % ND == 0, "prefetch depth and ring depth must divide the unrolled loop");
    constexpr int UN = ND > PD ? ND : PD;
    constexpr int OX = kWave - 2 * HX, OY = NW * SR - 2 * H;
    // strips start on odd rows (coarse rows are the upper rows of their pairs), residual planes live in
    // three slots addressed with a period of four planes
    static_assert(!RST || (H % 2 == 1 && OY % 2 == 0 && UN % 4 == 0 && DR % 2 == 0), "tile shape does not fit the restriction");
    __shared__ R bnd[2][NSET][NW * STAGES][2][kWave];
    // first row pair of every strip's residual plane, for the strip above: [step parity][strip][colour][lane]
    __shared__ R rbn[2][RST ? NW : 1][2][kWave];
    // hand-over between the stages: [step parity][strip][u | f][colour][row pair][lane]; a wavefront of
    // the second stage reads exactly what the wavefront of the first stage with the same strip wrote
    __shared__ R hand[2][STAGES > 1 ? NW : 1][2][2][J][kWave];

    const int lane = threadIdx.x & (kWave - 1);
    // wave-uniform by construction; read as a scalar so that the stages branch without exec masks
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
    // Two stages: the colour passes a strip sits out (see the passes below) leave the edge strips with less to do, and
    // a time step lasts as long as the busiest SIMD; wavefront i of a block runs on SIMD i % 4, so the strips are dealt
    // such that every SIMD gets the same number of colour passes a step (6 of 8).  Four bits a wavefront, the first NW
    // wavefronts are the first stage.  A speed matter only: any bijection gives the same values.
    // Tiles of eight strips only: with six the step is short enough to be a chain of latencies, and the tests cost what
    // the spared passes save (129^3 with the correction: 55.2 -> 56.5 us, 65^3: 27.5 -> 28.1).
    constexpr bool DEALT = STAGES == 2 && SR == 2 && ITERS == 1 && NW == 8;
    constexpr unsigned long long kStrips = 0x6543'1702'6427'5310ull; // 0 1 3 5 7 2 4 6 | 2 0 7 1 3 4 5 6
    const int stage = STAGES > 1 ? wave / NW : 0;
    const int w = DEALT ? (int)((kStrips >> (4 * wave)) & 15u) : (STAGES > 1 ? wave % NW : wave);
    // Blocks are dealt to the 8 XCDs round-robin (block b and b+8 share an L2).  Give every XCD a
    // contiguous run of tiles, so that the halo columns and the partial cache lines two
    // neighbouring tiles both touch are fetched from HBM once and hit in that XCD's L2 the second
    // time.  A speed matter only: any placement computes the same values.
    const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx = tile % ntx, ty = ty0 + tile / ntx; // of nty tile rows; ty0: the first one of this launch (a band of a large plane)
    const int x = tx * OX - HX + lane;
    const int ly0 = w * SR;               // first strip row inside the tile
    const int y0 = ty * OY - H + ly0;     // its array row
    const bool xin = x >= 0 && x < g.nx;
    // which row of a pair holds this lane's colour-0 point: row 2j + s0
    const int s0 = (x + y0 + g.y0) & 1;

    // Per row: byte offset of the column inside a plane for loads (past the plane for lanes and
    // rows outside the grid, which then read 0) and for stores (past the plane unless the point
    // belongs to the tile's interior), and the class of the column when it does not depend on z
    // (a lateral face, or OUTER outside the grid).
    constexpr unsigned kDrop = 0x80000000u;
    const unsigned plane_bytes = (unsigned)(g.pitch * (long long)g.ny * (long long)sizeof(R));
    // in == nullptr stands for a field of zeros: a descriptor of length 0 makes every load return 0
    const R *inb = in ? in : f;
    const unsigned in_bytes = in ? plane_bytes : 0u;
    unsigned off[SR], soff[SR];
    int rowcls[SR];
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const int y = y0 + r, yg = y + g.y0;
        const bool live = xin && y >= 0 && y < g.ny;
        const bool keep = live && lane >= HX && lane < kWave - HX && ly0 + r >= H && ly0 + r < NW * SR - H;
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
    // byte map: in-plane byte offset per row (element offset: one byte a point), and OUTER (01) in the
    // two bits of every row that lies outside the grid, where the dropped load returns 0 = INNER
    unsigned coff[MAP ? SR : 1];
    int deadbits = 0, shc[2][J];
    if constexpr (MAP) {
        static_assert(MGRT_CELL_OUTER == 1 && MGRT_CELL_INNER == 0, "dropped class loads must read as a mask of OUTER");
        const unsigned cplane = (unsigned)(g.pitch * (long long)g.ny);
        (void)cplane;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            coff[r] = off[r] == kDrop ? kDrop : off[r] / (unsigned)sizeof(R);
            if (off[r] == kDrop)
                deadbits |= MGRT_CELL_OUTER << (2 * r);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) { // row 2j + (s0 ^ c) holds the colour-c point of pair j
            shc[0][j] = 2 * (2 * j + s0);
            shc[1][j] = 2 * (2 * j + (s0 ^ 1));
        }
    }
    bool interior[2][J], latdir[2][J], latfix[2][J]; // latfix: lateral Dirichlet column
#pragma unroll
    for (int j = 0; j < J; ++j) { // row 2j + s0 of pair j holds the colour-0 point
        const int c0 = s0 ? rowcls[2 * j + 1] : rowcls[2 * j];
        const int c1 = s0 ? rowcls[2 * j] : rowcls[2 * j + 1];
        interior[0][j] = c0 < 0;
        interior[1][j] = c1 < 0;
        latdir[0][j] = c0 == MGRT_CELL_DIRICHLET;
        latdir[1][j] = c1 == MGRT_CELL_DIRICHLET;
        latfix[0][j] = latdir[0][j] && !interior[0][j];
        latfix[1][j] = latdir[1][j] && !interior[1][j];
    }

    // Restriction: where this lane's coarse points go (one per row pair: its upper row), whether they lie
    // on a lateral face of the grid, and where the two rows of a strip that can lie next to a seam
    // between tile rows go in r (rows H and H + 1 of a tile below a seam, row NW*SR - H - 1 above one).
    constexpr int RS0 = H % SR, RS1 = (H + 1) % SR;
    static_assert(!RST || (NW * SR - H - 1) % SR == RS1 || (NW * SR - H - 1) % SR == RS0, "seam rows");
    unsigned cof[RST ? J : 1], rof[2];
    bool injl[RST ? J : 1];
    if constexpr (RST) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int r = 2 * j + 1, l = ly0 + r, y = y0 + r, yg = y + g.y0;
            const int cj = (yg >> 1) - cf.y0; // row of the coarse array
            const bool own = xin && y >= 0 && y < g.ny && lane >= HX && lane < kWave - HX && (x & 1) == 0 &&
                             ((l >= H + 2 && l < NW * SR - H) || (l == H && ty == 0)) && cj >= cf.ylo && cj < cf.yhi;
            cof[j] = own ? ((unsigned)(x >> 1) + (unsigned)cf.pitch * (unsigned)cj) * (unsigned)sizeof(R) : kDrop;
            injl[j] = x == 0 || x == g.nx - 1 || yg == 0 || yg == g.gny - 1;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = k ? RS1 : RS0, l = ly0 + r;
            const bool seam = (l == H || l == H + 1) ? ty > 0 : (l == NW * SR - H - 1 ? ty + 1 < nty : false);
            rof[k] = seam ? soff[r] : kDrop;
        }
    }

    const long long slab = g.slab;
    const int zlast = g.nz - 1;
    const int zlim = zcount < g.nz ? zcount : g.nz;
    const int wu = w > 0 ? w - 1 : 0, wd = w + 1 < NW ? w + 1 : NW - 1;
    // Residual and restriction without smoothing have no recurrence in z, so the march can be cut: the block with
    // blockIdx.y = b forms coarse planes [K0, K1) only.  It starts at fine plane zs (the residual of plane 2*K0 - 1 is
    // the first it needs: planes from 2*K0 - 2 on, rounded down to the unrolled loop) and writes the seam rows of the
    // residual planes [2*K0, 2*K1), so that every plane of r has one writer; what it computes for other planes (from
    // the incomplete ring at its start, in the rest of its last group of steps) goes to a descriptor of length 0.
    constexpr bool CHUNKS = RST && ITERS == 0;
    int K0 = 0, K1 = cf.nz, zs = 0;
    if constexpr (CHUNKS) {
        if (cf.kchunk > 0) {
            K0 = (int)blockIdx.y * cf.kchunk;
            K1 = K0 + cf.kchunk < cf.nz ? K0 + cf.kchunk : cf.nz;
            zs = K0 > 0 ? (2 * K0 - 2) & ~(UN - 1) : 0;
        }
    }

    auto march = [&](auto stage_tag) {
    constexpr int STG = decltype(stage_tag)::value;
    constexpr bool FIRST = STG == 0, LAST = STG == STAGES - 1; // FIRST loads from memory, LAST stores to it
    constexpr bool COR = COR_ && FIRST;                        // the correction is applied where planes come in from memory
    constexpr int lag = STG * LAG;
    constexpr int WB = STG * NW; // this stage's wavefronts in bnd
    bool pair_needed[NP > 0 ? NP : 1][J]; // two stages: the passes of this stage that this strip's row pairs take part in
#pragma unroll
    for (int d = 1; d <= NP; ++d)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int P = STG * NP + d, lo = w * SR + 2 * j;
            pair_needed[d - 1][j] = lo + 1 >= P && lo <= NW * SR - 1 - P;
        }
    // per step; the coarse stores of every second step of RST are left out, which only makes the counted
    // waits stricter than needed
    // ITERS == 0 (residual and restriction alone) stores no smoothed plane
    constexpr int kStores = LAST ? (NP > 0 ? SR : 0) + (RST ? 2 : RES ? SR : 0) : 0;
    constexpr int kLoads = MAP ? 3 * SR : 2 * SR;                     // per step
    constexpr int kYounger = kStores + (PD - 1) * (kLoads + kStores); // see wait_slot
    static_assert(kYounger < 64, "too many vector-memory operations in flight for a counted wait");
    MGRT_STAMP_DECL();
    R C[2][ND][J], F[2][ND][J];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int j = 0; j < J; ++j) C[c][d][j] = F[c][d][j] = 0;
    // RST: residuals of the last three planes, [slot][colour][row pair], pair J being the first pair of
    // the strip below.  Odd planes alternate between slots 0 and 1, even planes use slot 2 (an even plane
    // is the centre of one coarse plane only, an odd one serves two).
    R Rr[RST ? 3 : 1][2][J + 1];
#pragma unroll
    for (int t = 0; t < (RST ? 3 : 1); ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j <= J; ++j) Rr[t][c][j] = 0;
    R *const rcb = (R *)cf.rc;
    const unsigned rcplane_bytes = RST ? (unsigned)(cf.pitch * (long long)cf.ny * (long long)sizeof(R)) : 0u;

    // ---- coarse-grid correction: offsets, weights, the two coarse planes in registers ----------
    constexpr int NCR = SR / 2 + 1; // coarse rows under SR fine rows that start on an even row
    const R *eb = (const R *)cf.e;
    const int klast = cf.nz - 1;
    const unsigned cplane_bytes = COR ? (unsigned)(cf.pitch * (long long)cf.ny * (long long)sizeof(R)) : 0u;
    unsigned eoff[NCR][2];
    R Ea[NCR][2], Eb[NCR][2], pe[NCR][2]; // coarse planes z/2 and z/2+1 (clamped), and the prefetched one
    R wa1 = 0, wa2 = 0, wa4 = 0, wu1 = 0, wu2 = 0, wu4 = 0;
    if constexpr (COR) {
        // as the interpolation kernel: index x/2 and, clamped to the array, x/2 + 1; likewise in y
        const int xc = x < 0 ? 0 : (x < g.nx ? x : g.nx - 1);
        const int ci = xc / 2, ci1 = ci + 1 < cf.nx ? ci + 1 : ci;
        const int cj0 = ((y0 + g.y0) >> 1) - cf.y0;
#pragma unroll
        for (int m = 0; m < NCR; ++m) {
            const int cj = cj0 + m < 0 ? 0 : (cj0 + m < cf.ny ? cj0 + m : cf.ny - 1);
            eoff[m][0] = ((unsigned)ci + (unsigned)cf.pitch * (unsigned)cj) * (unsigned)sizeof(R);
            eoff[m][1] = ((unsigned)ci1 + (unsigned)cf.pitch * (unsigned)cj) * (unsigned)sizeof(R);
        }
        // weights (1-wu)*c and wu*c for c = 1, 1/2, 1/4, all exact: wu = 0.5 * (x % 2)
        const R wu = 0.5 * (xc % 2);
        wa1 = 1.0 - wu, wa2 = 0.5 * (1.0 - wu), wa4 = 0.25 * (1.0 - wu);
        wu1 = wu, wu2 = 0.5 * wu, wu4 = 0.25 * wu;
        const desc_t r0 = plane_desc(eb, cplane_bytes), r1 = plane_desc(eb + cf.slab * (klast < 1 ? klast : 1), cplane_bytes);
#pragma unroll
        for (int m = 0; m < NCR; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                buf_load_hidden(Eb[m][q], r0, eoff[m][q]); // becomes plane z/2 at step 0
                buf_load_hidden(pe[m][q], r1, eoff[m][q]);
                Ea[m][q] = 0;
            }
    }
    auto pin_coarse = [&](R(&a)[NCR][2]) { // orders every later use of `a` after the wait before it
        if constexpr (COR) {
#pragma unroll
            for (int m = 0; m < NCR; ++m)
                asm volatile("" : "+v"(a[m][0]), "+v"(a[m][1]));
        }
    };
    R pu[PD][SR], pf[PD][SR];
    int pk[PD][MAP ? SR : 1], K[MAP ? ND : 1]; // class bytes on their way in, packed classes of the planes in flight
    const unsigned cells_bytes = MAP ? (unsigned)(g.pitch * (long long)g.ny) : 0u;
    auto pin_cells = [&](int(&a)[MAP ? SR : 1]) { // as pin_coarse
        if constexpr (MAP) {
#pragma unroll
            for (int r = 0; r < SR; ++r)
                asm volatile("" : "+v"(a[r]));
        }
    };
    if constexpr (MAP) {
#pragma unroll
        for (int d = 0; d < ND; ++d) K[d] = 0x55555555; // OUTER everywhere: planes below the grid
    }
    if constexpr (FIRST) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const long long zo = slab * (zs + k < zlast ? zs + k : zlast);
            const desc_t ru = plane_desc(inb + zo, in_bytes), rf = plane_desc(f + zo, plane_bytes);
#pragma unroll
            for (int r = 0; r < SR; ++r) {
                buf_load_hidden(pu[k][r], ru, off[r]);
                buf_load_hidden(pf[k][r], rf, off[r]);
            }
            if constexpr (MAP) {
                const desc_t rk = plane_desc(g.cells + zo, cells_bytes);
#pragma unroll
                for (int r = 0; r < SR; ++r)
                    buf_load_hidden_u8(pk[k][r], rk, coff[r]);
            }
        }
    }
    auto drain0 = [&]() {
        if constexpr (FIRST) {
#pragma unroll
            for (int k = 0; k < PD; ++k) {
                wait_slot<0>(pu[k], pf[k]);
                pin_cells(pk[k]);
            }
        }
        pin_coarse(pe);
        pin_coarse(Eb);
    };
    drain0(); // the loop below starts from landed registers (see `drain`)

    // planes entering this stage's ring: the last one finishes plane nz-1; all stages together take
    // nsteps_all time steps, every one of them with a barrier that every wavefront of the block joins
    // (a chunk: coarse plane K1 - 1 is stored in step 2*K1 + 2)
    const int nsteps = CHUNKS ? 2 * K1 + 3 : g.nz + (RST ? DR + 2 : RES ? DR : NP); // RST: the last coarse plane follows two steps later
    const int nsteps_all = nsteps + (STAGES - 1) * LAG;
    // One group of UN consecutive time steps, unrolled so that every ring index is a constant.
    // FAST groups are those whose steps all work on planes strictly between the z faces and below
    // zlim, prefetch planes that exist, and store a plane: no per-step conditions are left and the
    // class of a point is a per-lane constant.
    auto group = [&](auto fast_tag, const int g0) { // g0: first time step of the group, counted for the block
        constexpr bool FAST = decltype(fast_tag)::value;
        const int i0 = g0 - lag; // the same, counted for this stage: plane i0 enters its ring
        // Plane addresses of a fast group advance by one slab per step: keep running pointers
        // (two scalar adds each) instead of a 64-bit multiply per array and step.
        // The descriptors themselves are the running state: base address in words 0 and 1.
        desc_t d_in = plane_desc(inb + slab * (i0 + PD), in_bytes), d_f = plane_desc(f + slab * (i0 + PD), plane_bytes);
        desc_t d_out = plane_desc(NP > 0 ? out + slab * (i0 - NP) : (R *)f, plane_bytes);
        desc_t d_res = plane_desc(RES ? res + slab * (i0 - DR) : out, plane_bytes);
        // coarse plane written at the group's first step: fine plane i0 - 2 - DR is its centre
        desc_t d_rc = plane_desc(RST ? rcb + cf.slab * ((i0 - 2 - DR) / 2) : (R *)out, rcplane_bytes);
        // coarse plane prefetched at the group's first (even) step: the upper one of fine plane i0 + PD
        desc_t d_e = plane_desc(COR ? eb + cf.slab * ((i0 + PD) / 2 + 1) : eb, cplane_bytes);
        desc_t d_k = plane_desc(MAP ? g.cells + slab * (i0 + PD) : (const unsigned char *)f, cells_bytes);
        const unsigned long long slab_bytes = (unsigned long long)slab * sizeof(R);
        auto advance = [&](desc_t &d, unsigned long long by) {
            const unsigned long long a = (((unsigned long long)(unsigned)d.y << 32) | (unsigned)d.x) + by;
            d.x = (int)(unsigned)a;
            d.y = (int)(unsigned)(a >> 32); // addresses stay below 2^48: the upper half of word 1 stays 0
        };
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u; // plane entering the ring; g0 is a multiple of UN
            if (FAST || g0 + u < nsteps_all) {
#define PH(d) ((((u) - (d)) % ND + ND) % ND) /* ring slot of depth d, i.e. of plane i - d */
                const int ps = u % PD; // prefetch slot of plane i (a constant once the loop is unrolled)
                // ---- plane i: prefetch registers -> ring slot -----------------------------------
                // In a fast group every step issues 2*SR loads then kStores stores, so the slot's
                // loads have kStores stores and PD-1 whole steps behind them; elsewhere steps may
                // skip their stores, so wait for everything.
                // With COR the slots of even planes carry the 2*NCR coarse loads as well, so an odd
                // plane's slot has that many more operations behind it.
                if (FAST) {
                    MGRT_STAMP_BEGIN();
                }
                constexpr int kYoungerOdd = kYounger + (COR ? 2 * NCR : 0);
                static_assert(!COR || kYoungerOdd + 2 * SR + 2 * NCR < 64, "too many vector-memory operations in flight");
                if constexpr (FIRST) {
                    if (!FAST)
                        wait_slot<0>(pu[ps], pf[ps]);
                    else if (u & 1) // folds once the loop is unrolled
                        wait_slot<kYoungerOdd>(pu[ps], pf[ps]);
                    else
                        wait_slot<kYounger>(pu[ps], pf[ps]);
                    pin_cells(pk[ps]);
                    if constexpr (MAP) {
                        int packed = deadbits;
#pragma unroll
                        for (int r = 0; r < SR; ++r)
                            packed |= pk[ps][r] << (2 * r);
                        K[PH(0)] = packed;
                        asm volatile("" : "+v"(K[PH(0)])); // made before the slot is requested again
                    }
                }
                if (FAST) {
                    MGRT_STAMP(0); // waiting for the prefetched plane
                }
                R cor[SR];
                if constexpr (COR) {
                    const int pz = u & 1; // parity of plane i
                    if (pz == 0) { // planes (i/2, i/2 + 1): the upper one has just landed
                        pin_coarse(pe);
#pragma unroll
                        for (int m = 0; m < NCR; ++m)
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                Ea[m][q] = Eb[m][q];
                                Eb[m][q] = pe[m][q];
                            }
                        pin_coarse(Eb); // real copies, made before pe is requested again
                    }
                    const int zcls = FAST ? (int)MGRT_CELL_INNER
                                          : (i == 0 ? g.face[4] : (i == zlast ? g.face[5] : (int)MGRT_CELL_INNER));
#pragma unroll
                    for (int r = 0; r < SR; ++r) {
                        const int py = r & 1, m = r >> 1;
                        // c1 = (1-ww)(1-wv), c2 = ww(1-wv), c3 = (1-ww)wv, c4 = ww*wv with ww = py/2, wv = pz/2
                        const R n10 = py + pz == 0 ? wa1 : (py + pz == 1 ? wa2 : wa4);
                        const R n11 = py + pz == 0 ? wu1 : (py + pz == 1 ? wu2 : wu4);
                        const R n12 = py == 0 ? (R)0 : (pz == 0 ? wa2 : wa4), n13 = py == 0 ? (R)0 : (pz == 0 ? wu2 : wu4);
                        const R n20 = pz == 0 ? (R)0 : (py == 0 ? wa2 : wa4), n21 = pz == 0 ? (R)0 : (py == 0 ? wu2 : wu4);
                        const R n22 = py + pz == 2 ? wa4 : (R)0, n23 = py + pz == 2 ? wu4 : (R)0;
                        const R val = dot4<R>(Ea[m][0], Ea[m][1], Ea[m + 1][0], Ea[m + 1][1], n10, n11, n12, n13) +
                                      dot4<R>(Eb[m][0], Eb[m][1], Eb[m + 1][0], Eb[m + 1][1], n20, n21, n22, n23);
                        const int cls = rowcls[r] >= 0 ? rowcls[r] : zcls;
                        const bool border = cls == MGRT_CELL_DIRICHLET || cls == MGRT_CELL_NEUMANN;
                        cor[r] = (border ? Ea[m][0] : val) * 4;
                    }
                }
                if constexpr (FIRST) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const R a0 = COR ? pu[ps][2 * j] + cor[COR ? 2 * j : 0] : pu[ps][2 * j];
                    const R a1 = COR ? pu[ps][2 * j + 1] + cor[COR ? 2 * j + 1 : 0] : pu[ps][2 * j + 1];
                    const R b0 = pf[ps][2 * j], b1 = pf[ps][2 * j + 1];
                    C[0][PH(0)][j] = s0 ? a1 : a0;
                    C[1][PH(0)][j] = s0 ? a0 : a1;
                    F[0][PH(0)][j] = s0 ? b1 : b0;
                    F[1][PH(0)][j] = s0 ? b0 : b1;
                    // The slot's registers must be dead before they are requested again below:
                    // pin the four selects above the refill (asm volatile statements keep their
                    // order), otherwise the compiler keeps the old values alive in copies it makes
                    // ahead of the wait.
                    asm volatile("" : "+v"(C[0][PH(0)][j]), "+v"(C[1][PH(0)][j]), "+v"(F[0][PH(0)][j]), "+v"(F[1][PH(0)][j]));
                }
                }
                if constexpr (FIRST) { // plane i + PD -> prefetch slot ps
                    const int zn = i + PD;
                    const long long zo = slab * (zn < zlast ? zn : zlast);
                    const desc_t ru = FAST ? d_in : plane_desc(inb + zo, in_bytes);
                    const desc_t rf = FAST ? d_f : plane_desc(f + zo, plane_bytes);
#pragma unroll
                    for (int r = 0; r < SR; ++r) {
                        buf_load_hidden(pu[ps][r], ru, off[r]);
                        buf_load_hidden(pf[ps][r], rf, off[r]);
                    }
                    if constexpr (MAP) {
                        const desc_t rk = FAST ? d_k : plane_desc(g.cells + zo, cells_bytes);
#pragma unroll
                        for (int r = 0; r < SR; ++r)
                            buf_load_hidden_u8(pk[ps][r], rk, coff[r]);
                    }
                    if constexpr (COR) {
                        if ((u & 1) == 0) { // the upper coarse plane of fine plane i + PD
                            const int kn = (zn < zlast ? zn : zlast) / 2 + 1;
                            const desc_t re = FAST ? d_e : plane_desc(eb + cf.slab * (kn < klast ? kn : klast), cplane_bytes);
#pragma unroll
                            for (int m = 0; m < NCR; ++m) {
                                buf_load_hidden(pe[m][0], re, eoff[m][0]);
                                buf_load_hidden(pe[m][1], re, eoff[m][1]);
                            }
                        }
                    }
                }
                // ---- publish, for every pass, the strip's first and last row of the colour it
                // reads.  Wherever a neighbouring strip needs the first (last) row, that row holds
                // that colour.  None of these values changes during the step.
                const int par = u & 1;
#pragma unroll
                for (int d = 1; d <= NP; ++d) {
                    const int oc = ((d - 1) & 1) ^ 1;
                    bnd[par][d - 1][WB + w][0][lane] = C[oc][PH(d)][0];
                    bnd[par][d - 1][WB + w][1][lane] = C[oc][PH(d)][J - 1];
                }
                if (RES) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        bnd[par][NSET - 2 + c][WB + w][0][lane] = C[c][PH(DR)][0];
                        bnd[par][NSET - 2 + c][WB + w][1][lane] = C[c][PH(DR)][J - 1];
                    }
                }
                if (FAST) {
                    MGRT_STAMP(1); // ring insert, requests for the next plane, publishing the edge rows
                }
                __syncthreads();
                if (FAST) {
                    MGRT_STAMP(2); // the barrier
                }
                if constexpr (!FIRST) {
                    // plane i as the stage before left it at the end of the previous step, already
                    // split by colour (same strip, same lanes: same row parity)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            C[c][PH(0)][j] = hand[par ^ 1][w][0][c][j][lane];
                            F[c][PH(0)][j] = hand[par ^ 1][w][1][c][j][lane];
                        }
                }
                // ---- the passes: pass d = colour (d-1)%2 of iteration (d+1)/2 on plane i - d ----
#pragma unroll
                for (int d = 1; d <= NP; ++d) {
                    const int c = (d - 1) & 1;
                    const bool sc = (s0 ^ c) != 0; // odd row of the pair holds the colour-c point
                    const int z = i - d;
                    if (FAST || (z >= 0 && z < zlim)) {
                        const R up = bnd[par][d - 1][WB + wu][1][lane];
                        const R dn = bnd[par][d - 1][WB + wd][0][lane];
                        const int zcls = z == 0 ? g.face[4] : (z == zlast ? g.face[5] : MGRT_CELL_INNER);
                        const bool zin = zcls == MGRT_CELL_INNER, zdir = zcls == MGRT_CELL_DIRICHLET;
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            // Two stages of eight strips: every colour pass makes one more ring of the tile's halo worthless; pass
                            // number P of all stages together is owed rows P .. NW*SR - 1 - P only, nothing that is
                            // stored can depend on the others, and a row pair outside that range sits the pass out
                            // (wave-uniform).
                            if (DEALT && !pair_needed[d - 1][j])
                                continue;
                            const R o = C[c ^ 1][PH(d)][j]; // other colour, same pair
                            const R xm = from_lower_lane(o), xp = from_upper_lane(o);
                            const R below = j > 0 ? C[c ^ 1][PH(d)][j > 0 ? j - 1 : 0] : up;
                            const R above = j + 1 < J ? C[c ^ 1][PH(d)][j + 1 < J ? j + 1 : 0] : dn;
                            const R yp = sc ? above : o; // u[y+1]
                            const R ym = sc ? o : below; // u[y-1]
                            const R own = C[c][PH(d)][j];
                            const R fv = F[c][PH(d)][j];
                            // mg_1.cl:35-42: neighbours summed in R in this order, times the
                            // double literal 1.0/6.0, rounded to R
                            const R val = (1.0 / 6.0) * (xp + xm + yp + ym + C[c][PH(d - 1)][j] + C[c][PH(d + 1)][j] - fv);
                            // mg_1.cl:69; for omega == 1 the double expression reduces exactly to val + 0*own
                            // evaluated in R, and that to one fused multiply-add (the product, a zero or a NaN, is
                            // exact, so the one rounding is that of the sum: same bits, v_fmac_f32 with the constant
                            // inline)
                            R gs = OMEGA1 ? fma_r((R)0, own, val) : (R)(val * omega + (1.0 - omega) * own);
                            // Keep the update unconditional: left alone, the compiler sinks it
                            // under a branch on take_gs, and the exec-mask blocks that result
                            // serialise the independent updates of a step and their LDS reads.
                            asm("" : "+v"(gs));
                            bool take_gs = FAST ? interior[c][j] : (interior[c][j] && zin);
                            bool take_f = FAST ? latfix[c][j] : (interior[c][j] ? zdir : latdir[c][j]);
                            if constexpr (MAP) { // the class of this very point, from the map
                                const int cls = (K[PH(d)] >> shc[c][j]) & 3;
                                take_gs = cls == MGRT_CELL_INNER;
                                take_f = cls == MGRT_CELL_DIRICHLET;
                            }
                            // From the second iteration on a lateral Dirichlet point already holds its f (the pass of the
                            // first iteration put it there, bit for bit), so between the z faces one select is enough.
                            const bool settled = FAST && !MAP && STG * NP + d > 2;
                            C[c][PH(d)][j] = take_gs ? gs : (settled ? own : (take_f ? fv : own));
                        }
                    }
                }
                if (FAST) {
                    MGRT_STAMP(3); // neighbour rows from LDS and the colour passes
                }
                // ---- plane i - NP is final: store the tile's interior ----------------------------
                const int zf = i - NP;
                if constexpr (!LAST) {
                    // on to the next stage, which takes it up after the next barrier
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            hand[par][w][0][c][j][lane] = C[c][PH(NP)][j];
                            hand[par][w][1][c][j][lane] = F[c][PH(NP)][j];
                        }
                } else if (NP > 0 && (FAST || (zf >= 0 && zf <= zlast))) {
                    const desc_t ro = FAST ? d_out : plane_desc(out + slab * zf, plane_bytes);
#pragma unroll
                    for (int r = 0; r < SR; ++r)
                        buf_store_hidden(((r & 1) ^ s0) ? C[1][PH(NP)][r / 2] : C[0][PH(NP)][r / 2], ro, soff[r]);
                }
                // ---- restriction: coarse plane (i - 2 - DR) / 2, whose upper fine plane got its residual in
                // the step before (every second step) ------------------------------------------------
#define RSLOT(k) ((((u) - DR - (k)) % 2 + 2) % 2 ? (((((u) - DR - (k)) % UN + UN) % UN) >> 1) & 1 : 2) /* slot of plane i - DR - k */
                if constexpr (RST) {
                    // the pair below this strip's last pair, of the plane the step before left
                    Rr[RSLOT(1)][0][J] = rbn[par ^ 1][wd][0][lane];
                    Rr[RSLOT(1)][1][J] = rbn[par ^ 1][wd][1][lane];
                    const int zc = i - 2 - DR; // centre plane
                    if (((u + 1 + DR) & 1) != 0 && (FAST || (zc >= 0 && zc <= zlast))) {
                        const bool zface = zc <= 0 || zc >= zlast;
                        desc_t rd = FAST ? d_rc : plane_desc(rcb + cf.slab * (zc / 2), rcplane_bytes);
                        if (CHUNKS && !(zc >= 2 * K0 && zc < 2 * K1))
                            rd.z = 0; // not this chunk's plane: a descriptor of length 0 drops the stores
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            // planes z+ (P), z (Z), z- (M); colour 0 (A) holds the rows of the coarse points on
                            // even lanes and the rows below them on odd lanes, colour 1 (B) the other way round
#define RA(k, jj) Rr[RSLOT(k)][0][jj]
#define RB(k, jj) Rr[RSLOT(k)][1][jj]
#define UP(v) from_upper_lane(v)
#define LO(v) from_lower_lane(v)
                            // mg_1.cl:162-189, term by term
                            const R corners = UP(RA(1, j + 1)) + LO(RA(1, j + 1)) + UP(RA(1, j)) + LO(RA(1, j)) +
                                              UP(RA(3, j + 1)) + LO(RA(3, j + 1)) + UP(RA(3, j)) + LO(RA(3, j));
                            const R edges = UP(RA(2, j + 1)) + LO(RA(2, j + 1)) + UP(RA(2, j)) + LO(RA(2, j)) +
                                            RB(3, j + 1) + RB(3, j) + UP(RB(3, j)) + LO(RB(3, j)) +
                                            RB(1, j + 1) + RB(1, j) + UP(RB(1, j)) + LO(RB(1, j));
                            const R faces = RB(2, j + 1) + RB(2, j) + UP(RB(2, j)) + LO(RB(2, j)) + RA(1, j) + RA(3, j);
                            // The weights are powers of two, so with 4-byte elements every product is exact in double and
                            // a fused multiply-add rounds what the reference's separate multiply and add round: the
                            // same bits in four double-precision instructions instead of seven.  (8-byte elements keep
                            // the separate form: there a product can fall into the subnormal range and round.)
                            R val;
                            if constexpr (sizeof(R) == 4) {
                                double t = 1.0 / 64.0 * (double)corners;
                                t = __builtin_fma((double)edges, 1.0 / 32.0, t);
                                t = __builtin_fma((double)faces, 1.0 / 16.0, t);
                                val = (R)__builtin_fma((double)RA(2, j), 1.0 / 8.0, t);
                            } else {
                                val = 1.0 / 64.0 * corners + 1.0 / 32.0 * edges + 1.0 / 16.0 * faces + 1.0 / 8.0 * (RA(2, j));
                            }
                            asm("" : "+v"(val));
                            buf_store_hidden(injl[j] || zface ? RA(2, j) : val, rd, cof[j]);
#undef RA
#undef RB
#undef UP
#undef LO
                        }
                    }
                }
                // ---- residual of plane i - DR, whose neighbours are all final ---------------------
                if (RES) {
                    const int z = i - DR;
                    if (FAST || (z >= 0 && z <= zlast)) {
                        const int zcls = z == 0 ? g.face[4] : (z == zlast ? g.face[5] : MGRT_CELL_INNER);
                        const bool zin = zcls == MGRT_CELL_INNER, zdir = zcls == MGRT_CELL_DIRICHLET;
                        R rv[2][J];
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const bool sc = (s0 ^ c) != 0;
                            // first / last row of the other colour in the neighbouring strips
                            const R up = bnd[par][NSET - 2 + (c ^ 1)][WB + wu][1][lane];
                            const R dn = bnd[par][NSET - 2 + (c ^ 1)][WB + wd][0][lane];
#pragma unroll
                            for (int j = 0; j < J; ++j) {
                                const R o = C[c ^ 1][PH(DR)][j];
                                const R xm = from_lower_lane(o), xp = from_upper_lane(o);
                                const R below = j > 0 ? C[c ^ 1][PH(DR)][j > 0 ? j - 1 : 0] : up;
                                const R above = j + 1 < J ? C[c ^ 1][PH(DR)][j + 1 < J ? j + 1 : 0] : dn;
                                const R yp = sc ? above : o, ym = sc ? o : below;
                                const R own = C[c][PH(DR)][j], fv = F[c][PH(DR)][j];
                                // mg_1.cl:46-53: integer literal, pure R arithmetic
                                R vi = fv - (xp + xm + yp + ym + C[c][PH(DR - 1)][j] + C[c][PH(DR + 1)][j] - 6 * own);
                                asm("" : "+v"(vi));
                                const R vd = fv - own;
                                const bool is_in = FAST ? interior[c][j] : (interior[c][j] && zin);
                                const bool is_dir = FAST ? latfix[c][j] : (interior[c][j] ? zdir : latdir[c][j]);
                                rv[c][j] = is_in ? vi : (is_dir ? vd : (R)0);
                            }
                        }
                        desc_t rr = FAST ? d_res : plane_desc(res + slab * z, plane_bytes);
                        if (CHUNKS && !(z >= 2 * K0 && z < 2 * K1))
                            rr.z = 0; // as for the coarse plane above
                        if constexpr (RST) {
                            // keep the plane, hand its first pair to the strip above, write the seam rows
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
#pragma unroll
                                for (int j = 0; j < J; ++j) Rr[RSLOT(0)][c][j] = rv[c][j];
                                rbn[par][w][c][lane] = rv[c][0];
                            }
                            buf_store_hidden(((RS0 & 1) ^ s0) ? rv[1][RS0 / 2] : rv[0][RS0 / 2], rr, rof[0]);
                            buf_store_hidden(((RS1 & 1) ^ s0) ? rv[1][RS1 / 2] : rv[0][RS1 / 2], rr, rof[1]);
                        } else {
#pragma unroll
                            for (int r = 0; r < SR; ++r)
                                buf_store_hidden(((r & 1) ^ s0) ? rv[1][r / 2] : rv[0][r / 2], rr, soff[r]);
                        }
                    }
                }
#undef RSLOT
#undef PH
                if (FAST) {
                    MGRT_STAMP(4); // stores / hand-over (and the residual pass)
                    MGRT_STAMP_STEP();
                }
                if (FAST && (FIRST || LAST)) {
                    if (MAP)
                        advance(d_k, (unsigned long long)slab);
                    advance(d_in, slab_bytes), advance(d_f, slab_bytes);
                    if (NP > 0)
                        advance(d_out, slab_bytes);
                    if (RES)
                        advance(d_res, slab_bytes);
                    if (RST && ((u + 1 + DR) & 1) != 0)
                        advance(d_rc, (unsigned long long)cf.slab * sizeof(R));
                    if (COR && (u & 1) == 0)
                        advance(d_e, (unsigned long long)cf.slab * sizeof(R));
                }
            }
        }
    };

    // Registers with a hidden load pending must not be touched before their wait.  Inside the fast
    // loop that holds by construction (tools/check_hidden_loads.py audits the generated code for
    // copies); everywhere else the compiler may copy loop-carried registers at a loop latch, so
    // every other path drains all prefetches before it reaches one.
    auto drain = [&]() {
        if constexpr (FIRST) {
#pragma unroll
            for (int k = 0; k < PD; ++k) {
                wait_slot<0>(pu[k], pf[k]);
                pin_cells(pk[k]);
            }
        }
        pin_coarse(pe);
    };
    // first step of a group: deepest plane (i0 - NP, or i0 - DR with the residual pass) >= 1; last
    // step: shallowest plane i0 + UN - 2 below the top face and zlim, prefetched plane
    // i0 + UN - 1 + PD <= zlast
    auto is_fast = [&](int i0) {
        if (CHUNKS && i0 >= nsteps)
            return false; // the chunk ends here
        return i0 >= (RES ? DR : NP) + 1 && i0 + UN - 2 < zlim && i0 + UN - 2 < zlast &&
               i0 + UN - 1 + PD + (COR ? 2 : 0) <= zlast; // COR: the coarse plane above the prefetched fine one exists
    };
    // The fast groups run in a loop of their own: with a single path through the loop body the
    // register allocator keeps every ring and prefetch register in place across the back edge
    // (a shared loop made it copy the freshly requested prefetch registers at the latch, which
    // forces a full vmcnt(0) drain per group).
    // Every wavefront of the block walks the same sequence of groups (one barrier per time step each);
    // whether a group is a fast one is each stage's own matter.
    for (int g0 = zs; g0 < nsteps_all;) {
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
    MGRT_STAMP_END(wave, lane);
    }; // march

    if constexpr (STAGES == 1) {
        march(std::integral_constant<int, 0>{});
    } else {
        if (stage == 0)
            march(std::integral_constant<int, 0>{});
        else
            march(std::integral_constant<int, 1>{});
    }
}

template <class R, int ITERS, int SR, int NW, int PD, bool RES = false, bool COR = false, int STAGES = 1, bool MAP = false,
          bool RST = false>
mgrt_status launch(const mgrt_grid *g, R *out, const R *in, const R *f, R omega, int zcount, hipStream_t hs,
                   R *res = nullptr, const mgrt_grid *cg = nullptr, const R *e = nullptr, R *rc = nullptr, int cylo = 0,
                   int cyhi = 1 << 30, int kchunk = 0)
{
    constexpr int H = 2 * ITERS * STAGES + (RES ? 1 : 0), OX = kWave - 2 * (H + (RST ? 1 : 0)), OY = NW * SR - 2 * H;
    static_assert(OY > 0, "tile has no interior");
    static_assert(NW * STAGES * kWave <= 1024, "block too large");
    // every strip starts on an even row of its slab (the corrected input relies on it)
    static_assert(!COR || (OY % 2 == 0 && H % 2 == 0), "strips must start on even rows");
    const int ntx = (g->nx + OX - 1) / OX, nty = (g->ny + OY - 1) / OY;
    CoarseField cf = {};
    if (COR || RST) {
        cf.e = e, cf.rc = rc, cf.pitch = cg->pitch, cf.slab = cg->slab;
        cf.nx = cg->nx, cf.ny = cg->ny, cf.nz = cg->nz, cf.y0 = cg->y0;
        cf.ylo = cylo > 0 ? cylo : 0, cf.yhi = cyhi < cg->ny ? cyhi : cg->ny;
        cf.kchunk = RST && ITERS == 0 ? kchunk : 0;
    }
    const int nchunks = cf.kchunk > 0 ? (cf.nz + cf.kchunk - 1) / cf.kchunk : 1; // blocks along z
    // A plane with more tiles than the part runs at once is swept band by band, each band a launch of whole tile rows
    // that all start together.  In one launch the tiles of the later rounds start whenever a CU falls free, at z = 0 beside
    // neighbours that are hundreds of planes further on, and no halo row or column is found in the L2 any more
    // (1025^3 in double precision: 1.35 times the algorithmic bytes against 1.17 at 513^3, 4.4 TB/s of real traffic).
    auto kernel = omega == (R)1 ? &k_smooth3d_fused<R, ITERS, SR, NW, PD, true, RES, COR, STAGES, MAP, RST>
                                : &k_smooth3d_fused<R, ITERS, SR, NW, PD, false, RES, COR, STAGES, MAP, RST>;
    static int resident_of[2] = {0, 0}; // blocks of this kernel a CU holds at once, for either value of OMEGA1
    int &resident = resident_of[omega == (R)1];
    if (!resident) {
        int nb = 0;
        MGRT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NW * STAGES * kWave, 0));
        resident = nb > 0 ? nb : 1;
    }
    const long long room = (long long)runtime().num_cu * resident;
    int nbands = 1;
    if ((long long)ntx * nty > room && runtime().sweep_bands) {
        const int rows_at_once = room / ntx > 0 ? (int)(room / ntx) : 1;
        nbands = (nty + rows_at_once - 1) / rows_at_once;
    }
    const int band_rows = (nty + nbands - 1) / nbands;
    runtime().last_sweep_launches = 0;
    for (int ty0 = 0; ty0 < nty; ty0 += band_rows) {
        const int rows = nty - ty0 < band_rows ? nty - ty0 : band_rows;
        kernel<<<dim3(ntx * rows, nchunks), NW * STAGES * kWave, 0, hs>>>(*g, in, out, f, res, omega, zcount, ntx, nty, ty0, cf);
        ++runtime().last_sweep_launches;
    }
    MGRT_HIP(hipGetLastError());
    return MGRT_OK;
}

template <class R> struct Cfg;
template <> struct Cfg<float> {
    static constexpr int SR1 = 4, NW1 = 8, PF1 = 4; // one iteration per sweep: 4 planes prefetched
    static constexpr int SR2 = 4, NW2 = 8, PF2 = 2; // two iterations per sweep: 2 planes prefetched
};
template <> struct Cfg<double> {
    static constexpr int SR1 = 4, NW1 = 4, PF1 = 4;
    static constexpr int SR2 = 4, NW2 = 8, PF2 = 2; // 238-246 registers: one block of 8 waves per CU
};

// Whether the two-stage sweep of a latency-bound level takes six strips a stage (tiles of 64 x 12 with
// 56 x 4 points of their own) rather than eight: when that still leaves every tile a CU to itself.
bool two_stage_six_strips(const mgrt_grid &g)
{
    const long long tiles = (long long)((g.nx + 55) / 56) * ((g.ny + 3) / 4);
    return tiles <= runtime().num_cu;
}

// res == nullptr: the sweep alone; otherwise the sweep with the residual pass.
template <class R>
mgrt_status smooth_sweep(const mgrt_grid *g, R *out, const R *in, const R *f, R omega, int iters, int zcount,
                         mgrt_stream s, R *res = nullptr)
{
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(g, "mgrt_smooth_sweep")) return st;
    MGRT_REQUIRE(out && f, "mgrt_smooth_sweep: null field");
    MGRT_REQUIRE(out != in && out != f, "mgrt_smooth_sweep: out must not alias in or f");
    MGRT_REQUIRE(!res || (res != out && res != in && res != f), "mgrt_smooth_sweep_residual: r must not alias another field");
    MGRT_REQUIRE(zcount >= 0 && zcount <= g->nz, "mgrt_smooth_sweep: zcount %d outside [0,%d]", zcount, g->nz);
    if (g->cells && (g->dim != 3 || res))
        return fail(MGRT_E_UNSUPPORTED, "mgrt_smooth_sweep: with a byte cell map only the plain 3D sweep exists, use "
                                        "mgrt_smooth_pass");
    if (iters != 1 && iters != 2)
        return fail(MGRT_E_UNSUPPORTED, "mgrt_smooth_sweep: iters must be 1 or 2, got %d", iters);
    if (neumann_extension(g))
        return fail(MGRT_E_UNSUPPORTED, "mgrt_smooth_sweep: the first-order Neumann condition is in mgrt_smooth_pass only");
    hipStream_t hs = (hipStream_t)s;
    if (g->dim != 3)
        return smooth_sweep_2d(g, nullptr, out, res, in, nullptr, f, omega, iters, hs);
    if (!plane_offsets_fit(g, sizeof(R)))
        return fail(MGRT_E_UNSUPPORTED, "mgrt_smooth_sweep: a z plane of 2 GiB or more is beyond the 32-bit in-plane "
                                        "offsets of this kernel, use mgrt_smooth_pass");
    using C = Cfg<R>;
    // strips of two rows halve the work a wave does per step (measured 25-30 % faster from 257^2 down)
    bool small = latency_bound_level(*g);
    if (!small && !g->cells && !res) {
        // A wide but short slab (the share of one of many ranks) can give too few tiles of four-row strips
        // for the CUs while the tiles of two-row strips still fit in one round: take those
        // (1025 x 136 rows in double precision: 114 against 171 tiles, 12 % off the cycle of that slab).
        const int h = 2 * iters, ntx = (g->nx + kWave - 2 * h - 1) / (kWave - 2 * h);
        const int rows_small = (iters == 1 ? 16 : sizeof(R) == 4 ? 16 : 24) - 2 * h;
        const int big = ntx * ((g->ny + 32 - 2 * h - 1) / (32 - 2 * h)), thin = ntx * ((g->ny + rows_small - 1) / rows_small);
        small = 16 * big <= 9 * runtime().num_cu && thin <= runtime().num_cu;
        // 8-byte elements, two iterations: that shape is one stage of twelve two-row strips
        if (small && sizeof(R) == 8 && iters == 2)
            return launch<R, 2, 2, 12, 3>(g, out, in, f, omega, zcount, hs);
    }
    if (g->cells) { // any domain: classes from the byte map (8-byte elements: strips of two rows, for registers)
        if constexpr (sizeof(R) == 4) {
            if (!small)
                return iters == 1 ? launch<R, 1, 4, 8, 4, false, false, 1, true>(g, out, in, f, omega, zcount, hs)
                                  : launch<R, 2, 4, 8, 2, false, false, 1, true>(g, out, in, f, omega, zcount, hs);
        }
        return iters == 1 ? launch<R, 1, 2, 8, 4, false, false, 1, true>(g, out, in, f, omega, zcount, hs)
                          : launch<R, 2, 2, 8, 3, false, false, 1, true>(g, out, in, f, omega, zcount, hs);
    }
    if (res) {
        // Strips of four rows only with 4-byte elements (registers).  Shapes chosen by timing whole
        // cycles: kernel-only timings on warm caches favour other shapes that lose in the cycle.
        if (iters == 1)
            return small || sizeof(R) == 8 ? launch<R, 1, 2, 8, 3, true>(g, out, in, f, omega, zcount, hs, res)
                                           : launch<R, 1, sizeof(R) == 4 ? 4 : 2, 8, 3, true>(g, out, in, f, omega, zcount, hs, res);
        return small || sizeof(R) == 8 ? launch<R, 2, 2, 8, 4, true>(g, out, in, f, omega, zcount, hs, res)
                                       : launch<R, 2, sizeof(R) == 4 ? 4 : 2, 8, 4, true>(g, out, in, f, omega, zcount, hs, res);
    }
    if (iters == 1)
        return small ? launch<R, 1, 2, 8, 4>(g, out, in, f, omega, zcount, hs)
                     : launch<R, 1, C::SR1, C::NW1, C::PF1>(g, out, in, f, omega, zcount, hs);
    // Latency-bound levels: strips of two rows, as two stages of one iteration each: a step of the march
    // then costs what the wavefronts of a SIMD issue in it, and a wavefront issues half as much (105 vs
    // 119 us at 257^3).  Where tiles of six strips a stage still give every tile a CU of its own, those are
    // taken: three wavefronts a SIMD instead of four (129^3: 49 vs 55 us, 65^3: 29 vs 35 and 37 for one
    // stage, 33^3: 20 vs 22 and 23).  4-byte elements: the form that takes two planes a time step
    // (mgrt_smooth_pair.hip: 257^3 107 -> 95 us); 8-byte elements: one plane a step, 124-126 registers with
    // four planes prefetched, under the 128 that 16 wavefronts leave.
    if (small) {
        const bool six = two_stage_six_strips(*g), wide = (long long)g->nx * g->ny >= 100LL * 100LL;
        if constexpr (sizeof(R) == 4) {
            // from 60^2 columns up; below that its longer lag costs what its shorter steps gain (65^3: 25.3 against
            // 27.0 us, 33^3: 17.6 against 17.0)
            if ((long long)g->nx * g->ny >= 60LL * 60LL)
                return smooth_sweep_pair(g, out, in, f, omega, zcount, six, hs);
        }
        if (!six)
            return launch<R, 1, 2, 8, 4, false, false, 2>(g, out, in, f, omega, zcount, hs);
        return wide ? launch<R, 1, 2, 6, 2, false, false, 2>(g, out, in, f, omega, zcount, hs)
                    : launch<R, 1, 2, 6, 4, false, false, 2>(g, out, in, f, omega, zcount, hs);
    }
    return launch<R, 2, C::SR2, C::NW2, C::PF2>(g, out, in, f, omega, zcount, hs);
}

// out = S^iters(u + 4*P(e)): the correction feeding the post-smoother without a pass of its own.
template <class R>
mgrt_status correct_smooth_sweep(const mgrt_grid *fine, const mgrt_grid *coarse, R *out, const R *u, const R *e,
                                 const R *f, R omega, int iters, int zcount, mgrt_stream s)
{
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(fine, "mgrt_correct_smooth_sweep")) return st;
    if (mgrt_status st = check_grid(coarse, "mgrt_correct_smooth_sweep")) return st;
    MGRT_REQUIRE(out && u && e && f, "mgrt_correct_smooth_sweep: null field");
    MGRT_REQUIRE(out != u && out != f && out != e, "mgrt_correct_smooth_sweep: out must not alias an input");
    MGRT_REQUIRE(fine->dim == coarse->dim, "mgrt_correct_smooth_sweep: grids of different dimension");
    MGRT_REQUIRE(coarse->nx == (fine->nx + 1) / 2 && coarse->gny == (fine->gny + 1) / 2,
                 "mgrt_correct_smooth_sweep: coarse grid %dx%d is not the half of %dx%d", coarse->nx, coarse->gny, fine->nx,
                 fine->gny);
    MGRT_REQUIRE(zcount >= 0 && zcount <= fine->nz, "mgrt_correct_smooth_sweep: zcount %d outside [0,%d]", zcount, fine->nz);
    if (fine->cells || coarse->cells)
        return fail(MGRT_E_UNSUPPORTED, "mgrt_correct_smooth_sweep: byte cell maps are not supported");
    if (neumann_extension(fine))
        return fail(MGRT_E_UNSUPPORTED, "mgrt_correct_smooth_sweep: the first-order Neumann condition is in mgrt_smooth_pass only");
    if (iters != 1 && iters != 2)
        return fail(MGRT_E_UNSUPPORTED, "mgrt_correct_smooth_sweep: iters must be 1 or 2, got %d", iters);
    // the coarse rows the fine rows interpolate from must be held, as for mgrt_correct (with an even
    // fine extent the last row would read one coarse row too far, as the reference does)
    const int clo = fine->y0 / 2 - coarse->y0, chi = (fine->ny - 1 + fine->y0 + 1) / 2 - coarse->y0;
    MGRT_REQUIRE(clo >= 0 && chi < coarse->ny, "mgrt_correct_smooth_sweep: coarse slab does not hold rows %d..%d",
                 clo + coarse->y0, chi + coarse->y0);
    if (fine->dim == 3) {
        MGRT_REQUIRE(coarse->nz == (fine->nz + 1) / 2, "mgrt_correct_smooth_sweep: coarse grid has %d planes for %d",
                     coarse->nz, fine->nz);
        if (!correct_sweep_3d_supported(fine, sizeof(R), iters))
            return fail(MGRT_E_UNSUPPORTED, "mgrt_correct_smooth_sweep: in 3D, iters == 2, odd extents, an even slab offset and "
                                            "at least 100^2 columns (8-byte elements: planes of more tiles than the device has CUs) only");
        if constexpr (sizeof(R) == 4) {
            using C = Cfg<R>;
            // the shapes of the plain sweep on each kind of level, two planes prefetched
            if (!latency_bound_level(*fine))
                return launch<R, 2, C::SR2, C::NW2, C::PF2, false, true>(fine, out, u, f, omega, zcount, (hipStream_t)s, nullptr,
                                                                         coarse, e);
            // latency-bound levels: two stages of one iteration each, the correction in the first
            if (two_stage_six_strips(*fine))
                return launch<R, 1, 2, 6, 2, false, true, 2>(fine, out, u, f, omega, zcount, (hipStream_t)s, nullptr, coarse, e);
            return launch<R, 1, 2, 8, 2, false, true, 2>(fine, out, u, f, omega, zcount, (hipStream_t)s, nullptr, coarse, e);
        }
        if constexpr (sizeof(R) == 8) {
            // 8-byte elements, planes swept at the rate of memory: strips of two rows, twelve a tile (with four-row
            // strips the coarse values are a hundred registers too many).  The tile has 16 rows of its own in 24 where
            // the plain sweep has 24 in 32, but the separate correction it spares reads and writes the whole field
            // once more.  The one kernel here that the compiler gives scratch memory (24-28 registers of 168): its
            // spill traffic never touches a register with a load pending (tools/check_hidden_loads.py models the
            // counter with the scratch operations in it), and it is faster with them than the two launches it replaces.
            return launch<R, 2, 2, 12, 2, false, true>(fine, out, u, f, omega, zcount, (hipStream_t)s, nullptr, coarse, e);
        }
        return fail(MGRT_E_UNSUPPORTED, "mgrt_correct_smooth_sweep: no 3D kernel for this element size");
    }
    return smooth_sweep_2d(fine, coarse, out, nullptr, u, e, f, omega, iters, (hipStream_t)s);
}

template <class R> struct Step;
template <> struct Step<float> {
    static constexpr auto residual = mgrt_residual_f32;
    static constexpr auto restrict_ = mgrt_restrict_f32;
};
template <> struct Step<double> {
    static constexpr auto residual = mgrt_residual_f64;
    static constexpr auto restrict_ = mgrt_restrict_f64;
};

// out = S^iters(in) and rc = Restrict(f - A out): the pre-smoother with everything the coarse level
// needs from this one.
template <class R>
mgrt_status smooth_sweep_residual_restrict(const mgrt_grid *fine, const mgrt_grid *coarse, R *out, R *rc, R *r, const R *in,
                                           const R *f, R omega, int iters, int zcount, mgrt_stream s)
{
    const char *who = "mgrt_smooth_sweep_residual_restrict";
    if (mgrt_status st = require_init()) return st;
    if (mgrt_status st = check_grid(fine, who)) return st;
    if (mgrt_status st = check_grid(coarse, who)) return st;
    MGRT_REQUIRE(rc && f, "%s: null field", who);
    MGRT_REQUIRE((!out || (out != in && out != f && rc != out)) && rc != in && rc != f, "%s: outputs must not alias another field", who);
    MGRT_REQUIRE(fine->dim == coarse->dim, "%s: grids of different dimension", who);
    MGRT_REQUIRE(coarse->nx == (fine->nx + 1) / 2 && coarse->gny == (fine->gny + 1) / 2,
                 "%s: coarse grid %dx%d is not the half of %dx%d", who, coarse->nx, coarse->gny, fine->nx, fine->gny);
    MGRT_REQUIRE(zcount >= 0 && zcount <= fine->nz, "%s: zcount %d outside [0,%d]", who, zcount, fine->nz);
    if (fine->cells || coarse->cells)
        return fail(MGRT_E_UNSUPPORTED, "%s: byte cell maps are not supported", who);
    if (neumann_extension(fine))
        return fail(MGRT_E_UNSUPPORTED, "%s: the first-order Neumann condition is in mgrt_smooth_pass only", who);
    if (iters < 0 || iters > 2)
        return fail(MGRT_E_UNSUPPORTED, "%s: iters must be 0, 1 or 2, got %d", who, iters);
    MGRT_REQUIRE(iters > 0 ? out != nullptr : in != nullptr, "%s: null field", who);
    if (fine->nx % 2 == 0 || fine->ny % 2 == 0)
        return fail(MGRT_E_UNSUPPORTED, "%s: grids with odd extents only", who);
    if (fine->y0 != 0 || fine->ny != fine->gny || coarse->y0 != 0 || coarse->ny != coarse->gny)
        return fail(MGRT_E_UNSUPPORTED, "%s: whole grids only, not y-slabs", who);
    if (fine->dim == 2) {
        if (iters == 0)
            return fail(MGRT_E_UNSUPPORTED, "%s: in 2D, iters must be 1 or 2", who);
        return smooth_sweep_2d(fine, coarse, out, rc, in, nullptr, f, omega, iters, (hipStream_t)s);
    }
    MGRT_REQUIRE(coarse->nz == (fine->nz + 1) / 2, "%s: coarse grid has %d planes for %d", who, coarse->nz, fine->nz);
    if (!restrict_sweep_3d_supported(fine, sizeof(R), iters))
        return fail(MGRT_E_UNSUPPORTED, "%s: in 3D, whole grids with odd extents; with iters == 2 more than 300^2 columns and "
                                        "4-byte elements, with iters == 0 any element size; not iters == 1", who);
    MGRT_REQUIRE(r && r != out && r != rc && r != in && r != f, "%s: in 3D, r must be a work array of its own", who);
    // The sweep forms every coarse row except the one on each seam between two rows of tiles; it leaves
    // the three fine residual rows around such a seam in r, and the restriction of those coarse rows
    // alone follows.
    const int parts = runtime().sweep_parts; // both, unless a timing harness asked for one alone
    hipStream_t hs = (hipStream_t)s;
    int oy = 0;
    mgrt_status st = MGRT_OK;
    if (iters == 0) {
        // no smoothing: the residual of `in` and its restriction.  On bandwidth-bound levels as one march
        // over in and f (one ring of halo; 8-byte elements: two planes prefetched, for registers); on the
        // others, where a march costs its step latency whatever it computes, tile by tile through LDS
        // (nothing passes through r, no seams).
        if (latency_bound_level(*fine)) {
            if (!(parts & 1))
                return MGRT_OK;
            // narrow planes: one launch through LDS (65^3: 7 us against 5 + 5); in between the two streaming
            // kernels, the residual passing through r (257^3: 39 + 14 us against 100 for a march)
            if ((long long)fine->nx * fine->ny < 100LL * 100LL)
                return residual_restrict_tiles(fine, coarse, rc, in, f, hs);
            // ... as long as r stays in the last-level cache between the two.  A residual field of more than about
            // 100 MB does not (257^3 in double precision: 151 us for the two kernels, 2.8 times the 53 us of single
            // precision), and there the march wins once it is cut along z into as many chunks of coarse planes as
            // leave every block a CU of its own (no recurrence in z without smoothing): 92 us, bit for bit the same.
            // Where r does stay in cache it is no faster (257^3 float 53.5 against 53.0 us, 193^3 double 47.5
            // against 47.7).
            if ((long long)fine->nx * fine->ny * fine->nz * (long long)sizeof(R) > (96LL << 20)) {
                oy = 32 - 2;
                const int tiles = ((fine->nx + 59) / 60) * ((fine->ny + oy - 1) / oy);
                const int nchunks = runtime().num_cu / tiles > 1 ? runtime().num_cu / tiles : 1;
                const int kchunk = (coarse->nz + nchunks - 1) / nchunks;
                st = launch<R, 0, 4, 8, sizeof(R) == 4 ? 4 : 2, true, false, 1, false, true>(fine, nullptr, in, f, omega, zcount, hs, r, coarse,
                                                                                             nullptr, rc, 0, 1 << 30, kchunk);
                if (st != MGRT_OK)
                    return st;
                const int nseams = (fine->ny + oy - 1) / oy - 1;
                return restrict_rows(fine, coarse, rc, r, oy / 2, oy / 2, nseams, hs);
            }
            if (mgrt_status st2 = Step<R>::residual(fine, r, in, f, 1, nullptr, 0, fine->ny, s))
                return st2;
            return Step<R>::restrict_(fine, coarse, rc, r, 0, coarse->ny, s);
        }
        oy = 32 - 2;
        if (parts & 1)
            st = launch<R, 0, 4, 8, sizeof(R) == 4 ? 4 : 2, true, false, 1, false, true>(fine, nullptr, in, f, omega, zcount, hs, r, coarse, nullptr, rc);
    } else if constexpr (sizeof(R) == 4) {
        constexpr int SR = 4, NW = 8, H = 5;
        oy = NW * SR - 2 * H;
        if (parts & 1)
            st = launch<R, 2, SR, NW, 4, true, false, 1, false, true>(fine, out, in, f, omega, zcount, hs, r, coarse, nullptr, rc);
    } else {
        return fail(MGRT_E_UNSUPPORTED, "%s: no 3D kernel for this element size", who);
    }
    if (st != MGRT_OK)
        return st;
    const int nseams = (fine->ny + oy - 1) / oy - 1;
    return parts & 2 ? restrict_rows(fine, coarse, rc, r, oy / 2, oy / 2, nseams, hs) : MGRT_OK;
}

// The restriction-carrying sweep (iters == 2, 4-byte elements) or the residual + restriction march
// (iters == 0, out unused) on a whole grid or a y-slab: rows [cylo, cyhi) of the coarse array from the
// residual of the smoothed field (of `in`) on the fine rows around them.  Every fine row those coarse rows
// read, its neighbours and (iters == 2) the four halo rows two iterations consume must lie inside the arrays
// and hold valid values on entry.
template <class R>
mgrt_status fused_residual_restrict_rows(const mgrt_grid *fine, const mgrt_grid *coarse, R *out, R *rc, R *r, const R *in,
                                         const R *f, R omega, int iters, int zcount, int cylo, int cyhi, hipStream_t hs)
{
    const char *who = "fused_residual_restrict_rows";
    MGRT_REQUIRE(fine->dim == 3 && !fine->cells && !neumann_extension(fine) && fine->nx % 2 == 1 && fine->gny % 2 == 1 &&
                     fine->nz % 2 == 1 && fine->nz >= 5 && fine->y0 % 2 == 0 && plane_offsets_fit(fine, sizeof(R)),
                 "%s: 3D analytic boxes with odd extents, slabs at even offsets", who);
    MGRT_REQUIRE(iters == 0 || (iters == 2 && sizeof(R) == 4 && out), "%s: no iteration, or two with 4-byte elements", who);
    MGRT_REQUIRE(0 <= cylo && cylo <= cyhi && cyhi <= coarse->ny, "%s: coarse rows [%d,%d) outside [0,%d)", who, cylo, cyhi,
                 coarse->ny);
    if (cylo == cyhi && iters == 0)
        return MGRT_OK;
    if (cylo < cyhi) {
        // fine rows 2j - 1 .. 2j + 1 of a coarse row j, their stencils and the sweep's halo, as far as the grid goes
        const int need = 2 + 2 * iters;
        const int glo = 2 * (cylo + coarse->y0) - need, ghi = 2 * (cyhi - 1 + coarse->y0) + need; // global fine rows
        MGRT_REQUIRE((glo > 0 ? glo : 0) >= fine->y0 && (ghi < fine->gny - 1 ? ghi : fine->gny - 1) <= fine->y0 + fine->ny - 1,
                     "%s: the fine slab does not hold the rows coarse rows [%d,%d) read", who, cylo, cyhi);
    }
    int OY;
    mgrt_status st;
    if (iters == 0) {
        // strips of two rows where tiles of four-row strips would leave most CUs without one (thin slabs)
        const int ntx = (fine->nx + 59) / 60;
        const bool thin = 2 * ntx * ((fine->ny + 29) / 30) <= runtime().num_cu;
        OY = (thin ? 16 : 32) - 2;
        st = thin ? launch<R, 0, 2, 8, 4, true, false, 1, false, true>(fine, nullptr, in, f, (R)1, fine->nz, hs, r, coarse, nullptr, rc,
                                                                    cylo, cyhi)
                  : launch<R, 0, 4, 8, sizeof(R) == 4 ? 4 : 2, true, false, 1, false, true>(fine, nullptr, in, f, (R)1, fine->nz, hs, r,
                                                                                         coarse, nullptr, rc, cylo, cyhi);
    } else if constexpr (sizeof(R) == 4) {
        OY = 32 - 2 * 5;
        st = launch<R, 2, 4, 8, 4, true, false, 1, false, true>(fine, out, in, f, omega, zcount, hs, r, coarse, nullptr, rc, cylo, cyhi);
    } else {
        return fail(MGRT_E_UNSUPPORTED, "%s: no kernel", who);
    }
    if (st != MGRT_OK)
        return st;
    // coarse rows on the seams between rows of tiles: fine row k * OY of the arrays, k = 1 ..
    const int nty = (fine->ny + OY - 1) / OY, base = fine->y0 / 2 - coarse->y0;
    int k0 = 1, k1 = nty - 1;
    while (k0 <= k1 && base + k0 * (OY / 2) < cylo) ++k0;
    while (k1 >= k0 && base + k1 * (OY / 2) >= cyhi) --k1;
    return restrict_rows(fine, coarse, rc, r, base + k0 * (OY / 2), OY / 2, k1 - k0 + 1, hs);
}

} // namespace

namespace mgrt {
mgrt_status residual_restrict_march(const mgrt_grid *fine, const mgrt_grid *coarse, float *rc, float *r, const float *in,
                                    const float *f, int cylo, int cyhi, hipStream_t hs)
{
    return fused_residual_restrict_rows<float>(fine, coarse, nullptr, rc, r, in, f, 1.0f, 0, fine->nz, cylo, cyhi, hs);
}
mgrt_status residual_restrict_march(const mgrt_grid *fine, const mgrt_grid *coarse, double *rc, double *r, const double *in,
                                    const double *f, int cylo, int cyhi, hipStream_t hs)
{
    return fused_residual_restrict_rows<double>(fine, coarse, nullptr, rc, r, in, f, 1.0, 0, fine->nz, cylo, cyhi, hs);
}
mgrt_status sweep_residual_restrict_rows(const mgrt_grid *fine, const mgrt_grid *coarse, float *out, float *rc, float *r,
                                         const float *in, const float *f, float omega, int zcount, int cylo, int cyhi,
                                         hipStream_t hs)
{
    return fused_residual_restrict_rows<float>(fine, coarse, out, rc, r, in, f, omega, 2, zcount, cylo, cyhi, hs);
}
} // namespace mgrt

extern "C" mgrt_status mgrt_smooth_sweep_residual_restrict_f32(const mgrt_grid *fine, const mgrt_grid *coarse, float *out,
                                                               float *rc, float *r, const float *in, const float *f,
                                                               float omega, int iters, int zcount, mgrt_stream s)
{
    return smooth_sweep_residual_restrict<float>(fine, coarse, out, rc, r, in, f, omega, iters, zcount, s);
}

extern "C" mgrt_status mgrt_smooth_sweep_residual_restrict_f64(const mgrt_grid *fine, const mgrt_grid *coarse, double *out,
                                                               double *rc, double *r, const double *in, const double *f,
                                                               double omega, int iters, int zcount, mgrt_stream s)
{
    return smooth_sweep_residual_restrict<double>(fine, coarse, out, rc, r, in, f, omega, iters, zcount, s);
}

extern "C" int mgrt_smooth_sweep_residual_restrict_supported(const mgrt_grid *fine, int elem_size, int iters)
{
    if (!fine || fine->cells || iters < 0 || iters > 2 || (elem_size != 4 && elem_size != 8) || neumann_extension(fine) ||
        fine->nx % 2 == 0 || fine->gny % 2 == 0 || fine->y0 != 0 || fine->ny != fine->gny)
        return 0;
    return fine->dim == 2 ? iters > 0 : restrict_sweep_3d_supported(fine, (size_t)elem_size, iters);
}

extern "C" int mgrt_correct_smooth_sweep_supported(const mgrt_grid *fine, int elem_size, int iters)
{
    if (!fine || fine->cells || (iters != 1 && iters != 2) || (elem_size != 4 && elem_size != 8) || neumann_extension(fine))
        return 0;
    if (fine->dim == 2)
        return fine->nx % 2 == 1 && fine->gny % 2 == 1;
    return correct_sweep_3d_supported(fine, (size_t)elem_size, iters);
}

extern "C" mgrt_status mgrt_correct_smooth_sweep_f32(const mgrt_grid *fine, const mgrt_grid *coarse, float *out,
                                                     const float *u, const float *e, const float *f, float omega,
                                                     int iters, int zcount, mgrt_stream s)
{
    return correct_smooth_sweep<float>(fine, coarse, out, u, e, f, omega, iters, zcount, s);
}

extern "C" mgrt_status mgrt_correct_smooth_sweep_f64(const mgrt_grid *fine, const mgrt_grid *coarse, double *out,
                                                     const double *u, const double *e, const double *f, double omega,
                                                     int iters, int zcount, mgrt_stream s)
{
    return correct_smooth_sweep<double>(fine, coarse, out, u, e, f, omega, iters, zcount, s);
}

extern "C" mgrt_status mgrt_smooth_sweep_f32(const mgrt_grid *g, float *out, const float *in, const float *f,
                                             float omega, int iters, int zcount, mgrt_stream s)
{
    return smooth_sweep<float>(g, out, in, f, omega, iters, zcount, s);
}

extern "C" mgrt_status mgrt_smooth_sweep_residual_f32(const mgrt_grid *g, float *out, float *r, const float *in,
                                                      const float *f, float omega, int iters, int zcount,
                                                      mgrt_stream s)
{
    if (!r)
        return fail(MGRT_E_INVALID, "mgrt_smooth_sweep_residual: null field");
    return smooth_sweep<float>(g, out, in, f, omega, iters, zcount, s, r);
}

extern "C" mgrt_status mgrt_smooth_sweep_residual_f64(const mgrt_grid *g, double *out, double *r, const double *in,
                                                      const double *f, double omega, int iters, int zcount,
                                                      mgrt_stream s)
{
    if (!r)
        return fail(MGRT_E_INVALID, "mgrt_smooth_sweep_residual: null field");
    return smooth_sweep<double>(g, out, in, f, omega, iters, zcount, s, r);
}

extern "C" mgrt_status mgrt_smooth_sweep_f64(const mgrt_grid *g, double *out, const double *in, const double *f,
                                             double omega, int iters, int zcount, mgrt_stream s)
{
    return smooth_sweep<double>(g, out, in, f, omega, iters, zcount, s);
}
