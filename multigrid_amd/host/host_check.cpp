// host_check.cpp -- test driver for the C++ classes: runs one solve through the public interface
// and dumps the fields, so that a test can compare them with an independent computation.
//
//   host_check OUT DIM NX NY NZ MODE OMEGA A1 A2 V ITERS FULLZ DOMAIN
//     MODE    mg | fmg | sor | steps   (steps: one cycle spelled out with the step methods)
//             slab2mg | slab3mg | slab2fmg   3D box: the solve by 2 or 3 distributed solvers, one thread each,
//                                            every one holding its rows only (MultigridSolver3D::distribute)
//     DOMAIN  box | map                (map: a domain with Neumann and outer cells, through
//                                       MappedBorderHandler; the same rule as in the test)
//   OUT receives: int32 nx, ny, nz, elem_size; then f, u, r, err as dense arrays; then 5 doubles:
//   L2Norm(r), LInfNorm(r), Average(u), ExactL2Norm(r), ExactLInfNorm(r).
#include "auxiliary.h"
#include "functionhandler.h"
#include "rectangularborderhandler.h"
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifndef M_PI
#define M_PI 3.141592653589793238462643
#endif

namespace {

real f3(real x, real y, real z) { return -3.0 * M_PI * M_PI * sin(M_PI * x) * sin(M_PI * y) * sin(M_PI * z); }
real s3(real x, real y, real z) { return sin(M_PI * x) * sin(M_PI * y) * sin(M_PI * z); }
real f2(real x, real y) { return -M_PI * M_PI * 2 * sin(M_PI * x) * sin(M_PI * y); }
real s2(real x, real y) { return sin(M_PI * x) * sin(M_PI * y); }

// x faces Dirichlet, y faces and z = 0 Neumann, top Dirichlet, a Dirichlet ring around an outer core
int rule(int x, int y, int z, int nx, int ny, int nz, bool threeD)
{
    if (x == 0 || x == nx - 1)
        return MGRT_CELL_DIRICHLET;
    if (y == 0 || y == ny - 1 || (threeD && z == 0))
        return MGRT_CELL_NEUMANN;
    if (threeD && z == nz - 1)
        return MGRT_CELL_DIRICHLET;
    const int r2 = (x - nx / 2) * (x - nx / 2) + (y - ny / 2) * (y - ny / 2);
    if (nx > 5 && r2 < (nx / 5) * (nx / 5))
        return r2 < (nx / 8) * (nx / 8) ? MGRT_CELL_OUTER : MGRT_CELL_DIRICHLET;
    return MGRT_CELL_INNER;
}

struct Holey3D : mgrt::MappedBorderHandler3D {
    virtual CellType cellType(int x, int y, int z, int nx, int ny, int nz) const
    {
        return (CellType)rule(x, y, z, nx, ny, nz, true);
    }
};
struct Holey2D : mgrt::MappedBorderHandler2D {
    virtual CellType cellType(int x, int y, int nx, int ny) const { return (CellType)rule(x, y, 0, nx, ny, 1, false); }
};

struct Args {
    std::string out, mode, domain;
    int dim, nx, ny, nz, a1, a2, v, iters, fullz;
    real omega;
};

template <class Array> void put(FILE *fp, const Array &a, size_t n)
{
    if (fwrite(a.data(), sizeof(real), n, fp) != n)
        throw std::runtime_error("short write");
}

// ---- ranks as threads of this process: a callback transport through host memory ------------------
struct Hub {
    explicit Hub(int n) : world(n), box(n * n), full(n * n, 0), arrived(0), generation(0), sum(0) {}
    const int world;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::vector<char> > box; // [from * world + to]: what `from` has posted for `to`
    std::vector<int> full;
    int arrived, generation;             // a barrier that can be used again
    std::vector<char> gathered;
    double sum;
    // the last to arrive runs `last` (under the lock), then all leave
    template <class F> void barrier(F last)
    {
        std::unique_lock<std::mutex> lk(m);
        const int gen = generation;
        if (++arrived == world) {
            last();
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else
            while (gen == generation)
                cv.wait(lk);
    }
};
struct Rank {
    Hub *hub;
    int rank;
};
void nothing() {}

int cb_sendrecv(void *ctx, int n, const int *peers, void *const *send, void *const *recv, const size_t *bytes)
{
    Rank *me = (Rank *)ctx;
    Hub &h = *me->hub;
    for (int k = 0; k < n; ++k) {
        const int slot = me->rank * h.world + peers[k];
        std::vector<char> tmp(bytes[k]);
        if (mgrt_download(tmp.data(), send[k], bytes[k], 0) != MGRT_OK)
            return 1;
        std::unique_lock<std::mutex> lk(h.m);
        while (h.full[slot])
            h.cv.wait(lk);
        h.box[slot].swap(tmp);
        h.full[slot] = 1;
        h.cv.notify_all();
    }
    for (int k = 0; k < n; ++k) {
        const int slot = peers[k] * h.world + me->rank;
        std::vector<char> tmp;
        {
            std::unique_lock<std::mutex> lk(h.m);
            while (!h.full[slot])
                h.cv.wait(lk);
            tmp.swap(h.box[slot]);
            h.full[slot] = 0;
            h.cv.notify_all();
        }
        if (tmp.size() != bytes[k] || mgrt_upload(recv[k], tmp.data(), bytes[k], 0) != MGRT_OK)
            return 1;
    }
    return 0;
}

int cb_allgather(void *ctx, const void *send, void *recv, size_t bytes)
{
    Rank *me = (Rank *)ctx;
    Hub &h = *me->hub;
    std::vector<char> mine(bytes);
    if (mgrt_download(mine.data(), send, bytes, 0) != MGRT_OK)
        return 1;
    {
        std::unique_lock<std::mutex> lk(h.m);
        h.gathered.resize(bytes * h.world);
        memcpy(&h.gathered[bytes * me->rank], mine.data(), bytes);
    }
    h.barrier(nothing); // all parts are in
    const bool ok = mgrt_upload(recv, h.gathered.data(), bytes * h.world, 0) == MGRT_OK;
    h.barrier(nothing); // all have read before the next call writes
    return ok ? 0 : 1;
}

int cb_allreduce_sum(void *ctx, double *value)
{
    Rank *me = (Rank *)ctx;
    Hub &h = *me->hub;
    double v = 0;
    if (mgrt_download(&v, value, sizeof v, 0) != MGRT_OK)
        return 1;
    // summed in rank order so that every run gives the same bits
    static std::vector<double> parts;
    {
        std::unique_lock<std::mutex> lk(h.m);
        parts.resize(h.world);
        parts[me->rank] = v;
    }
    h.barrier([&h]() {
        h.sum = 0;
        for (int r = 0; r < h.world; ++r)
            h.sum += parts[r];
    });
    const double total = h.sum;
    h.barrier(nothing);
    return mgrt_upload(value, &total, sizeof total, 0) == MGRT_OK ? 0 : 1;
}

// The solve of `mode` by `world` distributed solvers; returns the whole field gathered on the host.
std::vector<real> solve_distributed(const Args &a, int world, const std::string &mode, const std::vector<real> &f_whole)
{
    Hub hub(world);
    std::vector<real> u_whole((size_t)a.nx * a.ny * a.nz, (real)0);
    std::vector<std::string> errors(world);
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r)
        threads.push_back(std::thread([&, r]() {
            try {
                Rank me = {&hub, r};
                mgrt_transport t = mgrt_transport();
                t.kind = MGRT_TRANSPORT_CALLBACK;
                t.rank = r, t.nranks = world;
                t.ctx = &me;
                t.sendrecv = cb_sendrecv, t.allgather = cb_allgather, t.allreduce_sum = cb_allreduce_sum;
                ParallelepipedalBorderHandler box;
                MultigridSolver3D solver("mg_1.cl", box);
                solver.setFullZRange(a.fullz != 0);
                solver.distribute(t, a.ny, /*crossover*/ 9 * 9 * 9);
                int lo, hi, own_lo, own_hi;
                solver.slabRows(a.nx, a.nz, &lo, &hi, &own_lo, &own_hi);
                const int rows = hi - lo;
                std::vector<real> fl((size_t)a.nx * rows * a.nz), ul(fl.size(), (real)0);
                for (int z = 0; z < a.nz; ++z)
                    memcpy(&fl[(size_t)z * rows * a.nx], &f_whole[((size_t)z * a.ny + lo) * a.nx], sizeof(real) * rows * a.nx);
                Buffer3D f(a.nx, rows, a.nz, fl.data());
                Buffer3D u(a.nx, rows, a.nz, ul.data());
                if (mode == "mg")
                    solver.mg(u, f, a.omega, a.a1, a.a2, a.v, a.iters);
                else
                    u = solver.fmg(f, a.omega, a.a1, a.a2, a.v, a.iters);
                solver.wait();
                TridimArray<real> got = u.read(solver.queue());
                for (int z = 0; z < a.nz; ++z) // rows of different ranks do not overlap
                    memcpy(&u_whole[((size_t)z * a.ny + own_lo) * a.nx], got.data() + ((size_t)z * rows + (own_lo - lo)) * a.nx,
                           sizeof(real) * (own_hi - own_lo) * a.nx);
            } catch (const std::exception &e) {
                errors[r] = e.what();
            }
        }));
    for (int r = 0; r < world; ++r)
        threads[r].join();
    for (int r = 0; r < world; ++r)
        if (!errors[r].empty())
            throw std::runtime_error("rank " + std::to_string(r) + ": " + errors[r]);
    return u_whole;
}

int run3(const Args &a)
{
    ParallelepipedalBorderHandler box;
    Holey3D holey;
    BorderHandler3D &h = a.domain == "map" ? static_cast<BorderHandler3D &>(holey) : box;
    MultigridSolver3D solver("mg_1.cl", h);
    solver.setFullZRange(a.fullz != 0);
    FunctionHandler3D fh(f3, s3, s3);
    const real dh = 1.0 / (a.nx - 1);
    Buffer3D f = fh.discretize_func(a.nx, a.ny, a.nz, dh, h);
    Buffer3D u = Buffer3D::empty(a.nx, a.ny, a.nz, solver.queue());
    if (a.mode == "mg")
        solver.mg(u, f, a.omega, a.a1, a.a2, a.v, a.iters);
    else if (a.mode == "fmg")
        u = solver.fmg(f, a.omega, a.a1, a.a2, a.v, a.iters);
    else if (a.mode.compare(0, 4, "slab") == 0 && a.mode.size() > 5) {
        if (a.domain != "box")
            throw std::runtime_error("distributed solves take box domains");
        TridimArray<real> fh_host = f.read(solver.queue());
        const std::vector<real> f_whole(fh_host.data(), fh_host.data() + (size_t)a.nx * a.ny * a.nz);
        std::vector<real> got = solve_distributed(a, a.mode[4] - '0', a.mode.substr(5), f_whole);
        u = Buffer3D(a.nx, a.ny, a.nz, got.data());
    } else if (a.mode == "sor")
        solver.smoother_iterate(u, f, a.omega, a.a1);
    else if (a.mode == "steps") {
        // one two-grid cycle from the public step methods
        const int cx = (a.nx + 1) / 2, cy = (a.ny + 1) / 2, cz = (a.nz + 1) / 2;
        Buffer3D tmp = Buffer3D::empty(a.nx, a.ny, a.nz, solver.queue());
        Buffer3D rc(cx, cy, cz), e(cx, cy, cz);
        solver.smoother_iterate(u, f, a.omega, a.a1);
        solver.zero_mem(e);
        solver.compute_residuals(tmp, u, f);
        solver.restrict(rc, tmp);
        solver.smoother_iterate(e, rc, a.omega, a.a1 + a.a2);
        solver.correct_residual(tmp, u, e);
        std::swap(u, tmp);
        solver.smoother_iterate(u, f, a.omega, a.a2);
    } else
        throw std::runtime_error("unknown mode " + a.mode);
    solver.zero_out(u);
    Buffer3D exact = fh.discretize_sol(a.nx, a.ny, a.nz, dh, h);
    Buffer3D err = Difference(u, exact, solver.queue());
    Buffer3D r = Buffer3D::empty(a.nx, a.ny, a.nz, solver.queue());
    solver.compute_residuals(r, u, f);
    solver.wait();
    const double norms[5] = {L2Norm(r, solver.queue()), LInfNorm(r, solver.queue()), Average(u, solver.queue()),
                             ExactL2Norm(r, a.nz, solver.queue()), ExactLInfNorm(r, a.nz, solver.queue())};
    FILE *fp = fopen(a.out.c_str(), "wb");
    if (!fp)
        throw std::runtime_error("cannot open " + a.out);
    const int head[4] = {a.nx, a.ny, a.nz, (int)sizeof(real)};
    fwrite(head, sizeof head, 1, fp);
    const size_t n = (size_t)a.nx * a.ny * a.nz;
    put(fp, f.read(solver.queue()), n);
    put(fp, u.read(solver.queue()), n);
    put(fp, r.read(solver.queue()), n);
    put(fp, err.read(solver.queue()), n);
    fwrite(norms, sizeof norms, 1, fp);
    fclose(fp);
    return 0;
}

int run2(const Args &a)
{
    RectangularBorderHandler box;
    Holey2D holey;
    BorderHandler2D &h = a.domain == "map" ? static_cast<BorderHandler2D &>(holey) : box;
    MultigridSolver2D solver("mg_0.cl", h);
    FunctionHandler2D fh(f2, s2, s2);
    const real dh = 1.0 / (a.nx - 1);
    Buffer2D f = fh.discretize_func(a.nx, a.ny, dh, h);
    Buffer2D u = Buffer2D::empty(a.nx, a.ny, solver.queue());
    if (a.mode == "mg")
        solver.mg(u, f, a.omega, a.a1, a.a2, a.v, a.iters);
    else if (a.mode == "fmg")
        u = solver.fmg(f, a.omega, a.a1, a.a2, a.v, a.iters);
    else if (a.mode == "sor")
        solver.smoother_iterate(u, f, a.omega, a.a1);
    else if (a.mode == "debug") {
        solver.m_debugPrintResiduals = true;
        solver.mg(u, f, a.omega, a.a1, a.a2, a.v, a.iters);
    } else
        throw std::runtime_error("unknown mode " + a.mode);
    solver.zero_out(u);
    Buffer2D exact = fh.discretize_sol(a.nx, a.ny, dh, h);
    Buffer2D err = Difference(u, exact, solver.queue());
    Buffer2D r = Buffer2D::empty(a.nx, a.ny, solver.queue());
    solver.compute_residuals(r, u, f);
    solver.wait();
    const double norms[5] = {L2Norm(r, solver.queue()), LInfNorm(r, solver.queue()), Average(u, solver.queue()),
                             ExactL2Norm(r, 1, solver.queue()), ExactLInfNorm(r, 1, solver.queue())};
    FILE *fp = fopen(a.out.c_str(), "wb");
    if (!fp)
        throw std::runtime_error("cannot open " + a.out);
    const int head[4] = {a.nx, a.ny, 1, (int)sizeof(real)};
    fwrite(head, sizeof head, 1, fp);
    const size_t n = (size_t)a.nx * a.ny;
    put(fp, f.read(solver.queue()), n);
    put(fp, u.read(solver.queue()), n);
    put(fp, r.read(solver.queue()), n);
    put(fp, err.read(solver.queue()), n);
    fwrite(norms, sizeof norms, 1, fp);
    fclose(fp);
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc != 14) {
        std::cerr << "usage: host_check OUT DIM NX NY NZ MODE OMEGA A1 A2 V ITERS FULLZ DOMAIN" << std::endl;
        return 2;
    }
    Args a;
    a.out = argv[1];
    a.dim = atoi(argv[2]);
    a.nx = atoi(argv[3]);
    a.ny = atoi(argv[4]);
    a.nz = atoi(argv[5]);
    a.mode = argv[6];
    a.omega = (real)atof(argv[7]);
    a.a1 = atoi(argv[8]);
    a.a2 = atoi(argv[9]);
    a.v = atoi(argv[10]);
    a.iters = atoi(argv[11]);
    a.fullz = atoi(argv[12]);
    a.domain = argv[13];
    try {
        return a.dim == 2 ? run2(a) : run3(a);
    } catch (const std::exception &e) {
        std::cerr << "host_check: " << e.what() << std::endl;
        return 1;
    }
}
