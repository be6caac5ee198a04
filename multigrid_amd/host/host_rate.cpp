// host_rate.cpp -- what a caller of the C++ classes sees when the fields start and end in host
// memory: upload of the right-hand side, `cycles` V(2,2) cycles, download of the solution.
//   host_rate N CYCLES      (3D N^3, float or double by build, all z planes smoothed)
#include "auxiliary.h"
#include "functionhandler.h"
#include "rectangularborderhandler.h"
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <vector>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 513, cycles = argc > 2 ? atoi(argv[2]) : 10;
    try {
        ParallelepipedalBorderHandler box;
        MultigridSolver3D solver("mg_1.cl", box);
        solver.setFullZRange(true);
        const size_t count = (size_t)n * n * n;
        std::vector<real> host(count);
        const double pi = 3.14159265358979323846, h = 1.0 / (n - 1);
        for (int k = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) {
                    const bool face = i == 0 || j == 0 || k == 0 || i == n - 1 || j == n - 1 || k == n - 1;
                    const double s = sin(pi * i * h) * sin(pi * j * h) * sin(pi * k * h);
                    host[i + (size_t)n * (j + (size_t)n * k)] = (real)(face ? s : -3.0 * pi * pi * s * h * h);
                }
        // warm the plan and the graph with a first solve
        {
            Buffer3D f(n, n, n, host.data());
            Buffer3D u = Buffer3D::empty(n, n, n, solver.queue());
            solver.mg(u, f, 1.0, 2, 2, 1, 1);
            solver.wait();
        }
        const double t0 = now();
        Buffer3D f(n, n, n, host.data());
        Buffer3D u = Buffer3D::empty(n, n, n, solver.queue());
        const double t1 = now();
        solver.mg(u, f, 1.0, 2, 2, 1, cycles);
        solver.wait();
        const double t2 = now();
        TridimArray<real> sol = u.read(solver.queue());
        const double t3 = now();
        const double dof = (double)count * cycles;
        std::cout << "n " << n << " cycles " << cycles << " elem_bytes " << sizeof(real) << "\n"
                  << "upload_s " << t1 - t0 << " (" << count * sizeof(real) / (t1 - t0) / 1e9 << " GB/s)\n"
                  << "cycles_s " << t2 - t1 << " (" << dof / (t2 - t1) / 1e9 << " GDOF/s resident)\n"
                  << "download_s " << t3 - t2 << " (" << count * sizeof(real) / (t3 - t2) / 1e9 << " GB/s)\n"
                  << "total_s " << t3 - t0 << " (" << dof / (t3 - t0) / 1e9 << " GDOF/s including the transfers)\n"
                  << "centre " << sol(n / 2, n / 2, n / 2) << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "host_rate: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
