// functionhandler.cpp -- grid sampling (functionhandler.cpp:25-101).
// Coordinates are (real)i/(n-1); inner cells get f*dh*dh, Dirichlet cells the boundary value,
// Neumann cells boundary*dh, outer cells are left as allocated (0 here).  The exact solution is
// sampled on every cell that is not outer.
#include "functionhandler.h"
#include <vector>

namespace {

enum Kind { Inner = 0, Outer = 1, Dirichlet = 2, Neumann = 3 }; // values of BorderHandler*::CellType

// Fills a dense x-fastest array: `cls(i,j,k)` classifies a cell, `rhs(x,y,z)` and `bnd(x,y,z)` are the
// two user functions in unit-cube coordinates.  With solution_only the array holds bnd on every
// cell that is not outer.
template <class Cls, class F, class B>
std::vector<real> sample(int nx, int ny, int nz, real dh, bool solution_only, Cls cls, F rhs, B bnd)
{
    std::vector<real> a((size_t)nx * ny * nz, (real)0);
    for (int k = 0; k < nz; ++k) {
        const real z = nz > 1 ? (real)(k) / (nz - 1) : (real)0;
        for (int j = 0; j < ny; ++j) {
            const real y = (real)(j) / (ny - 1);
            for (int i = 0; i < nx; ++i) {
                const real x = (real)(i) / (nx - 1);
                real &v = a[i + (size_t)nx * (j + (size_t)ny * k)];
                const int t = cls(i, j, k);
                if (solution_only) {
                    if (t != Outer)
                        v = bnd(x, y, z);
                } else if (t == Inner) {
                    v = rhs(x, y, z) * dh * dh;
                } else if (t == Dirichlet) {
                    v = bnd(x, y, z);
                } else if (t == Neumann) {
                    v = bnd(x, y, z) * dh;
                }
            }
        }
    }
    return a;
}

} // namespace

Buffer2D FunctionHandler2D::discretize_func(int dimx, int dimy, real dh, const BorderHandler2D &bord)
{
    std::vector<real> a = sample(
        dimx, dimy, 1, dh, false, [&](int i, int j, int) { return (int)bord.cellType(i, j, dimx, dimy); },
        [&](real x, real y, real) { return m_pFunc(x, y); }, [&](real x, real y, real) { return m_pBord(x, y); });
    return Buffer2D(dimx, dimy, a.data(), Buffer2D::ReadOnly);
}

Buffer2D FunctionHandler2D::discretize_sol(int dimx, int dimy, real dh, const BorderHandler2D &bord)
{
    std::vector<real> a = sample(
        dimx, dimy, 1, dh, true, [&](int i, int j, int) { return (int)bord.cellType(i, j, dimx, dimy); },
        [&](real, real, real) { return (real)0; }, [&](real x, real y, real) { return m_pSol(x, y); });
    return Buffer2D(dimx, dimy, a.data(), Buffer2D::ReadOnly);
}

Buffer3D FunctionHandler3D::discretize_func(int dimx, int dimy, int dimz, real dh, const BorderHandler3D &bord)
{
    std::vector<real> a = sample(
        dimx, dimy, dimz, dh, false, [&](int i, int j, int k) { return (int)bord.cellType(i, j, k, dimx, dimy, dimz); },
        [&](real x, real y, real z) { return m_pFunc(x, y, z); }, [&](real x, real y, real z) { return m_pBord(x, y, z); });
    return Buffer3D(dimx, dimy, dimz, a.data(), Buffer3D::ReadOnly);
}

Buffer3D FunctionHandler3D::discretize_sol(int dimx, int dimy, int dimz, real dh, const BorderHandler3D &bord)
{
    std::vector<real> a = sample(
        dimx, dimy, dimz, dh, true, [&](int i, int j, int k) { return (int)bord.cellType(i, j, k, dimx, dimy, dimz); },
        [&](real, real, real) { return (real)0; }, [&](real x, real y, real z) { return m_pSol(x, y, z); });
    return Buffer3D(dimx, dimy, dimz, a.data(), Buffer3D::ReadOnly);
}
