// rectangularborderhandler.cpp -- box domains and map-backed general domains.
#include "rectangularborderhandler.h"

namespace mgrt {

void CellMapCache::upload(Entry &e)
{
    e.device = deviceAlloc(e.host.size());
    MGRT_CHECK(mgrt_upload(e.device.get(), e.host.data(), e.host.size(), 0));
    MGRT_CHECK(mgrt_sync(0));
}

void MappedBorderHandler2D::setarg(int, KernelArgs &ker, int dimx, int dimy)
{
    const CellMapCache::Entry &e =
        m_cache.get(dimx, dimy, 1, [&](int x, int y, int) { return (int)cellType(x, y, dimx, dimy); });
    ker.grid.cells = static_cast<const unsigned char *>(e.device.get());
    ker.host_cells = e.host.data();
}

void MappedBorderHandler3D::setarg(int, KernelArgs &ker, int dimx, int dimy, int dimz)
{
    const CellMapCache::Entry &e =
        m_cache.get(dimx, dimy, dimz, [&](int x, int y, int z) { return (int)cellType(x, y, z, dimx, dimy, dimz); });
    ker.grid.cells = static_cast<const unsigned char *>(e.device.get());
    ker.host_cells = e.host.data();
}

} // namespace mgrt

// The box handlers describe their domain by its six face classes (rectangularborderhandler.cpp:
// 46-68,89-114 build the same thing as a per-point array): no map is stored or read.
static void allDirichlet(mgrt::KernelArgs &ker)
{
    for (int i = 0; i < 6; ++i)
        ker.grid.face[i] = MGRT_CELL_DIRICHLET;
    ker.grid.cells = 0;
    ker.host_cells = 0;
}

void RectangularBorderHandler::setarg(int, mgrt::KernelArgs &ker, int, int) { allDirichlet(ker); }

BorderHandler2D::CellType RectangularBorderHandler::cellType(int x, int y, int dimx, int dimy) const
{
    const bool inside = x >= 0 && y >= 0 && x < dimx && y < dimy;
    const bool rim = x == 0 || y == 0 || x == dimx - 1 || y == dimy - 1;
    return rim ? CellDirichlet : inside ? CellInner : CellOuter; // rectangularborderhandler.cpp:26-31
}

void ParallelepipedalBorderHandler::setarg(int, mgrt::KernelArgs &ker, int, int, int) { allDirichlet(ker); }

BorderHandler3D::CellType ParallelepipedalBorderHandler::cellType(int x, int y, int z, int dimx, int dimy, int dimz) const
{
    const bool rim = x == 0 || y == 0 || z == 0 || x == dimx - 1 || y == dimy - 1 || z == dimz - 1;
    // rectangularborderhandler.cpp:70-75 (its z test reads `z >= dimz-1`, which the rim test has
    // already taken: same classes as here)
    const bool inside = x >= 0 && y >= 0 && z >= 0 && x < dimx && y < dimy && z < dimz;
    return rim ? CellDirichlet : inside ? CellInner : CellOuter;
}
