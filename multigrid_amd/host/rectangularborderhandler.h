// rectangularborderhandler.h -- the two box domains of the reference
// (rectangularborderhandler.h:28-59): every face Dirichlet, everything else inner; and a base for
// handlers of arbitrary domains that derives the cell map from cellType().
#ifndef RECTANGULARBORDERHANDLER_H
#define RECTANGULARBORDERHANDLER_H

#include "multigridsolver2D.h"
#include "multigridsolver3D.h"
#include <map>
#include <vector>

namespace mgrt {
// Per-extent cache of byte maps (host copy + device copy), the counterpart of the handler-owned
// m_bufferMap of the reference (rectangularborderhandler.h:36,58): built on first use, kept for the
// handler's lifetime.
class CellMapCache
{
public:
    struct Entry {
        std::vector<unsigned char> host;
        std::shared_ptr<void> device;
    };
    // classify(x, y, z) -> MGRT_CELL_* is called once per point of a grid not seen before
    template <class F> const Entry &get(int w, int h, int d, F classify)
    {
        const detail::Dims key = {w, h, d};
        std::map<detail::Dims, Entry>::iterator it = m_maps.find(key);
        if (it == m_maps.end()) {
            Entry e;
            e.host.resize((size_t)w * h * d);
            size_t i = 0;
            for (int z = 0; z < d; ++z)
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < w; ++x)
                        e.host[i++] = (unsigned char)classify(x, y, z);
            upload(e);
            it = m_maps.insert(std::make_pair(key, e)).first;
        }
        return it->second;
    }

private:
    static void upload(Entry &e);
    std::map<detail::Dims, Entry> m_maps;
};

// Handlers of general domains derive from these and implement cellType() only.
class MappedBorderHandler2D : public BorderHandler2D
{
public:
    virtual void setarg(int arg, KernelArgs &ker, int dimx, int dimy);

private:
    CellMapCache m_cache;
};

class MappedBorderHandler3D : public BorderHandler3D
{
public:
    virtual void setarg(int arg, KernelArgs &ker, int dimx, int dimy, int dimz);

private:
    CellMapCache m_cache;
};
} // namespace mgrt

class RectangularBorderHandler : public BorderHandler2D
{
public:
    virtual void setarg(int arg, mgrt::KernelArgs &ker, int dimx, int dimy);
    virtual CellType cellType(int x, int y, int dimx, int dimy) const;
};

class ParallelepipedalBorderHandler : public BorderHandler3D
{
public:
    virtual void setarg(int arg, mgrt::KernelArgs &ker, int dimx, int dimy, int dimz);
    virtual CellType cellType(int x, int y, int z, int dimx, int dimy, int dimz) const;
};

#endif // RECTANGULARBORDERHANDLER_H
