// auxiliary.cpp -- norms, difference, bitmap I/O.
#include "auxiliary.h"
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <vector>

using mgrt::Stream;
using mgrt::detail::Field;

namespace {

mgrt_grid denseGrid(const Field &b, int depth)
{
    mgrt_grid g;
    std::memset(&g, 0, sizeof g);
    g.dim = depth > 1 ? 3 : 2; // a flat pass over the elements: the classes play no part
    g.nx = b.width();
    g.ny = b.height();
    g.nz = depth;
    g.pitch = g.nx;
    g.slab = (long long)g.nx * g.ny;
    g.gny = g.ny;
    for (int i = 0; i < 6; ++i)
        g.face[i] = MGRT_CELL_DIRICHLET;
    return g;
}

real treeL2(const Field &b, Stream &q)
{
    real r = 0;
    MGRT_CHECK(MGRT_K(l2norm_reftree)(b(), (long long)b.count(), &r, q.get()));
    return r;
}
real treeLInf(const Field &b, Stream &q)
{
    real r = 0;
    MGRT_CHECK(MGRT_K(linfnorm_reftree)(b(), (long long)b.count(), &r, q.get()));
    return r;
}
real treeAverage(const Field &b, Stream &q)
{
    real r = 0;
    MGRT_CHECK(MGRT_K(sum_reftree)(b(), (long long)b.count(), &r, q.get()));
    return r / b.count(); // auxiliary.h:52,93
}
void difference(const Field &out, const Field &a, const Field &b, int depth, Stream &q)
{
    const mgrt_grid g = denseGrid(a, depth);
    MGRT_CHECK(MGRT_K(diff)(&g, out(), a(), b(), q.get()));
}
double exact(const Field &in, int depth, Stream &q, bool maxabs)
{
    const mgrt_grid g = denseGrid(in, depth);
    std::shared_ptr<void> d = mgrt::deviceAlloc(sizeof(double));
    double *dp = static_cast<double *>(d.get());
    if (maxabs)
        MGRT_CHECK(MGRT_K(maxabs)(&g, in(), dp, 0, g.ny, q.get()));
    else
        MGRT_CHECK(MGRT_K(sumsq)(&g, in(), dp, 0, g.ny, q.get()));
    double h = 0;
    MGRT_CHECK(mgrt_download(&h, dp, sizeof h, q.get()));
    return maxabs ? h : std::sqrt(h);
}

} // namespace

real L2Norm(const Buffer2D &in, Stream &q) { return treeL2(in, q); }
real LInfNorm(const Buffer2D &in, Stream &q) { return treeLInf(in, q); }
real Average(Buffer2D &in, Stream &q) { return treeAverage(in, q); }
Buffer2D Difference(const Buffer2D &a, const Buffer2D &b, Stream &q)
{
    assert(a.width() == b.width() && a.height() == b.height());
    Buffer2D ans(a.width(), a.height());
    difference(ans, a, b, 1, q);
    return ans;
}

real L2Norm(const Buffer3D &in, Stream &q) { return treeL2(in, q); }
real LInfNorm(const Buffer3D &in, Stream &q) { return treeLInf(in, q); }
real Average(Buffer3D &in, Stream &q) { return treeAverage(in, q); }
Buffer3D Difference(const Buffer3D &a, const Buffer3D &b, Stream &q)
{
    assert(a.width() == b.width() && a.height() == b.height() && a.depth() == b.depth());
    Buffer3D ans(a.width(), a.height(), a.depth());
    difference(ans, a, b, a.depth(), q);
    return ans;
}

double ExactL2Norm(const Field &in, int depth, Stream &q) { return exact(in, depth, q, false); }
double ExactLInfNorm(const Field &in, int depth, Stream &q) { return exact(in, depth, q, true); }

// ---- work-group shape (auxiliary.cpp:33-117) --------------------------------------------------------
namespace {

const size_t kMaxItemsPerDim = 1024, kMaxGroup = 1024; // hipDeviceProp_t of gfx950

// divisors of n that a dimension of a group may have, largest first
std::vector<size_t> divisors(size_t n)
{
    std::vector<size_t> d;
    for (size_t i = n; i >= 1; --i)
        if (n % i == 0 && i <= kMaxItemsPerDim)
            d.push_back(i);
    return d;
}

// largest choice for dimension `k` that leaves the later dimensions a choice within `room`
bool choose(const std::vector<std::vector<size_t> > &div, size_t k, size_t room, size_t *out)
{
    for (size_t i = 0; i < div[k].size(); ++i) {
        const size_t v = div[k][i];
        if (v > room)
            continue;
        if (k + 1 == div.size() || choose(div, k + 1, room / v, out)) {
            out[k] = v;
            return true;
        }
    }
    return false;
}

} // namespace

mgrt::NDRange getBestWorkspaceDim(mgrt::NDRange wsDim)
{
    const size_t nd = wsDim.dimensions();
    if (nd == 0)
        return mgrt::NullRange;
    std::vector<std::vector<size_t> > div(nd);
    for (size_t i = 0; i < nd; ++i) {
        if (wsDim[i] == 0)
            return mgrt::NullRange;
        div[i] = divisors(wsDim[i]);
    }
    size_t dims[3] = {1, 1, 1};
    if (!choose(div, 0, kMaxGroup, dims))
        return mgrt::NullRange;
    return nd == 1 ? mgrt::NDRange(dims[0]) : nd == 2 ? mgrt::NDRange(dims[0], dims[1]) : mgrt::NDRange(dims[0], dims[1], dims[2]);
}

// ---- 8-bit BMP (auxiliary.cpp:181-303) -----------------------------------------------------------
// Layout written: 14-byte file header, 40-byte info header, 256-entry grey palette (4 bytes each),
// rows padded to a multiple of 4 pixels, pixel = 255*|v|/max|v| truncated to a byte.
namespace {

void put16(std::vector<unsigned char> &v, unsigned x)
{
    v.push_back(x & 255);
    v.push_back((x >> 8) & 255);
}
void put32(std::vector<unsigned char> &v, uint32_t x)
{
    put16(v, x & 65535);
    put16(v, x >> 16);
}
uint32_t get32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
unsigned get16(const unsigned char *p) { return p[0] | (p[1] << 8); }
const unsigned kSignature = 19778; // "BM"
const uint32_t kHeaders = 14 + 40, kPalette = 256 * 4;

} // namespace

Buffer2D fromBitmap(const char *filename)
{
    std::ifstream in(filename, std::ios::binary);
    if (!in)
        throw std::runtime_error(std::string(filename) + " Does not exists");
    unsigned char h[54];
    in.read(reinterpret_cast<char *>(h), sizeof h);
    const uint32_t compression = get32(h + 30);
    const int32_t w = (int32_t)get32(h + 18), hgt = (int32_t)get32(h + 22);
    if (compression != 0)
        throw std::runtime_error("UNSUPPORTED: Bmp compression");
    if (hgt < 0)
        throw std::runtime_error("UNSUPPORTED: Bmp negative height");
    if (in.fail() || get16(h) != kSignature)
        throw std::runtime_error(std::string());
    if (get16(h + 28) != 8)
        throw std::runtime_error("Bitmap not monochrome");
    in.seekg(get32(h + 10), std::ios::beg);
    // pixels are read as one dense w*h block of signed bytes, as the reference does
    // (auxiliary.cpp:240-245: no row padding on input, char -> real / 255)
    std::vector<char> px((size_t)w * hgt);
    in.read(px.data(), (std::streamsize)px.size());
    BidimArray<real> ans(w, hgt);
    for (int i = 0; i < hgt; ++i)
        for (int j = 0; j < w; ++j)
            ans(j, i) = static_cast<real>(px[j + (size_t)i * w]) / 255.0;
    return Buffer2D(w, hgt, ans.data());
}

void toBitmap(const Buffer2D &in, Stream &q, const char *filename)
{
    std::ofstream out(filename, std::ios::binary);
    if (!out)
        throw std::runtime_error(std::string(filename) + " does not exists");
    const BidimArray<real> a = in.read(q);
    const int w = in.width(), h = in.height(), padded = w + (4 - w % 4) % 4;
    std::vector<unsigned char> f;
    f.reserve(kHeaders + kPalette + (size_t)padded * h);
    put16(f, kSignature);
    put32(f, kHeaders + kPalette + (uint32_t)padded * h);
    put16(f, 0);
    put16(f, 0);
    put32(f, kHeaders + kPalette);
    put32(f, 40);
    put32(f, (uint32_t)w);
    put32(f, (uint32_t)h);
    put16(f, 1);
    put16(f, 8);
    put32(f, 0);
    put32(f, (uint32_t)padded * h);
    put32(f, 1);
    put32(f, 1);
    put32(f, 256);
    put32(f, 256);
    for (int i = 0; i < 256; ++i)
        for (int c = 0; c < 4; ++c)
            f.push_back((unsigned char)i);
    const real top = LInfNorm(in, q);
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < padded; ++i)
            // the unqualified fabs of auxiliary.cpp:298 is the double one: scale in double, truncate
            f.push_back(i < w ? (unsigned char)(int)(255 * std::fabs((double)a(i, j)) / (double)top) : 0);
    out.write(reinterpret_cast<const char *>(f.data()), (std::streamsize)f.size());
}
