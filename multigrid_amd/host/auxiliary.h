// auxiliary.h -- norms, difference and bitmap I/O on buffers; interface of auxiliary.h:27-112
// (cl::CommandQueue& -> mgrt::Stream&).
//
// L2Norm, LInfNorm and Average reproduce the reference's reduction tree exactly: groups of four
// folded in `real`, up to three trailing elements dropped per level (auxiliary.cpp:119-135,
// common_kernels.cl:13-53).  ExactL2Norm / ExactLInfNorm are additions that read every element
// and accumulate in double.
#ifndef AUXILIARY_H
#define AUXILIARY_H

#include "buffer.h"

real L2Norm(const Buffer2D &in, mgrt::Stream &q);
real LInfNorm(const Buffer2D &in, mgrt::Stream &q);
real Average(Buffer2D &in, mgrt::Stream &q);
Buffer2D Difference(const Buffer2D &a, const Buffer2D &b, mgrt::Stream &q);

real L2Norm(const Buffer3D &in, mgrt::Stream &q);
real LInfNorm(const Buffer3D &in, mgrt::Stream &q);
real Average(Buffer3D &in, mgrt::Stream &q);
Buffer3D Difference(const Buffer3D &a, const Buffer3D &b, mgrt::Stream &q);

double ExactL2Norm(const mgrt::detail::Field &in, int depth, mgrt::Stream &q);
double ExactLInfNorm(const mgrt::detail::Field &in, int depth, mgrt::Stream &q);

// The work-group shape the reference launches a range of `wsDim` work-items with (auxiliary.cpp:33-117):
// per dimension the largest divisor of the extent within the device's limits, dimension 0 first, the
// product kept within the largest group the device takes.  The kernels under this interface use fixed
// tile shapes and do not ask; the function is here for callers that name it.  The limits are those of
// a gfx950 device: 1024 work-items per dimension and per group.
mgrt::NDRange getBestWorkspaceDim(mgrt::NDRange wsDim);

// 8-bit grey-scale BMP files (auxiliary.cpp:209-303)
Buffer2D fromBitmap(const char *filename);
void toBitmap(const Buffer2D &in, mgrt::Stream &q, const char *filename);

#endif // AUXILIARY_H
