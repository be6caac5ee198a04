// mgrt_stamps.hip -- diagnostic build of the fused 3D sweep (make stamps, tools/stamps.py): the product's
// kernel source with its stamp hooks defined.  Every wavefront of the middle block sums, over the fast
// time steps of its march, the shader-clock time it spends in each segment of a step; the sums land in
// a device array that mgrt_debug_stamps() copies out.  Not part of libmgrt.so.
#include <hip/hip_runtime.h>

__device__ unsigned long long mgrt_stamps[16][8];

#define MGRT_STAMP_DECL() unsigned long long acc[8] = {}, tprev = 0
#define MGRT_STAMP_BEGIN() (tprev = __builtin_amdgcn_s_memtime())
#define MGRT_STAMP(k)                                                                              \
    do {                                                                                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
        acc[k] += t_ - tprev;                                                                      \
        tprev = t_;                                                                                \
    } while (0)
#define MGRT_STAMP_STEP() (acc[7] += 1)
#define MGRT_STAMP_END(wave, lane)                                                                 \
    do {                                                                                           \
        if (blockIdx.x == gridDim.x / 2 && (lane) == 0)                                            \
            for (int k_ = 0; k_ < 8; ++k_)                                                         \
                mgrt_stamps[(wave)&15][k_] = acc[k_];                                              \
    } while (0)

#include "../../multigrid_amd/csrc/mgrt_smooth_fused.hip"

extern "C" mgrt_status mgrt_debug_stamps(unsigned long long *out)
{
    MGRT_HIP(hipDeviceSynchronize());
    MGRT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(mgrt_stamps), sizeof(unsigned long long) * 16 * 8));
    return MGRT_OK;
}
