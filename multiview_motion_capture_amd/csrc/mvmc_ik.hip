// Temporary placeholder so the library links while the IK kernel is being written.
#include "mvmc_common.h"
extern "C" int mvmc_ik_solve(const mvmcSkeleton*, const double*, const double*, const int32_t*, int, int, int, int,
                             const double*, const uint8_t*, int, int, double*, double*, double*, mvmcStream_t) {
    return MVMC_ERR_UNSUPPORTED;
}
