// Shared device/host helpers of the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvmc.h"

#define MVMC_WAVE 64

#define MVMC_CHECK_LAUNCH()                                  \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return MVMC_ERR_LAUNCH; \
    } while (0)

// OpenPose-25 row of each COCO-17 joint (pose_def.py:72-96 vs :111-137).
__device__ __constant__ const int kOp25ToCoco17[17] = {0, 16, 15, 18, 17, 5, 2, 6, 3, 7, 4, 12, 9, 13, 10, 14, 11};

// Non-contracted double ops: used wherever the reference's NumPy expression rounds
// after every multiply/add and the result feeds a float32 store (bit-exact D / S).
__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ float fmulr(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float faddr(float a, float b) { return __fadd_rn(a, b); }

// Wave-wide (64 lanes) sum with a fixed butterfly order (deterministic).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
