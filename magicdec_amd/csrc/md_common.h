// Shared helpers for the gfx950 kernels of libmagicdec_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/magicdec_hip.h"
#include "../../include/magicdec_hip_dev.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MD_WAVE 64

// thread-local error message (host side)
void md_set_error(const char* fmt, ...);

#define MD_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            md_set_error(__VA_ARGS__);          \
            return MD_ERR_INVALID_ARG;          \
        }                                       \
    } while (0)

#define MD_CHECK_LAUNCH(what)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            md_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));  \
            return MD_ERR_LAUNCH;                                                 \
        }                                                                         \
    } while (0)

// hipFuncSetAttribute is per (function, device): "done" flags are kept per device ordinal so that a process driving
// several GPUs sets the attribute on each of them (one process per GPU is the deployment, but the C ABI allows more)
#define MD_MAX_DEVICES 64
struct MdPerDeviceOnce {
    bool done[MD_MAX_DEVICES] = {};
    // true exactly once per device for the calling host thread's current device
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MD_MAX_DEVICES) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
    void undo() {
        int d = 0;
        if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < MD_MAX_DEVICES) done[d] = false;
    }
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
// round-to-nearest-even float -> bf16 (hardware v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float x) { return (bf16_t)x; }
__device__ __forceinline__ float bf16_to_f32(bf16_t x) { return (float)x; }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
