// Shared device/host helpers for libdin_hip.so (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/din_hip.h"

// ---- error plumbing: integer status + thread-local message, never throws -------------------------
void din_set_error(const char* fmt, ...);
#define DIN_FAIL(code, ...) do { din_set_error(__VA_ARGS__); return (code); } while (0)
#define DIN_REQUIRE(cond, ...) do { if (!(cond)) DIN_FAIL(DIN_E_ARG, __VA_ARGS__); } while (0)
#define DIN_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- bf16 <-> f32 (round-to-nearest-even; NaN preserved) -----------------------------------------
typedef uint16_t bf16_t;
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int EPC = 4;                  // elements per 16-byte chunk
    __device__ static float ld(const float* p) { return *p; }
    __device__ static void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float load_as_f32(const void* base, int dtype, int64_t i) {
    return dtype == DIN_F32 ? ((const float*)base)[i] : bf16_to_f32(((const bf16_t*)base)[i]);
}
__device__ __forceinline__ void store_from_f32(void* base, int dtype, int64_t i, float v) {
    if (dtype == DIN_F32) ((float*)base)[i] = v; else ((bf16_t*)base)[i] = f32_to_bf16(v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int grid_1d(int64_t n, int block, int cap = 256 * 8) {
    int64_t g = ceil_div64(n, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
