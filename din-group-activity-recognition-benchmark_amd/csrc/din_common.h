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

// ---- tuning / test options: set through the C ABI (din_set_option), never read from the process environment -------------------
// The library promised "stateless and re-entrant" while ~45 getenv() calls per launch let a stray environment variable pick kernels
// (VERDICT r4 / ADVICE r3).  Every switch is now a named option: unset by default (the shipped choice), set by a test or a tuning tool with
// din_set_option(name, value).  DIN_OPT("NAME") costs one atomic load per use: the call site keeps a pointer to the option's slot.
#include <atomic>
struct din_option_slot { std::atomic<const char*> value; };
din_option_slot* din_option_register(const char* name);                    // din_error.cpp (find or create; slots live for the process)
#define DIN_OPT(name) ([]() -> const char* { static din_option_slot* s_ = din_option_register(name); \
                                             return s_->value.load(std::memory_order_acquire); }())

// Dynamic-LDS limit of a kernel above 64 KiB.  hipFuncSetAttribute is a slow host call and its effect is PER DEVICE: raise once per
// (calling thread, current device, kernel), not per launch -- and not once per process, which would leave a second device of the same
// process at the 64 KiB default.
#include <unordered_map>
static inline void din_raise_lds(const void* fn, size_t lds) {
    static thread_local std::unordered_map<uint64_t, size_t> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& g = granted[(uint64_t)(uintptr_t)fn * 0x9E3779B97F4A7C15ull + (uint64_t)dev];
    if (g < lds) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        g = lds;
    }
}

// ---- bf16 <-> f32 (round-to-nearest-even; NaN preserved) -----------------------------------------
typedef uint16_t bf16_t;
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even with NaN quieting: the gfx950 hardware conversion (v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_hw{lo, hi}, bf16x2_hw));
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

// XCD-aware workgroup order.  The dispatcher deals workgroup w (linear id, x fastest) to XCD w % 8, each XCD with a private 4 MiB L2
// (MI355X_MICROARCH.md, Workgroup dispatch); neighbouring tiles of a conv re-read the same pixels / filter slabs, so every XCD is
// given a CONTIGUOUS range of the logical tile order instead of every 8th tile (bijective for any total).  Speed only: a different
// placement is slower, never wrong.  DIN_XCD_REMAP=0 at build time restores the plain order.
#ifndef DIN_XCD_REMAP
#define DIN_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcd_remap(int w, int total) {
#if DIN_XCD_REMAP
    constexpr int NX = 8;
    const int q = total / NX, r = total % NX;
    const int x = w % NX, j = w / NX;
    return x < r ? x * (q + 1) + j : r * (q + 1) + (x - r) * q + j;
#else
    return w;
#endif
}
// logical (x, y) block coordinates of a 2-D grid after the remap (x fastest)
__device__ __forceinline__ void xcd_block(int& bx, int& by) {
    const int l = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    by = l / (int)gridDim.x;
    bx = l - by * (int)gridDim.x;
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

// counter-based dropout: the backward regenerates the forward's keep-mask from (seed, element index)
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
    // splitmix64 finaliser -> 32 random bits
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = x ^ (x >> 31);
    return (uint32_t)(x >> 32);
}
// mask seed of a launch: the by-value seed plus the optional device-side per-step offset (graph-captured steps), mod 2^63
__device__ __forceinline__ uint64_t fold_seed(uint64_t seed, const uint64_t* off) {
    return off ? ((seed + *off) & 0x7FFFFFFFFFFFFFFFull) : seed;
}
__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t idx, float p) {
    if (p <= 0.f) return 1.f;
    float u = (float)(mix32(seed ^ ((uint64_t)idx * 0xD1342543DE82EF95ull)) >> 8) * (1.0f / 16777216.0f);
    return u >= p ? 1.f / (1.f - p) : 0.f;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int grid_1d(int64_t n, int block, int cap = 256 * 8) {
    int64_t g = ceil_div64(n, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
