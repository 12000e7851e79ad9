// Where does the energy of an MFMA kernel go on a power-capped MI355X?  Sustained bf16 MFMA rate (v_mfma_f32_16x16x32_bf16, random operands) of a
// loop that has NO barriers and no global stores, as data movement is added at the ratios of the shipped 128 x 192 gather tile
// (per wave and 24 MFMAs: 16 ds_read_b128 fragment reads, 5 LDS-DMA transfers of 1 KiB from an L2-resident buffer):
//   shape test: 16x16x32 vs 32x32x16 on register operands only;  then 16x16x32 with R fragment reads per 12 MFMAs (R = 0, 2, 4, 8, 16) from LDS,
//   then R = 8 plus D LDS-DMA transfers per 24 MFMAs (D = 0, 2, 5, 10).
// 256 CUs x 2 workgroups x 8 waves (four waves per SIMD, 80 KiB LDS per workgroup: the shipped occupancy).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power_probe mfma_power_probe.hip && ./mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(uint32_t lds_addr, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

template <int SHAPE>   // register operands only.  0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(512, 2) void mfma_regs(const u32x4* __restrict__ ops, float* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ops[(wave * 8 + i) * 64 + lane]; b[i] = ops[(wave * 8 + 4 + i) * 64 + lane]; }
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 c[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(i + r) & 3]), __builtin_bit_cast(bf16x8, b[i & 3]), c[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) s += c[i][0] + c[i][3];
    } else {
        f32x16 c[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + r) & 3]), __builtin_bit_cast(bf16x8, b[i & 3]), c[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) s += c[i][0] + c[i][15];
    }
    if (s == 1234.5f) sink[0] = s;
}

// 24 MFMAs (16x16x32) per iteration and wave, R fragment reads per 12 MFMAs from this workgroup's LDS, D LDS-DMA transfers per iteration
template <int R, int D>
__global__ __launch_bounds__(512, 2) void mfma_lds(const u32x4* __restrict__ ops, const char* __restrict__ src, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    for (int i = threadIdx.x; i < 80 * 1024 / 16; i += 512) lds[i] = ops[i & 2047];       // random operand bits everywhere
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 1 << 20, 0x00020000);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    u32x4 f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = lds[(wave * 16 + i) * 64 + lane];
    f32x4 c[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int off = (blockIdx.x * 8 + wave) * 4096;
    for (int it = 0; it < iters; ++it) {
        const int slot = it & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < R; ++i) f[(h * 8 + i) & 15] = lds[((slot * 16 + (wave + i) & 15) * 16 + h * 8 + i) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 12; ++i)
                c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f[(h * 8 + (i & 1)) & 15]), __builtin_bit_cast(bf16x8, f[(h * 8 + 2 + i % 6) & 15]), c[i], 0, 0, 0);
        }
#pragma unroll
        for (int dI = 0; dI < D; ++dI)
            lds_dma16(__builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((((it + 2) & 3) * 16 * 16 + (wave * D + dI) % 256) * 1024 % (80 * 1024))), rs, (off + dI * 1024 + lane * 16) & ((1 << 20) - 1), 0);
        off += D * 1024 * 7;
        if (D > 0 && (it & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += c[i][0] + c[i][3];
    if (s == 1234.5f) sink[0] = s;
}
static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <typename F> double timed(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 2;
}

int main() {
    const size_t n = 2048 * 8;                               // bf16 elements of the operand pool (2048 x 16 B)
    std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> h(n);
    for (auto& v : h) v = bf16(nd(rng));
    u32x4* d; float* sink; char* src;
    hipMalloc(&d, n * 2); hipMalloc(&sink, 64); hipMalloc(&src, 1 << 20);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (1 << 20); i += n * 2) hipMemcpy(src + i, h.data(), n * 2, hipMemcpyHostToDevice);
    const int iters = 1500000;                               // >= 1 s per launch: the power governor needs it
    const double fl = 24.0 * 16 * 16 * 32 * 2 * iters * 512.0 * 8;        // 24 MFMAs per iteration and wave, 512 workgroups x 8 waves
    printf("register operands only, N(0,1):   16x16x32 %7.1f TFLOP/s    32x32x16 %7.1f TFLOP/s\n",
           fl / (timed([&] { hipLaunchKernelGGL(mfma_regs<0>, dim3(512), dim3(512), 0, 0, d, sink, iters); }) * 1e-3) / 1e12,
           fl / (timed([&] { hipLaunchKernelGGL(mfma_regs<1>, dim3(512), dim3(512), 0, 0, d, sink, iters); }) * 1e-3) / 1e12);
    fflush(stdout);
    auto run = [&](auto kern, const char* what) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        const double ms = timed([&] { hipLaunchKernelGGL(kern, dim3(512), dim3(512), 80 * 1024, 0, d, src, sink, iters); });
        printf("%-72s %8.1f ms %7.1f TFLOP/s\n", what, ms, fl / (ms * 1e-3) / 1e12); fflush(stdout);
    };
    run(mfma_lds<0, 0>, "16x16x32, no fragment reads, no DMA");
    run(mfma_lds<2, 0>, "  + 2 ds_read_b128 per 12 MFMAs");
    run(mfma_lds<4, 0>, "  + 4 ds_read_b128 per 12 MFMAs");
    run(mfma_lds<8, 0>, "  + 8 ds_read_b128 per 12 MFMAs (the shipped 32 x 96 wave tile)");
    run(mfma_lds<16, 0>, "  + 16 ds_read_b128 per 12 MFMAs");
    run(mfma_lds<8, 2>, "  8 reads + 2 LDS-DMA transfers (1 KiB, L2-resident source) per 24 MFMAs");
    run(mfma_lds<8, 5>, "  8 reads + 5 transfers per 24 MFMAs (the shipped 128 x 192 tile)");
    run(mfma_lds<8, 10>, "  8 reads + 10 transfers per 24 MFMAs");
    run(mfma_lds<4, 2>, "  4 reads + 2 transfers per 24 MFMAs");
    return 0;
}
