// Why do three different kernels read the 768-channel block-entry tensor of Mixed_6 at the same ~3.95 TB/s (profiles/r04_gather_knockout_1x1.txt)?
// A [M][768] bf16 tensor (row = pixel, 1536 bytes) is pulled into LDS by LDS-DMA exactly as a 1x1 conv with a 128-pixel tile does it: a workgroup
// (8 waves, two per CU... 512 persistent workgroups) owns 128 consecutive rows and walks the row in SEG-byte column slabs, one slab = 128 rows x SEG
// bytes per step, NS steps in flight.  SEG = 128 is the conv's 64-channel k-step; 256 / 512 / 1536 read the same bytes in wider pieces of a row
// (1536 = whole rows: a contiguous 192 KiB stream per tile).  Nothing is computed; only the achieved TB/s is read.
//   hipcc --offload-arch=gfx950 -O3 -o slab_stream_probe slab_stream_probe.hip && ./slab_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void lds_dma16(uint32_t lds_addr, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

// SEG bytes of a row per step; a step moves 128 rows x SEG bytes = SEG / 8 KiB... = (128 * SEG / 1024) wave-level transfers, TR per wave
// NF: filter transfers per wave and step from an L2-resident 295 KB bank (the conv's 192 x 64-channel slab = 3 per wave); BAR: one s_barrier per step
template <int SEG, int NS, int NF = 0, bool BAR = false>
__global__ __launch_bounds__(512, 2) void slab_kernel(const char* src, int rows, int pitch, uint32_t* sink, const char* bank = nullptr, int fpitch = 1536) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TR = 128 * SEG / 1024 / 8;                       // transfers per wave and step (SEG = 128: 2)
    constexpr int LPR = SEG / 16;                                  // lanes per row
    constexpr int RPT = 64 / LPR > 0 ? 64 / LPR : 1;               // rows per transfer (SEG <= 1024)
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int ntiles = rows / 128, nsteps = pitch / SEG;
    unsigned acc = 0;
    int inflight = 0;
    __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(bank ? bank : src), 0, 192 * 2048, 0x00020000);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = (long long)tile * 128 * pitch;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + base, 0, 128 * pitch, 0x00020000);
        for (int s = 0; s < nsteps; ++s) {
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                int voff;
                if constexpr (SEG <= 1024) {
                    const int row = (wid * TR + t) * RPT + lane / LPR;          // 8 waves x TR transfers x RPT rows = 128 rows
                    voff = row * pitch + (lane % LPR) * 16;
                } else {                                                      // whole rows: a transfer is 1 KiB of one row, 1.5 transfers per row
                    const int piece = (wid * TR + t);                         // 1 KiB pieces of the tile's contiguous 192 KiB
                    voff = piece * 1024 + lane * 16;
                }
                lds_dma16(__builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((((s % NS) * 8 + wid) * TR + t) * 1024 % (72 * 1024))), rs, voff,
                          SEG <= 1024 ? s * SEG : 0);
            }
#pragma unroll
            for (int t = 0; t < NF; ++t)
                lds_dma16(__builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(40 * 1024 + (((s % NS) * 8 + wid) * NF + t) * 1024 % (32 * 1024))), rsF,
                          ((wid * NF + t) * 8 + (lane >> 3)) * fpitch + (lane & 7) * 16, (s % 12) * 128);
            if (++inflight >= NS) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * (TR + NF)) : "memory"); }
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename K> double run(K kern, const char* src, int rows, int pitch, uint32_t* sink, int reps, const char* bank = nullptr, int fpitch = 1536) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(512), dim3(512), 72 * 1024, 0, src, rows, pitch, sink, bank, fpitch);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(512), dim3(512), 72 * 1024, 0, src, rows, pitch, sink, bank, fpitch);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)rows * pitch * reps / (ms * 1e-3) / 1e12;
}

int main() {
    const int rows = 321984 / 128 * 128, pitch = 1536;              // 96 frames x 43 x 78 pixels x 768 bf16 channels = 494 MB (> the 256 MB Infinity Cache)
    char* src; uint32_t* sink;
    hipMalloc(&src, (size_t)rows * pitch); hipMemset(src, 1, (size_t)rows * pitch); hipMalloc(&sink, 64);
    const int reps = 300;
    printf("[%d][768] bf16 = %.0f MB through LDS-DMA, 512 workgroups x 8 waves, 128-row tiles; TB/s by bytes of a row per step and steps in flight\n", rows, rows * 1536.0 / 1e6);
    printf("  128 B (the conv's 64-channel k-step), 1 / 2 / 3 in flight: %5.2f %5.2f %5.2f\n", run(slab_kernel<128, 1>, src, rows, pitch, sink, reps), run(slab_kernel<128, 2>, src, rows, pitch, sink, reps), run(slab_kernel<128, 3>, src, rows, pitch, sink, reps));
    printf("  256 B, 1 / 2 in flight:                                   %5.2f %5.2f\n", run(slab_kernel<256, 1>, src, rows, pitch, sink, reps), run(slab_kernel<256, 2>, src, rows, pitch, sink, reps));
    printf("  512 B, 1 / 2 in flight:                                   %5.2f %5.2f\n", run(slab_kernel<512, 1>, src, rows, pitch, sink, reps), run(slab_kernel<512, 2>, src, rows, pitch, sink, reps));
    printf("  whole rows (contiguous 192 KiB per tile), 24 transfers per wave, 1 in flight: %5.2f\n", run(slab_kernel<1536, 1>, src, rows, pitch, sink, reps));
    char* bank; hipMalloc(&bank, 192 * 2048); hipMemset(bank, 2, 192 * 2048);
    printf("  the conv's step (128 B slabs), pixel TB/s only:   + s_barrier per step, 1 / 2 in flight: %5.2f %5.2f\n", run(slab_kernel<128, 1, 0, true>, src, rows, pitch, sink, reps), run(slab_kernel<128, 2, 0, true>, src, rows, pitch, sink, reps));
    printf("                                                    + 3 filter transfers per wave (L2), no barrier, 1 / 2 in flight: %5.2f %5.2f\n", run(slab_kernel<128, 1, 3, false>, src, rows, pitch, sink, reps, bank), run(slab_kernel<128, 2, 3, false>, src, rows, pitch, sink, reps, bank));
    printf("                                                    + both, 1 / 2 in flight: %5.2f %5.2f\n", run(slab_kernel<128, 1, 3, true>, src, rows, pitch, sink, reps, bank), run(slab_kernel<128, 2, 3, true>, src, rows, pitch, sink, reps, bank));
    for (int fp : {1536, 1536 + 128, 1536 + 256, 1536 + 64, 2048}) printf("  both, filter bank row pitch %4d B, 1 / 2 in flight: %5.2f %5.2f\n", fp, run(slab_kernel<128, 1, 3, true>, src, rows, pitch, sink, reps, bank, fp), run(slab_kernel<128, 2, 3, true>, src, rows, pitch, sink, reps, bank, fp));
    return 0;
}
