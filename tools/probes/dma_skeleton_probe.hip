// What does one step of an LDS-DMA ring cost when nothing else runs?  8 waves x 256 workgroups (one per CU), per step and wave: NTR wave-level
// LDS-DMA transfers (1 KiB each), s_waitcnt vmcnt(2 NTR), s_barrier.  Variants: transfers out of range (no memory access) / from a small buffer (L2
// hits) / streaming from a large buffer (HBM); without the barrier; without the transfers; with plain register loads instead of LDS-DMA.
//   hipcc --offload-arch=gfx950 -O3 -o dma_skeleton_probe dma_skeleton_probe.hip && ./dma_skeleton_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void lds_dma16(uint32_t lds_addr, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

template <int NTR, int MODE, bool BARRIER, bool DMA>
__global__ __launch_bounds__(512, 1) void probe(const char* src, long long bytes, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds = (uint32_t)(uintptr_t)smem;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)(bytes > 0x7fffffffll ? 0x7fffffffll : bytes), 0x00020000);
    unsigned acc = 0;
    const long long span = MODE == 2 ? bytes : (1ll << 20);                     // MODE 1: 1 MiB window (L2), MODE 2: the whole buffer (HBM)
    long long base = ((long long)blockIdx.x * 8 + wid) * 65536 % span;
    for (int s = 0; s < steps; ++s) {
        if (DMA) {
#pragma unroll
            for (int i = 0; i < NTR; ++i) {
                int voff = MODE == 0 ? (int)0x80000000u : (int)((base + i * 1024 + lane * 16) % span);
                lds_dma16(lds + (uint32_t)(((s & 3) * 8 * NTR + wid * NTR + i) * 1024), rs, voff, 0);
            }
            base += 8 * 65536 * 256 % span + NTR * 1024;
            if (base >= span) base -= span;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NTR) : "memory");
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename K> float run(K kern, const char* src, long long bytes, int steps, unsigned* sink, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, src, bytes, steps, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, src, bytes, steps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / steps;          // ns per step
}

int main() {
    const long long bytes = 1ll << 31;
    char* src; unsigned* sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 64);
    const int steps = 2000;
    constexpr int NTR = 5;
    const size_t lds = 4 * 8 * NTR * 1024;
    printf("per step (ns; x2.4 = cycles), 8 waves x %d transfers of 1 KiB = %d KiB per CU and step\n", NTR, 8 * NTR);
    printf("barrier only                      %8.1f\n", run(probe<NTR, 0, true, false>, src, bytes, steps, sink, lds));
    printf("DMA out of range, no barrier      %8.1f\n", run(probe<NTR, 0, false, true>, src, bytes, steps, sink, lds));
    printf("DMA out of range + barrier        %8.1f\n", run(probe<NTR, 0, true, true>, src, bytes, steps, sink, lds));
    printf("DMA 1 MiB window (L2) + barrier   %8.1f\n", run(probe<NTR, 1, true, true>, src, bytes, steps, sink, lds));
    printf("DMA streaming (HBM) + barrier     %8.1f   -> %.2f TB/s\n", run(probe<NTR, 2, true, true>, src, bytes, steps, sink, lds),
           256.0 * 8 * NTR * 1024 / run(probe<NTR, 2, true, true>, src, bytes, steps, sink, lds) / 1e3);
    printf("DMA streaming (HBM), no barrier   %8.1f\n", run(probe<NTR, 2, false, true>, src, bytes, steps, sink, lds));
    printf("2 transfers: out of range+barrier %8.1f   L2 %8.1f   HBM %8.1f\n", run(probe<2, 0, true, true>, src, bytes, steps, sink, lds),
           run(probe<2, 1, true, true>, src, bytes, steps, sink, lds), run(probe<2, 2, true, true>, src, bytes, steps, sink, lds));
    return 0;
}
