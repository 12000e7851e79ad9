// Probe: what does ds_read_b64_tr_b16 return per lane?  LDS holds u16 value == its element index.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t* out, int mode) {
    __shared__ uint16_t sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (uint16_t)i;
    __syncthreads();
    int lane = threadIdx.x;
    uint32_t addr;
    if (mode == 0) addr = lane * 8;                                   // lane-linear
    else { int li = lane & 15, lg = lane >> 4; addr = ((4 * lg + (li >> 2)) * 144 + (li & 3) * 4) * 2; }  // my wgrad pattern, row stride 144 elems
    addr += (uint32_t)(uintptr_t)sm;
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[lane * 2] = r.x; out[lane * 2 + 1] = r.y;
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 8);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        uint32_t h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u\n", l, h[2*l] & 0xffff, h[2*l] >> 16, h[2*l+1] & 0xffff, h[2*l+1] >> 16);
    }
    return 0;
}
