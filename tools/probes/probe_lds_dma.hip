// Probe: buffer_load_dwordx4 ... lds  -- placement (M0 base + lane*16) and what out-of-range lanes write.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* src, int nbytes, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = 0xFFFFFFFFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, nbytes, 0x00020000);
    int lane = threadIdx.x;
    // lanes 0..47 in range with a permuted source chunk (lane ^ 3), lanes 48..63 out of range
    unsigned voff = lane < 48 ? (unsigned)((lane ^ 3) * 16) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + 64), 16, (int)voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = sm[i];
}
int main() {
    uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
    uint32_t *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 4096, o);
    uint32_t r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    printf("dwords before the DMA window (should stay 0xFFFFFFFF): %08x %08x\n", r[62], r[63]);
    for (int l = 0; l < 64; l += 1) if (l < 6 || (l >= 46 && l < 52) || l == 63)
        printf("slot %2d: %08x %08x %08x %08x\n", l, r[64 + 4*l], r[65 + 4*l], r[66 + 4*l], r[67 + 4*l]);
    printf("after window: %08x\n", r[64 + 256]);
    return 0;
}
