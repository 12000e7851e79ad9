// HBM-bound helpers of the DIN stage-2 path for gfx950: image prep, casts, layout changes, Adam (pools / resize: pool.hip).
// All tensors are NHWC with (pixel stride, channel offset); one thread handles a 4-channel group of one
// pixel so that every access is an 8-/16-byte coalesced vector along the channel axis.
#include "din_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- 4-channel vector load/store in either storage type --------------------------------------------
__device__ __forceinline__ f32x4 ld4(const void* base, int dtype, int64_t i) {
    if (dtype == DIN_F32) return *reinterpret_cast<const f32x4*>((const float*)base + i);
    uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)base + i);
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                 __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
__device__ __forceinline__ void st4(void* base, int dtype, int64_t i, f32x4 v) {
    if (dtype == DIN_F32) { *reinterpret_cast<f32x4*>((float*)base + i) = v; return; }
    uint2 r = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *reinterpret_cast<uint2*>((bf16_t*)base + i) = r;
}

// 8-channel (16-byte) bf16 vectors for the hot pool kernels
struct f32x8 { f32x4 lo, hi; };
__device__ __forceinline__ f32x8 ld8_bf16(const void* base, int64_t i) {
    uint4 r = *reinterpret_cast<const uint4*>((const bf16_t*)base + i);
    f32x8 o;
    o.lo = f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    o.hi = f32x4{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
    return o;
}
__device__ __forceinline__ void st8_bf16(void* base, int64_t i, const f32x8& v) {
    uint4 r = {pack_bf16x2(v.lo[0], v.lo[1]), pack_bf16x2(v.lo[2], v.lo[3]), pack_bf16x2(v.hi[0], v.hi[1]), pack_bf16x2(v.hi[2], v.hi[3])};
    *reinterpret_cast<uint4*>((bf16_t*)base + i) = r;
}

// ---- Row P: prep_images (utils.py:8-19) --------------------------------------------------------------
__device__ __forceinline__ float prep1(float x) {
    // three separately rounded fp32 operations, as in the reference (div, sub, mul)
    float y = __fdiv_rn(x, 255.0f);
    y = __fsub_rn(y, 0.5f);
    return __fmul_rn(y, 2.0f);
}
__global__ void prep_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = prep1(in[i]);
}
// NCHW (u8|f32) -> NHWC padded to cpad channels.  One thread per pixel: 3 strided (but wave-coalesced) plane reads,
// one 16-byte pixel write.
template <typename TIN>
__global__ void prep_nhwc_kernel(const TIN* __restrict__ in, void* __restrict__ out, int out_dtype, int nb, int h, int w, int cpad) {
    int64_t hw = (int64_t)h * w, total = (int64_t)nb * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i / hw, pix = i - n * hw;
        const TIN* src = in + n * 3 * hw + pix;
        float r = prep1((float)src[0]), g = prep1((float)src[hw]), b = prep1((float)src[2 * hw]);
        st4(out, out_dtype, i * cpad, f32x4{r, g, b, 0.f});
        for (int c = 4; c < cpad; c += 4) st4(out, out_dtype, i * cpad + c, f32x4{0.f, 0.f, 0.f, 0.f});
    }
}

// ---- casts / layout --------------------------------------------------------------------------------------
__global__ void cast_kernel(const void* __restrict__ src, int sdt, void* __restrict__ dst, int ddt, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        store_from_f32(dst, ddt, i, load_as_f32(src, sdt, i));
}
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, float a, float b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}
// out (+)= x * scalar[idx]   (scalar lives on the device: the learnable beta of dynamic_infer_module.py:42-44,145)
__global__ void scale_by_param_kernel(const float* __restrict__ x, const float* __restrict__ scalar, int idx, float* __restrict__ out,
                                      int accumulate, int64_t n) {
    const float sc = scalar[idx];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (accumulate ? out[i] : 0.f) + x[i] * sc;
}
// out[idx] += <x, y>
__global__ void dot_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int idx, int64_t n) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i] * y[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + idx, s);
}
__global__ void grad_cast_mask_kernel(const float* __restrict__ g, const void* __restrict__ y, void* __restrict__ out, int dtype,
                                      int64_t pixels, int c, int ldy, int yoff, int ldo, int ooff, int use_mask) {
    const int c4 = c >> 2;
    int64_t total = pixels * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c4);
        int64_t p = i / c4;
        f32x4 v = *reinterpret_cast<const f32x4*>(g + p * c + cg * 4);
        if (use_mask) {
            f32x4 yy = ld4(y, dtype, p * ldy + yoff + cg * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = yy[e] > 0.f ? v[e] : 0.f;
        }
        st4(out, dtype, p * ldo + ooff + cg * 4, v);
    }
}
// tiled transposes between NHWC (storage dtype) and NCHW fp32 (API-parity views only; not on the training path)
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ in, int dtype, int64_t hw, int c, int ld, int coff, float* __restrict__ out) {
    __shared__ float tile[32][33];
    int n = blockIdx.z;
    int64_t p0 = (int64_t)blockIdx.x * 32;
    int c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int64_t p = p0 + r; int cc = c0 + threadIdx.x;
        tile[r][threadIdx.x] = (p < hw && cc < c) ? load_as_f32(in, dtype, (n * hw + p) * ld + coff + cc) : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int cc = c0 + r; int64_t p = p0 + threadIdx.x;
        if (cc < c && p < hw) out[((int64_t)n * c + cc) * hw + p] = tile[threadIdx.x][r];
    }
}
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, int64_t hw, int c, void* __restrict__ out, int dtype, int ld, int coff) {
    __shared__ float tile[32][33];
    int n = blockIdx.z;
    int64_t p0 = (int64_t)blockIdx.x * 32;
    int c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int cc = c0 + r; int64_t p = p0 + threadIdx.x;
        tile[r][threadIdx.x] = (cc < c && p < hw) ? in[((int64_t)n * c + cc) * hw + p] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        int64_t p = p0 + r; int cc = c0 + threadIdx.x;
        if (p < hw && cc < c) store_from_f32(out, dtype, (n * hw + p) * ld + coff + cc, tile[threadIdx.x][r]);
    }
}
// out[b,t,i,:] = i < n_per_clip[b] ? x[b,t,i,:] : 0   (the padding actors of a Collective clip; also the backward of itself)
__global__ void mask_actors_kernel(const float* __restrict__ x, const int32_t* __restrict__ n_per_clip, float* __restrict__ out, int t, int n,
                                   int c, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c;                       // (b, t, actor)
        const int actor = (int)(row % n);
        const int b = (int)(row / ((int64_t)t * n));
        out[i] = actor < n_per_clip[b] ? x[i] : 0.f;
    }
}
__global__ void frame_index_kernel(int32_t* out, int bt, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bt * n) out[i] = i / n;
}

// fused Adam (torch.optim.Adam semantics: L2 weight decay folded into the gradient, bias-corrected moments)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// multi-tensor form: one launch for the whole parameter list.  ptrs[t] = {p, g, m, v} of tensor t; workgroup b updates elements
// [chunk_index[b] * chunk, +chunk) of tensor chunk_tensor[b]
__global__ __launch_bounds__(256) void adam_multi_kernel(const uint64_t* __restrict__ ptrs, const int64_t* __restrict__ sizes,
                                                         const int32_t* __restrict__ chunk_tensor, const int32_t* __restrict__ chunk_index,
                                                         int chunk, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                         float bc2_sqrt, float gscale) {
    const int t = chunk_tensor[blockIdx.x];
    float* __restrict__ p = reinterpret_cast<float*>(ptrs[4 * t + 0]);
    const float* __restrict__ g = reinterpret_cast<const float*>(ptrs[4 * t + 1]);
    float* __restrict__ m = reinterpret_cast<float*>(ptrs[4 * t + 2]);
    float* __restrict__ v = reinterpret_cast<float*>(ptrs[4 * t + 3]);
    const int64_t n = sizes[t];
    const int64_t i0 = (int64_t)chunk_index[blockIdx.x] * chunk;
    int64_t i1 = i0 + chunk;
    if (i1 > n) i1 = n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        float gi = g[i] * gscale;
        float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

__global__ void counter_add_kernel(uint64_t* counter, uint64_t delta) { *counter = (*counter + delta) & 0x7FFFFFFFFFFFFFFFull; }

}  // namespace

extern "C" {

int din_prep_images_f32(const float* in, float* out, int64_t n, void* stream) {
    DIN_REQUIRE(in && out && n >= 0, "prep_images: bad argument");
    if (n == 0) return DIN_OK;
    hipLaunchKernelGGL(prep_f32_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, as_stream(stream), in, out, n);
    DIN_CHECK_LAUNCH("prep_images_f32");
    return DIN_OK;
}
int din_prep_images_nhwc(const void* in, int in_is_u8, void* out, int out_dtype, int nb, int h, int w, int cpad, void* stream) {
    DIN_REQUIRE(in && out && nb > 0 && h > 0 && w > 0, "prep_images_nhwc: bad argument");
    DIN_REQUIRE(cpad >= 4 && cpad % 4 == 0, "prep_images_nhwc: cpad must be a multiple of 4");
    int64_t total = (int64_t)nb * h * w;
    if (in_is_u8) hipLaunchKernelGGL(prep_nhwc_kernel<uint8_t>, dim3(grid_1d(total, 256, 8192)), dim3(256), 0, as_stream(stream), (const uint8_t*)in, out, out_dtype, nb, h, w, cpad);
    else hipLaunchKernelGGL(prep_nhwc_kernel<float>, dim3(grid_1d(total, 256, 8192)), dim3(256), 0, as_stream(stream), (const float*)in, out, out_dtype, nb, h, w, cpad);
    DIN_CHECK_LAUNCH("prep_images_nhwc");
    return DIN_OK;
}

int din_grad_cast_mask(const float* g, const void* y, void* out, int dtype, int64_t pixels, int c, int ldy, int yoff, int ldo,
                       int ooff, int use_mask, void* stream) {
    DIN_REQUIRE(g && out && (!use_mask || y), "grad_cast_mask: null pointer");
    DIN_REQUIRE(c % 4 == 0 && ldy % 4 == 0 && yoff % 4 == 0 && ldo % 4 == 0 && ooff % 4 == 0, "grad_cast_mask: alignment");
    int64_t total = pixels * (c / 4);
    hipLaunchKernelGGL(grad_cast_mask_kernel, dim3(grid_1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), g, y, out, dtype, pixels, c, ldy, yoff, ldo, ooff, use_mask);
    DIN_CHECK_LAUNCH("grad_cast_mask");
    return DIN_OK;
}
int din_boxes_frame_index(int32_t* out, int bt, int n, void* stream) {
    DIN_REQUIRE(out && bt > 0 && n > 0, "boxes_frame_index: bad argument");
    hipLaunchKernelGGL(frame_index_kernel, dim3((bt * n + 255) / 256), dim3(256), 0, as_stream(stream), out, bt, n);
    DIN_CHECK_LAUNCH("boxes_frame_index");
    return DIN_OK;
}
int din_counter_add(uint64_t* counter, uint64_t delta, void* stream) {
    DIN_REQUIRE(counter, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, as_stream(stream), counter, delta);
    DIN_CHECK_LAUNCH("counter_add");
    return DIN_OK;
}
int din_axpby(const float* x, const float* y, float* out, float alpha, float beta, int64_t n, void* stream) {
    DIN_REQUIRE(x && out && n >= 0, "axpby: bad argument");
    if (n == 0) return DIN_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, as_stream(stream), x, y, out, alpha, beta, n);
    DIN_CHECK_LAUNCH("axpby");
    return DIN_OK;
}
int din_mask_actors(const float* x, const int32_t* n_per_clip, int b, int t, int n, int c, float* out, void* stream) {
    DIN_REQUIRE(x && n_per_clip && out && b > 0 && t > 0 && n > 0 && c > 0, "mask_actors: bad argument");
    const int64_t total = (int64_t)b * t * n * c;
    hipLaunchKernelGGL(mask_actors_kernel, dim3(grid_1d(total, 256)), dim3(256), 0, as_stream(stream), x, n_per_clip, out, t, n, c, total);
    DIN_CHECK_LAUNCH("mask_actors");
    return DIN_OK;
}
int din_scale_by_param(const float* x, const float* scalar, int idx, float* out, int accumulate, int64_t n, void* stream) {
    DIN_REQUIRE(x && scalar && out && n >= 0 && idx >= 0, "scale_by_param: bad argument");
    if (n == 0) return DIN_OK;
    hipLaunchKernelGGL(scale_by_param_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, as_stream(stream), x, scalar, idx, out, accumulate, n);
    DIN_CHECK_LAUNCH("scale_by_param");
    return DIN_OK;
}
int din_dot_accum(const float* x, const float* y, float* out, int idx, int64_t n, void* stream) {
    DIN_REQUIRE(x && y && out && n >= 0 && idx >= 0, "dot_accum: bad argument");
    if (n == 0) return DIN_OK;
    hipLaunchKernelGGL(dot_kernel, dim3(grid_1d(n, 256, 512)), dim3(256), 0, as_stream(stream), x, y, out, idx, n);
    DIN_CHECK_LAUNCH("dot_accum");
    return DIN_OK;
}
int din_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
    DIN_REQUIRE(src && dst && n >= 0, "cast: bad argument");
    if (n == 0) return DIN_OK;
    hipLaunchKernelGGL(cast_kernel, dim3(grid_1d(n, 256, 8192)), dim3(256), 0, as_stream(stream), src, src_dtype, dst, dst_dtype, n);
    DIN_CHECK_LAUNCH("cast");
    return DIN_OK;
}
int din_nhwc_to_nchw_f32(const void* in, int dtype, int nb, int h, int w, int c, int ld, int coff, float* out, void* stream) {
    DIN_REQUIRE(in && out && nb > 0 && h > 0 && w > 0 && c > 0, "nhwc_to_nchw: bad argument");
    int64_t hw = (int64_t)h * w;
    dim3 grid((unsigned)ceil_div64(hw, 32), (unsigned)((c + 31) / 32), (unsigned)nb);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(32, 8), 0, as_stream(stream), in, dtype, hw, c, ld, coff, out);
    DIN_CHECK_LAUNCH("nhwc_to_nchw");
    return DIN_OK;
}
int din_nchw_f32_to_nhwc(const float* in, int nb, int h, int w, int c, void* out, int dtype, int ld, int coff, void* stream) {
    DIN_REQUIRE(in && out && nb > 0 && h > 0 && w > 0 && c > 0, "nchw_to_nhwc: bad argument");
    int64_t hw = (int64_t)h * w;
    dim3 grid((unsigned)ceil_div64(hw, 32), (unsigned)((c + 31) / 32), (unsigned)nb);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(32, 8), 0, as_stream(stream), in, hw, c, out, dtype, ld, coff);
    DIN_CHECK_LAUNCH("nchw_to_nhwc");
    return DIN_OK;
}
int din_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float grad_scale, void* stream) {
    DIN_REQUIRE(p && g && m && v && n >= 0 && step >= 1, "adam_step: bad argument");
    if (n == 0) return DIN_OK;
    float bc1 = 1.f - powf(beta1, (float)step), bc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_1d(n, 256, 4096)), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2, grad_scale);
    DIN_CHECK_LAUNCH("adam_step");
    return DIN_OK;
}

int din_adam_step_multi(const uint64_t* ptrs, const int64_t* sizes, const int32_t* chunk_tensor, const int32_t* chunk_index, int nchunks,
                        int chunk_elems, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                        void* stream) {
    DIN_REQUIRE(ptrs && sizes && chunk_tensor && chunk_index && nchunks >= 0 && chunk_elems > 0 && step >= 1, "adam_step_multi: bad argument");
    if (nchunks == 0) return DIN_OK;
    float bc1 = 1.f - powf(beta1, (float)step), bc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nchunks), dim3(256), 0, as_stream(stream), ptrs, sizes, chunk_tensor, chunk_index, chunk_elems,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
    DIN_CHECK_LAUNCH("adam_step_multi");
    return DIN_OK;
}

}  // extern "C"
