// HBM-bound helpers of the DIN stage-2 path for gfx950: image prep, pools, bilinear resize, casts, Adam.
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

// ---- pools ---------------------------------------------------------------------------------------------
struct PoolK {
    int nb, h, w, c, oh, ow, k, stride, pad, ldi, cioff, ldo, cooff, dtype;
};
__device__ __forceinline__ PoolK mk(const din_pool_desc& d) {
    return PoolK{d.nb, d.h, d.w, d.c, d.oh, d.ow, d.k, d.stride, d.pad, d.ldi, d.cioff, d.ldo, d.cooff, d.dtype};
}

// Forward optionally records, per pooled element, which window tap won (first maximum in scan order = PyTorch's tie rule)
// as one byte: tap index r*k+s, or 255 when the winner is <= 0 (the fused ReLU backward would zero its gradient anyway).
__global__ void maxpool_fwd_kernel(din_pool_desc d, const void* __restrict__ in, void* __restrict__ out, uint8_t* __restrict__ amax) {
    const int c4 = d.c >> 2;
    int64_t total = (int64_t)d.nb * d.oh * d.ow * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c4);
        int64_t p = i / c4;
        int ox = (int)(p % d.ow);
        int64_t q = p / d.ow;
        int oy = (int)(q % d.oh), n = (int)(q / d.oh);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int am[4] = {0, 0, 0, 0};
        for (int r = 0; r < d.k; ++r) {
            int iy = oy * d.stride - d.pad + r;
            if (iy < 0 || iy >= d.h) continue;
            for (int s = 0; s < d.k; ++s) {
                int ix = ox * d.stride - d.pad + s;
                if (ix < 0 || ix >= d.w) continue;
                f32x4 v = ld4(in, d.dtype, ((int64_t)(n * d.h + iy) * d.w + ix) * d.ldi + d.cioff + cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[e] > m[e]) { m[e] = v[e]; am[e] = r * d.k + s; }   // first max wins on ties
            }
        }
        st4(out, d.dtype, p * d.ldo + d.cooff + cg * 4, m);
        if (amax) {
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= (uint32_t)(m[e] > 0.f ? am[e] : 255) << (8 * e);
            *reinterpret_cast<uint32_t*>(amax + p * d.c + cg * 4) = pk;
        }
    }
}

// Backward from the saved map (gather form, no atomics, no re-read of the input): each input element visits the <= ceil(k/s)^2
// windows that contain it and takes the gradient where the recorded tap is itself.
__global__ void maxpool_bwd_amax_kernel(din_pool_desc d, const uint8_t* __restrict__ amax, const void* __restrict__ dout,
                                        void* __restrict__ din_, int accumulate) {
    const int c4 = d.c >> 2;
    int64_t total = (int64_t)d.nb * d.h * d.w * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c4);
        int64_t p = i / c4;
        int ix = (int)(p % d.w);
        int64_t q = p / d.w;
        int iy = (int)(q % d.h), n = (int)(q / d.h);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        int oy_hi = (iy + d.pad) / d.stride, ox_hi = (ix + d.pad) / d.stride;
        int oy_lo = (iy + d.pad - d.k + d.stride) / d.stride, ox_lo = (ix + d.pad - d.k + d.stride) / d.stride;
        if (iy + d.pad - d.k + 1 < 0) oy_lo = 0;
        if (ix + d.pad - d.k + 1 < 0) ox_lo = 0;
        if (oy_hi >= d.oh) oy_hi = d.oh - 1;
        if (ox_hi >= d.ow) ox_hi = d.ow - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const uint32_t tap = (uint32_t)((iy - (oy * d.stride - d.pad)) * d.k + (ix - (ox * d.stride - d.pad)));
                int64_t po = (int64_t)(n * d.oh + oy) * d.ow + ox;
                uint32_t pk = *reinterpret_cast<const uint32_t*>(amax + po * d.c + cg * 4);
                if (((pk & 0xff) != tap) && (((pk >> 8) & 0xff) != tap) && (((pk >> 16) & 0xff) != tap) && ((pk >> 24) != tap)) continue;
                f32x4 go = ld4(dout, d.dtype, po * d.ldo + d.cooff + cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += ((pk >> (8 * e)) & 0xff) == tap ? go[e] : 0.f;
            }
        int64_t self_off = p * d.ldi + d.cioff + cg * 4;
        if (accumulate) g += ld4(din_, d.dtype, self_off);
        st4(din_, d.dtype, self_off, g);
    }
}

// Map-free backward (recomputes each window's arg-max): kept for callers that did not save the map.
__global__ void maxpool_bwd_kernel(din_pool_desc d, const void* __restrict__ in, const void* __restrict__ dout,
                                   void* __restrict__ din_, int relu_mask, int accumulate) {
    const int c4 = d.c >> 2;
    int64_t total = (int64_t)d.nb * d.h * d.w * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c4);
        int64_t p = i / c4;
        int ix = (int)(p % d.w);
        int64_t q = p / d.w;
        int iy = (int)(q % d.h), n = (int)(q / d.h);
        int64_t self_off = p * d.ldi + d.cioff + cg * 4;
        f32x4 xv = ld4(in, d.dtype, self_off);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        int oy_hi = (iy + d.pad) / d.stride, ox_hi = (ix + d.pad) / d.stride;
        int oy_lo = (iy + d.pad - d.k + d.stride) / d.stride, ox_lo = (ix + d.pad - d.k + d.stride) / d.stride;
        if (iy + d.pad - d.k + 1 < 0) oy_lo = 0;
        if (ix + d.pad - d.k + 1 < 0) ox_lo = 0;
        if (oy_hi >= d.oh) oy_hi = d.oh - 1;
        if (ox_hi >= d.ow) ox_hi = d.ow - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                bool win[4] = {true, true, true, true};
                for (int r = 0; r < d.k; ++r) {
                    int yy = oy * d.stride - d.pad + r;
                    if (yy < 0 || yy >= d.h) continue;
                    for (int s = 0; s < d.k; ++s) {
                        int xx = ox * d.stride - d.pad + s;
                        if (xx < 0 || xx >= d.w) continue;
                        if (yy == iy && xx == ix) continue;
                        f32x4 v = ld4(in, d.dtype, ((int64_t)(n * d.h + yy) * d.w + xx) * d.ldi + d.cioff + cg * 4);
                        bool before = (yy < iy) || (yy == iy && xx < ix);
#pragma unroll
                        for (int e = 0; e < 4; ++e) win[e] = win[e] && (before ? v[e] < xv[e] : v[e] <= xv[e]);
                    }
                }
                f32x4 go = ld4(dout, d.dtype, ((int64_t)(n * d.oh + oy) * d.ow + ox) * d.ldo + d.cooff + cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += win[e] ? go[e] : 0.f;
            }
        if (relu_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = xv[e] > 0.f ? g[e] : 0.f;
        }
        if (accumulate) { f32x4 o = ld4(din_, d.dtype, self_off); g += o; }
        st4(din_, d.dtype, self_off, g);
    }
}

// flags: DIN_CONV_BIAS adds bias[c] after the average, DIN_CONV_RELU clamps -- the epilogue of a 1x1 conv that was commuted in
// front of the pool (avgpool(conv1x1(x)) == conv1x1(avgpool(x)): both linear, zero padding maps to zero)
__global__ void avgpool_fwd_kernel(din_pool_desc d, const void* __restrict__ in, void* __restrict__ out,
                                   const float* __restrict__ bias, int flags) {
    const int c4 = d.c >> 2;
    int64_t total = (int64_t)d.nb * d.oh * d.ow * c4;
    const float inv = 1.f / (float)(d.k * d.k);                       // count_include_pad=True
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c4);
        int64_t p = i / c4;
        int ox = (int)(p % d.ow);
        int64_t q = p / d.ow;
        int oy = (int)(q % d.oh), n = (int)(q / d.oh);
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < d.k; ++r) {
            int iy = oy * d.stride - d.pad + r;
            if (iy < 0 || iy >= d.h) continue;
            for (int s = 0; s < d.k; ++s) {
                int ix = ox * d.stride - d.pad + s;
                if (ix < 0 || ix >= d.w) continue;
                s4 += ld4(in, d.dtype, ((int64_t)(n * d.h + iy) * d.w + ix) * d.ldi + d.cioff + cg * 4);
            }
        }
        s4 = s4 * inv;
        if (flags & DIN_CONV_BIAS) s4 += *reinterpret_cast<const f32x4*>(bias + cg * 4);
        if (flags & DIN_CONV_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] = fmaxf(s4[e], 0.f);
        }
        st4(out, d.dtype, p * d.ldo + d.cooff + cg * 4, s4);
    }
}
__global__ void avgpool_fwd8_kernel(din_pool_desc d, const void* __restrict__ in, void* __restrict__ out,
                                    const float* __restrict__ bias, int flags) {
    const int c8 = d.c >> 3;
    int64_t total = (int64_t)d.nb * d.oh * d.ow * c8;
    const float inv = 1.f / (float)(d.k * d.k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c8);
        int64_t p = i / c8;
        int ox = (int)(p % d.ow);
        int64_t q = p / d.ow;
        int oy = (int)(q % d.oh), n = (int)(q / d.oh);
        f32x8 a; a.lo = f32x4{0.f, 0.f, 0.f, 0.f}; a.hi = a.lo;
        for (int r = 0; r < d.k; ++r) {
            int iy = oy * d.stride - d.pad + r;
            if (iy < 0 || iy >= d.h) continue;
            for (int s = 0; s < d.k; ++s) {
                int ix = ox * d.stride - d.pad + s;
                if (ix < 0 || ix >= d.w) continue;
                f32x8 v = ld8_bf16(in, ((int64_t)(n * d.h + iy) * d.w + ix) * d.ldi + d.cioff + cg * 8);
                a.lo += v.lo; a.hi += v.hi;
            }
        }
        a.lo = a.lo * inv; a.hi = a.hi * inv;
        if (flags & DIN_CONV_BIAS) {
            a.lo += *reinterpret_cast<const f32x4*>(bias + cg * 8);
            a.hi += *reinterpret_cast<const f32x4*>(bias + cg * 8 + 4);
        }
        if (flags & DIN_CONV_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { a.lo[e] = fmaxf(a.lo[e], 0.f); a.hi[e] = fmaxf(a.hi[e], 0.f); }
        }
        st8_bf16(out, p * d.ldo + d.cooff + cg * 8, a);
    }
}
__global__ void avgpool_bwd8_kernel(din_pool_desc d, const void* __restrict__ dout, void* __restrict__ din_,
                                    const void* __restrict__ mask, int accumulate) {
    const int c8 = d.c >> 3;
    int64_t total = (int64_t)d.nb * d.h * d.w * c8;
    const float inv = 1.f / (float)(d.k * d.k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c8);
        int64_t p = i / c8;
        int ix = (int)(p % d.w);
        int64_t q = p / d.w;
        int iy = (int)(q % d.h), n = (int)(q / d.h);
        f32x8 g; g.lo = f32x4{0.f, 0.f, 0.f, 0.f}; g.hi = g.lo;
        for (int r = 0; r < d.k; ++r) {
            int ty = iy + d.pad - r;
            if (ty < 0 || ty % d.stride) continue;
            int oy = ty / d.stride;
            if (oy >= d.oh) continue;
            for (int s = 0; s < d.k; ++s) {
                int tx = ix + d.pad - s;
                if (tx < 0 || tx % d.stride) continue;
                int ox = tx / d.stride;
                if (ox >= d.ow) continue;
                f32x8 v = ld8_bf16(dout, ((int64_t)(n * d.oh + oy) * d.ow + ox) * d.ldo + d.cooff + cg * 8);
                g.lo += v.lo; g.hi += v.hi;
            }
        }
        g.lo = g.lo * inv; g.hi = g.hi * inv;
        int64_t off = p * d.ldi + d.cioff + cg * 8;
        if (mask) {
            f32x8 y = ld8_bf16(mask, off);
#pragma unroll
            for (int e = 0; e < 4; ++e) { g.lo[e] = y.lo[e] > 0.f ? g.lo[e] : 0.f; g.hi[e] = y.hi[e] > 0.f ? g.hi[e] : 0.f; }
        }
        if (accumulate) { f32x8 o = ld8_bf16(din_, off); g.lo += o.lo; g.hi += o.hi; }
        st8_bf16(din_, off, g);
    }
}
// 8-wide max-pool backward from the arg-max map
__global__ void maxpool_bwd_amax8_kernel(din_pool_desc d, const uint8_t* __restrict__ amax, const void* __restrict__ dout,
                                         void* __restrict__ din_, int accumulate) {
    const int c8 = d.c >> 3;
    int64_t total = (int64_t)d.nb * d.h * d.w * c8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c8);
        int64_t p = i / c8;
        int ix = (int)(p % d.w);
        int64_t q = p / d.w;
        int iy = (int)(q % d.h), n = (int)(q / d.h);
        f32x8 g; g.lo = f32x4{0.f, 0.f, 0.f, 0.f}; g.hi = g.lo;
        int oy_hi = (iy + d.pad) / d.stride, ox_hi = (ix + d.pad) / d.stride;
        int oy_lo = (iy + d.pad - d.k + d.stride) / d.stride, ox_lo = (ix + d.pad - d.k + d.stride) / d.stride;
        if (iy + d.pad - d.k + 1 < 0) oy_lo = 0;
        if (ix + d.pad - d.k + 1 < 0) ox_lo = 0;
        if (oy_hi >= d.oh) oy_hi = d.oh - 1;
        if (ox_hi >= d.ow) ox_hi = d.ow - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const uint32_t tap = (uint32_t)((iy - (oy * d.stride - d.pad)) * d.k + (ix - (ox * d.stride - d.pad)));
                int64_t po = (int64_t)(n * d.oh + oy) * d.ow + ox;
                uint2 pk = *reinterpret_cast<const uint2*>(amax + po * d.c + cg * 8);
                const uint32_t t4 = tap * 0x01010101u;
                // any byte equal to tap?  (x ^ t4) has a zero byte
                uint32_t xa = pk.x ^ t4, xb = pk.y ^ t4;
                bool any = (((xa - 0x01010101u) & ~xa & 0x80808080u) | ((xb - 0x01010101u) & ~xb & 0x80808080u)) != 0u;
                if (!any) continue;
                f32x8 go = ld8_bf16(dout, po * d.ldo + d.cooff + cg * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g.lo[e] += ((pk.x >> (8 * e)) & 0xff) == tap ? go.lo[e] : 0.f;
                    g.hi[e] += ((pk.y >> (8 * e)) & 0xff) == tap ? go.hi[e] : 0.f;
                }
            }
        int64_t self_off = p * d.ldi + d.cioff + cg * 8;
        if (accumulate) { f32x8 o = ld8_bf16(din_, self_off); g.lo += o.lo; g.hi += o.hi; }
        st8_bf16(din_, self_off, g);
    }
}
__global__ void avgpool_bwd_kernel(din_pool_desc d, const void* __restrict__ dout, void* __restrict__ din_,
                                   const void* __restrict__ mask, int accumulate) {
    const int c4 = d.c >> 2;
    int64_t total = (int64_t)d.nb * d.h * d.w * c4;
    const float inv = 1.f / (float)(d.k * d.k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c4);
        int64_t p = i / c4;
        int ix = (int)(p % d.w);
        int64_t q = p / d.w;
        int iy = (int)(q % d.h), n = (int)(q / d.h);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < d.k; ++r) {
            int ty = iy + d.pad - r;
            if (ty < 0 || ty % d.stride) continue;
            int oy = ty / d.stride;
            if (oy >= d.oh) continue;
            for (int s = 0; s < d.k; ++s) {
                int tx = ix + d.pad - s;
                if (tx < 0 || tx % d.stride) continue;
                int ox = tx / d.stride;
                if (ox >= d.ow) continue;
                g += ld4(dout, d.dtype, ((int64_t)(n * d.oh + oy) * d.ow + ox) * d.ldo + d.cooff + cg * 4);
            }
        }
        g = g * inv;
        int64_t off = p * d.ldi + d.cioff + cg * 4;
        if (mask) {
            f32x4 y = ld4(mask, d.dtype, off);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
        }
        if (accumulate) g += ld4(din_, d.dtype, off);
        st4(din_, d.dtype, off, g);
    }
}

// bilinear resize, align_corners=True (infer_model.py:169): src = dst*(in-1)/(out-1)
__device__ __forceinline__ void bil_coord(int o, int in, int out, int& i0, int& i1, float& l) {
    float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    float src = sc * (float)o;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + 1 < in ? i0 + 1 : in - 1;
    l = src - (float)i0;
}
__global__ void bilinear_fwd_kernel(din_pool_desc d, const void* __restrict__ in, void* __restrict__ out) {
    const int c4 = d.c >> 2;
    int64_t total = (int64_t)d.nb * d.oh * d.ow * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c4);
        int64_t p = i / c4;
        int ox = (int)(p % d.ow);
        int64_t q = p / d.ow;
        int oy = (int)(q % d.oh), n = (int)(q / d.oh);
        int y0, y1, x0, x1; float ly, lx;
        bil_coord(oy, d.h, d.oh, y0, y1, ly);
        bil_coord(ox, d.w, d.ow, x0, x1, lx);
        auto at = [&](int y, int x) { return ld4(in, d.dtype, ((int64_t)(n * d.h + y) * d.w + x) * d.ldi + d.cioff + cg * 4); };
        f32x4 top = at(y0, x0) * (1.f - lx) + at(y0, x1) * lx;
        f32x4 bot = at(y1, x0) * (1.f - lx) + at(y1, x1) * lx;
        st4(out, d.dtype, p * d.ldo + d.cooff + cg * 4, top * (1.f - ly) + bot * ly);
    }
}
// gather-form backward: each input cell sums the contributions of the output cells whose 2x2 footprint touches it.
// Because the map is monotone, candidate outputs for input row y are those with y0 in {y-1, y}; we scan the (small)
// output range bounded by the inverse scale.
__global__ void bilinear_bwd_kernel(din_pool_desc d, const void* __restrict__ dout, void* __restrict__ din_,
                                    const void* __restrict__ mask, int accumulate) {
    const int c4 = d.c >> 2;
    int64_t total = (int64_t)d.nb * d.h * d.w * c4;
    const float scy = d.oh > 1 ? (float)(d.h - 1) / (float)(d.oh - 1) : 0.f;
    const float scx = d.ow > 1 ? (float)(d.w - 1) / (float)(d.ow - 1) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % c4);
        int64_t p = i / c4;
        int ix = (int)(p % d.w);
        int64_t q = p / d.w;
        int iy = (int)(q % d.h), n = (int)(q / d.h);
        // output rows whose source coordinate lies in (iy-1, iy+1)
        int oy_lo = scy > 0.f ? (int)floorf((float)(iy - 1) / scy) : 0, oy_hi = scy > 0.f ? (int)ceilf((float)(iy + 1) / scy) : d.oh - 1;
        int ox_lo = scx > 0.f ? (int)floorf((float)(ix - 1) / scx) : 0, ox_hi = scx > 0.f ? (int)ceilf((float)(ix + 1) / scx) : d.ow - 1;
        if (oy_lo < 0) oy_lo = 0;
        if (ox_lo < 0) ox_lo = 0;
        if (oy_hi > d.oh - 1) oy_hi = d.oh - 1;
        if (ox_hi > d.ow - 1) ox_hi = d.ow - 1;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float ly;
            bil_coord(oy, d.h, d.oh, y0, y1, ly);
            float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float lx;
                bil_coord(ox, d.w, d.ow, x0, x1, lx);
                float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx == 0.f) continue;
                g += ld4(dout, d.dtype, ((int64_t)(n * d.oh + oy) * d.ow + ox) * d.ldo + d.cooff + cg * 4) * (wy * wx);
            }
        }
        int64_t off = p * d.ldi + d.cioff + cg * 4;
        if (mask) {
            f32x4 y = ld4(mask, d.dtype, off);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
        }
        if (accumulate) g += ld4(din_, d.dtype, off);
        st4(din_, d.dtype, off, g);
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

inline bool wide8(const din_pool_desc* d) {
    return d->dtype == DIN_BF16 && d->c % 8 == 0 && d->ldi % 8 == 0 && d->ldo % 8 == 0 && d->cioff % 8 == 0 && d->cooff % 8 == 0;
}

int check_pool(const din_pool_desc* d, const char* what) {
    DIN_REQUIRE(d != nullptr, "%s: null descriptor", what);
    DIN_REQUIRE(d->dtype == DIN_F32 || d->dtype == DIN_BF16, "%s: bad dtype", what);
    DIN_REQUIRE(d->c % 4 == 0 && d->ldi % 4 == 0 && d->ldo % 4 == 0 && d->cioff % 4 == 0 && d->cooff % 4 == 0,
                "%s: channels/strides/offsets must be multiples of 4", what);
    DIN_REQUIRE(d->nb > 0 && d->h > 0 && d->w > 0 && d->oh > 0 && d->ow > 0 && d->c > 0, "%s: empty tensor", what);
    return DIN_OK;
}

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

int din_maxpool_fwd(const din_pool_desc* d, const void* in, void* out, uint8_t* argmax, void* stream) {
    if (int e = check_pool(d, "maxpool_fwd")) return e;
    DIN_REQUIRE(in && out, "maxpool_fwd: null pointer");
    DIN_REQUIRE(d->k * d->k < 255, "maxpool_fwd: window too large for the byte arg-max map");
    int64_t total = (int64_t)d->nb * d->oh * d->ow * (d->c / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), *d, in, out, argmax);
    DIN_CHECK_LAUNCH("maxpool_fwd");
    return DIN_OK;
}
int din_maxpool_bwd(const din_pool_desc* d, const void* in, const uint8_t* argmax, const void* dout, void* din_, int relu_mask,
                    int accumulate, void* stream) {
    if (int e = check_pool(d, "maxpool_bwd")) return e;
    DIN_REQUIRE((in || argmax) && dout && din_, "maxpool_bwd: null pointer");
    int64_t total = (int64_t)d->nb * d->h * d->w * (d->c / 4);
    if (argmax) {
        DIN_REQUIRE(relu_mask, "maxpool_bwd: the arg-max map encodes the fused ReLU mask; relu_mask must be set");
        if (wide8(d)) hipLaunchKernelGGL(maxpool_bwd_amax8_kernel, dim3(grid_1d(total / 2, 256, 32768)), dim3(256), 0, as_stream(stream), *d, argmax, dout, din_, accumulate);
        else hipLaunchKernelGGL(maxpool_bwd_amax_kernel, dim3(grid_1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), *d, argmax, dout, din_, accumulate);
    } else {
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), *d, in, dout, din_, relu_mask, accumulate);
    }
    DIN_CHECK_LAUNCH("maxpool_bwd");
    return DIN_OK;
}
int din_avgpool_fwd(const din_pool_desc* d, const void* in, void* out, const float* bias, int flags, void* stream) {
    if (int e = check_pool(d, "avgpool_fwd")) return e;
    DIN_REQUIRE(in && out, "avgpool_fwd: null pointer");
    DIN_REQUIRE(!(flags & ~(DIN_CONV_BIAS | DIN_CONV_RELU)), "avgpool_fwd: only BIAS / RELU flags");
    DIN_REQUIRE(!(flags & DIN_CONV_BIAS) || bias, "avgpool_fwd: BIAS flag without bias");
    int64_t total = (int64_t)d->nb * d->oh * d->ow * (d->c / 4);
    if (wide8(d)) hipLaunchKernelGGL(avgpool_fwd8_kernel, dim3(grid_1d(total / 2, 256, 32768)), dim3(256), 0, as_stream(stream), *d, in, out, bias, flags);
    else hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), *d, in, out, bias, flags);
    DIN_CHECK_LAUNCH("avgpool_fwd");
    return DIN_OK;
}
int din_avgpool_bwd(const din_pool_desc* d, const void* dout, void* din_, const void* mask, int accumulate, void* stream) {
    if (int e = check_pool(d, "avgpool_bwd")) return e;
    DIN_REQUIRE(dout && din_, "avgpool_bwd: null pointer");
    int64_t total = (int64_t)d->nb * d->h * d->w * (d->c / 4);
    if (wide8(d)) hipLaunchKernelGGL(avgpool_bwd8_kernel, dim3(grid_1d(total / 2, 256, 32768)), dim3(256), 0, as_stream(stream), *d, dout, din_, mask, accumulate);
    else hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), *d, dout, din_, mask, accumulate);
    DIN_CHECK_LAUNCH("avgpool_bwd");
    return DIN_OK;
}
int din_bilinear_fwd(const din_pool_desc* d, const void* in, void* out, void* stream) {
    if (int e = check_pool(d, "bilinear_fwd")) return e;
    DIN_REQUIRE(in && out, "bilinear_fwd: null pointer");
    int64_t total = (int64_t)d->nb * d->oh * d->ow * (d->c / 4);
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid_1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), *d, in, out);
    DIN_CHECK_LAUNCH("bilinear_fwd");
    return DIN_OK;
}
int din_bilinear_bwd(const din_pool_desc* d, const void* dout, void* din_, const void* mask, int accumulate, void* stream) {
    if (int e = check_pool(d, "bilinear_bwd")) return e;
    DIN_REQUIRE(dout && din_, "bilinear_bwd: null pointer");
    int64_t total = (int64_t)d->nb * d->h * d->w * (d->c / 4);
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), *d, dout, din_, mask, accumulate);
    DIN_CHECK_LAUNCH("bilinear_bwd");
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
int din_axpby(const float* x, const float* y, float* out, float alpha, float beta, int64_t n, void* stream) {
    DIN_REQUIRE(x && out && n >= 0, "axpby: bad argument");
    if (n == 0) return DIN_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, as_stream(stream), x, y, out, alpha, beta, n);
    DIN_CHECK_LAUNCH("axpby");
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

}  // extern "C"
