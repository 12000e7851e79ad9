// Pools and bilinear resize of the DIN stage-2 backbone path for gfx950 (HBM-bound; no MFMA work here).
//   max_pool2d (backbone.py MyVGG16 / torchvision Inception3 stem + InceptionB), avg_pool2d count_include_pad (InceptionA/C
//   branch_pool), F.interpolate(bilinear, align_corners=True) (infer_model.py:165-172 multiscale fuse).
// All tensors are NHWC views (pixel stride, channel offset).  One thread = V channels of one pixel: V = 8 (16-byte bf16 vectors) when
// the view allows it, else V = 4 (8-/16-byte vectors, either storage type).  What makes these kernels run at the memory system's
// speed rather than the ALU's / the latency's:
//   * the linear index -> (n, y, x, channel group) decode uses host-precomputed multiply-shift reciprocals (three 64-bit
//     divisions per element cost more than the element's memory traffic);
//   * the 3x3 / 2x2 windows are template constants: all taps are issued as independent, predicated loads (clamped address +
//     select) instead of a data-dependent loop with one load in flight;
//   * workgroups walk the tensor in XCD-contiguous order (din_common.h xcd_remap) so the rows a window shares with its
//     neighbours are L2 hits.
#include "din_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- index decode ---------------------------------------------------------------------------------------------------------------
struct FastDiv { uint32_t mul; int shift; int d; };      // q = n / d for 0 <= n < 2^31:  mul == 0 -> d == 1
FastDiv make_fastdiv(int d) {
    FastDiv f{0u, 0, d};
    if (d <= 1) return f;
    int l = 0;
    while ((1ll << l) < d) ++l;                           // ceil(log2 d) >= 1
    f.mul = (uint32_t)(((1ull << (31 + l)) + (uint64_t)d - 1) / (uint64_t)d);   // ceil(2^(31+l) / d) <= 2^32 - 1 for d >= 2
    f.shift = l - 1;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) { return f.mul ? (__umulhi(n, f.mul) >> f.shift) : n; }

struct Dec3 { FastDiv c, w, h; int fast; };              // element id -> (channel group, x, y, n); fast: total < 2^31
Dec3 make_dec(int cgroups, int w, int h, int64_t total) { return Dec3{make_fastdiv(cgroups), make_fastdiv(w), make_fastdiv(h), total < (1ll << 31) ? 1 : 0}; }
__device__ __forceinline__ void decode(const Dec3& dd, int64_t i, int& cg, int& x, int& y, int& n, int64_t& p) {
    if (dd.fast) {
        const uint32_t ii = (uint32_t)i;
        const uint32_t pp = fdiv(ii, dd.c);
        cg = (int)(ii - pp * (uint32_t)dd.c.d);
        const uint32_t q = fdiv(pp, dd.w);
        x = (int)(pp - q * (uint32_t)dd.w.d);
        const uint32_t nn = fdiv(q, dd.h);
        y = (int)(q - nn * (uint32_t)dd.h.d);
        n = (int)nn; p = (int64_t)pp;
    } else {
        p = i / dd.c.d; cg = (int)(i - p * dd.c.d);
        int64_t q = p / dd.w.d; x = (int)(p - q * dd.w.d);
        n = (int)(q / dd.h.d); y = (int)(q - (int64_t)n * dd.h.d);
    }
}
#define DIN_GRID_STRIDE(i, total) \
    for (int64_t i = (int64_t)xcd_remap((int)blockIdx.x, (int)gridDim.x) * blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

// ---- V-channel vectors ----------------------------------------------------------------------------------------------------------
template <int V> struct Vec { float v[V]; };
template <int V> __device__ __forceinline__ Vec<V> vzero() { Vec<V> r; for (int e = 0; e < V; ++e) r.v[e] = 0.f; return r; }
template <int V> __device__ __forceinline__ Vec<V> vload(const void* base, int dtype, int64_t i);
template <> __device__ __forceinline__ Vec<4> vload<4>(const void* base, int dtype, int64_t i) {
    Vec<4> o;
    if (dtype == DIN_F32) {
        f32x4 t = *reinterpret_cast<const f32x4*>((const float*)base + i);
        o.v[0] = t[0]; o.v[1] = t[1]; o.v[2] = t[2]; o.v[3] = t[3];
    } else {
        uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)base + i);
        o.v[0] = __uint_as_float(r.x << 16); o.v[1] = __uint_as_float(r.x & 0xffff0000u);
        o.v[2] = __uint_as_float(r.y << 16); o.v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
    return o;
}
template <> __device__ __forceinline__ Vec<8> vload<8>(const void* base, int, int64_t i) {       // bf16 only
    uint4 r = *reinterpret_cast<const uint4*>((const bf16_t*)base + i);
    Vec<8> o;
    o.v[0] = __uint_as_float(r.x << 16); o.v[1] = __uint_as_float(r.x & 0xffff0000u);
    o.v[2] = __uint_as_float(r.y << 16); o.v[3] = __uint_as_float(r.y & 0xffff0000u);
    o.v[4] = __uint_as_float(r.z << 16); o.v[5] = __uint_as_float(r.z & 0xffff0000u);
    o.v[6] = __uint_as_float(r.w << 16); o.v[7] = __uint_as_float(r.w & 0xffff0000u);
    return o;
}
template <int V> __device__ __forceinline__ void vstore(void* base, int dtype, int64_t i, const Vec<V>& a);
template <> __device__ __forceinline__ void vstore<4>(void* base, int dtype, int64_t i, const Vec<4>& a) {
    if (dtype == DIN_F32) { *reinterpret_cast<f32x4*>((float*)base + i) = f32x4{a.v[0], a.v[1], a.v[2], a.v[3]}; return; }
    *reinterpret_cast<uint2*>((bf16_t*)base + i) = uint2{pack_bf16x2(a.v[0], a.v[1]), pack_bf16x2(a.v[2], a.v[3])};
}
template <> __device__ __forceinline__ void vstore<8>(void* base, int, int64_t i, const Vec<8>& a) {
    *reinterpret_cast<uint4*>((bf16_t*)base + i) =
        uint4{pack_bf16x2(a.v[0], a.v[1]), pack_bf16x2(a.v[2], a.v[3]), pack_bf16x2(a.v[4], a.v[5]), pack_bf16x2(a.v[6], a.v[7])};
}
// V arg-max bytes
template <int V> __device__ __forceinline__ void amax_load(const uint8_t* p, uint32_t (&w)[V / 4]);
template <> __device__ __forceinline__ void amax_load<4>(const uint8_t* p, uint32_t (&w)[1]) { w[0] = *reinterpret_cast<const uint32_t*>(p); }
template <> __device__ __forceinline__ void amax_load<8>(const uint8_t* p, uint32_t (&w)[2]) {
    uint2 t = *reinterpret_cast<const uint2*>(p); w[0] = t.x; w[1] = t.y;
}

// ---- max pool -------------------------------------------------------------------------------------------------------------------
// Forward optionally records, per pooled element, which window tap won (first maximum in scan order = PyTorch's tie rule) as one
// byte: tap index r*k+s, or 255 when the winner is <= 0 (the fused ReLU backward would zero its gradient anyway).
// K = 0: runtime window size (loop form).
template <int V, int K>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ in, void* __restrict__ out,
                                                          uint8_t* __restrict__ amax) {
    const int64_t total = (int64_t)d.nb * d.oh * d.ow * (d.c / V);
    const int k = K ? K : d.k;
    DIN_GRID_STRIDE(i, total) {
        int cg, ox, oy, n; int64_t p;
        decode(dd, i, cg, ox, oy, n, p);
        Vec<V> m; int am[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { m.v[e] = -INFINITY; am[e] = 0; }
        const int y0 = oy * d.stride - d.pad, x0 = ox * d.stride - d.pad;
        const int64_t img = (int64_t)n * d.h;
        if (K) {
            Vec<V> t[K ? K * K : 1];
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const int iy = min(max(y0 + r, 0), d.h - 1), ix = min(max(x0 + s, 0), d.w - 1);
                    t[r * K + s] = vload<V>(in, d.dtype, ((img + iy) * d.w + ix) * d.ldi + d.cioff + cg * V);
                }
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const bool ok = y0 + r >= 0 && y0 + r < d.h && x0 + s >= 0 && x0 + s < d.w;
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        if (ok && t[r * K + s].v[e] > m.v[e]) { m.v[e] = t[r * K + s].v[e]; am[e] = r * K + s; }   // first max wins on ties
                }
        } else {
            for (int r = 0; r < k; ++r) {
                const int iy = y0 + r;
                if (iy < 0 || iy >= d.h) continue;
                for (int s = 0; s < k; ++s) {
                    const int ix = x0 + s;
                    if (ix < 0 || ix >= d.w) continue;
                    Vec<V> t = vload<V>(in, d.dtype, ((img + iy) * d.w + ix) * d.ldi + d.cioff + cg * V);
#pragma unroll
                    for (int e = 0; e < V; ++e) if (t.v[e] > m.v[e]) { m.v[e] = t.v[e]; am[e] = r * k + s; }
                }
            }
        }
        vstore<V>(out, d.dtype, p * d.ldo + d.cooff + cg * V, m);
        if (amax) {
            uint32_t pk[V / 4];
#pragma unroll
            for (int q = 0; q < V / 4; ++q) {
                pk[q] = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[q] |= (uint32_t)(m.v[q * 4 + e] > 0.f ? am[q * 4 + e] : 255) << (8 * e);
            }
            if (V == 4) *reinterpret_cast<uint32_t*>(amax + p * d.c + cg * V) = pk[0];
            else *reinterpret_cast<uint2*>(amax + p * d.c + cg * V) = uint2{pk[0], pk[V / 4 - 1]};
        }
    }
}

// Backward from the saved map (gather form, no atomics, no re-read of the input): each input element visits the <= ceil(k/s)^2
// windows that contain it and takes the gradient where the recorded tap is itself.  NW = windows per axis (template) or 0 (loop).
template <int V, int NW>
__global__ __launch_bounds__(256) void maxpool_bwd_amax_kernel(din_pool_desc d, Dec3 dd, const uint8_t* __restrict__ amax,
                                                               const void* __restrict__ dout, void* __restrict__ din_, int accumulate) {
    const int64_t total = (int64_t)d.nb * d.h * d.w * (d.c / V);
    DIN_GRID_STRIDE(i, total) {
        int cg, ix, iy, n; int64_t p;
        decode(dd, i, cg, ix, iy, n, p);
        Vec<V> g = vzero<V>();
        int oy_hi = (iy + d.pad) / d.stride, ox_hi = (ix + d.pad) / d.stride;
        int oy_lo = (iy + d.pad - d.k + d.stride) / d.stride, ox_lo = (ix + d.pad - d.k + d.stride) / d.stride;
        if (iy + d.pad - d.k + 1 < 0) oy_lo = 0;
        if (ix + d.pad - d.k + 1 < 0) ox_lo = 0;
        if (oy_hi >= d.oh) oy_hi = d.oh - 1;
        if (ox_hi >= d.ow) ox_hi = d.ow - 1;
        const int64_t img = (int64_t)n * d.oh;
        auto visit = [&](int oy, int ox, bool ok, uint32_t (&pk)[V / 4], Vec<V>& go, uint32_t& tap) {
            const int oyc = min(max(oy, 0), d.oh - 1), oxc = min(max(ox, 0), d.ow - 1);
            const int64_t po = (img + oyc) * d.ow + oxc;
            amax_load<V>(amax + po * d.c + cg * V, pk);
            go = vload<V>(dout, d.dtype, po * d.ldo + d.cooff + cg * V);
            tap = ok ? (uint32_t)((iy - (oy * d.stride - d.pad)) * d.k + (ix - (ox * d.stride - d.pad))) : 254u;   // 254: never recorded
        };
        if (NW) {
            uint32_t pk[NW ? NW * NW : 1][V / 4], tap[NW ? NW * NW : 1];
            Vec<V> go[NW ? NW * NW : 1];
#pragma unroll
            for (int a = 0; a < NW; ++a)
#pragma unroll
                for (int b = 0; b < NW; ++b) {
                    const int oy = oy_hi - a, ox = ox_hi - b;
                    visit(oy, ox, oy >= oy_lo && ox >= ox_lo, pk[a * NW + b], go[a * NW + b], tap[a * NW + b]);
                }
#pragma unroll
            for (int t = 0; t < NW * NW; ++t)
#pragma unroll
                for (int e = 0; e < V; ++e) g.v[e] += ((pk[t][e / 4] >> (8 * (e & 3))) & 0xff) == tap[t] ? go[t].v[e] : 0.f;
        } else {
            for (int oy = oy_lo; oy <= oy_hi; ++oy)
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    uint32_t pk[V / 4], tap; Vec<V> go;
                    visit(oy, ox, true, pk, go, tap);
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] += ((pk[e / 4] >> (8 * (e & 3))) & 0xff) == tap ? go.v[e] : 0.f;
                }
        }
        const int64_t self_off = p * d.ldi + d.cioff + cg * V;
        if (accumulate) { Vec<V> o = vload<V>(din_, d.dtype, self_off);
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] += o.v[e]; }
        vstore<V>(din_, d.dtype, self_off, g);
    }
}


// k = 3, stride 2, pad 0 (the backbones' only overlapping max-pool): one thread owns a 2x2 block of input pixels.  Its four pixels
// lie in the same (up to) four windows (oy in {a-1, a}, ox in {b-1, b}), so the arg-max bytes and gradients of those windows are
// loaded once and serve four outputs -- a quarter of the L2 traffic of the one-pixel-per-thread form.
template <int V>
__global__ __launch_bounds__(256) void maxpool_bwd_amax_k3s2_kernel(din_pool_desc d, Dec3 dd, const uint8_t* __restrict__ amax,
                                                                    const void* __restrict__ dout, void* __restrict__ din_, int accumulate) {
    const int hb = (d.h + 1) >> 1, wb = (d.w + 1) >> 1;
    const int64_t total = (int64_t)d.nb * hb * wb * (d.c / V);
    DIN_GRID_STRIDE(i, total) {
        int cg, bx, by, n; int64_t pblk;
        decode(dd, i, cg, bx, by, n, pblk);
        uint32_t pk[4][V / 4]; Vec<V> go[4]; bool wok[4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int oy = by - 1 + a, ox = bx - 1 + b;
                wok[a * 2 + b] = oy >= 0 && oy < d.oh && ox >= 0 && ox < d.ow;
                const int oyc = min(max(oy, 0), d.oh - 1), oxc = min(max(ox, 0), d.ow - 1);
                const int64_t po = ((int64_t)n * d.oh + oyc) * d.ow + oxc;
                amax_load<V>(amax + po * d.c + cg * V, pk[a * 2 + b]);
                go[a * 2 + b] = vload<V>(dout, d.dtype, po * d.ldo + d.cooff + cg * V);
            }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int iy = 2 * by + dy, ix = 2 * bx + dx;
                if (iy >= d.h || ix >= d.w) continue;
                Vec<V> g = vzero<V>();
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        // window (by-1+a, bx-1+b) starts at input (2(by-1+a), 2(bx-1+b)): this pixel is its tap (dy + 2(1-a), dx + 2(1-b))
                        const int ty = dy + 2 * (1 - a), tx = dx + 2 * (1 - b);
                        if (ty > 2 || tx > 2) continue;                       // compile-time: pixel outside that window
                        const uint32_t tap = wok[a * 2 + b] ? (uint32_t)(ty * 3 + tx) : 254u;
#pragma unroll
                        for (int e = 0; e < V; ++e) g.v[e] += ((pk[a * 2 + b][e / 4] >> (8 * (e & 3))) & 0xff) == tap ? go[a * 2 + b].v[e] : 0.f;
                    }
                const int64_t self_off = (((int64_t)n * d.h + iy) * d.w + ix) * d.ldi + d.cioff + cg * V;
                if (accumulate) { Vec<V> o = vload<V>(din_, d.dtype, self_off);
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] += o.v[e]; }
                vstore<V>(din_, d.dtype, self_off, g);
            }
    }
}

// Map-free backward (recomputes each window's arg-max): for callers that did not save the map.
__global__ void maxpool_bwd_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ in, const void* __restrict__ dout,
                                   void* __restrict__ din_, int relu_mask, int accumulate) {
    constexpr int V = 4;
    const int64_t total = (int64_t)d.nb * d.h * d.w * (d.c / V);
    DIN_GRID_STRIDE(i, total) {
        int cg, ix, iy, n; int64_t p;
        decode(dd, i, cg, ix, iy, n, p);
        const int64_t self_off = p * d.ldi + d.cioff + cg * V;
        Vec<V> xv = vload<V>(in, d.dtype, self_off);
        Vec<V> g = vzero<V>();
        int oy_hi = (iy + d.pad) / d.stride, ox_hi = (ix + d.pad) / d.stride;
        int oy_lo = (iy + d.pad - d.k + d.stride) / d.stride, ox_lo = (ix + d.pad - d.k + d.stride) / d.stride;
        if (iy + d.pad - d.k + 1 < 0) oy_lo = 0;
        if (ix + d.pad - d.k + 1 < 0) ox_lo = 0;
        if (oy_hi >= d.oh) oy_hi = d.oh - 1;
        if (ox_hi >= d.ow) ox_hi = d.ow - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                bool win[V] = {true, true, true, true};
                for (int r = 0; r < d.k; ++r) {
                    int yy = oy * d.stride - d.pad + r;
                    if (yy < 0 || yy >= d.h) continue;
                    for (int s = 0; s < d.k; ++s) {
                        int xx = ox * d.stride - d.pad + s;
                        if (xx < 0 || xx >= d.w) continue;
                        if (yy == iy && xx == ix) continue;
                        Vec<V> v = vload<V>(in, d.dtype, ((int64_t)(n * d.h + yy) * d.w + xx) * d.ldi + d.cioff + cg * V);
                        bool before = (yy < iy) || (yy == iy && xx < ix);
#pragma unroll
                        for (int e = 0; e < V; ++e) win[e] = win[e] && (before ? v.v[e] < xv.v[e] : v.v[e] <= xv.v[e]);
                    }
                }
                Vec<V> go = vload<V>(dout, d.dtype, ((int64_t)(n * d.oh + oy) * d.ow + ox) * d.ldo + d.cooff + cg * V);
#pragma unroll
                for (int e = 0; e < V; ++e) g.v[e] += win[e] ? go.v[e] : 0.f;
            }
        if (relu_mask) {
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] = xv.v[e] > 0.f ? g.v[e] : 0.f;
        }
        if (accumulate) { Vec<V> o = vload<V>(din_, d.dtype, self_off);
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] += o.v[e]; }
        vstore<V>(din_, d.dtype, self_off, g);
    }
}

// ---- average pool (count_include_pad) -------------------------------------------------------------------------------------------
// BOX3: k = 3, stride 1, pad 1 (the only shape the backbones use): forward and backward are the same zero-padded box filter.
// The box rows are summed top-to-bottom, taps left-to-right (the order of the loop form), so both forms give identical bits.
template <int V>
__device__ __forceinline__ Vec<V> box3(const void* __restrict__ src, int dtype, int64_t img, int h, int w, int y, int x, int ld, int off) {
    Vec<V> t[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int yy = min(max(y + r - 1, 0), h - 1), xx = min(max(x + s - 1, 0), w - 1);
            t[r * 3 + s] = vload<V>(src, dtype, ((img + yy) * w + xx) * ld + off);
        }
    Vec<V> a = vzero<V>();
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const bool ok = y + r - 1 >= 0 && y + r - 1 < h && x + s - 1 >= 0 && x + s - 1 < w;
#pragma unroll
            for (int e = 0; e < V; ++e) a.v[e] += ok ? t[r * 3 + s].v[e] : 0.f;
        }
    return a;
}
// flags: DIN_CONV_BIAS adds bias[c] after the average, DIN_CONV_RELU clamps -- the epilogue of a 1x1 conv that was commuted in
// front of the pool (avgpool(conv1x1(x)) == conv1x1(avgpool(x)): both linear, zero padding maps to zero)
template <int V, bool BOX3>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ in, void* __restrict__ out,
                                                          const float* __restrict__ bias, int flags) {
    const int64_t total = (int64_t)d.nb * d.oh * d.ow * (d.c / V);
    const float inv = 1.f / (float)(d.k * d.k);
    DIN_GRID_STRIDE(i, total) {
        int cg, ox, oy, n; int64_t p;
        decode(dd, i, cg, ox, oy, n, p);
        Vec<V> a;
        if (BOX3) {
            a = box3<V>(in, d.dtype, (int64_t)n * d.h, d.h, d.w, oy, ox, d.ldi, d.cioff + cg * V);
        } else {
            a = vzero<V>();
            for (int r = 0; r < d.k; ++r) {
                int iy = oy * d.stride - d.pad + r;
                if (iy < 0 || iy >= d.h) continue;
                for (int s = 0; s < d.k; ++s) {
                    int ix = ox * d.stride - d.pad + s;
                    if (ix < 0 || ix >= d.w) continue;
                    Vec<V> t = vload<V>(in, d.dtype, ((int64_t)(n * d.h + iy) * d.w + ix) * d.ldi + d.cioff + cg * V);
#pragma unroll
                    for (int e = 0; e < V; ++e) a.v[e] += t.v[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = a.v[e] * inv;
            if (flags & DIN_CONV_BIAS) v += bias[cg * V + e];
            if (flags & DIN_CONV_RELU) v = fmaxf(v, 0.f);
            a.v[e] = v;
        }
        vstore<V>(out, d.dtype, p * d.ldo + d.cooff + cg * V, a);
    }
}
template <int V, bool BOX3>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ dout, void* __restrict__ din_,
                                                          const void* __restrict__ mask, int accumulate) {
    const int64_t total = (int64_t)d.nb * d.h * d.w * (d.c / V);
    const float inv = 1.f / (float)(d.k * d.k);
    DIN_GRID_STRIDE(i, total) {
        int cg, ix, iy, n; int64_t p;
        decode(dd, i, cg, ix, iy, n, p);
        Vec<V> g;
        if (BOX3) {
            // taps r = 0..2 of the loop form visit output rows iy+1, iy, iy-1: mirror the box so the summation order is unchanged
            Vec<V> t[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int yy = min(max(iy + 1 - r, 0), d.oh - 1), xx = min(max(ix + 1 - s, 0), d.ow - 1);
                    t[r * 3 + s] = vload<V>(dout, d.dtype, (((int64_t)n * d.oh + yy) * d.ow + xx) * d.ldo + d.cooff + cg * V);
                }
            g = vzero<V>();
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const bool ok = iy + 1 - r >= 0 && iy + 1 - r < d.oh && ix + 1 - s >= 0 && ix + 1 - s < d.ow;
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] += ok ? t[r * 3 + s].v[e] : 0.f;
                }
        } else {
            g = vzero<V>();
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
                    Vec<V> t = vload<V>(dout, d.dtype, ((int64_t)(n * d.oh + oy) * d.ow + ox) * d.ldo + d.cooff + cg * V);
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] += t.v[e];
                }
            }
        }
        const int64_t off = p * d.ldi + d.cioff + cg * V;
#pragma unroll
        for (int e = 0; e < V; ++e) g.v[e] *= inv;
        if (mask) { Vec<V> y = vload<V>(mask, d.dtype, off);
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] = y.v[e] > 0.f ? g.v[e] : 0.f; }
        if (accumulate) { Vec<V> o = vload<V>(din_, d.dtype, off);
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] += o.v[e]; }
        vstore<V>(din_, d.dtype, off, g);
    }
}

// ---- 3x3 / 1 / pad 1 box filter, column strips ---------------------------------------------------------------------------------------
// One thread owns R consecutive rows of one (frame, column, channel group) and keeps the three taps of rows y0-1 .. y0+R in registers
// (bf16: still packed, 4 registers per tap): a strip costs 3 (R + 2) vector loads for R outputs (3.75 per output at R = 8) instead of
// 9, all issued before the first sum is needed.  Every output is summed in exactly the order of the one-thread-per-output kernels
// (= ATen's avg_pool2d loop: rows top-to-bottom, taps left-to-right; backward mirrored), so the results are bit-identical to them --
// and the commuted branch_pool layers keep the reference's ReLU decisions.
// Forward (epilogue: * 1/9, + bias, ReLU) and backward (the box filter is its own transpose; epilogue: * 1/9, ReLU mask, accumulate)
// share the strip; `src`/`dst` pixel strides and channel offsets come from the caller.
template <int V> struct RawTap;
template <> struct RawTap<8> {                                          // 8 bf16 channels, kept packed
    uint4 r;
    __device__ __forceinline__ void load(const void* base, int, int64_t i) { r = *reinterpret_cast<const uint4*>((const bf16_t*)base + i); }
    __device__ __forceinline__ void keep(bool ok) { r.x = ok ? r.x : 0u; r.y = ok ? r.y : 0u; r.z = ok ? r.z : 0u; r.w = ok ? r.w : 0u; }
    __device__ __forceinline__ float get(int e) const {
        const uint32_t w = e < 2 ? r.x : e < 4 ? r.y : e < 6 ? r.z : r.w;
        return (e & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
    }
};
template <> struct RawTap<4> {                                          // 4 channels, fp32 or bf16: converted at load
    Vec<4> r;
    __device__ __forceinline__ void load(const void* base, int dtype, int64_t i) { r = vload<4>(base, dtype, i); }
    __device__ __forceinline__ void keep(bool ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r.v[e] = ok ? r.v[e] : 0.f;
    }
    __device__ __forceinline__ float get(int e) const { return r.v[e]; }
};

template <int V, int R, bool BWD>
__global__ __launch_bounds__(256) void avgpool3_strip_kernel(int nb, int h, int w, int cgroups, Dec3 dd, int dtype, const void* __restrict__ src,
                                                             int lds_, int soff, void* __restrict__ dst, int ldd, int doff,
                                                             const float* __restrict__ bias, int flags, const void* __restrict__ mask,
                                                             int accumulate) {
    const int strips = (h + R - 1) / R;
    const int64_t total = (int64_t)nb * strips * w * cgroups;
    DIN_GRID_STRIDE(i, total) {
        int cg, x, sy, n; int64_t p;
        decode(dd, i, cg, x, sy, n, p);                                 // dd built with (cgroups, w, strips)
        const int y0 = sy * R;
        float bv[V];                                                    // the bias of this channel group, fetched once per strip
#pragma unroll
        for (int e = 0; e < V; ++e) bv[e] = (!BWD && (flags & DIN_CONV_BIAS)) ? bias[cg * V + e] : 0.f;
        RawTap<V> t[R + 2][3];
#pragma unroll
        for (int r = 0; r < R + 2; ++r) {
            const int yc = min(max(y0 + r - 1, 0), h - 1);
            const int64_t rowp = ((int64_t)n * h + yc) * w;
            t[r][0].load(src, dtype, (rowp + max(x - 1, 0)) * lds_ + soff + cg * V);
            t[r][1].load(src, dtype, (rowp + x) * lds_ + soff + cg * V);
            t[r][2].load(src, dtype, (rowp + min(x + 1, w - 1)) * lds_ + soff + cg * V);
        }
        const bool lok = x > 0, rok = x + 1 < w;
        // taps outside the map become +0.0 ONCE here (4 selects per tap) instead of a select per use (8 channels x 9 taps x R outputs): the sums
        // below still add them, in the same order -- same bits
#pragma unroll
        for (int r = 0; r < R + 2; ++r) {
            const bool yok = y0 + r - 1 >= 0 && y0 + r - 1 < h;
            t[r][0].keep(yok && lok); t[r][1].keep(yok); t[r][2].keep(yok && rok);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = y0 + r;
            if (y >= h) break;
            Vec<V> g = vzero<V>();
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int q = BWD ? 2 - rr : rr;                        // backward visits rows y+1, y, y-1 and columns x+1, x, x-1
#pragma unroll
                for (int ss = 0; ss < 3; ++ss) {
                    const int c = BWD ? 2 - ss : ss;
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] += t[r + q][c].get(e);
                }
            }
            const int64_t off = (((int64_t)n * h + y) * w + x) * ldd + doff + cg * V;
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] *= (1.f / 9.f);
            if (!BWD) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float v = g.v[e];
                    if (flags & DIN_CONV_BIAS) v += bv[e];
                    if (flags & DIN_CONV_RELU) v = fmaxf(v, 0.f);
                    g.v[e] = v;
                }
            } else {
                if (mask) { Vec<V> yv = vload<V>(mask, dtype, off);
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] = yv.v[e] > 0.f ? g.v[e] : 0.f; }
                if (accumulate) { Vec<V> o = vload<V>(dst, dtype, off);
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] += o.v[e]; }
            }
            vstore<V>(dst, dtype, off, g);
        }
    }
}

// ---- 3x3 / stride 2 max pool, column strips (the three max-pools of Inception-v3) ------------------------------------------------------------
// One thread owns R consecutive OUTPUT rows of one (frame, output column, channel group): consecutive windows share a row, so the strip
// needs 2R + 1 input rows x 3 columns = 6.75 loads per output at R = 4 instead of 9, all issued before the first comparison.  Scan order per
// window = the one-thread-per-output kernel's (rows top-to-bottom, taps left-to-right, first maximum wins): bit-identical values and arg-max bytes.
template <int R>
__global__ __launch_bounds__(256) void maxpool3s2_strip_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ in, void* __restrict__ out,
                                                               uint8_t* __restrict__ amax) {
    constexpr int V = 8;
    const int strips = (d.oh + R - 1) / R, cgroups = d.c / V;
    const int64_t total = (int64_t)d.nb * strips * d.ow * cgroups;
    DIN_GRID_STRIDE(i, total) {
        int cg, ox, sy, n; int64_t p;
        decode(dd, i, cg, ox, sy, n, p);                                // dd built with (cgroups, ow, strips)
        const int oy0 = sy * R;
        const int y0 = oy0 * 2 - d.pad, x0 = ox * 2 - d.pad;
        RawTap<V> t[2 * R + 1][3];
#pragma unroll
        for (int r = 0; r < 2 * R + 1; ++r) {
            const int iy = min(max(y0 + r, 0), d.h - 1);
            const int64_t rowp = ((int64_t)n * d.h + iy) * d.w;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ix = min(max(x0 + c, 0), d.w - 1);
                t[r][c].load(in, d.dtype, (rowp + ix) * d.ldi + d.cioff + cg * V);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int oy = oy0 + q;
            if (oy >= d.oh) break;
            float m[V]; int am[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { m[e] = -INFINITY; am[e] = 0; }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int iy = y0 + 2 * q + r, ix = x0 + c;
                    const bool ok = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float v = t[2 * q + r][c].get(e);
                        if (ok && v > m[e]) { m[e] = v; am[e] = r * 3 + c; }
                    }
                }
            Vec<V> mv;
#pragma unroll
            for (int e = 0; e < V; ++e) mv.v[e] = m[e];
            const int64_t po = ((int64_t)n * d.oh + oy) * d.ow + ox;
            vstore<V>(out, d.dtype, po * d.ldo + d.cooff + cg * V, mv);
            if (amax) {
                uint32_t pk[2] = {0u, 0u};
#pragma unroll
                for (int e = 0; e < V; ++e) pk[e >> 2] |= (uint32_t)(m[e] > 0.f ? am[e] : 255) << (8 * (e & 3));
                *reinterpret_cast<uint2*>(amax + po * d.c + cg * V) = uint2{pk[0], pk[1]};
            }
        }
    }
}

// ---- 3x3 / stride 2 / pad 0 max pool, bf16, workgroups inside rows ------------------------------------------------------------------------
// The element-per-thread kernels above are VALU-bound, not HBM-bound, on the stem's big maps (measured: 3.7 - 5 TB/s where a plain
// streaming kernel reaches 6.2): a three-level index decode plus 64-bit address arithmetic per tap (quarter-rate integer multiplies), and
// compare + two selects per element and tap.  These two kernels take both away:
//   * one workgroup = 256 items of one output row (forward) / of one pair of input rows (backward): every row base is a scalar, a thread
//     only derives (x, channel group) from its position in the row -- one multiply-shift; one item per thread, no loop;
//   * forward: a tap's bf16 becomes the fp32 word (bf16 << 16) | (15 - tap), so ONE v_max3_f32 folds two taps into the running maximum
//     and carries the arg-max with it: among equal bf16 values the larger payload = the earlier tap wins (PyTorch's first-maximum rule);
//     for winners <= 0 the payload is not used (the map stores 255 there), and the pooled value is the word's upper half either way.
//     [fp32 denormals are preserved in this library's kernels (float_denorm_mode_32 = 3), so +0 | payload orders as a positive number.]
// Arg-max bytes and gradients are bit for bit those of the kernels above, the pooled values equal as numbers: a window whose maximum is zero
// and which holds both -0.0 and +0.0 pools to +0.0 here and to the first zero in scan order there (its map byte is 255 either way)
// (tests/test_gpu_kernels.py::test_maxpool_row_kernels_are_bit_identical).
__device__ __forceinline__ uint32_t max3_f32_bits(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void maxpool3s2_row_fwd_kernel(din_pool_desc d, FastDiv cgd, const bf16_t* __restrict__ in,
                                                                   bf16_t* __restrict__ out, uint8_t* __restrict__ amax) {
    // one item per thread, no loop: a wave that loops waits for its previous stores before its next loads return (gfx9 counts both in vmcnt)
    const int cgs = d.c >> 3, items = d.ow * cgs, chunks = (items + BLOCK - 1) / BLOCK;
    const int lin = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int row = lin / chunks, chunk = lin - row * chunks;             // row = (n, oy)
    const int n = row / d.oh, oy = row - n * d.oh;
    const char* rp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) rp[r] = (const char*)(in + ((int64_t)(n * d.h + 2 * oy + r) * d.w) * d.ldi + d.cioff);
    char* orow = (char*)(out + (int64_t)row * d.ow * d.ldo + d.cooff);
    uint8_t* arow = amax ? amax + (int64_t)row * d.ow * d.c : nullptr;
    const uint32_t pix = (uint32_t)d.ldi * 2u;                             // bytes per input pixel
    const int j = chunk * BLOCK + (int)threadIdx.x;
    if (j < items) {
        const uint32_t ox = fdiv((uint32_t)j, cgd), cg = (uint32_t)j - ox * (uint32_t)cgs;
        const uint32_t off = ox * 2u * pix + cg * 16u;
        uint4 t[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) t[r][s] = *reinterpret_cast<const uint4*>(rp[r] + (off + (uint32_t)s * pix));
        uint32_t m[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t lo[9], hi[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const uint4& v = t[k / 3][k % 3];
                const uint32_t x = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
                lo[k] = (x << 16) | (uint32_t)(15 - k);
                hi[k] = (x & 0xffff0000u) | (uint32_t)(15 - k);
            }
            uint32_t a = max3_f32_bits(lo[0], lo[1], lo[2]), b = max3_f32_bits(hi[0], hi[1], hi[2]);
#pragma unroll
            for (int k = 3; k < 9; k += 2) { a = max3_f32_bits(a, lo[k], lo[k + 1]); b = max3_f32_bits(b, hi[k], hi[k + 1]); }
            m[2 * q] = a; m[2 * q + 1] = b;
        }
        uint4 o;
        o.x = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u); o.y = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
        o.z = __builtin_amdgcn_perm(m[5], m[4], 0x07060302u); o.w = __builtin_amdgcn_perm(m[7], m[6], 0x07060302u);
        *reinterpret_cast<uint4*>(orow + (ox * (uint32_t)d.ldo * 2u + cg * 16u)) = o;
        if (arow) {
            uint32_t pk[2] = {0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t tap = (int32_t)m[e] > 0xffff ? ((m[e] & 15u) ^ 15u) : 255u;      // bf16 part > 0
                pk[e >> 2] |= tap << (8 * (e & 3));
            }
            *reinterpret_cast<uint2*>(arow + (ox * (uint32_t)d.c + cg * 8u)) = uint2{pk[0], pk[1]};
        }
    }
}

// Backward from the arg-max map: a workgroup owns input rows (2 by, 2 by + 1) of one frame; a thread a 2x2 pixel block x 8 channels, as in
// maxpool_bwd_amax_k3s2_kernel (same window visit order: the fp32 sums round identically).
// One item per thread, no loop (grid = rows x chunks of BLOCK items).  Measured alternatives (tools/pool_bench.py, 64-channel map): a row walk per
// workgroup 814 us, the same walk with the next item's windows requested before the current item's arithmetic and stores (ping-pong registers,
// 98 VGPRs) 869 us, one item per thread 769 us.  Knock-outs of this kernel: stores only 438 us, loads only 344 us, arithmetic only 390 us.
typedef uint32_t pool_u32x4 __attribute__((ext_vector_type(4)));
template <bool ACC> struct PoolWin { uint2 am[4]; uint4 go[4]; uint4 prev[ACC ? 4 : 1]; };
template <int BLOCK, bool ACC>
__global__ __launch_bounds__(BLOCK, ACC ? 5 : 8) void maxpool3s2_row_bwd_kernel(din_pool_desc d, FastDiv cgd, const uint8_t* __restrict__ amax,
                                                                      const bf16_t* __restrict__ dout, bf16_t* __restrict__ din_) {
    const int hb = (d.h + 1) >> 1, wb = (d.w + 1) >> 1;
    const int cgs = d.c >> 3, items = wb * cgs, chunks = (items + BLOCK - 1) / BLOCK;
    const int lin = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int row = lin / chunks, chunk = lin - row * chunks;             // row = (n, by)
    const int n = row / hb, by = row - n * hb;
    const uint8_t* arow[2]; const char* grow[2]; bool rok[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = by - 1 + a;
        rok[a] = oy >= 0 && oy < d.oh;
        const int64_t po = (int64_t)(n * d.oh + min(max(oy, 0), d.oh - 1)) * d.ow;
        arow[a] = amax + po * d.c;
        grow[a] = (const char*)(dout + po * d.ldo + d.cooff);
    }
    // the gradient rows as buffer resources: a pixel beyond the row (odd widths) or a row beyond the image (odd heights) is an out-of-range
    // offset / an empty resource, and the hardware drops the store -- every thread issues the same four stores, no branch around them
    char* irow[2]; __amdgpu_buffer_rsrc_t irs[2];
    const int row_bytes = d.w * d.ldi * 2 - d.cioff * 2;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int iy = 2 * by + dy;
        irow[dy] = (char*)(din_ + ((int64_t)(n * d.h + min(iy, d.h - 1)) * d.w) * d.ldi + d.cioff);
        irs[dy] = __builtin_amdgcn_make_buffer_rsrc(irow[dy], 0, iy < d.h ? row_bytes : 0, 0x00020000);
    }
    const uint32_t ipix = (uint32_t)d.ldi * 2u, opix = (uint32_t)d.ldo * 2u;
    auto fetch = [&](int j, PoolWin<ACC>& w) {                                         // raw loads only: nothing here waits for them
        const uint32_t bx = fdiv((uint32_t)j, cgd), cg = (uint32_t)j - bx * (uint32_t)cgs;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t oxc = (uint32_t)min(max((int)bx - 1 + b, 0), d.ow - 1);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                w.am[a * 2 + b] = *reinterpret_cast<const uint2*>(arow[a] + (oxc * (uint32_t)d.c + cg * 8u));
                w.go[a * 2 + b] = *reinterpret_cast<const uint4*>(grow[a] + (oxc * opix + cg * 16u));
            }
        }
        if (ACC) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                w.prev[p] = *reinterpret_cast<const uint4*>(irow[p >> 1] + (min(2u * bx + (uint32_t)(p & 1), (uint32_t)d.w - 1u) * ipix + cg * 16u));
        }
    };
    auto finish = [&](int j, const PoolWin<ACC>& w) {
        const uint32_t bx = fdiv((uint32_t)j, cgd), cg = (uint32_t)j - bx * (uint32_t)cgs;
        uint32_t pk[4][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = (int)bx - 1 + b;
            const bool cok = ox >= 0 && ox < d.ow;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bool ok = cok && rok[a];
                pk[a * 2 + b][0] = ok ? w.am[a * 2 + b].x : 0xfefefefeu;          // 254: a tap no window records
                pk[a * 2 + b][1] = ok ? w.am[a * 2 + b].y : 0xfefefefeu;
            }
        }
        // two channels (one gradient dword) at a time across the four pixels: few live registers, no divergent control flow before the stores
        uint32_t od[4][4];                                                        // [pixel dy * 2 + dx][dword]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float f[4][2];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t x = q == 0 ? w.go[v].x : q == 1 ? w.go[v].y : q == 2 ? w.go[v].z : w.go[v].w;
                f[v][0] = __uint_as_float(x << 16); f[v][1] = __uint_as_float(x & 0xffff0000u);
            }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float g0 = 0.f, g1 = 0.f;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int ty = dy + 2 * (1 - a), tx = dx + 2 * (1 - b);   // this pixel's tap in window (by - 1 + a, bx - 1 + b)
                            if (ty > 2 || tx > 2) continue;                           // compile time
                            const uint32_t tap = (uint32_t)(ty * 3 + tx), word = pk[a * 2 + b][q >> 1];
                            g0 += ((word >> (16 * (q & 1))) & 0xffu) == tap ? f[a * 2 + b][0] : 0.f;
                            g1 += ((word >> (16 * (q & 1) + 8)) & 0xffu) == tap ? f[a * 2 + b][1] : 0.f;
                        }
                    if (ACC) {
                        const uint4& o = w.prev[ACC ? dy * 2 + dx : 0];
                        const uint32_t x = q == 0 ? o.x : q == 1 ? o.y : q == 2 ? o.z : o.w;
                        g0 += __uint_as_float(x << 16); g1 += __uint_as_float(x & 0xffff0000u);
                    }
                    od[dy * 2 + dx][q] = pack_bf16x2(g0, g1);
                }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_raw_buffer_store_b128(pool_u32x4{od[p][0], od[p][1], od[p][2], od[p][3]}, irs[p >> 1],
                                                   (int)((2u * bx + (uint32_t)(p & 1)) * ipix + cg * 16u), 0, 0);
    };
    const int j = chunk * BLOCK + (int)threadIdx.x;
    if (j < items) { PoolWin<ACC> w; fetch(j, w); finish(j, w); }
}

// ---- bilinear resize, align_corners=True (infer_model.py:169): src = dst*(in-1)/(out-1) -----------------------------------------
__device__ __forceinline__ void bil_coord(int o, int in, int out, int& i0, int& i1, float& l) {
    float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    float src = sc * (float)o;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + 1 < in ? i0 + 1 : in - 1;
    l = src - (float)i0;
}
template <int V>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ in, void* __restrict__ out) {
    const int64_t total = (int64_t)d.nb * d.oh * d.ow * (d.c / V);
    DIN_GRID_STRIDE(i, total) {
        int cg, ox, oy, n; int64_t p;
        decode(dd, i, cg, ox, oy, n, p);
        int y0, y1, x0, x1; float ly, lx;
        bil_coord(oy, d.h, d.oh, y0, y1, ly);
        bil_coord(ox, d.w, d.ow, x0, x1, lx);
        auto at = [&](int y, int x) { return vload<V>(in, d.dtype, ((int64_t)(n * d.h + y) * d.w + x) * d.ldi + d.cioff + cg * V); };
        const Vec<V> a = at(y0, x0), b = at(y0, x1), c = at(y1, x0), e2 = at(y1, x1);
        Vec<V> o;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float top = a.v[e] * (1.f - lx) + b.v[e] * lx;
            const float bot = c.v[e] * (1.f - lx) + e2.v[e] * lx;
            o.v[e] = top * (1.f - ly) + bot * ly;
        }
        vstore<V>(out, d.dtype, p * d.ldo + d.cooff + cg * V, o);
    }
}
// Up-sampling form of the forward (out >= in on both axes, the multiscale fuse: 43x78 -> 87x157): one thread per SOURCE cell
// (y0, x0, channel group).  It loads the cell's four corners once and produces every output pixel whose source coordinate falls into
// the cell (about (oh/h) x (ow/w) = 4 of them), so a corner is read 4 times from L2 instead of ~16 and every load feeds ~4 stores.
// The outputs of a cell are found with the SAME expression bil_coord uses ((int)(sc * o) == y0), so the result is bit-identical to the
// per-output kernel.
template <int V>
__global__ __launch_bounds__(256) void bilinear_fwd_cells_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ in, void* __restrict__ out) {
    const int64_t total = (int64_t)d.nb * d.h * d.w * (d.c / V);
    const float scy = d.oh > 1 ? (float)(d.h - 1) / (float)(d.oh - 1) : 0.f, scx = d.ow > 1 ? (float)(d.w - 1) / (float)(d.ow - 1) : 0.f;
    DIN_GRID_STRIDE(i, total) {
        int cg, x0, y0, n; int64_t p;
        decode(dd, i, cg, x0, y0, n, p);
        const int y1 = y0 + 1 < d.h ? y0 + 1 : d.h - 1, x1 = x0 + 1 < d.w ? x0 + 1 : d.w - 1;
        auto at = [&](int y, int x) { return vload<V>(in, d.dtype, ((int64_t)(n * d.h + y) * d.w + x) * d.ldi + d.cioff + cg * V); };
        const Vec<V> a = at(y0, x0), b = at(y0, x1), c = at(y1, x0), e2 = at(y1, x1);
        // candidate outputs: o with (int)(sc * o) == cell; start one below the estimate, stop at the first one beyond
        int oy = scy > 0.f ? (int)((float)y0 / scy) - 1 : 0, ox_first = scx > 0.f ? (int)((float)x0 / scx) - 1 : 0;
        if (oy < 0) oy = 0;
        if (ox_first < 0) ox_first = 0;
        for (; oy < d.oh; ++oy) {
            int ty0, ty1; float ly;
            bil_coord(oy, d.h, d.oh, ty0, ty1, ly);
            if (ty0 < y0) continue;
            if (ty0 > y0) break;
            for (int ox = ox_first; ox < d.ow; ++ox) {
                int tx0, tx1; float lx;
                bil_coord(ox, d.w, d.ow, tx0, tx1, lx);
                if (tx0 < x0) continue;
                if (tx0 > x0) break;
                Vec<V> o;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float top = a.v[e] * (1.f - lx) + b.v[e] * lx;
                    const float bot = c.v[e] * (1.f - lx) + e2.v[e] * lx;
                    o.v[e] = top * (1.f - ly) + bot * ly;
                }
                vstore<V>(out, d.dtype, ((int64_t)(n * d.oh + oy) * d.ow + ox) * d.ldo + d.cooff + cg * V, o);
            }
        }
    }
}
// gather-form backward: each input cell sums the contributions of the output cells whose 2x2 footprint touches it.  Because the map
// is monotone, candidate outputs for input row y are those with source coordinate in (y-1, y+1): a short output range per axis.
template <int V>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ dout, void* __restrict__ din_,
                                                           const void* __restrict__ mask, int accumulate) {
    const int64_t total = (int64_t)d.nb * d.h * d.w * (d.c / V);
    const float scy = d.oh > 1 ? (float)(d.h - 1) / (float)(d.oh - 1) : 0.f;
    const float scx = d.ow > 1 ? (float)(d.w - 1) / (float)(d.ow - 1) : 0.f;
    DIN_GRID_STRIDE(i, total) {
        int cg, ix, iy, n; int64_t p;
        decode(dd, i, cg, ix, iy, n, p);
        int oy_lo = scy > 0.f ? (int)floorf((float)(iy - 1) / scy) : 0, oy_hi = scy > 0.f ? (int)ceilf((float)(iy + 1) / scy) : d.oh - 1;
        int ox_lo = scx > 0.f ? (int)floorf((float)(ix - 1) / scx) : 0, ox_hi = scx > 0.f ? (int)ceilf((float)(ix + 1) / scx) : d.ow - 1;
        if (oy_lo < 0) oy_lo = 0;
        if (ox_lo < 0) ox_lo = 0;
        if (oy_hi > d.oh - 1) oy_hi = d.oh - 1;
        if (ox_hi > d.ow - 1) ox_hi = d.ow - 1;
        Vec<V> g = vzero<V>();
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
                Vec<V> t = vload<V>(dout, d.dtype, ((int64_t)(n * d.oh + oy) * d.ow + ox) * d.ldo + d.cooff + cg * V);
                const float wgt = wy * wx;
#pragma unroll
                for (int e = 0; e < V; ++e) g.v[e] += t.v[e] * wgt;
            }
        }
        const int64_t off = p * d.ldi + d.cioff + cg * V;
        if (mask) { Vec<V> y = vload<V>(mask, d.dtype, off);
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] = y.v[e] > 0.f ? g.v[e] : 0.f; }
        if (accumulate) { Vec<V> o = vload<V>(din_, d.dtype, off);
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] += o.v[e]; }
        vstore<V>(din_, d.dtype, off, g);
    }
}

// Gather-form backward with the candidate scan hoisted: the <= MAXC output rows / columns whose 2x2 footprint can touch input row iy /
// column ix get their weights computed ONCE per axis (bil_coord divides; the nested form above recomputes it MAXC^2 times), then only the
// non-zero (row, column) pairs are loaded.  Same expressions, same summation order (rows outer, columns inner) -> bit-identical.
template <int V, int MAXC>
__global__ __launch_bounds__(256) void bilinear_bwd_hoisted_kernel(din_pool_desc d, Dec3 dd, const void* __restrict__ dout, void* __restrict__ din_,
                                                                   const void* __restrict__ mask, int accumulate) {
    const int64_t total = (int64_t)d.nb * d.h * d.w * (d.c / V);
    const float scy = d.oh > 1 ? (float)(d.h - 1) / (float)(d.oh - 1) : 0.f;
    const float scx = d.ow > 1 ? (float)(d.w - 1) / (float)(d.ow - 1) : 0.f;
    DIN_GRID_STRIDE(i, total) {
        int cg, ix, iy, n; int64_t p;
        decode(dd, i, cg, ix, iy, n, p);
        int oy_lo = scy > 0.f ? (int)floorf((float)(iy - 1) / scy) : 0, oy_hi = scy > 0.f ? (int)ceilf((float)(iy + 1) / scy) : d.oh - 1;
        int ox_lo = scx > 0.f ? (int)floorf((float)(ix - 1) / scx) : 0, ox_hi = scx > 0.f ? (int)ceilf((float)(ix + 1) / scx) : d.ow - 1;
        if (oy_lo < 0) oy_lo = 0;
        if (ox_lo < 0) ox_lo = 0;
        if (oy_hi > d.oh - 1) oy_hi = d.oh - 1;
        if (ox_hi > d.ow - 1) ox_hi = d.ow - 1;
        float wys[MAXC], wxs[MAXC];
#pragma unroll
        for (int a = 0; a < MAXC; ++a) {
            const int oy = oy_lo + a, ox = ox_lo + a;
            int i0, i1; float l;
            wys[a] = 0.f; wxs[a] = 0.f;
            if (oy <= oy_hi) { bil_coord(oy, d.h, d.oh, i0, i1, l); wys[a] = (i0 == iy ? 1.f - l : 0.f) + (i1 == iy ? l : 0.f); }
            if (ox <= ox_hi) { bil_coord(ox, d.w, d.ow, i0, i1, l); wxs[a] = (i0 == ix ? 1.f - l : 0.f) + (i1 == ix ? l : 0.f); }
        }
        Vec<V> g = vzero<V>();
#pragma unroll
        for (int a = 0; a < MAXC; ++a) {
            if (wys[a] == 0.f) continue;
#pragma unroll
            for (int b = 0; b < MAXC; ++b) {
                if (wxs[b] == 0.f) continue;
                Vec<V> t = vload<V>(dout, d.dtype, ((int64_t)(n * d.oh + oy_lo + a) * d.ow + ox_lo + b) * d.ldo + d.cooff + cg * V);
                const float wgt = wys[a] * wxs[b];
#pragma unroll
                for (int e = 0; e < V; ++e) g.v[e] += t.v[e] * wgt;
            }
        }
        const int64_t off = p * d.ldi + d.cioff + cg * V;
        if (mask) { Vec<V> y = vload<V>(mask, d.dtype, off);
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] = y.v[e] > 0.f ? g.v[e] : 0.f; }
        if (accumulate) { Vec<V> o = vload<V>(din_, d.dtype, off);
#pragma unroll
            for (int e = 0; e < V; ++e) g.v[e] += o.v[e]; }
        vstore<V>(din_, d.dtype, off, g);
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
inline int avgpool_strip_rows() { const char* e = DIN_OPT("DIN_AVGPOOL_STRIP"); return e ? atoi(e) : 1; }   // 0: one thread per output (round 1)
inline bool is_box3(const din_pool_desc* d) { return d->k == 3 && d->stride == 1 && d->pad == 1 && d->oh == d->h && d->ow == d->w; }
inline int pool_grid_cap() { const char* e = DIN_OPT("DIN_POOL_GRID_CAP"); return e ? atoi(e) : 32768; }
inline bool maxpool_rows() { const char* e = DIN_OPT("DIN_MAXPOOL_ROWS"); return e ? atoi(e) != 0 : true; }

}  // namespace

#define POOL_LAUNCH(kern, total, ...) hipLaunchKernelGGL(kern, dim3(grid_1d(total, 256, pool_grid_cap())), dim3(256), 0, as_stream(stream), __VA_ARGS__)

extern "C" {

int din_maxpool_fwd(const din_pool_desc* d, const void* in, void* out, uint8_t* argmax, void* stream) {
    if (int e = check_pool(d, "maxpool_fwd")) return e;
    DIN_REQUIRE(in && out, "maxpool_fwd: null pointer");
    DIN_REQUIRE(d->k * d->k < 254, "maxpool_fwd: window too large for the byte arg-max map");
    const int v = wide8(d) ? 8 : 4;
    const int64_t total = (int64_t)d->nb * d->oh * d->ow * (d->c / v);
    const Dec3 dd = make_dec(d->c / v, d->ow, d->oh, total);
    // column strips for the wide maps (measured, tools/pool_bench.py: 192 ch 570 -> 537 us, 288 ch 285 -> 201 us; 64 ch 864 -> 1027 us: not there);
    // DIN_MAXPOOL_STRIP=0 / 2: never / always
    const char* ms = DIN_OPT("DIN_MAXPOOL_STRIP");
    const int strip_mode = ms ? atoi(ms) : 1;
    // row kernels (scalar row bases, max3 with the tap in the low mantissa bits): tools/pool_bench.py; DIN_MAXPOOL_ROWS=0: the kernels below
    const bool strip = v == 8 && d->k == 3 && d->stride == 2 && strip_mode != 0 && (d->c >= 128 || strip_mode == 2);
    // measured (same box, us): 64 ch 851 -> 653, 192 ch 539 (strip) -> 502, 288 ch 207 (strip) -> 182
    if (v == 8 && d->k == 3 && d->stride == 2 && d->pad == 0 && 2 * d->oh + 1 <= d->h && 2 * d->ow + 1 <= d->w && maxpool_rows()) {
        const int items = d->ow * (d->c / 8);
        hipLaunchKernelGGL(maxpool3s2_row_fwd_kernel<256>, dim3(d->nb * d->oh * ((items + 255) / 256)), dim3(256), 0, as_stream(stream), *d, make_fastdiv(d->c / 8),
                               (const bf16_t*)in, (bf16_t*)out, argmax);
        DIN_CHECK_LAUNCH("maxpool_fwd");
        return DIN_OK;
    }
    if (strip) {
        constexpr int R = 4;
        const int strips = (d->oh + R - 1) / R;
        const int64_t tot = (int64_t)d->nb * strips * d->ow * (d->c / v);
        const Dec3 ds = make_dec(d->c / v, d->ow, strips, tot);
        POOL_LAUNCH((maxpool3s2_strip_kernel<R>), tot, *d, ds, in, out, argmax);
        DIN_CHECK_LAUNCH("maxpool_fwd");
        return DIN_OK;
    }
    if (v == 8) {
        if (d->k == 3) POOL_LAUNCH((maxpool_fwd_kernel<8, 3>), total, *d, dd, in, out, argmax);
        else if (d->k == 2) POOL_LAUNCH((maxpool_fwd_kernel<8, 2>), total, *d, dd, in, out, argmax);
        else POOL_LAUNCH((maxpool_fwd_kernel<8, 0>), total, *d, dd, in, out, argmax);
    } else {
        if (d->k == 3) POOL_LAUNCH((maxpool_fwd_kernel<4, 3>), total, *d, dd, in, out, argmax);
        else if (d->k == 2) POOL_LAUNCH((maxpool_fwd_kernel<4, 2>), total, *d, dd, in, out, argmax);
        else POOL_LAUNCH((maxpool_fwd_kernel<4, 0>), total, *d, dd, in, out, argmax);
    }
    DIN_CHECK_LAUNCH("maxpool_fwd");
    return DIN_OK;
}
int din_maxpool_bwd(const din_pool_desc* d, const void* in, const uint8_t* argmax, const void* dout, void* din_, int relu_mask,
                    int accumulate, void* stream) {
    if (int e = check_pool(d, "maxpool_bwd")) return e;
    DIN_REQUIRE((in || argmax) && dout && din_, "maxpool_bwd: null pointer");
    DIN_REQUIRE(d->stride >= 1 && d->k >= 1, "maxpool_bwd: bad window");
    if (argmax) {
        DIN_REQUIRE(relu_mask, "maxpool_bwd: the arg-max map encodes the fused ReLU mask; relu_mask must be set");
        const int v = wide8(d) ? 8 : 4;
        if (d->k == 3 && d->stride == 2 && d->pad == 0) {
            const int hb = (d->h + 1) / 2, wb = (d->w + 1) / 2;
            if (v == 8 && maxpool_rows()) {
                        const int items = wb * (d->c / 8);
                const FastDiv cgd = make_fastdiv(d->c / 8);
                const bf16_t* go = (const bf16_t*)dout; bf16_t* gi = (bf16_t*)din_;
                const dim3 grid(d->nb * hb * ((items + 255) / 256));
                if (accumulate) hipLaunchKernelGGL((maxpool3s2_row_bwd_kernel<256, true>), grid, dim3(256), 0, as_stream(stream), *d, cgd, argmax, go, gi);
                else hipLaunchKernelGGL((maxpool3s2_row_bwd_kernel<256, false>), grid, dim3(256), 0, as_stream(stream), *d, cgd, argmax, go, gi);
                DIN_CHECK_LAUNCH("maxpool_bwd");
                return DIN_OK;
            }
            const int64_t totalb = (int64_t)d->nb * hb * wb * (d->c / v);
            const Dec3 ddb = make_dec(d->c / v, wb, hb, totalb);
            if (v == 8) POOL_LAUNCH((maxpool_bwd_amax_k3s2_kernel<8>), totalb, *d, ddb, argmax, dout, din_, accumulate);
            else POOL_LAUNCH((maxpool_bwd_amax_k3s2_kernel<4>), totalb, *d, ddb, argmax, dout, din_, accumulate);
            DIN_CHECK_LAUNCH("maxpool_bwd");
            return DIN_OK;
        }
        const int64_t total = (int64_t)d->nb * d->h * d->w * (d->c / v);
        const Dec3 dd = make_dec(d->c / v, d->w, d->h, total);
        const int nw = (d->k + d->stride - 1) / d->stride;       // windows per axis that can contain one input element
        if (v == 8) {
            if (nw == 2) POOL_LAUNCH((maxpool_bwd_amax_kernel<8, 2>), total, *d, dd, argmax, dout, din_, accumulate);
            else if (nw == 1) POOL_LAUNCH((maxpool_bwd_amax_kernel<8, 1>), total, *d, dd, argmax, dout, din_, accumulate);
            else POOL_LAUNCH((maxpool_bwd_amax_kernel<8, 0>), total, *d, dd, argmax, dout, din_, accumulate);
        } else {
            if (nw == 2) POOL_LAUNCH((maxpool_bwd_amax_kernel<4, 2>), total, *d, dd, argmax, dout, din_, accumulate);
            else if (nw == 1) POOL_LAUNCH((maxpool_bwd_amax_kernel<4, 1>), total, *d, dd, argmax, dout, din_, accumulate);
            else POOL_LAUNCH((maxpool_bwd_amax_kernel<4, 0>), total, *d, dd, argmax, dout, din_, accumulate);
        }
    } else {
        const int64_t total = (int64_t)d->nb * d->h * d->w * (d->c / 4);
        const Dec3 dd = make_dec(d->c / 4, d->w, d->h, total);
        POOL_LAUNCH(maxpool_bwd_kernel, total, *d, dd, in, dout, din_, relu_mask, accumulate);
    }
    DIN_CHECK_LAUNCH("maxpool_bwd");
    return DIN_OK;
}
int din_avgpool_fwd(const din_pool_desc* d, const void* in, void* out, const float* bias, int flags, void* stream) {
    if (int e = check_pool(d, "avgpool_fwd")) return e;
    DIN_REQUIRE(in && out, "avgpool_fwd: null pointer");
    DIN_REQUIRE(!(flags & ~(DIN_CONV_BIAS | DIN_CONV_RELU)), "avgpool_fwd: only BIAS / RELU flags");
    DIN_REQUIRE(!(flags & DIN_CONV_BIAS) || bias, "avgpool_fwd: BIAS flag without bias");
    const int v = wide8(d) ? 8 : 4;
    const int64_t total = (int64_t)d->nb * d->oh * d->ow * (d->c / v);
    const Dec3 dd = make_dec(d->c / v, d->ow, d->oh, total);
    const bool b3 = is_box3(d);
    if (b3 && avgpool_strip_rows() > 0) {
        constexpr int R = 8;
        const int strips = (d->h + R - 1) / R;
        const int64_t tot = (int64_t)d->nb * strips * d->w * (d->c / v);
        const Dec3 ds = make_dec(d->c / v, d->w, strips, tot);
        if (v == 8) POOL_LAUNCH((avgpool3_strip_kernel<8, R, false>), tot, d->nb, d->h, d->w, d->c / v, ds, d->dtype, in, d->ldi, d->cioff, out, d->ldo, d->cooff, bias, flags, (const void*)nullptr, 0);
        else POOL_LAUNCH((avgpool3_strip_kernel<4, R, false>), tot, d->nb, d->h, d->w, d->c / v, ds, d->dtype, in, d->ldi, d->cioff, out, d->ldo, d->cooff, bias, flags, (const void*)nullptr, 0);
        DIN_CHECK_LAUNCH("avgpool_fwd");
        return DIN_OK;
    }
    if (v == 8) { if (b3) POOL_LAUNCH((avgpool_fwd_kernel<8, true>), total, *d, dd, in, out, bias, flags); else POOL_LAUNCH((avgpool_fwd_kernel<8, false>), total, *d, dd, in, out, bias, flags); }
    else { if (b3) POOL_LAUNCH((avgpool_fwd_kernel<4, true>), total, *d, dd, in, out, bias, flags); else POOL_LAUNCH((avgpool_fwd_kernel<4, false>), total, *d, dd, in, out, bias, flags); }
    DIN_CHECK_LAUNCH("avgpool_fwd");
    return DIN_OK;
}
int din_avgpool_bwd(const din_pool_desc* d, const void* dout, void* din_, const void* mask, int accumulate, void* stream) {
    if (int e = check_pool(d, "avgpool_bwd")) return e;
    DIN_REQUIRE(dout && din_, "avgpool_bwd: null pointer");
    DIN_REQUIRE(d->stride >= 1, "avgpool_bwd: bad stride");
    const int v = wide8(d) ? 8 : 4;
    const int64_t total = (int64_t)d->nb * d->h * d->w * (d->c / v);
    const Dec3 dd = make_dec(d->c / v, d->w, d->h, total);
    const bool b3 = is_box3(d);
    if (b3 && avgpool_strip_rows() > 0) {                               // the box filter is symmetric: same strip, source = dout view, destination = din view
        constexpr int R = 8;
        const int strips = (d->h + R - 1) / R;
        const int64_t tot = (int64_t)d->nb * strips * d->w * (d->c / v);
        const Dec3 ds = make_dec(d->c / v, d->w, strips, tot);
        if (v == 8) POOL_LAUNCH((avgpool3_strip_kernel<8, R, true>), tot, d->nb, d->h, d->w, d->c / v, ds, d->dtype, dout, d->ldo, d->cooff, din_, d->ldi, d->cioff, (const float*)nullptr, 0, mask, accumulate);
        else POOL_LAUNCH((avgpool3_strip_kernel<4, R, true>), tot, d->nb, d->h, d->w, d->c / v, ds, d->dtype, dout, d->ldo, d->cooff, din_, d->ldi, d->cioff, (const float*)nullptr, 0, mask, accumulate);
        DIN_CHECK_LAUNCH("avgpool_bwd");
        return DIN_OK;
    }
    if (v == 8) { if (b3) POOL_LAUNCH((avgpool_bwd_kernel<8, true>), total, *d, dd, dout, din_, mask, accumulate); else POOL_LAUNCH((avgpool_bwd_kernel<8, false>), total, *d, dd, dout, din_, mask, accumulate); }
    else { if (b3) POOL_LAUNCH((avgpool_bwd_kernel<4, true>), total, *d, dd, dout, din_, mask, accumulate); else POOL_LAUNCH((avgpool_bwd_kernel<4, false>), total, *d, dd, dout, din_, mask, accumulate); }
    DIN_CHECK_LAUNCH("avgpool_bwd");
    return DIN_OK;
}
int din_bilinear_fwd(const din_pool_desc* d, const void* in, void* out, void* stream) {
    if (int e = check_pool(d, "bilinear_fwd")) return e;
    DIN_REQUIRE(in && out, "bilinear_fwd: null pointer");
    const int v = wide8(d) ? 8 : 4;
    const char* ce = DIN_OPT("DIN_BILINEAR_CELLS");
    const int cells_env = ce ? atoi(ce) : 1;
    if (cells_env && d->oh >= d->h && d->ow >= d->w && d->h > 1 && d->w > 1) {          // up-sampling: one thread per source cell
        const int64_t total = (int64_t)d->nb * d->h * d->w * (d->c / v);
        const Dec3 dd = make_dec(d->c / v, d->w, d->h, total);
        if (v == 8) POOL_LAUNCH(bilinear_fwd_cells_kernel<8>, total, *d, dd, in, out);
        else POOL_LAUNCH(bilinear_fwd_cells_kernel<4>, total, *d, dd, in, out);
        DIN_CHECK_LAUNCH("bilinear_fwd");
        return DIN_OK;
    }
    const int64_t total = (int64_t)d->nb * d->oh * d->ow * (d->c / v);
    const Dec3 dd = make_dec(d->c / v, d->ow, d->oh, total);
    if (v == 8) POOL_LAUNCH(bilinear_fwd_kernel<8>, total, *d, dd, in, out);
    else POOL_LAUNCH(bilinear_fwd_kernel<4>, total, *d, dd, in, out);
    DIN_CHECK_LAUNCH("bilinear_fwd");
    return DIN_OK;
}
int din_bilinear_bwd(const din_pool_desc* d, const void* dout, void* din_, const void* mask, int accumulate, void* stream) {
    if (int e = check_pool(d, "bilinear_bwd")) return e;
    DIN_REQUIRE(dout && din_, "bilinear_bwd: null pointer");
    const int v = wide8(d) ? 8 : 4;
    const int64_t total = (int64_t)d->nb * d->h * d->w * (d->c / v);
    const Dec3 dd = make_dec(d->c / v, d->w, d->h, total);
    // candidate window per axis: outputs with source coordinate in (i - 1, i + 1) -> at most 2 / scale + 3 of them
    const float scy = d->oh > 1 ? (float)(d->h - 1) / (float)(d->oh - 1) : 0.f, scx = d->ow > 1 ? (float)(d->w - 1) / (float)(d->ow - 1) : 0.f;
    const char* he = DIN_OPT("DIN_BILINEAR_HOIST");
    const bool hoist = (he ? atoi(he) != 0 : true) && scy > 0.f && scx > 0.f && 2.f / scy + 3.f <= 8.f && 2.f / scx + 3.f <= 8.f;
    if (hoist) {
        if (v == 8) POOL_LAUNCH((bilinear_bwd_hoisted_kernel<8, 8>), total, *d, dd, dout, din_, mask, accumulate);
        else POOL_LAUNCH((bilinear_bwd_hoisted_kernel<4, 8>), total, *d, dd, dout, din_, mask, accumulate);
    }
    else if (v == 8) POOL_LAUNCH(bilinear_bwd_kernel<8>, total, *d, dd, dout, din_, mask, accumulate);
    else POOL_LAUNCH(bilinear_bwd_kernel<4>, total, *d, dd, dout, din_, mask, accumulate);
    DIN_CHECK_LAUNCH("bilinear_bwd");
    return DIN_OK;
}

}  // extern "C"
