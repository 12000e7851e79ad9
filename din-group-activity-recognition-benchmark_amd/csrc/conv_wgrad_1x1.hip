// Weight gradients of the 1x1 convs that read ONE tensor (the block-entry convs of an InceptionA block: branch1x1, branch5x5_1 +
// branch3x3dbl_1, branch_pool; backbone/backbone.py:44-58 through torchvision) in ONE launch.
//
// Each of these gradients is a [cout_b x Cin] contraction over millions of pixels: memory-bound, and run one by one they read the block
// input (192-288 channels, 0.5-0.75 GB for 96 frames) three times.  Here the input tile is brought in once, the gradient operands of all
// sources next to it, and -- as in conv_wgrad_halo.hip -- persistent workgroups keep their block of dW (all sources' rows x Cin, fp32) in
// registers across all their tiles: HBM traffic = every tensor once.
//   * the sources are dealt to two workgroup classes (blockIdx.y; <= 128 filter rows = 8 row tiles each, host-chosen); both classes read
//     the same input tile (second read from L2);
//   * 64-pixel stages, a 3-slot LDS ring filled by LDS-DMA two stages ahead (counted vmcnt), 16 waves as 4 row groups x 4 column groups:
//     wave (rg, cg) owns row tiles 2 rg, 2 rg + 1 of its class and the 16-column tiles cg, cg + 4, ... of the input channels;
//   * the G image of a stage is the concatenation of per-source [pixel][cout_b] images (each its own LDS-DMA stream and its own swizzle);
//   * k (pixel) order of a 32-pixel k-step as in conv_wgrad_small_kernel (read rd, lane group g4, sub-row q -> pixel 16 rd + 4 g4 + q):
//     every 32-lane half of a ds_read_b64_tr_b16 covers 8 consecutive pixels, conflict-free with the pitch-specific unit swizzles;
//   * result: one fp32 slab per persistent workgroup index, [rows_pad][Cin]; the host reduces each source's rows with
//     conv_wgrad_reduce_kernel (BatchNorm scale, <W, dW> dot) into that layer's dW.
#include "din_common.h"
#include "conv_wgrad.h"
#include <unordered_map>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace din_wgrad {
namespace {

__device__ __forceinline__ u32x2 tr_read(uint32_t byte_addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(byte_addr) : "memory");
    return r;
}

// XOR for the chunk-PAIR index (one pair = 32 bytes = one 16-channel column / row tile) of pixel px, by the pixel pitch in 16-byte chunks:
// the eight consecutive pixels a 32-lane half reads must land in eight different 32-byte bank groups of the 256-byte bank row.
//   pitch mod 256 B == 0 (16, 32 chunks): all eight collide -> px & 7;   == 128 (8, 24): pairs collide four ways -> (px >> 1) & 3;
//   == 64 / 192 (4, 12, 20, 36): two ways -> (px >> 2) & 1;   == 32 / 96 / 160 / 224 (2, 6, 10, 14, ...): none.
__device__ __forceinline__ int pair_swz(int chunks, int px) {
    const int m = chunks & 15;
    if (m == 0) return px & 7;
    if (m == 8) return (px >> 1) & 3;
    if (m == 4 || m == 12) return (px >> 2) & 1;
    return 0;
}

constexpr int NW = 16, SPX = 64, NS = 3;

// CPP: 16-byte chunks per input pixel (Cin / 8)
template <int CPP>
__global__ __launch_bounds__(1024, 1) void conv_wgrad_1x1_multi_kernel(Wg1x1K p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NCT = CPP / 2, TJ = (NCT + 3) / 4, TI = 2;
    constexpr int XBYTES = SPX * CPP * 16, NSLOT_X = XBYTES / 1024, NTR_X = (NSLOT_X + NW - 1) / NW;
    static_assert(CPP % 2 == 0 && XBYTES % 1024 == 0 && NTR_X <= 3, "whole column tiles / whole 1 KiB transfers / vmcnt cases");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cls = blockIdx.y, rg = wid >> 2, cg = wid & 3;
    const int gbytes = p.cls_gbytes[cls];                               // G image of a stage for this class (whole KiB)
    const int stage = XBYTES + p.gbytes_max;                            // (both classes use the same slot size)
    const int nslot_g = gbytes >> 10;

    // ---- DMA plans -----------------------------------------------------------------------------------------------------------------
    int relX[NTR_X];
#pragma unroll
    for (int i = 0; i < NTR_X; ++i) {
        const int id = (wid + NW * i) * 64 + lane;
        const int px = id / CPP, slot = id - px * CPP;
        const int cc = slot ^ (pair_swz(CPP, px) * 2);
        relX[i] = id < SPX * CPP ? px * p.ldi * 2 + cc * 16 : -1;
    }
    // G: transfer t (1 KiB) of this class belongs to source cls_src[cls][s] with t in [t0_s, t0_s + 64 * cg_s / 64); one transfer per wave
    int relG = -1, gsrc = 0;
    if (wid < nslot_g) {
        int t = wid, s = 0;
        for (; s < 2; ++s) {
            const int si = p.cls_src[cls][s];
            if (si < 0) break;
            const int n = p.src[si].cout >> 3;                          // chunks per pixel = KiB transfers per 64-pixel stage
            if (t < n) { gsrc = si; break; }
            t -= n;
        }
        const int cgs = p.src[gsrc].cout >> 3;
        const int id = t * 64 + lane, px = id / cgs, slot = id - px * cgs;
        const int cc = slot ^ (pair_swz(cgs, px) * 2);
        relG = px * p.src[gsrc].ld * 2 + p.src[gsrc].coff * 2 + cc * 16;
    }
    gsrc = __builtin_amdgcn_readfirstlane(gsrc);
    const int ntiles = (p.M + SPX - 1) / SPX;
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t ldsW = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(wid * 1024));
    // buffer resources are rebuilt per stage with the stage's first pixel as base (64-bit scalar arithmetic): tensors beyond 2 GiB
    // (T = 10 clips: 320 frames) stay addressable with 32-bit lane offsets, and rows beyond M fall outside num_records -> zeros
    const char* xbase = reinterpret_cast<const char*>(p.x) + p.cioff * 2;
    const char* gbase = reinterpret_cast<const char*>(p.src[gsrc].g);
    const long long xrow = (long long)p.ldi * 2, grow_b = (long long)p.src[gsrc].ld * 2;

    auto issue = [&](int buf, int tile) {
        const uint32_t dX = ldsW + (uint32_t)(buf * stage), dG = dX + (uint32_t)XBYTES;
        const long long m0 = (long long)tile * SPX;
        long long left = (long long)p.M - m0;
        if (left > SPX) left = SPX;
        __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase + m0 * xrow), 0, (int)(left * xrow - p.cioff * 2), 0x00020000);
#pragma unroll
        for (int i = 0; i < NTR_X; ++i) {
            if (wid + NW * i < NSLOT_X)                                // (wave-uniform)
                lds_dma16(dX + (uint32_t)(i * 1024 * NW), rsX, relX[i] >= 0 ? relX[i] : (int)OOB, 0);
        }
        if (wid < nslot_g) {
            __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gbase + m0 * grow_b), 0, (int)(left * grow_b), 0x00020000);
            lds_dma16(dG, rsG, relG >= 0 ? relG : (int)OOB, 0);
        }
    };
    // transfers THIS wave issues per stage (wave-uniform): its share of the X image + at most one of the G image
    int ndma = wid < nslot_g ? 1 : 0;
#pragma unroll
    for (int i = 0; i < NTR_X; ++i) ndma += (wid + NW * i < NSLOT_X) ? 1 : 0;
    ndma = __builtin_amdgcn_readfirstlane(ndma);

    f32x4 acc[TI][TJ], accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    // ---- fragment addressing ---------------------------------------------------------------------------------------------------------
    const int i16 = lane & 15, g4 = lane >> 4;
    const int xq = g4 * 4 + (i16 >> 2);
    const int csel = (i16 & 3) >> 1, chalf = (i16 & 1) * 8;
    // this wave's two row tiles: class-local tile index -> (source, tile inside the source)
    uint32_t gaddr[TI]; int grow[TI]; int gpitch[TI];                   // LDS offset (stage-relative, k-step 0 read 0), slab row (-1: none), pitch in bytes
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        int lt = rg * 2 + i, base = 0, si = -1, li = 0;
        for (int s = 0; s < 2; ++s) {
            const int c = p.cls_src[cls][s];
            if (c < 0) break;
            const int nt = p.src[c].cout >> 4;                          // 16-row tiles of the source
            if (lt < nt) { si = c; li = lt; break; }
            lt -= nt; base += SPX * (p.src[c].cout >> 3) * 16;
        }
        if (si < 0) { si = p.cls_src[cls][0]; li = 0; base = 0; grow[i] = -1; }
        else grow[i] = p.src[si].row0 + li * 16;
        const int cgs = p.src[si].cout >> 3;
        gpitch[i] = cgs * 16;
        gaddr[i] = (uint32_t)(XBYTES + base + (xq * cgs + ((li ^ pair_swz(cgs, xq)) * 2) + csel) * 16 + chalf);
    }
    uint32_t xaddr[TJ];
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
        const int j = min(cg + 4 * jj, NCT - 1);
        xaddr[jj] = (uint32_t)((xq * CPP + ((j ^ pair_swz(CPP, xq)) * 2) + csel) * 16 + chalf);
    }
    const bool do_bias = cg == 0;

    // ---- ring: stage s of this workgroup's tile sequence lives in slot s % 3; two stages are in flight while one is multiplied ---------
    int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int step = (int)gridDim.x;
    const int my = tile < ntiles ? (ntiles - tile + step - 1) / step : 0;      // stages of this workgroup
    if (my > 0) issue(0, tile);
    if (my > 1) issue(1, tile + step);
    int cur = 0;
    for (int s = 0; s < my; ++s) {
        // stage s landed when at most the (younger) stage s + 1 is still in flight
        if (s + 1 >= my || ndma == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ndma == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (ndma == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (ndma == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // ... for every wave; and everyone finished reading slot (s - 1) % 3
        asm volatile("" ::: "memory");
        if (s + 2 < my) issue(cur == 0 ? 2 : cur - 1, tile + (s + 2) * step);
        const uint32_t Sb = lds_base + (uint32_t)(cur * stage);
#pragma unroll
        for (int ks = 0; ks < SPX / 32; ++ks) {
            u32x4 gf[TI], xf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const uint32_t a = Sb + gaddr[i] + (uint32_t)(ks * 32 * gpitch[i]);
                const u32x2 lo = tr_read(a), hi = tr_read(a + (uint32_t)(16 * gpitch[i]));
                gf[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int jj = 0; jj < TJ; ++jj) {
                const u32x2 lo = tr_read(Sb + xaddr[jj] + (uint32_t)(ks * 32 * CPP * 16)), hi = tr_read(Sb + xaddr[jj] + (uint32_t)((ks * 32 + 16) * CPP * 16));
                xf[jj] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int jj = 0; jj < TJ; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[i]), __builtin_bit_cast(bf16x8, xf[jj]), acc[i][jj], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[i]), __builtin_bit_cast(bf16x8, ones), accb[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
    // ---- slab blockIdx.x: [rows_pad][Cin] fp32 -------------------------------------------------------------------------------------------
    float* dst = p.partial + (int64_t)blockIdx.x * p.rows_pad * p.Cin;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        if (grow[i] < 0) continue;
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
            const int j = cg + 4 * jj;
            if (j < NCT) {
                const int row = grow[i] + g4 * 4, kc = j * 16 + i16;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(int64_t)(row + e) * p.Cin + kc] = acc[i][jj][e];
            }
        }
    }
    if (do_bias && i16 == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (grow[i] < 0) continue;
            // slab row -> (source, channel): the source is the one whose row range holds grow[i]
            for (int s = 0; s < p.nsrc; ++s) {
                const int r = grow[i] - p.src[s].row0;
                if (r >= 0 && r < p.src[s].cout && p.src[s].dbias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(p.src[s].dbias + r + g4 * 4 + e, accb[i][e]);
                }
            }
        }
    }
#endif
}

template <typename K>
void raise_lds(K kern, size_t lds) { din_raise_lds(reinterpret_cast<const void*>(kern), lds); }

}  // namespace

// Deals the sources to the two classes (<= 2 sources and <= 128 rows each, every cout a multiple of 16, Cin in {192, 256, 288});
// false: this group does not fit the kernel (the caller launches the layers one by one).
bool plan_wgrad_1x1_multi(int nsrc, const int* couts, int cin, Wg1x1K* k) {
    if (nsrc < 2 || nsrc > 4 || !(cin == 192 || cin == 256 || cin == 288)) return false;
    for (int s = 0; s < nsrc; ++s) if (couts[s] % 16 != 0 || couts[s] <= 0 || couts[s] > 128) return false;
    // greedy: largest source first into the lighter class
    int order[4] = {0, 1, 2, 3};
    for (int a = 0; a < nsrc; ++a) for (int b = a + 1; b < nsrc; ++b) if (couts[order[b]] > couts[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
    int rows[2] = {0, 0}, cnt[2] = {0, 0}, cls_src[2][2] = {{-1, -1}, {-1, -1}};
    for (int a = 0; a < nsrc; ++a) {
        const int s = order[a];
        int c = rows[0] <= rows[1] ? 0 : 1;
        if (cnt[c] == 2 || rows[c] + couts[s] > 128) c ^= 1;
        if (cnt[c] == 2 || rows[c] + couts[s] > 128) return false;
        cls_src[c][cnt[c]++] = s; rows[c] += couts[s];
    }
    if (k) {
        int row0 = 0;
        for (int s = 0; s < nsrc; ++s) { k->src[s].row0 = row0; row0 += couts[s]; }
        k->rows_pad = row0;
        for (int c = 0; c < 2; ++c) {
            k->cls_src[c][0] = cls_src[c][0]; k->cls_src[c][1] = cls_src[c][1];
            k->cls_gbytes[c] = rows[c] / 8 * 1024;                     // 64 pixels x (rows / 8) chunks x 16 B
        }
        k->gbytes_max = k->cls_gbytes[0] > k->cls_gbytes[1] ? k->cls_gbytes[0] : k->cls_gbytes[1];
    }
    return true;
}

int launch_wgrad_1x1_multi(const Wg1x1K& k, int nwg, hipStream_t st) {
    dim3 grid(nwg, 2);
    const size_t lds = (size_t)NS * ((size_t)SPX * (k.Cin / 8) * 16 + k.gbytes_max);
    DIN_REQUIRE(lds <= 160 * 1024, "wgrad 1x1 multi: %zu bytes of LDS", lds);
    auto launch = [&](auto kern) {
        raise_lds(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, k);
    };
    if (k.Cin == 192) launch(conv_wgrad_1x1_multi_kernel<24>);
    else if (k.Cin == 256) launch(conv_wgrad_1x1_multi_kernel<32>);
    else if (k.Cin == 288) launch(conv_wgrad_1x1_multi_kernel<36>);
    else DIN_FAIL(DIN_E_ARG, "wgrad 1x1 multi: Cin %d not instantiated", k.Cin);
    return DIN_OK;
}

}  // namespace din_wgrad
