// Weight gradient of the narrow mid-network layers (Inception Mixed_5x / Mixed_6a: 5x5 48 -> 64, 3x3 64 -> 96, 3x3 96 -> 96; stride 1):
// millions of pixels, 55-83 thousand filter gradients.  The general wgrad kernels stream im2col(X) -- every input pixel once per tap -- against
// 64 / 96 filter rows, i.e. 43 FLOP per byte brought into LDS (conv_wgrad_ring_kernel<64|96,128>: 350-620 TFLOP/s).  Here, as in the stem
// kernel (conv_wgrad_small_kernel), an 8x32 (4x32) output tile brings in its G tile and its input HALO once (LDS-DMA, out-of-image ->
// hardware zeros), every (tap, channel) column is formed from the halo by transposing LDS reads, and persistent workgroups keep their whole
// block of dW in registers across all their tiles: HBM / L2 traffic = the two tensors once (the input twice: see below), ~500 FLOP per
// staged byte.
//   * dW is Cout x (taps * Cin) fp32 = 300-330 KB: too much for one workgroup's registers next to its fragments.  The filter rows are split
//     over CS = 2 workgroup classes (blockIdx.y) of BNT = Cout / 2 rows; both classes read the same halo (the input is the small operand
//     here: 48-96 channels) and their own half of G.
//   * sixteen waves (four per SIMD); wave w owns the 16-column tiles w, w + 16, ... of the (tap, ci) axis and all BNT rows of its class.
//   * k (pixel) order of a 32-pixel k-step = one tile row, exactly as in conv_wgrad_small_kernel: read rd, lane group g4, sub-row q ->
//     x = 16 rd + 4 g4 + q, so each 32-lane half of a ds_read_b64_tr_b16 covers 8 consecutive pixels; with the unit swizzles below these
//     are conflict-free for pixel pitches of 64 / 96 / 128 / 192 bytes.
//   * result: one fp32 slab per persistent workgroup index in the layout conv_wgrad_reduce_kernel expects ([cout_pad][kcols_pad], columns
//     (tap, ci)); class c writes rows [c BNT, (c + 1) BNT) of slab blockIdx.x.
// Replaces the autograd weight gradient of torchvision's BasicConv2d convs reached from backbone/backbone.py:44-77 (Mixed_5b..Mixed_6a).
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

// XOR applied to the (even) chunk-pair index of pixel `px` so that 8 consecutive pixels' 32-byte pieces fall into 8 different 32-byte bank
// groups: pitch 64 B (4 chunks): bit 2 of the pixel; 128 B (8): bits 1-2; 96 B (6) and 192 B (12): pitch * {0..7} mod 256 already differ
// for 96, and differ in halves of four for 192 (bit 2 of the pixel moves the second four by 32 B).
template <int CH> __device__ __forceinline__ int unit_swz(int px) {
    if constexpr (CH == 4) return ((px >> 2) & 1) * 2;
    else if constexpr (CH == 8) return ((px >> 1) & 3) * 2;
    else if constexpr (CH == 12) return ((px >> 2) & 1) * 2;
    else return 0;
}

// CPP: 16-byte chunks per input pixel (Cin / 8); BNT: filter rows per workgroup class; TH x 32 output pixels per tile
template <int CPP, int BNT, int KH, int KW, int TH>
__global__ __launch_bounds__(1024, 1) void conv_wgrad_halo_kernel(WgradK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = 16, TW = 32, NPX = TH * TW;
    constexpr int HWW = TW + KW - 1, HWH = TH + KH - 1, HPX = HWW * HWH, HC = HPX * CPP;
    constexpr int HBYTES = (HC * 16 + 1023) / 1024 * 1024, NSLOT_H = HBYTES / 1024, NTR_H = (NSLOT_H + NW - 1) / NW;
    constexpr int CG = BNT / 8, GBYTES = (NPX * CG * 16 + 1023) / 1024 * 1024, NSLOT_G = GBYTES / 1024, NTR_G = (NSLOT_G + NW - 1) / NW;
    constexpr int STAGE = HBYTES + GBYTES;
    constexpr int UPT = CPP / 2, NCT = KH * KW * UPT, TI = BNT / 16, TJ = (NCT + NW - 1) / NW;
    static_assert(CPP % 2 == 0 && BNT % 16 == 0, "whole 16-column / 16-row MFMA tiles");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cls = blockIdx.y;                                        // filter rows [cls * BNT, (cls + 1) * BNT)

    // ---- DMA plans (per lane, tile independent) ----------------------------------------------------------------------------
    int relH[NTR_H]; short hyv[NTR_H], hxv[NTR_H];
#pragma unroll
    for (int i = 0; i < NTR_H; ++i) {
        const int id = (wid + NW * i) * 64 + lane;
        const int hp = id / CPP, slot = id - hp * CPP;
        const int hy = hp / HWW, hx = hp - hy * HWW;
        const int cc = slot ^ unit_swz<CPP>(hx);                        // (the halo swizzle is a function of the COLUMN: see xaddr below)
        relH[i] = id < HC ? (hy * p.W + hx) * p.ldi * 2 + cc * 16 : -1;
        hyv[i] = (short)hy; hxv[i] = (short)hx;
    }
    int relG[NTR_G]; short gyv[NTR_G], gxv[NTR_G];
#pragma unroll
    for (int i = 0; i < NTR_G; ++i) {
        const int id = (wid + NW * i) * 64 + lane;
        const int t = id / CG, slot = id - t * CG;
        const int cc = slot ^ unit_swz<CG>(t);
        gyv[i] = (short)(t >> 5); gxv[i] = (short)(t & 31);
        relG[i] = (t < NPX && cls * BNT + cc * 8 + 7 < p.Cout) ? ((t >> 5) * p.OW + (t & 31)) * p.ldo * 2 + (cls * BNT + cc * 8) * 2 : -1;
    }
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y, ntiles = tiles_img * p.NB;
    const long long ximg = (long long)p.H * p.W * p.ldi * 2ll, gimg = (long long)p.OH * p.OW * p.ldo * 2ll;
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t ldsW = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(wid * 1024));

    auto issue = [&](int buf, int tile) {
        const int n = tile / tiles_img;
        const int tr = tile - n * tiles_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int gy0 = ty * TH - p.ph, gx0 = tx * TW - p.pw;
        __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.in)) + (long long)n * ximg, 0, (int)ximg, 0x00020000);
        __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.g)) + (long long)n * gimg, 0, (int)gimg, 0x00020000);
        const int baseX = (gy0 * p.W + gx0) * p.ldi * 2 + p.cioff * 2;
        const int baseG = ((ty * TH) * p.OW + tx * TW) * p.ldo * 2 + p.cooff * 2;
        const uint32_t dH = ldsW + (uint32_t)(buf * STAGE), dG = dH + (uint32_t)HBYTES;
#pragma unroll
        for (int i = 0; i < NTR_H; ++i) {
            if (wid + NW * i < NSLOT_H) {                              // (wave-uniform)
                const int gy = gy0 + hyv[i], gx = gx0 + hxv[i];
                const bool ok = relH[i] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                lds_dma16(dH + (uint32_t)(i * 1024 * NW), rsX, ok ? baseX + relH[i] : (int)OOB, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NTR_G; ++i) {
            if (wid + NW * i < NSLOT_G) {
                const bool ok = relG[i] >= 0 && ty * TH + gyv[i] < p.OH && tx * TW + gxv[i] < p.OW;
                lds_dma16(dG + (uint32_t)(i * 1024 * NW), rsG, ok ? baseG + relG[i] : (int)OOB, 0);
            }
        }
    };

    f32x4 acc[TI][TJ], accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = p.dbias != nullptr && wid == 0;
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    // ---- transpose-read addressing (per lane, tile independent): lane i16 of a 16-lane group supplies the 8-byte piece
    //      (k row = i16 >> 2, columns 4 * (i16 & 3) .. +3) of a 4-pixel x 16-channel block and receives column i16 --------------------
    const int i16 = lane & 15, g4 = lane >> 4;
    const int xq = g4 * 4 + (i16 >> 2);                                 // x inside the 16-pixel read (add 16 * rd)
    const int csel = (i16 & 3) >> 1, chalf = (i16 & 1) * 8;
    // LDS byte offsets (inside a stage) of this lane's pieces for k-step 0, read 0; k-step ks / read rd add the compile-time constants
    // ks * 32 * CG * 16 + rd * 16 * CG * 16 (G) and ks * HWW * CPP * 16 + rd * 16 * CPP * 16 (halo): both swizzles only look at pixel-index
    // bits that 8-pixel steps leave alone (G: the tile-linear pixel; halo: the halo COLUMN, so that rows and taps' row offsets drop out)
    uint32_t gaddr[TI], xaddr[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) gaddr[i] = (uint32_t)((xq * CG + ((i * 2) ^ unit_swz<CG>(xq)) + csel) * 16 + chalf);
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
        const int j = min(wid + NW * jj, NCT - 1);                      // (column tiles beyond NCT repeat the last one; never stored)
        const int tap = j / UPT, r = tap / KW, s2 = tap - r * KW;
        const int hx = xq + s2;
        xaddr[jj] = (uint32_t)(((r * HWW + hx) * CPP + (((j - tap * UPT) * 2) ^ unit_swz<CPP>(hx)) + csel) * 16 + chalf);
    }

    int cur = 0;
    int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (tile < ntiles) issue(0, tile);
    for (; tile < ntiles; tile += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // stage(cur) landed; everyone finished reading stage(cur ^ 1)
        asm volatile("" ::: "memory");
        if (tile + (int)gridDim.x < ntiles) issue(cur ^ 1, tile + gridDim.x);
        const uint32_t Hb = lds_base + (uint32_t)(cur * STAGE), Gb = Hb + (uint32_t)HBYTES;
        // (four waves per SIMD: the other waves' MFMAs cover this wave's fragment reads; a second fragment set does not fit 128 registers)
#pragma unroll
        for (int ks = 0; ks < TH; ++ks) {
            u32x4 gf[TI], xf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const u32x2 lo = tr_read(Gb + gaddr[i] + (uint32_t)(ks * 32 * CG * 16)), hi = tr_read(Gb + gaddr[i] + (uint32_t)((ks * 32 + 16) * CG * 16));
                gf[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int jj = 0; jj < TJ; ++jj) {
                const u32x2 lo = tr_read(Hb + xaddr[jj] + (uint32_t)(ks * HWW * CPP * 16)), hi = tr_read(Hb + xaddr[jj] + (uint32_t)((ks * HWW + 16) * CPP * 16));
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
        cur ^= 1;
    }
    // ---- this workgroup's rows of slab blockIdx.x: [cout_pad][kcols_pad] fp32, columns (tap, ci) ------------------------------------------
    float* dst = p.partial + (int64_t)blockIdx.x * p.cout_pad * p.kcols_pad;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
            const int j = wid + NW * jj;
            if (j < NCT) {
                const int co = cls * BNT + i * 16 + g4 * 4, kc = j * 16 + i16;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(int64_t)(co + e) * p.kcols_pad + kc] = acc[i][jj][e];
            }
        }
    if (do_bias && i16 == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int co = cls * BNT + i * 16 + g4 * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (co + e < p.Cout) atomicAdd(p.dbias + co + e, accb[i][e]);
        }
    }
#endif
}

template <typename K>
void raise_lds(K kern, size_t lds) { din_raise_lds(reinterpret_cast<const void*>(kern), lds); }

constexpr size_t stage_bytes(int cpp, int bnt, int kh, int kw, int th) {
    return (size_t)(((32 + kw - 1) * (th + kh - 1) * cpp * 16 + 1023) / 1024 * 1024) + (size_t)((th * 32 * (bnt / 8) * 16 + 1023) / 1024 * 1024);
}

}  // namespace

// shape table: which (cin, cout, kh, kw) run the halo weight-gradient kernel (stride 1, dilation 1, bf16)
bool wgrad_halo_shape(int cin, int cout, int kh, int kw, int* bnt) {
    int b = 0;
    if (kh == 5 && kw == 5 && cin == 48 && cout == 64) b = 32;
    else if (kh == 3 && kw == 3 && cin == 64 && cout == 96) b = 48;
    else if (kh == 3 && kw == 3 && cin == 96 && cout == 96) b = 48;
    if (bnt) *bnt = b;
    return b != 0;
}

int launch_wgrad_halo(const WgradK& k, int nwg, hipStream_t st) {
    dim3 grid(nwg, 2);
    auto launch = [&](auto kern, size_t lds) {
        raise_lds(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, k);
    };
    if (k.kh == 5 && k.Cin == 48 && k.Cout == 64) launch(conv_wgrad_halo_kernel<6, 32, 5, 5, 8>, 2 * stage_bytes(6, 32, 5, 5, 8));
    else if (k.kh == 3 && k.Cin == 64 && k.Cout == 96) launch(conv_wgrad_halo_kernel<8, 48, 3, 3, 8>, 2 * stage_bytes(8, 48, 3, 3, 8));
    else if (k.kh == 3 && k.Cin == 96 && k.Cout == 96) launch(conv_wgrad_halo_kernel<12, 48, 3, 3, 4>, 2 * stage_bytes(12, 48, 3, 3, 4));
    else DIN_FAIL(DIN_E_ARG, "wgrad halo kernel: shape %dx%d %d -> %d not instantiated", k.kh, k.kw, k.Cin, k.Cout);
    return DIN_OK;
}

}  // namespace din_wgrad
