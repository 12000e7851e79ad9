// Implicit-GEMM convolution for gfx950 (MI355X): fwd / dgrad share one gather kernel, wgrad has its own.
//
//   D[co][pix] = sum_k  Wpk[co][k] * im2col(X)[pix][k]          k = (r, s, ci)   (NHWC, ci contiguous)
//
// One template covers both storage types because the byte geometry is identical: every operand moves in
// 16-byte "chunks" along the reduction axis (4 fp32 or 8 bf16).  A lane feeds the MFMA one chunk:
//   bf16 : v_mfma_f32_16x16x32_bf16  -- the chunk is the lane's 8 k-values          (1 MFMA / chunk set)
//   fp32 : v_mfma_f32_16x16x4_f32    -- element j of the chunk feeds the j-th of 4 MFMAs (exact fp32)
// (the k order inside a k-step is permuted identically for both operands, which a sum does not care about).
//
// Tile: 128 pixels x BN (128|64) filters per 256-thread workgroup (4 waves as 2x2), k-step = 8 chunks
// (128 B per row), LDS double-buffered with an XOR swizzle (chunk ^ ((row>>1)&7)) that makes both the
// 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups conflict-free.  The MFMA's i index is the
// FILTER and j the PIXEL so that a lane ends up holding 4 consecutive output channels of one pixel
// (16-/8-byte NHWC stores).  Epilogue fuses bias, ReLU, the ReLU-backward mask, accumulation and the
// channel-offset write that makes torch.cat free.
//
// Replaces the arithmetic of torch.nn.Conv2d / nn.Linear reached from the reference at
// backbone/backbone.py:44-99, infer_model.py:184,190,226 and infer_module/dynamic_infer_module.py:149,191,195.
#include "din_common.h"
#include "conv_wgrad.h"
#include "conv_gather.h"
#include <atomic>
#include <unordered_map>
#include <mutex>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

using din_wgrad::WgradK;
using din_wgrad::lds_dma16;
using din_gather::ConvK;
using din_gather::out_pixel;
using din_gather::staged_tile_store;

namespace {

#ifndef DIN_GATHER_PRIO
#define DIN_GATHER_PRIO 0         // experiment: raised wave priority over the MFMA stream of the interleaved k-step
#endif
#ifndef DIN_GATHER_ILV
#define DIN_GATHER_ILV 1          // in-wave interleaved schedule of the FASTK gather loop (0: the compiler-scheduled loop, for A/B builds)
#endif
constexpr int BM = 128;      // pixels per workgroup tile
constexpr int KC = 8;        // 16-byte chunks per k-step (=> 128 B per tile row)
constexpr int NTHREADS = 256;

__device__ __forceinline__ int lds_slot(int row, int chunk) { return row * KC + (chunk ^ ((row >> 1) & 7)); }

template <typename T> struct Mma;
template <> struct Mma<float> {
    __device__ static void run(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    __device__ static void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// ------------------------------------------------------------------------------------------------
// fwd / dgrad gather kernels
// ------------------------------------------------------------------------------------------------
// One k-step of the FASTK loop: NA pixel-tile + NB filter-tile wave-level DMAs into consecutive STRIDE-byte slots of a ring stage, as
// ONE asm statement: M0 is written once and then advanced (declared clobbered instead of saved/restored around every transfer),
// 3 instructions per transfer instead of 5 -- the scalar unit is shared by the CU's 16 waves and was 40 % busy (SQ_ACTIVE_INST_SCA).
template <int NA, int NB, int STRIDE>
__device__ __forceinline__ void lds_dma_stage(uint32_t lds_addr, __amdgpu_buffer_rsrc_t rsA, const unsigned (&va)[NA], int soffA,
                                              __amdgpu_buffer_rsrc_t rsB, const int (&vb)[NB], int soffB) {
#define DIN_DMA_FIRST(V, R, S) "s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 " V ", " R ", " S " offen lds\n\t"
#define DIN_DMA_NEXT(V, R, S) "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 " V ", " R ", " S " offen lds\n\t"
    static_assert(NA == 2 && NB >= 1 && NB <= 3, "instantiated for the 8-wave 128-pixel tiles");
    if constexpr (NB == 1)
        asm volatile(DIN_DMA_FIRST("%6", "%1", "%3") DIN_DMA_NEXT("%7", "%1", "%3") DIN_DMA_NEXT("%8", "%2", "%4")
                     :: "s"(lds_addr), "s"(rsA), "s"(rsB), "s"(soffA), "s"(soffB), "n"(STRIDE), "v"(va[0]), "v"(va[1]), "v"(vb[0])
                     : "memory", "m0", "scc");
    else if constexpr (NB == 2)
        asm volatile(DIN_DMA_FIRST("%6", "%1", "%3") DIN_DMA_NEXT("%7", "%1", "%3") DIN_DMA_NEXT("%8", "%2", "%4") DIN_DMA_NEXT("%9", "%2", "%4")
                     :: "s"(lds_addr), "s"(rsA), "s"(rsB), "s"(soffA), "s"(soffB), "n"(STRIDE), "v"(va[0]), "v"(va[1]), "v"(vb[0]), "v"(vb[1])
                     : "memory", "m0", "scc");
    else
        asm volatile(DIN_DMA_FIRST("%6", "%1", "%3") DIN_DMA_NEXT("%7", "%1", "%3") DIN_DMA_NEXT("%8", "%2", "%4") DIN_DMA_NEXT("%9", "%2", "%4")
                     DIN_DMA_NEXT("%10", "%2", "%4")
                     :: "s"(lds_addr), "s"(rsA), "s"(rsB), "s"(soffA), "s"(soffB), "n"(STRIDE), "v"(va[0]), "v"(va[1]), "v"(vb[0]), "v"(vb[1]), "v"(vb[2])
                     : "memory", "m0", "scc");
#undef DIN_DMA_FIRST
#undef DIN_DMA_NEXT
}

// direct (un-staged) epilogue shared by both kernels: lane holds D[co0 + i*16 + (lane>>4)*4 + e][pix0 + j*16 + (lane&15)]
template <typename T, int TI, int TJ, int BN, int BMT = BM, int WM = 2, int WN = 2>
__device__ __forceinline__ void epilogue_direct(const ConvK& p, f32x4 (&acc)[TI][TJ], int co_tile, int px_tile, int wm, int wn, int lane, int split) {
    const int co_base = co_tile * BN + wn * (BN / WN) + (lane >> 4) * 4;
    const int px_base = px_tile * BMT + wm * (BMT / WM) + (lane & 15);
    if (p.splitk > 1) {
        // raw fp32 partial sums: partial[split][pix][cout_pad]   (cout_pad = n_co_tiles*BN)
        const int cpad = p.n_co_tiles * BN;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            int m = px_base + j * 16;
            if (m >= p.M) continue;
            float* dst = p.partial + ((int64_t)split * p.M + m) * cpad;
#pragma unroll
            for (int i = 0; i < TI; ++i) *reinterpret_cast<f32x4*>(dst + co_base + i * 16) = acc[i][j];
        }
        return;
    }
    T* __restrict__ outp = reinterpret_cast<T*>(p.out);
    const T* __restrict__ maskp = reinterpret_cast<const T*>(p.mask);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        int m = px_base + j * 16;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            int co = co_base + i * 16;
            if (co >= p.Cout) continue;
            f32x4 v = acc[i][j];
            const int64_t opx = out_pixel(p, m);
            int64_t o = opx * p.ldo + p.cooff + co;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (co + e >= p.Cout) break;
                float x = v[e];
                if (p.flags & DIN_CONV_BIAS) x += p.bias[co + e];
                if (p.flags & DIN_CONV_RELU) x = fmaxf(x, 0.f);
                if (p.flags & DIN_CONV_MASK) {
                    float y = Elem<T>::ld(maskp + opx * p.ldm + p.moff + co + e);
                    x = y > 0.f ? x : 0.f;
                }
                if (p.flags & DIN_CONV_ACCUM) x += Elem<T>::ld(outp + o + e);
                v[e] = x;
            }
            if (co + 3 < p.Cout && ((o & 3) == 0)) {
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(outp + o) = v;
                } else {
                    u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(outp + o) = pk;
                }
            } else {
                for (int e = 0; e < 4 && co + e < p.Cout; ++e) Elem<T>::st(outp + o + e, v[e]);
            }
        }
    }
}

// Generic addressing (strided dgrad / > 64 taps): plain loads with per-k-step coordinate arithmetic.
template <typename T, int BN>
__global__ __launch_bounds__(NTHREADS, 2) void conv_gather_generic_kernel(ConvK p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int TI = BN / 32, TJ = BM / 32, PA = BM / 32, PB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* smem = reinterpret_cast<u32x4*>(smem_raw);
    constexpr int BUF = (BM + BN) * KC;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int bx_, by_;
    xcd_block(bx_, by_);
    const int co_tile = bx_ % p.n_co_tiles, px_tile = bx_ / p.n_co_tiles;
    const int split = by_;
    const int ks_begin = split * p.ks_per_split;
    int ks_end = ks_begin + p.ks_per_split;
    if (ks_end > p.nk) ks_end = p.nk;
    const int cq = tid & 7, r0 = tid >> 3;
    int tyb[PA], txb[PA], nbase[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int m = px_tile * BM + r0 + 32 * i;
        tyb[i] = -(1 << 28); txb[i] = 0; nbase[i] = 0;
        if (m < p.M) {
            int n = m / (p.OH * p.OW);
            int rem = m - n * (p.OH * p.OW);
            int oy = rem / p.OW, ox = rem - oy * p.OW;
            tyb[i] = oy * p.ay + p.by; txb[i] = ox * p.ax + p.bx; nbase[i] = n * p.H * p.W;
        }
    }
    const T* __restrict__ inp = reinterpret_cast<const T*>(p.in);
    const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(p.w);
    u32x4 ga[PA], gb[PB];
    auto load_global = [&](int ks) {
        const int q = ks * KC + cq;
        const bool qok = q < p.Q;
        int tap = qok ? q / p.cpt : 0;
        int cc = q - tap * p.cpt;
        int r = tap / p.kw, s = tap - r * p.kw;
        const int dyk = r * p.cy, dxk = s * p.cx;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int ty = tyb[i] + dyk, tx = txb[i] + dxk;
            bool ok = qok && ty >= 0 && tx >= 0;
            int iy = ty, ix = tx;
            if (p.divy > 1) { iy = ty / p.divy; ok = ok && (iy * p.divy == ty); }
            if (p.divx > 1) { ix = tx / p.divx; ok = ok && (ix * p.divx == tx); }
            ok = ok && iy < p.H && ix < p.W;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(inp + (int64_t)(nbase[i] + iy * p.W + ix) * p.ldi + p.cioff + cc * EPC);
            ga[i] = v;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) gb[i] = wp[(int64_t)(co_tile * BN + r0 + 32 * i) * p.wld + ks * KC + cq];
    };
    auto store_lds = [&](int buf) {
        u32x4* A = smem + buf * BUF;
        u32x4* B = A + BM * KC;
#pragma unroll
        for (int i = 0; i < PA; ++i) A[lds_slot(r0 + 32 * i, cq)] = ga[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) B[lds_slot(r0 + 32 * i, cq)] = gb[i];
    };
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fchunk = lane >> 4;
    if (ks_begin < ks_end) {
        load_global(ks_begin);
        store_lds(0);
        __syncthreads();
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const int cur = (ks - ks_begin) & 1;
            const bool more = ks + 1 < ks_end;
            if (more) load_global(ks + 1);
            const u32x4* A = smem + cur * BUF;
            const u32x4* B = A + BM * KC;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 wf[TI], xf[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) wf[i] = B[lds_slot(wn * (BN / 2) + i * 16 + frow, kk * 4 + fchunk)];
#pragma unroll
                for (int j = 0; j < TJ; ++j) xf[j] = A[lds_slot(wm * (BM / 2) + j * 16 + frow, kk * 4 + fchunk)];
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
            }
            if (more) store_lds(cur ^ 1);
            __syncthreads();
        }
    }
    epilogue_direct<T, TI, TJ, BN>(p, acc, co_tile, px_tile, wm, wn, lane, split);
}

// Fast addressing (stride-1 gathers, <= 32 taps: every forward conv and every stride-1 dgrad).
//  * operands are fetched with BUFFER loads: one 32-bit byte offset per tile row, recomputed only when the tap changes;
//    padding taps get an out-of-range offset and the hardware returns zeros; the k-offset inside a tap is a scalar.
//    The resource base is moved to the first image of the tile so 32-bit offsets suffice for any tensor size.
//  * prefetch distance 2 through two register sets; LDS double buffer; one barrier per k-step.
//  * epilogue staged through LDS: bias/ReLU applied in registers, tile transposed in LDS, then 16-byte coalesced
//    stores with vector loads for the ReLU-backward mask and the accumulate input.
// Tile variants <BMT, BN, WM x WN waves, DEEP>: 128x128 (2x2), 128x64 (2x2) and 256x64 (4x1; twice the work per barrier for the
// short-K, latency-bound 64-filter layers; single register set to stay under 256 VGPRs, no LDS tables so two workgroups fit a CU).
// 512-thread variants 256x128 (4x2 waves) and 256x256 (4x2 waves, 64x128 wave tiles) raise the FLOPs per byte streamed into LDS
// from 64 to 85 / 128; measured throughput follows that ratio (128x128 ~770, 256x256 ~1150 TFLOP/s) although the L2->LDS path itself
// is not the limiter (profiles/r01_stream_probe.txt: 37 TB/s raw; the k-loop structure tops out at ~68 % of MFMA peak, and the MFMA
// rate drops a further ~27 % on real, toggling operands).
// Pipeline: an NS-stage ring of LDS stages of KCS 16-byte chunks per row, filled by LDS-DMA, NS-1 stages in flight: per stage ONE
// counted s_waitcnt vmcnt(N) (VMEM completes in order: N = DMAs of the younger stages) + ONE raw s_barrier (every wave's share of
// the stage landed, and every wave is done reading the stage about to be refilled).
// FASTK (bf16 8-wave tiles; host: whole k-steps per tap, no tap remap, k-order = taps inside channel chunks): every piece of per-step
// loader state is scalar except one validity select per tile row, the offsets of the NEXT transfer are prepared while the current
// stage is multiplied (so only the transfers themselves sit between the barrier and the MFMAs), and no other mode is compiled in.
// LANEK (with FASTK; bf16 8-wave tiles whose reduction channels are NOT whole k-steps per tap -- Conv2d_4a's 80, the 160-channel 7-tap layers of
// Mixed_6c / 6d): the same double-buffered, in-wave interleaved step, with the k-walk PER LANE: a lane's 16-byte chunk of a k-step is chunk
// q = 8 ks + cq of the flattened (tap, channel) axis, so every lane carries its own (tap, channel chunk) and forms its own tap delta / validity
// bit (~10 VALU per step); the filter side stays a scalar offset (the packed bank IS the flattened axis).  These launches used to run the
// general loop below -- compiler-scheduled, no interleaving: Conv2d_4a forward 718 TF where the same tile reaches 1004 TF on FASTK.
template <typename T, int BMT, int BN, int WM, int WN, int KCS, int NS, bool MULTI = false, bool FASTK = false, bool XSRC = false, bool LANEK = false>
// (second argument = waves per SIMD the register allocation must leave room for: the 8-wave 128-pixel bf16 tiles run TWO workgroups per CU
//  = four waves per SIMD = at most 128 VGPRs.  Left at 2, a harmless-looking edit -- round 4: the knock-out switches turned compile-time
//  constants -- moved the FASTK 128 x 192 instantiation from 125 to 131 registers: one workgroup per CU, 186 -> 259 us per launch, -1.6 ms per
//  step, caught only by diffing kernel_stats.csv against the previous round's.)
#ifdef DIN_EXPERIMENTS
// (experiment, round 4: four-wave 128-pixel tiles on 32-deep stages, THREE workgroups per CU = three barrier domains: <= 168 VGPRs)
__global__ __launch_bounds__(64 * WM * WN, (sizeof(T) == 2 && BMT == 128 && WM * WN == 8) ? 4 : ((sizeof(T) == 2 && BMT == 128 && WM * WN == 4 && KCS == 4 && NS == 2 && FASTK) ? 3 : 2)) void conv_gather_fast_kernel(ConvK p) {
#else
__global__ __launch_bounds__(64 * WM * WN, (sizeof(T) == 2 && BMT == 128 && WM * WN == 8) ? 4 : 2) void conv_gather_fast_kernel(ConvK p) {
#endif
#if defined(__HIP_DEVICE_COMPILE__)      // the host pass only needs the launch stub (the LDS-DMA builtin is device-only)
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BM = BMT;                                  // shadows the file-level default inside this kernel
    constexpr int KC = KCS;                                  // chunks per stage row (shadows the packing granularity of 8)
    constexpr int NT = 64 * WM * WN;                         // threads per workgroup
    constexpr int LR = NT / KC;                              // tile rows covered by one loader pass (KC chunk lanes per row)
    constexpr int RW = 64 / KC;                              // rows written by one wave-level DMA (1 KiB)
    // the loader covers LR rows per pass: a filter tile that is not a multiple (160 or 96 rows on 8 waves) is loaded as PB whole passes
    // into a padded LDS region; the surplus rows (next tile's filters or zeros) are never read
    constexpr int TI = BN / WN / 16, TJ = BM / WM / 16, PA = BM / LR, PB = (BN + LR - 1) / LR, BNP = PB * LR;
    static_assert(BM % LR == 0, "pixel tile must be whole loader passes");
    constexpr int BUF = (BM + BNP) * KC;                     // 16-byte units per stage
    auto lds_slot = [](int row, int chunk) { return row * KCS + (chunk ^ ((row >> 1) & (KCS - 1))); };   // conflict-free for 4 and 8
    constexpr int CPITCH = BN * (int)sizeof(T) + 16;         // epilogue tile row pitch (bytes): +16 B kills bank conflicts
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* smem = reinterpret_cast<u32x4*>(smem_raw);
    int* wtap_lds = reinterpret_cast<int*>(smem_raw + NS * BUF * 16);  // [32] tap -> tap of the packed bank (remap launches only)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    int bx_, by_;
    xcd_block(bx_, by_);
    const int co_tile = bx_ % p.n_co_tiles, px_tile = bx_ / p.n_co_tiles;
    const int split = by_;
    // host counts k-steps in units of 8 chunks (the packing granularity); this kernel steps KCS chunks
    const int ks_begin = split * p.ks_per_split * (8 / KCS);
    int ks_end = ks_begin + p.ks_per_split * (8 / KCS);
    {
        const int nk_s = (p.Q + KCS - 1) / KCS;
        if (ks_end > nk_s) ks_end = nk_s;
    }
    [[maybe_unused]] const int ks_x0 = ks_end;                 // XSRC: k-steps >= ks_x0 read the extra 1x1 source (never with split-K)
    if constexpr (XSRC) {
        static_assert(!MULTI && !FASTK && KCS == 8, "XSRC: general single-source loop");
        ks_end += p.xsteps;
    }
    const int ntaps = p.kh * p.kw;
    if (p.remap && tid < 32) wtap_lds[tid] = tid < ntaps ? (int)p.wtap[tid] : 0;
    // byte offset of tap t = (r,s) relative to tap (0,0): r*dA + s*dB, r = t / kw by an exact multiply-shift (t < 32)
    const int dA = p.cy * p.W * p.ldi * (int)sizeof(T), dB = p.cx * p.ldi * (int)sizeof(T);
    const int inv_kw = 65536 / p.kw + 1;
    auto tapdelta = [&](int t) { int r = (t * inv_kw) >> 16; return r * dA + (t - r * p.kw) * dB; };

    // ---- buffer resources ---------------------------------------------------------------------------------
    const int m_first = px_tile * BM;
    const int n_first = m_first / (p.OH * p.OW);                              // uniform
    const long long img_bytes = (long long)p.H * p.W * p.ldi * (long long)sizeof(T);
    const long long a_off = (long long)n_first * img_bytes;
    long long a_rem = p.in_bytes - a_off;
    if (a_rem > 0x7fffffffll) a_rem = 0x7fffffffll;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in)) + a_off, 0, (int)a_rem, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

    // ---- per-row state --------------------------------------------------------------------------------------
    // LDS-DMA writes lane-linearly (wave base + lane*16 B; measured: profiles/r01_probe_lds_dma.txt), i.e. lane -> (row lane>>3,
    // slot lane&7).  The XOR swizzle of the LDS image is therefore applied to the SOURCE: the lane fetches logical chunk
    // cq = slot ^ swizzle(row) (guide rule 21: linear destination + permuted source + same permutation on the read).
    const int r0 = tid / KC;
    const int cq = (tid % KC) ^ ((r0 >> 1) & (KC - 1));
    int pixoff[PA];                       // byte offset (from the resource base) of tap (0,0), channel cioff, chunk 0
                                          // (MULTI: pixel index relative to the first image of the tile, -1 = row beyond M)
    unsigned vmask[PA];                   // bit t set <=> tap t of this pixel lies inside the image (<= 32 taps)
    {
        const unsigned full_row = p.kw >= 32 ? 0xffffffffu : ((1u << p.kw) - 1u);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int m = m_first + r0 + LR * i;
            vmask[i] = 0u; pixoff[i] = 0;
            if (m < p.M) {
                int n = m / (p.OH * p.OW);
                int rem = m - n * (p.OH * p.OW);
                int oy = rem / p.OW, ox = rem - oy * p.OW;
                const int ty0 = oy * p.ay + p.by, tx0 = ox * p.ax + p.bx;
                // valid s: 0 <= tx0 + s*cx < W   (cx may be negative for dgrad) -- build the kw-bit column mask once
                unsigned cmask = 0u;
                for (int s2 = 0; s2 < p.kw; ++s2) {
                    int tx = tx0 + s2 * p.cx;
                    cmask |= (tx >= 0 && tx < p.W) ? (1u << s2) : 0u;
                }
                cmask &= full_row;
                unsigned mk = 0u;
                for (int r = 0; r < p.kh; ++r) {
                    int ty = ty0 + r * p.cy;
                    mk |= (ty >= 0 && ty < p.H) ? (cmask << (r * p.kw)) : 0u;
                }
                vmask[i] = mk;
                pixoff[i] = (((n - n_first) * p.H + ty0) * p.W + tx0) * p.ldi * (int)sizeof(T) + p.cioff * (int)sizeof(T);
                if constexpr (MULTI) pixoff[i] = m - n_first * (p.OH * p.OW);      // 1x1, stride 1: output pixel == input pixel
            } else if constexpr (MULTI) pixoff[i] = -1;
        }
    }
    int voffB[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) voffB[i] = ((co_tile * BN + r0 + LR * i) * p.wld + (p.remap ? 0 : cq)) * 16;
    if (p.remap) __syncthreads();                                             // wtap table visible (uniform branch)

    const bool tap_uniform = (p.cpt % KC) == 0;
    // uniform-tap bookkeeping (scalar).  korder: ks = cchunk * ntaps + tap
    int tap_s, cc_s;
    if (p.korder) { const int cch = ks_begin / ntaps; tap_s = ks_begin - cch * ntaps; cc_s = cch * KC; }
    else { tap_s = (ks_begin * KC) / p.cpt; cc_s = ks_begin * KC - tap_s * p.cpt; }
    unsigned voffA[PA];
    auto refresh_uniform = [&]() {
        const int td = tapdelta(tap_s) + cq * 16;
#pragma unroll
        for (int i = 0; i < PA; ++i) voffA[i] = (tap_s < ntaps && ((vmask[i] >> tap_s) & 1u)) ? (unsigned)(pixoff[i] + td) : OOB;
    };
    // per-lane bookkeeping for k-steps that straddle taps
    int tap_l = (ks_begin * KC + cq) / p.cpt, cc_l = ks_begin * KC + cq - tap_l * p.cpt;
    if (tap_uniform) refresh_uniform();

    // ---- multi-source bookkeeping (uniform): current source, chunk position inside it ----------------------------------------
    int sb = 0, cc_m = 0;
    [[maybe_unused]] auto open_source = [&](int bsrc) {
        const ConvK::Src& sr = p.src[bsrc];
        const long long imgb = (long long)p.H * p.W * sr.ld * (long long)sizeof(T);
        long long rem = sr.in_bytes - (long long)n_first * imgb;
        if (rem > 0x7fffffffll) rem = 0x7fffffffll;
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(sr.in)) + (long long)n_first * imgb, 0, (int)rem, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sr.w), 0, (int)sr.w_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < PB; ++i) voffB[i] = ((co_tile * BN + r0 + LR * i) * sr.wld + cq) * 16;
    };
    if constexpr (MULTI) {
        // position at this split's first k-step (split-K is never used with MULTI, but keep the walk general)
        int skip = ks_begin;
        while (sb < p.nsrc) {
            const int steps = (p.src[sb].cpt + KC - 1) / KC;
            if (skip < steps) break;
            skip -= steps; ++sb;
        }
        cc_m = skip * KC;
        if (sb < p.nsrc) open_source(sb);
    }

    // LDS byte address of this wave's RW rows of pass 0 in stage 0 (wave-uniform -> SGPR)
    const uint32_t ldsA0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(smem + (wid * RW) * KC));
    // XSRC: resources of the extra source (its tensor from the first image of the tile on, its packed 1x1 bank)
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsXA = rsA, rsXB = rsB;
    [[maybe_unused]] long long x_px0 = 0;
    if constexpr (XSRC) {
        const ConvK::Src& sr = p.src[0];
        x_px0 = (long long)n_first * (p.out_sy == 0 ? p.OH * p.OW : p.out_H * p.out_W);
        long long rem = sr.in_bytes - x_px0 * sr.ld * (long long)sizeof(T);
        if (rem > 0x7fffffffll) rem = 0x7fffffffll;
        rsXA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(sr.in)) + x_px0 * sr.ld * (long long)sizeof(T), 0,
                                                 (int)(rem > 0 ? rem : 0), 0x00020000);
        rsXB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sr.w), 0, (int)sr.w_bytes, 0x00020000);
    }
    // issue the DMA of k-step ks into stage `buf`: PA + PB wave-level 1-KiB transfers per wave, no VGPRs, no ds_write
    auto issue_dma = [&](int buf, int ks) {
        const uint32_t A = ldsA0 + (uint32_t)(buf * BUF * 16);
        const uint32_t B = A + (uint32_t)(BM * KC * 16);
        if constexpr (XSRC) {
            if (ks >= ks_x0) {
                const ConvK::Src& sr = p.src[0];
                const int cc = (ks - ks_x0) * KC;
                const bool cok = (cc + cq) < sr.cpt;
                const int chan = sr.coff * (int)sizeof(T) + (cc + cq) * 16, ldb = sr.ld * (int)sizeof(T);
#pragma unroll
                for (int i = 0; i < PA; ++i) {
                    const int m = m_first + r0 + LR * i;
                    unsigned vo = OOB;
                    if (cok && m < p.M) vo = (unsigned)((int)(out_pixel(p, m) - x_px0) * ldb + chan);
                    lds_dma16(A + (uint32_t)(LR * i * KC * 16), rsXA, (int)vo, 0);
                }
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    lds_dma16(B + (uint32_t)(LR * i * KC * 16), rsXB, ((co_tile * BN + r0 + LR * i) * sr.wld + cq) * 16, cc * 16);
                return;
            }
        }
        int remapB = 0;                                    // per-lane chunk offset (bytes) into the packed bank, remap mode only
        int korderB = 0;                                   // scalar chunk offset of this k-step in the packed bank (korder mode)
        if constexpr (MULTI) {
            const ConvK::Src& sr = p.src[sb < p.nsrc ? sb : 0];
            const bool cok = sb < p.nsrc && (cc_m + cq) < sr.cpt;         // chunk inside this source's channels
            const int chan = sr.coff * (int)sizeof(T) + (cc_m + cq) * 16;
            const int ldb = sr.ld * (int)sizeof(T);
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                unsigned vo = (cok && pixoff[i] >= 0) ? (unsigned)(pixoff[i] * ldb + chan) : OOB;
                lds_dma16(A + (uint32_t)(LR * i * KC * 16), rsA, (int)vo, 0);
            }
            const int soffB = cc_m * 16;
#pragma unroll
            for (int i = 0; i < PB; ++i) lds_dma16(B + (uint32_t)(LR * i * KC * 16), rsB, voffB[i], soffB);
            cc_m += KC;
            if (sb < p.nsrc && cc_m >= sr.cpt) { cc_m = 0; ++sb; if (sb < p.nsrc) open_source(sb); }
            return;
        }
        if (tap_uniform) {
            const int soff = cc_s * 16;
            if (p.remap) remapB = ((tap_s < ntaps ? wtap_lds[tap_s] : 0) * p.cpt + cc_s + cq) * 16;
            korderB = (tap_s * p.cpt + cc_s) * 16;
#pragma unroll
            for (int i = 0; i < PA; ++i)
                lds_dma16(A + (uint32_t)(LR * i * KC * 16), rsA, (int)voffA[i], soff);
            if (p.korder) {
                if (++tap_s == ntaps) { tap_s = 0; cc_s += KC; }
                refresh_uniform();
            } else {
                cc_s += KC;
                if (cc_s >= p.cpt) { cc_s = 0; ++tap_s; refresh_uniform(); }
            }
        } else {
            // k-step straddles taps: per-lane tap index; cost kept to ~5 VALU per row (bit test, add, select)
            const bool ok = tap_l < ntaps;
            const int td = tapdelta(tap_l) + cc_l * 16;
            const unsigned bit = ok ? (1u << tap_l) : 0u;
            if (p.remap) remapB = ok ? (wtap_lds[tap_l] * p.cpt + cc_l) * 16 : 0;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                unsigned vo = (vmask[i] & bit) ? (unsigned)(pixoff[i] + td) : OOB;
                lds_dma16(A + (uint32_t)(LR * i * KC * 16), rsA, (int)vo, 0);
            }
            cc_l += KC;
            if (p.cpt >= KC) {                       // at most one tap boundary per k-step: branch-free
                const bool wrap = cc_l >= p.cpt;
                cc_l -= wrap ? p.cpt : 0;
                tap_l += wrap ? 1 : 0;
            } else {
                while (cc_l >= p.cpt) { cc_l -= p.cpt; ++tap_l; }
            }
        }
        if (p.remap) {
#pragma unroll
            for (int i = 0; i < PB; ++i)
                lds_dma16(B + (uint32_t)(LR * i * KC * 16), rsB, voffB[i] + remapB, 0);
        } else {
            const int soffB = p.korder ? korderB : ks * KC * 16;
#pragma unroll
            for (int i = 0; i < PB; ++i)
                lds_dma16(B + (uint32_t)(LR * i * KC * 16), rsB, voffB[i], soffB);
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fchunk = lane >> 4;
    auto compute = [&](int cur) {
        const u32x4* A = smem + cur * BUF;
        const u32x4* B = A + BM * KC;
#pragma unroll
        for (int kk = 0; kk < KCS / 4; ++kk) {
            u32x4 wf[TI], xf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) wf[i] = B[lds_slot(wn * (BN / WN) + i * 16 + frow, kk * 4 + fchunk)];
#pragma unroll
            for (int j = 0; j < TJ; ++j) xf[j] = A[lds_slot(wm * (BM / WM) + j * 16 + frow, kk * 4 + fchunk)];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
        }
    };

    constexpr int NDMA = PA + PB;                                  // wave-level DMAs per stage per wave (issued unconditionally)
    static_assert((NS - 2) * NDMA <= 63, "vmcnt field");
    if constexpr (FASTK) {
        static_assert(!MULTI && NS == 2 && PA == 2, "FASTK: 128-pixel tiles whose loader covers 64 rows per pass, double-buffered");
        static_assert(!LANEK || KCS == 8, "LANEK: 8-chunk k-steps");
        // scalar walk over k-steps: ks -> (channel chunk ks / ntaps, tap ks % ntaps); td = byte delta of the tap, fa / fb = scalar byte
        // offsets of the step inside a pixel's channels / inside a packed filter row
        int tap = ks_begin % ntaps, tr = (tap * inv_kw) >> 16, tc = tap - tr * p.kw;
        int td = tr * dA + tc * dB;
        int fa = (ks_begin / ntaps) * KC * 16, fb = (tap * p.cpt) * 16 + fa;
        const int tap_row_wrap = dA - p.kw * dB, cpt16 = p.cpt * 16;
        unsigned pixq[PA];
#pragma unroll
        for (int i = 0; i < PA; ++i) pixq[i] = (unsigned)(pixoff[i] + cq * 16);
        unsigned va[PA];
        int soffA = 0, soffB = 0;
        const bool knockA = DIN_KNOCK(p.flags, 0x200);
        if (DIN_KNOCK(p.flags, 0x400)) {
#pragma unroll
            for (int i = 0; i < PB; ++i) voffB[i] = (int)OOB;
        }
        // LANEK: this lane's chunk of the flattened (tap, channel-chunk) axis at the walk's current step
        [[maybe_unused]] int tap_v = (ks_begin * KC + cq) / p.cpt, cc_v = (ks_begin * KC + cq) - tap_v * p.cpt;
        if constexpr (LANEK) fb = ks_begin * KC * 16;
        auto prepare = [&]() {                                     // offsets of the transfer for the walk's current step, then advance it
            if constexpr (LANEK) {
                const int r = (tap_v * inv_kw) >> 16;
                const int tdv = r * dA + (tap_v - r * p.kw) * dB + cc_v * 16;
                const unsigned bit = tap_v < ntaps ? (1u << tap_v) : 0u;         // (past the last tap: zeros; the bank's rows are padded to whole k-steps)
#pragma unroll
                for (int i = 0; i < PA; ++i) va[i] = ((vmask[i] & bit) && !knockA) ? (unsigned)(pixoff[i] + tdv) : OOB;
                soffA = 0; soffB = fb;
                fb += KC * 16;
                cc_v += KC;
                const bool wrap = cc_v >= p.cpt;                                 // cpt >= 8 (host): at most one tap boundary per k-step
                cc_v -= wrap ? p.cpt : 0;
                tap_v += wrap ? 1 : 0;
                return;
            }
            const unsigned bit = 1u << tap;
#pragma unroll
            for (int i = 0; i < PA; ++i) va[i] = ((vmask[i] & bit) && !knockA) ? pixq[i] + (unsigned)td : OOB;
            soffA = fa; soffB = fb;
            ++tap; ++tc; td += dB; fb += cpt16;
            if (tc == p.kw) { tc = 0; td += tap_row_wrap; }
            if (tap == ntaps) { tap = 0; td = 0; fa += KC * 16; fb = fa; }
        };
        if (ks_begin < ks_end) {
            prepare();
            lds_dma_stage<PA, PB, LR * KC * 16>(ldsA0, rsA, va, soffA, rsB, voffB, soffB);
            prepare();
            int cur = 0;
#if DIN_GATHER_ILV
            // In-wave schedule of a k-step (round 3, late): the compiler's own placement of compute() kept ONE fragment of lookahead
            // (ds_read x2 -> s_waitcnt lgkmcnt -> 2 MFMAs: twelve exposed LDS latencies per k-step) behind five back-to-back transfer
            // issues that stall the wave while the CU's vector-memory path drains the other seven waves' transfers.  Here: the fragments
            // of half-step 0 are requested first, the next stage's transfers and the fragment reads of half-step 1 sit one per MFMA between
            // the MFMAs of half-step 0 (pinned with sched_barrier; a second fragment register set, ~10 fragments live at the peak).
            // Same operands, same accumulation order: bit-identical results.
            auto step = [&](auto more_t) {
                constexpr bool MORE = decltype(more_t)::value;
                const u32x4* A = smem + cur * BUF;
                const u32x4* B = A + BM * KC;
                const uint32_t nx = ldsA0 + (uint32_t)((cur ^ 1) * BUF * 16);
                constexpr int NKK = KCS / 4, NMF = TI * TJ;
                u32x4 xf[2][TJ], wf[2][TI];
                auto rd = [&](int set, int kk, int f) {                  // fragment f of half-step kk: pixel rows first, then filter rows
                    if (f < TJ) xf[set][f] = A[lds_slot(wm * (BM / WM) + f * 16 + frow, kk * 4 + fchunk)];
                    else wf[set][f - TJ] = B[lds_slot(wn * (BN / WN) + (f - TJ) * 16 + frow, kk * 4 + fchunk)];
                };
#pragma unroll
                for (int f = 0; f < TI + TJ; ++f) rd(0, 0, f);
                __builtin_amdgcn_sched_barrier(0);
#if DIN_GATHER_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const int set = kk & 1;
                    const int ndma = (kk == 0 && MORE) ? NDMA : 0, nrd = (kk + 1 < NKK) ? TI + TJ : 0;
                    const int nitems = ndma + nrd, per = (nitems + NMF - 1) / NMF;
#pragma unroll
                    for (int mf = 0; mf < NMF; ++mf) {
                        Mma<T>::run(wf[set][mf / TJ], xf[set][mf % TJ], acc[mf / TJ][mf % TJ]);
#pragma unroll
                        for (int q = 0; q < per; ++q) {
                            const int it = mf * per + q;
                            if (it < ndma) {
                                if (it < PA) lds_dma16(nx + (uint32_t)(it * LR * KC * 16), rsA, (int)va[it < PA ? it : 0], soffA);
                                else lds_dma16(nx + (uint32_t)(it * LR * KC * 16), rsB, voffB[it >= PA ? it - PA : 0], soffB);
                            } else if (it < nitems) rd(set ^ 1, kk + 1, it - ndma);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#if DIN_GATHER_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            };
            // the last k-step (nothing left to request) is peeled: two accumulating variants merging inside one loop body cost a copy of
            // every accumulator per k-step and 192 VGPRs
            for (int ks = ks_begin; ks + 1 < ks_end; ++ks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (DIN_KNOCK(p.flags, 0x800)) lds_dma_stage<PA, PB, LR * KC * 16>(ldsA0 + (uint32_t)((cur ^ 1) * BUF * 16), rsA, va, soffA, rsB, voffB, soffB);
                else step(std::true_type{});
                prepare();
                cur ^= 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!DIN_KNOCK(p.flags, 0x800)) step(std::false_type{});
#else
            for (int ks = ks_begin; ks < ks_end; ++ks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (ks + 1 < ks_end) lds_dma_stage<PA, PB, LR * KC * 16>(ldsA0 + (uint32_t)((cur ^ 1) * BUF * 16), rsA, va, soffA, rsB, voffB, soffB);
                if (!DIN_KNOCK(p.flags, 0x800)) compute(cur);
                prepare();
                cur ^= 1;
            }
#endif
        }
    } else
    if (ks_begin < ks_end) {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (ks_begin + s0 < ks_end) issue_dma(s0, ks_begin + s0);
        int cur = 0, nxt = NS - 1;                                 // stage of ks / stage the next DMA goes to
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            // stage ks must have landed; younger stages (at most NS-2, fewer at the tail) may stay in flight
            int younger = ks_end - 1 - ks;
            if (younger > NS - 2) younger = NS - 2;
            if constexpr (NS >= 4) { if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory"); }
            if constexpr (NS >= 3) { if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * NDMA) : "memory"); }
            if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");                       // LDS contents changed behind the compiler's back
            if (ks + NS - 1 < ks_end) issue_dma(nxt, ks + NS - 1);
            compute(cur);
            cur = cur + 1 == NS ? 0 : cur + 1;
            nxt = nxt + 1 == NS ? 0 : nxt + 1;
        }
    }
    __syncthreads();                                                // all waves done with the stages: the epilogue reuses them

    // ---- epilogue ---------------------------------------------------------------------------------------------
    const bool aligned = (p.Cout % EPC == 0) && (p.cooff % EPC == 0) && (p.ldo % EPC == 0) &&
                         (!(p.flags & DIN_CONV_MASK) || ((p.ldm % EPC == 0) && (p.moff % EPC == 0)));
    if (p.splitk > 1 || !aligned) {
        epilogue_direct<T, TI, TJ, BN, BM, WM, WN>(p, acc, co_tile, px_tile, wm, wn, lane, split);
        return;
    }
    {
        const int co_l = wn * (BN / WN) + (lane >> 4) * 4;         // channel inside the tile
        const int px_l = wm * (BM / WM) + (lane & 15);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            const int co = co_tile * BN + co_l + i * 16;
            const bool cooked = p.craw <= 0 || co < p.craw;             // (craw is a multiple of 4: uniform over the lane's 4 channels)
            if ((p.flags & DIN_CONV_BIAS) && co < p.Cout && cooked) bv = *reinterpret_cast<const f32x4*>(p.bias + co);   // Cout % 4 == 0 here
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                f32x4 v = acc[i][j] + bv;
                if ((p.flags & DIN_CONV_RELU) && cooked) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                unsigned char* dst = smem_raw + (px_l + j * 16) * CPITCH + (co_l + i * 16) * (int)sizeof(T);
                if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(dst) = v;
                else *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        }
    }
    staged_tile_store<T, BM, BN, NT>(p, smem_raw, tid, co_tile, m_first);
#endif
}

// ---- image layer fed from raw uint8 frames (din_conv_desc::in_u8): the halo pixels of a tile are fetched as bytes from the three colour
//      planes, normalised exactly like utils.prep_images ((x / 255 - 0.5) * 2: three separately rounded fp32 operations, utils.py:8-19),
//      rounded to bf16 and written to the LDS position the LDS-DMA of a prepared NHWC tensor would have filled (16 bytes per pixel:
//      r, g, b and five zero channels; pixels outside the image are zero).  Lane-linear: halo pixel id = (wave + 4 i) * 64 + lane.
__device__ __forceinline__ float prep_u8(uint32_t v) {
    float y = __fdiv_rn((float)v, 255.0f);
    y = __fsub_rn(y, 0.5f);
    return __fmul_rn(y, 2.0f);
}
__device__ __forceinline__ void u8_lut_init(bf16_t* lut, int tid) {          // NTHREADS == 256: one entry per thread
    lut[tid] = (bf16_t)(pack_bf16x2(prep_u8((uint32_t)tid), 0.f) & 0xffffu);
}
template <int V> struct IcTag { static constexpr int value = V; };
template <int NTR>
struct U8Halo {
    // the three bytes of a pixel stay in separate registers until store(): nothing consumes them at load time, so the loads stay in flight
    // under the tile's MFMAs instead of being waited for where they are issued
    uint32_t r[NTR], g[NTR], b[NTR];
    uint32_t valid;                                      // bit i: pixel i lies inside the image
    // hyv / hxv: the lane's halo coordinates per transfer (the kernels' tile-independent DMA plans); inside[i]: the id is a halo pixel
    template <int NSLOT>
    __device__ __forceinline__ void load(const unsigned char* __restrict__ img, int n, int H, int W, int gy0, int gx0, int wid,
                                         const short (&hyv)[NTR], const short (&hxv)[NTR], const int (&inside)[NTR]) {
        // every lane ALWAYS loads (coordinates clamped into the image, validity kept as a bit): a load inside `if (inside)` merges with the zero
        // of the other path at the end of the branch, and the compiler waits for it right there -- five exposed memory latencies per tile
        // (vmcnt(2) / (1) / (0) after every pixel in the ISA; the image layer ran 5.5 us per tile = 2.7 TB/s because of it)
        const int64_t plane = (int64_t)H * W;
        const unsigned char* base = img + (int64_t)n * 3 * plane;
        valid = 0u;
#pragma unroll
        for (int i = 0; i < NTR; ++i) {
            const int gy = gy0 + hyv[i], gx = gx0 + hxv[i];
            const bool ok = wid + 4 * i < NSLOT && inside[i] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const unsigned char* q = base + (int64_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
            // untracked loads (inline asm): the compiler would zero-extend the bytes -- i.e. wait for them -- right here; the caller waits by
            // count (s_waitcnt vmcnt) before store() and calls landed()
            asm volatile("global_load_ubyte %0, %1, off" : "=&v"(r[i]) : "v"(q) : "memory");
            asm volatile("global_load_ubyte %0, %1, off" : "=&v"(g[i]) : "v"(q + plane) : "memory");
            asm volatile("global_load_ubyte %0, %1, off" : "=&v"(b[i]) : "v"(q + 2 * plane) : "memory");
            valid |= (ok ? 1u : 0u) << i;
        }
    }
    __device__ __forceinline__ void landed() {                       // after the caller's s_waitcnt: ties the registers to this point
#pragma unroll
        for (int i = 0; i < NTR; ++i) { asm volatile("" : "+v"(r[i])); asm volatile("" : "+v"(g[i])); asm volatile("" : "+v"(b[i])); }
    }
    // lut: the 256 normalised bf16 values (prep_u8 of every byte, built once per workgroup by u8_lut_init) -- three LDS reads per pixel
    // instead of three fp32 divisions
    template <int NSLOT>
    __device__ __forceinline__ void store(unsigned char* lds_buf, const bf16_t* lut, int wid, int lane) const {   // lds_buf: the halo buffer
#pragma unroll
        for (int i = 0; i < NTR; ++i) {
            if (wid + 4 * i < NSLOT) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if ((valid >> i) & 1u) {
                    v[0] = (uint32_t)lut[r[i]] | ((uint32_t)lut[g[i]] << 16);
                    v[1] = (uint32_t)lut[b[i]];
                }
                *reinterpret_cast<u32x4*>(lds_buf + ((wid + 4 * i) * 64 + lane) * 16) = v;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Small-channel 3x3 (stem) convolution: filters stationary in LDS + input HALO tiles, persistent workgroups.
// For layers with 32/64 reduction channels and <= 64 produced channels over millions of pixels (Inception Conv2d_2a/2b, fwd and
// dgrad) the implicit-GEMM kernel is bound by the L2->CU operand stream: im2col pulls every input pixel kh*kw times.  Here
//   * the whole filter bank of the launch (<= 36 KiB) is loaded into LDS once per workgroup and stays there;
//   * per 8x32 output tile the (8+kh-1) x (32+kw-1) input halo is brought in ONCE by LDS-DMA (1.33x the unique bytes instead of 9x),
//     out-of-image pixels as hardware zeros; all taps are formed from LDS with a swizzle that is conflict-free at every pixel
//     offset ((hp>>1)&3 for 64-byte pixels, hp&7 for 128-byte pixels; brute-forced against the ds_read_b128 lane groups);
//   * workgroups are persistent (grid = 2 per CU) and walk the tiles; with 64-byte pixels the next halo is in flight while the
//     current tile is multiplied; the epilogue is staged through the just-consumed halo buffer -> 16-byte coalesced stores with the
//     usual fused bias / ReLU / ReLU-backward mask / accumulate.
// bf16 only; stride 1, dilation 1; the gather geometry (ay = 1, by, cy = +-1) covers forward and (stride-1) dgrad.
// ------------------------------------------------------------------------------------------------
// NW: waves per workgroup (4, or 8 for the two variants whose 80 KiB of LDS allow ONE workgroup per CU -- the 64-filter forward and the
// 64-channel dgrad of Conv2d_2b: eight waves give every SIMD two waves; each wave then owns two instead of four 16-pixel segments)
// EPI (dgrad launches: the host picks it when DIN_CONV_MASK / DIN_CONV_ACCUM is set): the epilogue's extra operands are requested at the top
// of the tile and held in registers through the MFMA phase; without it they are fetched inside the store loop (and cost no registers --
// compiled into the forward variants the prefetch registers slowed Conv2d_2b's forward by a third).
template <int CPP, int BN, int NBUF, int KH, int KW, int ST, bool U8 = false, int NW = 4, bool EPI = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void conv_small_kernel(ConvK p) {
    constexpr int NTHREADS = 64 * NW;                                  // (shadows the file-level 256)
    static_assert(!U8 || NW == 4, "the uint8 halo loader is written for four waves");
    static_assert(!U8 || (CPP == 1 && NBUF == 2), "uint8 frames feed the image layer only");
#if defined(__HIP_DEVICE_COMPILE__)
    typedef bf16_t T;
    constexpr int TH = 8, TW = 32, NPX = TH * TW;
    constexpr int TI = BN / 16, TJ = 16 / NW, NTAPS = KH * KW;
    // MFMA k-slices (32 channels-of-taps each): CPP >= 4: SL slices per tap; CPP == 1 (image layer, 8 padded channels per pixel):
    // four taps share one slice, lane group g4 carries tap 4*slice + g4 (taps >= NTAPS hit zero filter chunks of the packed bank)
    constexpr int SL = CPP >= 4 ? CPP / 4 : 1, NSL = CPP >= 4 ? NTAPS * SL : (NTAPS + 3) / 4;
    constexpr int CPITCH = BN * 2 + 16;
    constexpr int HWW = (TW - 1) * ST + KW, HWH = (TH - 1) * ST + KH, HPX = HWW * HWH, HC = HPX * CPP;  // halo geometry (pixels, chunks)
    constexpr int WBYTES = NSL * BN * 4 * 16;
    constexpr int HBYTES = (HC * 16 + 1023) / 1024 * 1024;                                // whole 1-KiB DMA slots
    constexpr int NSLOT = HBYTES / 1024, NTR = (NSLOT + NW - 1) / NW;
    constexpr int NPASS = HBYTES / CPITCH >= NPX ? 1 : 2;                                  // epilogue passes through the staging buffer
    static_assert(HBYTES / CPITCH >= NPX / NPASS, "staging does not fit the halo buffer");
    constexpr int JN = TJ / NPASS;                                                         // pixel segments per wave per pass
    constexpr int CPR = BN / 8;                                                            // 16-byte chunks per produced pixel
    constexpr int NST = (NPX / NPASS) * CPR / NTHREADS * NPASS;                            // store instructions per wave per tile
    static_assert((NPX / NPASS) * CPR % NTHREADS == 0, "store loop must be uniform");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    u32x4* Wl = reinterpret_cast<u32x4*>(smem_raw);
    auto swz = [](int row) { return CPP == 1 ? 0 : CPP == 4 ? ((row >> 1) & 3) : (row & 7); };

    // ---- filters: [tap][co][chunk ^ swz(co)]  (CPP == 1: [slice][co][tap & 3]) ---------------------------------------------
    {
        const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(p.w);
        if (CPP >= 4) {
            for (int id = tid; id < NTAPS * BN * CPP; id += NTHREADS) {
                const int row = id / CPP, slot = id - row * CPP;
                const int tap = row / BN, co = row - tap * BN;
                const int cc = slot ^ swz(co);
                Wl[id] = wp[(int64_t)co * p.wld + tap * CPP + cc];
            }
        } else {
            for (int id = tid; id < NSL * BN * 4; id += NTHREADS) {
                const int row = id >> 2, g = id & 3;
                const int sl = row / BN, co = row - sl * BN;
                Wl[id] = wp[(int64_t)co * p.wld + sl * 4 + g];          // sl*4+g < wld: the packed row is zero beyond the last tap
            }
        }
    }
    // ---- halo DMA plan: transfer i of this wave covers chunk ids [(wid + 4 i) * 64, +64) ------------------------------------
    int rel[NTR];                                      // byte offset relative to the halo origin pixel (chunk already permuted)
    short hyv[NTR], hxv[NTR];
#pragma unroll
    for (int i = 0; i < NTR; ++i) {
        const int id = (wid + NW * i) * 64 + lane;
        const int hp = id / CPP, slot = id - hp * CPP;
        const int cc = slot ^ swz(hp);
        const int hy = hp / HWW, hx = hp - hy * HWW;
        rel[i] = id < HC ? (hy * p.W + hx) * p.ldi * 2 + cc * 16 : -1;
        hyv[i] = (short)hy; hxv[i] = (short)hx;
    }
    const int hy0 = p.by + (p.cy < 0 ? (KH - 1) * p.cy : 0), hx0 = p.bx + (p.cx < 0 ? (KW - 1) * p.cx : 0);   // halo origin - tile origin
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y, ntiles = tiles_img * p.NB;
    const long long img_bytes = (long long)p.H * p.W * p.ldi * 2ll;
    const long long oimg_bytes = (long long)p.OH * p.OW * p.ldo * 2ll, mimg_bytes = (long long)p.OH * p.OW * p.ldm * 2ll;
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t ldsH0 = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)WBYTES + (uint32_t)(wid * 1024));

    auto issue_halo = [&](int buf, int tile) {
        const int n = tile / tiles_img;
        const int tr = tile - n * tiles_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int gy0 = ty * TH * ST + hy0, gx0 = tx * TW * ST + hx0;
        // one image per resource: 32-bit offsets always suffice; out-of-image pixels get the out-of-range offset -> zeros
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.in)) + (long long)n * img_bytes, 0, (int)img_bytes, 0x00020000);
        const int base = (gy0 * p.W + gx0) * p.ldi * 2 + p.cioff * 2;
        const uint32_t dst = ldsH0 + (uint32_t)(buf * HBYTES);
#pragma unroll
        for (int i = 0; i < NTR; ++i) {
            if (wid + NW * i < NSLOT) {                                                  // uniform: slot inside the halo buffer
                const int gy = gy0 + hyv[i], gx = gx0 + hxv[i];
                const bool ok = rel[i] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                lds_dma16(dst + (uint32_t)(i * 1024 * NW), rs, ok ? base + rel[i] : (int)OOB, 0);
            }
        }
    };

    // uint8 frames: the next tile's halo bytes are fetched into registers where the DMA would be issued and written to the other halo
    // buffer after this tile's MFMAs (nobody reads that buffer between the two barriers around them)
    // two register sets: the bytes of tile t + 2G are requested at the top of tile t and written to LDS at the end of tile t + 1 -- a whole tile
    // period for the loads to land (requested and consumed inside ONE tile their ~2 us were exposed every tile: 5.5 us per tile, 2.7 TB/s)
    U8Halo<NTR> u8s[2];
    bf16_t* u8lut = reinterpret_cast<bf16_t*>(smem_raw + WBYTES + NBUF * HBYTES);       // 512 bytes behind the halo buffers (host adds them)
    if (U8) { u8_lut_init(u8lut, tid); __syncthreads(); }
    auto u8_load = [&](int tile, U8Halo<NTR>& u8h) {
        const int n = tile / tiles_img;
        const int tr = tile - n * tiles_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        u8h.template load<NSLOT>(p.u8, n, p.H, p.W, ty * TH * ST + hy0, tx * TW * ST + hx0, wid, hyv, hxv, rel);
    };
    const int frow = lane & 15, g4 = lane >> 4;
    int cur = 0;
    // persistent walk: round i covers tiles [i*G, (i+1)*G); inside a round every XCD takes a contiguous run (shared halo rows hit L2)
    int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    bool first = true;
    if (U8) {
        if (tile < ntiles) {
            u8_load(tile, u8s[0]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); u8s[0].landed();
            u8s[0].template store<NSLOT>(smem_raw + WBYTES, u8lut, wid, lane);
        }
        if (tile + (int)gridDim.x < ntiles) u8_load(tile + gridDim.x, u8s[1]);
    } else if (tile < ntiles) issue_halo(0, tile);
    __syncthreads();                                                                   // filters visible
    // bias once per workgroup (a load inside the tile loop is a compiler-visible wait that also drains the next halo's transfers)
    f32x4 biasv[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int co = i * 16 + g4 * 4;
        biasv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((p.flags & DIN_CONV_BIAS) && co < p.Cout) biasv[i] = *reinterpret_cast<const f32x4*>(p.bias + co);
        asm volatile("" : "+v"(biasv[i]));                             // consumed HERE: the compiler's wait for the load stays out of the loop
    }
    auto tile_body = [&](auto PARC) {                                                  // PAR: which uint8 register set this tile FILLS
        constexpr int PAR = decltype(PARC)::value;
        // in-order completion: the halo transfers of this tile are older than the (always NST) stores of the previous tile when
        // double-buffered, so the stores may stay in flight; single-buffered the transfers are the youngest -> drain everything
        if (U8) {}                                                                     // (no transfers: the halo was written to LDS by the waves themselves)
        else if (NBUF == 2 && !first) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        first = false;
        __builtin_amdgcn_s_barrier();                                                  // halo(tile) landed; everyone left the previous tile
        asm volatile("" ::: "memory");
        const bool have_next = tile + (int)gridDim.x < ntiles;
        if (U8) { if (tile + 2 * (int)gridDim.x < ntiles) u8_load(tile + 2 * gridDim.x, u8s[PAR]); }
        else if (NBUF == 2 && have_next) issue_halo(cur ^ 1, tile + gridDim.x);
        const u32x4* Hl = reinterpret_cast<const u32x4*>(smem_raw + WBYTES + cur * HBYTES);
        // ---- dgrad epilogue operands (ReLU mask of the produced pixels, accumulate input): requested HERE, a whole MFMA phase before the
        //      epilogue uses them.  Fetched inside the store loop their HBM latency (~2 us) was exposed once per tile: the Conv2d_2a dgrad
        //      ran 1083 us inside the training step, 871 us with the early request (and 680 us with no mask at all).
        constexpr int NIT = (NPX / NPASS) * CPR / NTHREADS;
        const int n = tile / tiles_img;
        const int trm = tile - n * tiles_img;
        const int ty = trm / tiles_x, tx = trm - ty * tiles_x;
        __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (long long)n * oimg_bytes, 0, (int)oimg_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>((p.flags & DIN_CONV_MASK) ? p.mask : p.out)) + (long long)n * mimg_bytes, 0,
            (p.flags & DIN_CONV_MASK) ? (int)mimg_bytes : 0, 0x00020000);
        auto out_pixel = [&](int ps, int it, int& opx, int& co) -> bool {              // pixel / channel chunk of store `it` of pass `ps`
            const int idx = it * NTHREADS + tid;
            const int srow = idx / CPR, c = idx - srow * CPR;
            const int w_ = srow / (JN * 16), rem = srow - w_ * (JN * 16);
            const int q = w_ * TJ + ps * JN + rem / 16;                                // segment of the tile
            const int gy = ty * TH + (q >> 1), gx = tx * TW + (q & 1) * 16 + (rem & 15);
            co = c * 8;
            opx = gy * p.OW + gx;
            return gy < p.OH && gx < p.OW && co < p.Cout;
        };
        u32x4 mkP[EPI ? NPASS : 1][EPI ? NIT : 1], oldP[EPI ? NPASS : 1][EPI ? NIT : 1];
        if (EPI && (p.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM))) {
#pragma unroll
            for (int ps = 0; ps < (EPI ? NPASS : 1); ++ps)
#pragma unroll
                for (int it = 0; it < (EPI ? NIT : 1); ++it) {
                    int opx, co;
                    const bool ok = out_pixel(ps, it, opx, co);
                    mkP[ps][it] = u32x4{0u, 0u, 0u, 0u}; oldP[ps][it] = u32x4{0u, 0u, 0u, 0u};
                    if (p.flags & DIN_CONV_MASK)
                        mkP[ps][it] = __builtin_amdgcn_raw_buffer_load_b128(rsM, ok ? (opx * p.ldm + p.moff + co) * 2 : (int)OOB, 0, 0);
                    if (p.flags & DIN_CONV_ACCUM)
                        oldP[ps][it] = __builtin_amdgcn_raw_buffer_load_b128(rsO, ok ? (opx * p.ldo + p.cooff + co) * 2 : (int)OOB, 0, 0);
                }
        }

        f32x4 acc[TI][TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ysgn = p.cy > 0 ? 1 : -1, xsgn = p.cx > 0 ? 1 : -1;
        const int yb = p.cy > 0 ? 0 : KH - 1, xb = p.cx > 0 ? 0 : KW - 1;
        if (CPP >= 4) {
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) {
                const int r = tap / KW, s2 = tap - r * KW;
                const int dy = yb + ysgn * r, dx = xb + xsgn * s2;
#pragma unroll
                for (int sl = 0; sl < SL; ++sl) {
                    u32x4 wf[TI], xf[TJ];
                    const int chunk = sl * 4 + g4;
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
                        const int co = i * 16 + frow;
                        wf[i] = Wl[(tap * BN + co) * CPP + (chunk ^ swz(co))];
                    }
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        const int q = wid * TJ + j;                                     // 16-pixel segment of the tile
                        const int hp = ((q >> 1) * ST + dy) * HWW + ((q & 1) * 16 + frow) * ST + dx;
                        xf[j] = Hl[hp * CPP + (chunk ^ swz(hp))];
                    }
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
                }
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                u32x4 wf[TI], xf[TJ];
                const int tap = min(sl * 4 + g4, NTAPS - 1);                           // surplus taps: finite data x zero filter
                const int r = tap / KW, s2 = tap - r * KW;
                const int dy = yb + ysgn * r, dx = xb + xsgn * s2;
#pragma unroll
                for (int i = 0; i < TI; ++i) wf[i] = Wl[(sl * BN + i * 16 + frow) * 4 + g4];
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int q = wid * TJ + j;
                    xf[j] = Hl[((q >> 1) * ST + dy) * HWW + ((q & 1) * 16 + frow) * ST + dx];
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
            }
        }
        if (U8 && have_next) {
            // the bytes of tile + G were requested a tile ago; younger: the previous tile's stores and this tile's 3 NTR byte loads (if any)
            if (tile + 2 * (int)gridDim.x < ntiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NTR) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            u8s[PAR ^ 1].landed();
            u8s[PAR ^ 1].template store<NSLOT>(smem_raw + WBYTES + (cur ^ 1) * HBYTES, u8lut, wid, lane);
        }
        __syncthreads();                                                               // all waves done reading halo(cur)
        // ---- epilogue: stage through the consumed halo buffer, then 16-byte coalesced buffer stores (always NST per wave) -----
        unsigned char* stg = smem_raw + WBYTES + cur * HBYTES;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const f32x4 bv = biasv[i];
                const int co = i * 16 + g4 * 4;
#pragma unroll
                for (int j = ps * JN; j < (ps + 1) * JN; ++j) {
                    f32x4 v = acc[i][j] + bv;
                    if (p.flags & DIN_CONV_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    const int srow = (wid * JN + (j - ps * JN)) * 16 + frow;             // staging row of this pixel
                    *reinterpret_cast<u32x2*>(stg + srow * CPITCH + co * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int opx, co;
                const bool ok = out_pixel(ps, it, opx, co);
                const int idx = it * NTHREADS + tid;
                const int srow = idx / CPR, c = idx - srow * CPR;
                u32x4 v = *reinterpret_cast<const u32x4*>(stg + srow * CPITCH + c * 16);
                const int o = ok ? (opx * p.ldo + p.cooff + co) * 2 : (int)OOB;
                if (p.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM)) {
                    u32x4 mk = {0u, 0u, 0u, 0u}, old = {0u, 0u, 0u, 0u};
                    if (EPI) { mk = mkP[EPI ? ps : 0][EPI ? it : 0]; old = oldP[EPI ? ps : 0][EPI ? it : 0]; }
                    else {
                        if (p.flags & DIN_CONV_MASK) mk = __builtin_amdgcn_raw_buffer_load_b128(rsM, ok ? (opx * p.ldm + p.moff + co) * 2 : (int)OOB, 0, 0);
                        if (p.flags & DIN_CONV_ACCUM) old = __builtin_amdgcn_raw_buffer_load_b128(rsO, o, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                        if (p.flags & DIN_CONV_MASK) {
                            if (!(__uint_as_float(mk[e] << 16) > 0.f)) lo = 0.f;
                            if (!(__uint_as_float(mk[e] & 0xffff0000u) > 0.f)) hi = 0.f;
                        }
                        if (p.flags & DIN_CONV_ACCUM) { lo += __uint_as_float(old[e] << 16); hi += __uint_as_float(old[e] & 0xffff0000u); }
                        v[e] = pack_bf16x2(lo, hi);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(v, rsO, o, 0, 0);                  // out-of-range offset -> dropped, still counted
            }
            if (ps + 1 < NPASS) __syncthreads();
        }
        if (NBUF == 2) cur ^= 1;
        else {
            __syncthreads();                                                            // staging buffer free again
            if (tile + (int)gridDim.x < ntiles) issue_halo(0, tile + gridDim.x);
        }
    };
    if constexpr (U8) {
        for (;;) {
            if (tile >= ntiles) break;
            tile_body(IcTag<0>{}); tile += gridDim.x;
            if (tile >= ntiles) break;
            tile_body(IcTag<1>{}); tile += gridDim.x;
        }
    } else {
        for (; tile < ntiles; tile += gridDim.x) tile_body(IcTag<0>{});
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Halo-tiled convolution for the mid-network multi-tap layers (3x3, 1x7, 7x1; stride 1; bf16; fwd and dgrad).
// The implicit-GEMM kernel streams every input pixel once per tap and the filter slab once per 128 pixels; measured, it runs at
// (FLOP per streamed byte) x ~9.6 TB/s.  Here a workgroup owns a TH x TW = 256-pixel output tile: per 64-channel block the input
// HALO is brought into LDS once (all taps are formed from it) and only the filter slab of one (tap, channel block) moves through
// a small LDS-DMA ring: 2-3x fewer streamed bytes per FLOP.  Both LDS images use a 160-byte row pitch (8 data chunks + 2 pad
// chunks the DMA leaves empty): 16 consecutive rows x one chunk are then conflict-free for ds_read_b128 at ANY row offset, so a
// tap is nothing but an immediate offset on the fragment reads -- no per-tap address arithmetic (the first version of this kernel
// spent 46 % of its wave cycles issuing address / bookkeeping instructions; SQ counters in profiles/r01_halo_probe.txt).
// Persistent workgroups (one per CU) walk the (tile, filter tile) items; the next block's halo (also across items) is in flight
// during the current block's taps, the slab ring runs NSW-1 tap steps ahead, one s_barrier per tap step, vmcnt counted by hand
// (every wave issues the same number of transfers; surplus ones fetch nothing into pad space).  Epilogue straight from the
// accumulators as always-issued buffer stores (bias / ReLU / ReLU-backward mask / accumulate).
// ------------------------------------------------------------------------------------------------
template <int BN, int KH, int KW, int TH, int TW, int NSW, int NWV_ = 8>
__global__ __launch_bounds__(64 * NWV_, 1) void conv_halo_kernel(ConvK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef bf16_t T;
    // 8 waves: wave = (pixel group wp of 64 pixels, filter half wc): two waves per SIMD, so one wave's waits (barrier, vmcnt, LDS
    // latency) hide behind the other's MFMAs -- with 4 waves (one per SIMD) the kernel ran at 20 % MFMA utilisation
    // NWV_ = 16: pixel groups of 32 instead of 64 pixels (four waves per SIMD; the LDS budget then allows a 2-slot filter ring only)
    // BN = 80 (Conv2d_4a's data gradient, 192 -> 80 channels): the first filter half takes 48 rows (three 16-row tiles), the second 32 (two) --
    // a sixth fewer MFMAs than padding the bank to 96 rows; the two kinds of wave alternate on every SIMD (waves go to SIMDs round-robin)
    constexpr int NWV = NWV_, NTAPS = KH * KW, TI = (BN + 31) / 32, TJ = 32 / NWV;
    constexpr int WCR = BN == 80 ? 48 : BN / 2;                       // rows of the first filter half
    constexpr int HWW = TW + KW - 1, HWH = TH + KH - 1, HPX = HWW * HWH;
    constexpr int PCH = 10, PB = PCH * 16;                            // row pitch: 10 chunks = 160 bytes
    constexpr int NTR_H = (HPX * PCH + 64 * NWV - 1) / (64 * NWV), HBYTES = NTR_H * 1024 * NWV;
    constexpr int NTR_W = (BN * PCH + 64 * NWV - 1) / (64 * NWV), WBYTES = NTR_W * 1024 * NWV;
    static_assert(TH * TW == 256 && TW % 16 == 0 && NTAPS > NSW && (BN % 32 == 0 || BN == 80), "tile shape");
    constexpr int NST = TI * TJ;                                       // epilogue stores per wave per item
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int frow = lane & 15, g4 = lane >> 4;
    const int wp = wid & (NWV / 2 - 1), wc = __builtin_amdgcn_readfirstlane(wid / (NWV / 2));
    const int tiw = wc == 0 ? TI : (BN - WCR) / 16;                    // 16-row filter tiles of this wave (scalar)
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t ldsWv = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(wid * 1024));

    // ---- DMA plans ------------------------------------------------------------------------------------------------------------
    // (the halo plan is recomputed per transfer at issue time -- once per 64-channel block; as per-lane arrays it cost ~60 VGPRs and
    //  pushed the tap loop into AGPR spills)
    int relW[NTR_W]; signed char wcc[NTR_W];
#pragma unroll
    for (int i = 0; i < NTR_W; ++i) {
        const int id = (wid + NWV * i) * 64 + lane;
        const int co = id / PCH, cc = id - co * PCH;
        relW[i] = (co < BN && cc < 8) ? co * p.wld * 16 + cc * 16 : -1;  // + filter-tile row offset + (tap * cpt + 8 b) * 16
        wcc[i] = (signed char)cc;
    }
    const int hy0 = p.by + (p.cy < 0 ? (KH - 1) * p.cy : 0), hx0 = p.bx + (p.cx < 0 ? (KW - 1) * p.cx : 0);
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y, nitems = tiles_img * p.NB * p.n_co_tiles;
    const long long img_bytes = (long long)p.H * p.W * p.ldi * 2ll;
    const long long oimg_bytes = (long long)p.OH * p.OW * p.ldo * 2ll, mimg_bytes = (long long)p.OH * p.OW * p.ldm * 2ll;
    const int nblk = (p.cpt + 7) >> 3;                                  // 64-channel blocks
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

    struct Item { int n, ty, tx, co_tile; };
    auto decode_item = [&](int item) {
        Item it;
        it.co_tile = item % p.n_co_tiles;
        const int tile = item / p.n_co_tiles;
        it.n = tile / tiles_img;
        const int tr = tile - it.n * tiles_img;
        it.ty = tr / tiles_x; it.tx = tr - it.ty * tiles_x;
        return it;
    };
    auto issue_halo = [&](int buf, const Item& it, int b) {
        const int gy0 = it.ty * TH + hy0, gx0 = it.tx * TW + hx0;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.in)) + (long long)it.n * img_bytes, 0, (int)img_bytes, 0x00020000);
        const int base = (gy0 * p.W + gx0) * p.ldi * 2 + (p.cioff + b * 64) * 2;
        const int nch = p.cpt - b * 8;                                  // chunks of this block that exist (>= 8: all)
        const uint32_t dst = ldsWv + (uint32_t)(buf * HBYTES);
#pragma unroll 2
        for (int i = 0; i < NTR_H; ++i) {
            const int id = (wid + NWV * i) * 64 + lane;
            const int hp = (int)(((unsigned)id * 52429u) >> 19);         // id / 10 for id < 81920
            const int cc = id - hp * PCH;
            const int hy = (int)(((unsigned)hp * (65536u / HWW + 1u)) >> 16), hx = hp - hy * HWW;   // hp / HWW (hp < 1024)
            const int gy = gy0 + hy, gx = gx0 + hx;
            const bool ok = hp < HPX && cc < 8 && cc < nch && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            lds_dma16(dst + (uint32_t)(i * 1024 * NWV), rs, ok ? base + (hy * p.W + hx) * p.ldi * 2 + cc * 16 : (int)OOB, 0);
        }
    };
    auto issue_w = [&](int slot, int co_tile, int b, int tap) {
        const int nch = p.cpt - b * 8;
        const int soff = (co_tile * BN * p.wld + tap * p.cpt + b * 8) * 16;
        const uint32_t dst = ldsWv + (uint32_t)(2 * HBYTES + slot * WBYTES);
#pragma unroll
        for (int i = 0; i < NTR_W; ++i) lds_dma16(dst + (uint32_t)(i * 1024 * NWV), rsW, (relW[i] >= 0 && wcc[i] < nch) ? relW[i] : (int)OOB, soff);
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment read bases (bytes): pixel segment j of this wave at tap (0,0) of the gather direction, chunk g4; filter row frow, chunk g4
    const int yb = p.cy > 0 ? 0 : KH - 1, xb = p.cx > 0 ? 0 : KW - 1;
    const int tstep_y = (p.cy > 0 ? 1 : -1) * HWW * PB, tstep_x = (p.cx > 0 ? 1 : -1) * PB;     // byte step of one tap row / column
    uint32_t xbase[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int q = wp * TJ + j;
        const int qy = TW == 32 ? (q >> 1) : q, qx = TW == 32 ? (q & 1) * 16 : 0;
        xbase[j] = (uint32_t)(((qy + yb) * HWW + qx + frow + xb) * PB + g4 * 16);
    }
    const uint32_t wbase = (uint32_t)(2 * HBYTES + (wc * WCR + frow) * PB + g4 * 16);

    constexpr int NBP = NWV == 16 ? 1 : 2;                             // (sixteen waves: 128 registers)
    f32x4 biasP[NBP][TI];                                               // bias of the first NBP filter tiles (see the epilogue)
#pragma unroll
    for (int ct = 0; ct < NBP; ++ct)
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int co = ct * BN + wc * WCR + i * 16 + g4 * 4;
            biasP[ct][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((p.flags & DIN_CONV_BIAS) && co < p.Cout) biasP[ct][i] = *reinterpret_cast<const f32x4*>(p.bias + co);
            asm volatile("" : "+v"(biasP[ct][i]));                      // consumed here: the compiler's wait stays out of the walk
        }
    // ---- (item, channel block) pair walk ----------------------------------------------------------------------------------------
    const int G = (int)gridDim.x;
    int item = xcd_remap((int)blockIdx.x, G);
    if (item >= nitems) return;
    Item cur = decode_item(item);
    int b = 0, hb = 0;
    int item_w = item, b_w = 0, tap_w = 0;                              // (item, block, tap) of the next filter slab to issue
    int co_w = cur.co_tile;
    bool w_live = true;
    auto advance_w = [&]() {
        if (++tap_w == NTAPS) {
            tap_w = 0;
            if (++b_w == nblk) { b_w = 0; item_w += G; w_live = item_w < nitems; if (w_live) co_w = item_w % p.n_co_tiles; }
        }
    };
    issue_halo(0, cur, 0);
    int wslot_issue = 0;
#pragma unroll
    for (int s0 = 0; s0 < NSW - 1; ++s0) {
        if (w_live) { issue_w(wslot_issue, co_w, b_w, tap_w); advance_w(); }
        wslot_issue = wslot_issue + 1 == NSW ? 0 : wslot_issue + 1;
    }
    int wslot = 0;                                                      // ring slot of the current step
    bool fresh_item = false;                                            // an epilogue's stores were issued since the last tap step
    for (;;) {
        int item_n = item, b_n = b + 1;
        if (b_n == nblk) { b_n = 0; item_n = item + G; }
        const bool has_next = item_n < nitems;
        const Item nxt = (b_n == 0 && has_next) ? decode_item(item_n) : cur;
        const uint32_t hoff = (uint32_t)(hb * HBYTES);
        const bool two = p.cpt - b * 8 > 4;                             // second 32-channel slice present
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            // ---- slab of this step landed?  Younger transfers that may stay in flight (in-order completion): the NSW-2 slabs issued
            //      after it, the halo issued at this pair's tap-0 step (taps 1 .. NSW-1), the stores of an epilogue issued since ----
            {
                constexpr int base_allow = (NSW - 2) * NTR_W;
                const bool halo_tap = tap >= 1 && tap <= NSW - 1;
                const bool store_tap = tap <= NSW - 2;
                constexpr int a_hs = base_allow + NTR_H + NST > 63 ? 63 : base_allow + NTR_H + NST;
                constexpr int a_s = base_allow + NST > 63 ? 63 : base_allow + NST;
                const bool hy_ = halo_tap && has_next, st_ = store_tap && fresh_item;
                if (!w_live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // tail of the walk: drain
                else if (hy_ && st_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(a_hs) : "memory");
                else if (hy_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(base_allow + NTR_H) : "memory");
                else if (st_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(a_s) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(base_allow) : "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (w_live) { issue_w(wslot_issue, co_w, b_w, tap_w); advance_w(); }
            wslot_issue = wslot_issue + 1 == NSW ? 0 : wslot_issue + 1;
            if (tap == 0 && has_next) issue_halo(hb ^ 1, nxt, b_n);
            // ---- MFMAs of (block b, tap): every fragment address = lane base + compile-time offset ------------------------------------
            const int r = tap / KW, s2 = tap - r * KW;
            const uint32_t xoff = hoff + (uint32_t)(r * tstep_y + s2 * tstep_x);
            const uint32_t woff = (uint32_t)(wslot * WBYTES);
            u32x4 wf[2][TI], xf[2][TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) if (i < tiw) wf[0][i] = *reinterpret_cast<const u32x4*>(smem_raw + wbase + woff + i * 16 * PB);
#pragma unroll
            for (int j = 0; j < TJ; ++j) xf[0][j] = *reinterpret_cast<const u32x4*>(smem_raw + xbase[j] + xoff);
            __builtin_amdgcn_sched_barrier(0);
            if (two) {
#pragma unroll
                for (int i = 0; i < TI; ++i) if (i < tiw) wf[1][i] = *reinterpret_cast<const u32x4*>(smem_raw + wbase + woff + i * 16 * PB + 64);
#pragma unroll
                for (int j = 0; j < TJ; ++j) xf[1][j] = *reinterpret_cast<const u32x4*>(smem_raw + xbase[j] + xoff + 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
                if (i < tiw) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j) Mma<T>::run(wf[0][i], xf[0][j], acc[i][j]);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (two) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    if (i < tiw) {
#pragma unroll
                        for (int j = 0; j < TJ; ++j) Mma<T>::run(wf[1][i], xf[1][j], acc[i][j]);
                    }
            }
            wslot = wslot + 1 == NSW ? 0 : wslot + 1;
            if (tap == NSW - 2) fresh_item = false;
        }
        // ---- end of the item: epilogue straight from the accumulators ----------------------------------------------------------------
        if (b == nblk - 1) {
            __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (long long)cur.n * oimg_bytes, 0, (int)oimg_bytes, 0x00020000);
            __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>((p.flags & DIN_CONV_MASK) ? p.mask : p.out)) + (long long)cur.n * mimg_bytes, 0,
                (p.flags & DIN_CONV_MASK) ? (int)mimg_bytes : 0, 0x00020000);
            // bias: preloaded before the walk for the first NBP filter tiles (a compiler-visible load here waits with vmcnt(0): it would
            // drain the slab ring and the next halo once per item)
            f32x4 bv[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int co = cur.co_tile * BN + wc * WCR + i * 16 + g4 * 4;
                if (cur.co_tile < NBP) bv[i] = cur.co_tile == 0 ? biasP[0][i] : biasP[NBP - 1][i];
                else { bv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; if ((p.flags & DIN_CONV_BIAS) && co < p.Cout) bv[i] = *reinterpret_cast<const f32x4*>(p.bias + co); }
            }
            // dgrad operands: ALL of the item's loads first, then all its stores (segment by segment the loads of segment j + 1 waited for the
            // stores of segment j -- in-order completion -- i.e. TJ exposed round trips per item)
            u32x2 mk[TJ][TI], old[TJ][TI];
            int off[TJ][TI];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int q = wp * TJ + j;
                const int qy = TW == 32 ? (q >> 1) : q, qx = TW == 32 ? (q & 1) * 16 : 0;
                const int gy = cur.ty * TH + qy, gx = cur.tx * TW + qx + frow;
                const bool pok = gy < p.OH && gx < p.OW;
                const int opx = gy * p.OW + gx;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const int co = cur.co_tile * BN + wc * WCR + i * 16 + g4 * 4;
                    const bool ok = pok && co < p.Cout;                      // Cout % 4 == 0 (host)
                    off[j][i] = ok ? (opx * p.ldo + p.cooff + co) * 2 : (int)OOB;
                    if (p.flags & DIN_CONV_MASK) mk[j][i] = __builtin_amdgcn_raw_buffer_load_b64(rsM, ok ? (opx * p.ldm + p.moff + co) * 2 : (int)OOB, 0, 0);
                    if (p.flags & DIN_CONV_ACCUM) old[j][i] = __builtin_amdgcn_raw_buffer_load_b64(rsO, off[j][i], 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    f32x4 v = acc[i][j] + bv[i];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.flags & DIN_CONV_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (p.flags & DIN_CONV_MASK) {
                        if (!(__uint_as_float(mk[j][i][0] << 16) > 0.f)) v[0] = 0.f;
                        if (!(__uint_as_float(mk[j][i][0] & 0xffff0000u) > 0.f)) v[1] = 0.f;
                        if (!(__uint_as_float(mk[j][i][1] << 16) > 0.f)) v[2] = 0.f;
                        if (!(__uint_as_float(mk[j][i][1] & 0xffff0000u) > 0.f)) v[3] = 0.f;
                    }
                    if (p.flags & DIN_CONV_ACCUM) {
                        v[0] += __uint_as_float(old[j][i][0] << 16); v[1] += __uint_as_float(old[j][i][0] & 0xffff0000u);
                        v[2] += __uint_as_float(old[j][i][1] << 16); v[3] += __uint_as_float(old[j][i][1] & 0xffff0000u);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}, rsO, off[j][i], 0, 0);
                }
            fresh_item = true;
        }
        if (!has_next) break;
        item = item_n; b = b_n; hb ^= 1;
        if (b == 0) cur = nxt;
    }
#endif
}

// split-K finish: out = epilogue(sum_s partial[s])
template <typename T>
__global__ void conv_splitk_finish_kernel(ConvK p, int cpad) {
    int64_t total = (int64_t)p.M * p.Cout;
    T* __restrict__ outp = reinterpret_cast<T*>(p.out);
    const T* __restrict__ maskp = reinterpret_cast<const T*>(p.mask);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int m = (int)(i / p.Cout), co = (int)(i - (int64_t)m * p.Cout);
        float x = 0.f;
        for (int s = 0; s < p.splitk; ++s) x += p.partial[((int64_t)s * p.M + m) * cpad + co];
        if (p.flags & DIN_CONV_BIAS) x += p.bias[co];
        if (p.flags & DIN_CONV_RELU) x = fmaxf(x, 0.f);
        const int64_t opx = out_pixel(p, m);
        if (p.flags & DIN_CONV_MASK) {
            float y = Elem<T>::ld(maskp + opx * p.ldm + p.moff + co);
            x = y > 0.f ? x : 0.f;
        }
        int64_t o = opx * p.ldo + p.cooff + co;
        if (p.flags & DIN_CONV_ACCUM) x += Elem<T>::ld(outp + o);
        Elem<T>::st(outp + o, x);
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing:  w[cout][cin][kh][kw] fp32 (* scale[cout]) -> T[rows_pad][nk*KC*EPC]
//   transposed = 0: row = co, k = (r,s,ci)      (fwd)
//   transposed = 1: row = ci, k = (r,s,co)      (dgrad)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void conv_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ out,
                                 int cout, int cin, int kh, int kw, int rows, int rows_pad, int inner, int inner_pad,
                                 int kelems, int transposed) {
    // inner = reduction channels per tap (cin or cout), inner_pad = padded to EPC; kelems = padded row length
    int64_t total = (int64_t)rows_pad * kelems;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int row = (int)(i / kelems), k = (int)(i - (int64_t)row * kelems);
        int tap = k / inner_pad, c = k - tap * inner_pad;
        float v = 0.f;
        if (row < rows && tap < kh * kw && c < inner) {
            int r = tap / kw, s = tap - r * kw;
            int co = transposed ? c : row, ci = transposed ? row : c;
            v = w[(((int64_t)co * cin + ci) * kh + r) * kw + s];
            if (scale) v *= scale[co];
        }
        Elem<T>::st(out + i, v);
    }
}

// 1x1 filter banks (the 26400 x 1024 embedding filter is re-packed twice per step: 108 MB in, 54 MB out per orientation): 64 x 64 tiles
// through LDS, reads coalesced along the filter's contiguous axis (ci), writes coalesced along the packed row -- instead of one element
// per thread with two 64-bit divisions (106 -> ~35 us per orientation).  out[row][k]: row = co (k = ci) or, transposed, row = ci (k = co).
template <typename T>
__global__ __launch_bounds__(256) void conv_pack_1x1_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ out,
                                                            int cout, int cin, int rows_pad, int kelems, int transposed) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64;                  // tile of the packed matrix
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;                // 64 x 4
    if (!transposed) {
#pragma unroll 4
        for (int rr = ty; rr < 64; rr += 4) {
            const int co = r0 + rr, ci = k0 + tx;
            float v = 0.f;
            if (co < cout && ci < cin) { v = w[(int64_t)co * cin + ci]; if (scale) v *= scale[co]; }
            if (co < rows_pad && ci < kelems) Elem<T>::st(out + (int64_t)co * kelems + ci, v);
        }
        return;
    }
    // transposed: packed row = ci, packed column = co; read w[co][ci] with ci fastest
#pragma unroll 4
    for (int cc = ty; cc < 64; cc += 4) {
        const int co = k0 + cc, ci = r0 + tx;
        float v = 0.f;
        if (co < cout && ci < cin) { v = w[(int64_t)co * cin + ci]; if (scale) v *= scale[co]; }
        tile[cc][tx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = ty; rr < 64; rr += 4) {
        const int ci = r0 + rr, co = k0 + tx;
        if (ci < rows_pad && co < kelems) Elem<T>::st(out + (int64_t)ci * kelems + co, tile[tx][rr]);
    }
}

// every filter bank of a backbone (both orientations) in one launch: workgroup b packs elements [chunk_index[b] * chunk, +chunk) of
// bank layer_of[b]; the table lives on the device and is built once (weights, scales and packed buffers keep their addresses)
__global__ __launch_bounds__(256) void conv_pack_multi_kernel(const din_pack_desc* __restrict__ table, const int32_t* __restrict__ layer_of,
                                                              const int32_t* __restrict__ chunk_index, int chunk) {
    const din_pack_desc d = table[layer_of[blockIdx.x]];
    const float* __restrict__ w = reinterpret_cast<const float*>(d.w);
    const float* __restrict__ scale = reinterpret_cast<const float*>(d.scale);
    const int64_t total = (int64_t)d.rows_pad * d.kelems;
    const int64_t i0 = (int64_t)chunk_index[blockIdx.x] * chunk;
    int64_t i1 = i0 + chunk;
    if (i1 > total) i1 = total;
    const int taps = d.kh * d.kw;
    // (banks below 2^31 elements -- every backbone bank -- take 32-bit index arithmetic: the 64-bit divisions cost more than the traffic)
    const bool small = total < (1ll << 31);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        int row, k;
        if (small) { row = (int)((uint32_t)i / (uint32_t)d.kelems); k = (int)((uint32_t)i - (uint32_t)row * (uint32_t)d.kelems); }
        else { row = (int)(i / d.kelems); k = (int)(i - (int64_t)row * d.kelems); }
        const int tap = k / d.inner_pad, c = k - tap * d.inner_pad;
        float v = 0.f;
        if (row < d.rows && tap < taps && c < d.inner) {
            const int r = tap / d.kw, s2 = tap - r * d.kw;
            const int co = d.transposed ? c : row, ci = d.transposed ? row : c;
            v = small ? w[(uint32_t)(((co * d.cin + ci) * d.kh + r) * d.kw + s2)] : w[(((int64_t)co * d.cin + ci) * d.kh + r) * d.kw + s2];
            if (scale) v *= scale[co];
        }
        if (d.dtype == DIN_F32) reinterpret_cast<float*>(d.out)[i] = v;
        else reinterpret_cast<bf16_t*>(d.out)[i] = f32_to_bf16(v);
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad:  dW[co][(r,s,ci)] = sum_pix G[pix][co] * im2col(X)[pix][(r,s,ci)]
// 128 (co) x 128 (k columns) tile per workgroup, reduction over a slice of the pixels; partials to a
// workspace [slice][cout_pad][kcols_pad] fp32, reduced (and un-permuted to [cout][cin][kh][kw]) afterwards.
// ------------------------------------------------------------------------------------------------
constexpr int WG_TILE = 128;

// fp32: 16 pixels per k-step, operands read with ds_read_b32 (lane k-index = pixel row)
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_f32_kernel(WgradK p) {
    constexpr int PK = 16;
    constexpr int RS = WG_TILE + 16;     // padded row (floats): 4 k-rows hit 4 disjoint bank ranges
    __shared__ float Gs[2][PK][RS];
    __shared__ float Xs[2][PK][RS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int bid, slice_;
    xcd_block(bid, slice_);
    const int k_tile = bid % p.n_k_tiles;
    const int co_tile = bid / p.n_k_tiles;
    const int slice = slice_;
    const int m_begin = slice * p.m_per_slice;
    int m_end = m_begin + p.m_per_slice;
    if (m_end > p.M) m_end = p.M;

    // loader: 16 rows x 32 chunks (of 4 floats) per operand -> 2 chunks per thread per operand
    const int cc = tid & 31, rr = tid >> 5;         // chunk column 0..31, row 0..7 (+8)
    // fixed k column of this thread's X chunk
    const int kcol = k_tile * WG_TILE + cc * 4;
    const bool kok = kcol < p.kcols;
    const int tap = kok ? kcol / p.cin_pad : 0;
    const int ci = kcol - tap * p.cin_pad;
    const int r = tap / p.kw, s = tap - r * p.kw;
    const bool ci_ok = kok && ci < p.Cin;           // Cin % 4 == 0 is enforced by the host unless cin_pad>Cin (conv1)
    const int gco = co_tile * WG_TILE + cc * 4;
    const float* __restrict__ inp = reinterpret_cast<const float*>(p.in);
    const float* __restrict__ gp = reinterpret_cast<const float*>(p.g);

    // pixel coordinates of the two rows this thread loads, advanced incrementally (no divisions in the loop)
    int pn[2], py[2], px[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m_begin + rr + 8 * i;
        int n = m / (p.OH * p.OW);
        int rem = m - n * (p.OH * p.OW);
        pn[i] = n; py[i] = rem / p.OW; px[i] = rem - py[i] * p.OW;
    }
    f32x4 xa[2], ga[2];
    auto load_global = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + rr + 8 * i;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
                int iy = py[i] * p.sh - p.ph + r * p.dh, ix = px[i] * p.sw - p.pw + s * p.dw;
                if (ci_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    const float* src = inp + (int64_t)((pn[i] * p.H + iy) * p.W + ix) * p.ldi + p.cioff + ci;
                    if (ci + 3 < p.Cin) xv = *reinterpret_cast<const f32x4*>(src);
                    else for (int e = 0; e < 4 && ci + e < p.Cin; ++e) xv[e] = src[e];
                }
                if (gco < p.Cout) {
                    const float* src = gp + (int64_t)m * p.ldo + p.cooff + gco;
                    if (gco + 3 < p.Cout) gv = *reinterpret_cast<const f32x4*>(src);
                    else for (int e = 0; e < 4 && gco + e < p.Cout; ++e) gv[e] = src[e];
                }
            }
            xa[i] = xv; ga[i] = gv;
            // advance this row by PK pixels
            px[i] += PK;
            while (px[i] >= p.OW) { px[i] -= p.OW; if (++py[i] == p.OH) { py[i] = 0; ++pn[i]; } }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&Xs[buf][rr + 8 * i][cc * 4]) = xa[i];
            *reinterpret_cast<f32x4*>(&Gs[buf][rr + 8 * i][cc * 4]) = ga[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fcol = lane & 15, frow = lane >> 4;
    if (m_begin < m_end) {
        load_global(m_begin);
        store_lds(0);
        __syncthreads();
        int it = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += PK, ++it) {
            const int cur = it & 1;
            const bool more = m0 + PK < m_end;
            if (more) load_global(m0 + PK);
#pragma unroll
            for (int kk = 0; kk < PK / 4; ++kk) {
                float gf[4], xf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) gf[i] = Gs[cur][kk * 4 + frow][wm * 64 + i * 16 + fcol];
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = Xs[cur][kk * 4 + frow][wn * 64 + j * 16 + fcol];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[i], xf[j], acc[i][j], 0, 0, 0);
            }
            if (more) store_lds(cur ^ 1);
            __syncthreads();
        }
    }
    // D[i = co][j = kcol]: lane holds co = ..+(lane>>4)*4+e, kcol = ..+(lane&15)
    float* dst = p.partial + (int64_t)slice * p.cout_pad * p.kcols_pad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int co = co_tile * WG_TILE + wm * 64 + i * 16 + (lane >> 4) * 4;
            int kc = k_tile * WG_TILE + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(int64_t)(co + e) * p.kcols_pad + kc] = acc[i][j][e];
        }
}

// bf16: 32 pixels per k-step; operands are stored [pixel][channel] in LDS (as they sit in HBM) and read with the
// gfx950 transpose read ds_read_b64_tr_b16, which hands lane i of a 16-lane group column i of a 4x16 block.
// k (pixel) order inside the k-step: lane group g, element e -> pixel 4g+e (e<4) or 16+4g+(e-4): the two
// 32-lane halves of each ds_read_b64 then cover 8 consecutive rows = one full 256-byte bank row (RS pad 32 B).
__device__ __forceinline__ u32x2 lds_tr_read(uint32_t byte_addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(byte_addr) : "memory");
    return r;
}

// v3: 64 pixels per k-step; operands arrive by LDS-DMA (buffer_load ... lds, issued through inline asm and counted by hand like the
// gather kernel): G: constant per-lane offset + scalar row offset; X: per-row offset advanced incrementally, out-of-image taps ->
// hardware zero.  LDS-DMA is lane-linear, so the tiles are UNPADDED [pixel][channel] images; bank conflicts of the transpose reads
// are avoided by rotating each row's 16-byte chunks by 2*(row & 7) -- applied on the SOURCE side (the lane fetches the logical chunk
// that belongs at its slot) and in the read addresses.  BCO in {64,96,128,160} filter rows x 128 k columns per workgroup; the bias
// gradient is fused in: the k_tile == 0 workgroups' kcol-half-0 waves also multiply their G fragments with an all-ones operand.
template <int BCO>
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_bf16_kernel(WgradK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PK = 64;
    constexpr int CG = BCO / 8, CX = WG_TILE / 8;                 // 16-byte chunks per G / X row
    constexpr int RBG = BCO * 2, RBX = WG_TILE * 2;               // row bytes (unpadded)
    constexpr int OPG = PK * RBG, OPX = PK * RBX, STAGE = OPG + OPX;
    constexpr int TI = BCO / 32;                                  // 16-row filter tiles per wave (wave tile = BCO/2 x 64)
    constexpr int GP = PK * CG / NTHREADS;                        // G DMA chunks per thread per stage (= BCO/32)
    constexpr int XP = PK * CX / NTHREADS;                        // X DMA chunks per thread per stage (= 4)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int bx_, by_;
    xcd_block(bx_, by_);
    const int k_tile = bx_ % p.n_k_tiles, co_tile = bx_ / p.n_k_tiles;
    const int slice = by_;
    const int m_begin = slice * p.m_per_slice;                   // multiple of PK
    int m_end = m_begin + p.m_per_slice;
    if (m_end > p.M) m_end = p.M;

    // ---- buffer resources (base moved to the slice's first image / first pixel so 32-bit offsets always suffice) ----
    const int ohw = p.OH * p.OW;
    const int n_first = m_begin / ohw;
    const long long img_bytes = (long long)p.H * p.W * p.ldi * 2ll;
    const long long x_off = (long long)n_first * img_bytes;
    long long x_rem = (long long)p.NB * img_bytes - x_off;
    if (x_rem > 0x7fffffffll) x_rem = 0x7fffffffll;
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in)) + x_off, 0, (int)x_rem, 0x00020000);
    const long long g_off = (long long)m_begin * p.ldo * 2ll;
    long long g_rem = (long long)p.M * p.ldo * 2ll - g_off;
    if (g_rem > 0x7fffffffll) g_rem = 0x7fffffffll;
    __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.g)) + g_off, 0, (int)g_rem, 0x00020000);

    // ---- G DMA: wave-level transfer t of this wave covers chunk ids [(wid + 4 t) * 64, +64): id -> (row id / CG, slot id % CG);
    //      the lane fetches logical chunk (slot - 2*(row & 7)) mod CG of that row ---------------------------------------------
    unsigned voffG[GP];
#pragma unroll
    for (int t = 0; t < GP; ++t) {
        const int id = (wid + 4 * t) * 64 + lane;
        const int grow = id / CG, slot = id - grow * CG;
        int gc = slot - 2 * (grow & 7);
        gc += gc < 0 ? CG : 0;
        gc += gc < 0 ? CG : 0;                                      // 2*(row&7) <= 14 may exceed CG = 8 or 12 once
        const int gco = co_tile * BCO + gc * 8;
        voffG[t] = (gco + 7 < p.Cout) ? (unsigned)((grow * p.ldo + p.cooff + gco) * 2) : OOB;   // Cout % 8 == 0 enforced
    }
    // ---- X DMA: transfer t covers rows (wid + 4 t) * 4 + (lane >> 4); rotation 2*(row & 7) is the same for all t, so the lane's
    //      logical chunk -- hence its (tap, ci) -- is fixed -------------------------------------------------------------------
    const int xrow0 = wid * 4 + (lane >> 4);                       // rows xrow0 + 16 t
    int xc = (lane & 15) - 2 * (xrow0 & 7);
    xc += xc < 0 ? CX : 0;
    const int kcol = k_tile * WG_TILE + xc * 8;
    const bool kok = kcol < p.kcols;
    const int tap = kok ? kcol / p.cin_pad : 0;
    const int ci = kcol - tap * p.cin_pad;
    const int tr_ = tap / p.kw, ts_ = tap - tr_ * p.kw;
    const bool ci_ok = kok && ci + 7 < p.Cin;                       // Cin % 8 == 0 enforced (conv1 uses the tail kernel)
    const int dy0 = -p.ph + tr_ * p.dh, dx0 = -p.pw + ts_ * p.dw;   // iy = oy*sh + dy0, ix = ox*sw + dx0
    const int step_bytes = p.sw * p.ldi * 2;                        // +1 output column
    int px[XP], py[XP];          // output coordinates of this thread's rows
    int rowoff[XP];              // byte offset of (n, iy, ix = dx0) for the current (n, oy): may be "virtual"
#pragma unroll
    for (int t = 0; t < XP; ++t) {
        int m = m_begin + xrow0 + 16 * t;
        int n = m / ohw;
        int rem = m - n * ohw;
        py[t] = rem / p.OW; px[t] = rem - py[t] * p.OW;
        rowoff[t] = (((n - n_first) * p.H + py[t] * p.sh + dy0) * p.W + dx0) * p.ldi * 2 + (p.cioff + ci) * 2;
    }
    const int row_jump = p.sh * p.W * p.ldi * 2;                    // +1 output row
    const int img_jump = (p.H - p.OH * p.sh) * p.W * p.ldi * 2;     // extra when wrapping to the next image

    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t ldsW = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(wid * 1024));   // this wave's first 1-KiB slot
    auto issue_dma = [&](int buf, int m0) {
        const uint32_t Gd = ldsW + (uint32_t)(buf * STAGE), Xd = Gd + (uint32_t)OPG;
        const int soffG = (m0 - m_begin) * p.ldo * 2;                // uniform
#pragma unroll
        for (int t = 0; t < GP; ++t) lds_dma16(Gd + (uint32_t)(t * 4096), rsG, (int)voffG[t], soffG);
#pragma unroll
        for (int t = 0; t < XP; ++t) {
            const int iy = py[t] * p.sh + dy0, ix = px[t] * p.sw + dx0;
            const bool ok = ci_ok && (m0 + xrow0 + 16 * t < m_end) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const unsigned vo = ok ? (unsigned)(rowoff[t] + px[t] * step_bytes) : OOB;
            lds_dma16(Xd + (uint32_t)(t * 4096), rsX, (int)vo, 0);
            // advance this row by PK output pixels
            px[t] += PK;
            while (px[t] >= p.OW) {
                px[t] -= p.OW; rowoff[t] += row_jump;
                if (++py[t] == p.OH) { py[t] = 0; rowoff[t] += img_jump; }
            }
        }
    };

    f32x4 acc[TI][4], accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = p.dbias != nullptr && k_tile == 0 && wn == 0;       // wave-uniform
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    // transpose-read addressing: lane i of a 16-lane group supplies the 8-byte piece (row 4*lg + (i>>2), cols 4*(i&3)..+3) of a
    // 16-column tile, i.e. chunk (tile_chunk + ((i&3)>>1)), half (i&1); rows of the second read are +16 (same row & 7 -> same rotation)
    const int li = lane & 15, lg = lane >> 4;
    const int prow = 4 * lg + (li >> 2);
    const int rot = 2 * (prow & 7);
    uint32_t colG[TI], colX[4];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        int ch = (wm * (BCO / 2) + i * 16) / 8 + rot;                         // even
        ch -= ch >= CG ? CG : 0;
        ch -= ch >= CG ? CG : 0;
        colG[i] = (uint32_t)(prow * RBG + (ch + ((li & 3) >> 1)) * 16 + (li & 1) * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ch = (wn * 64 + j * 16) / 8 + rot;
        ch -= ch >= CX ? CX : 0;
        colX[j] = (uint32_t)(prow * RBX + (ch + ((li & 3) >> 1)) * 16 + (li & 1) * 8);
    }
    auto compute = [&](int cur) {
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {                                       // two 32-pixel MFMA k-groups per stage
            const uint32_t Gb = lds_base + cur * STAGE + kg * 32 * RBG;
            const uint32_t Xb = lds_base + cur * STAGE + OPG + kg * 32 * RBX;
            u32x4 gf[TI], xf[4];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                u32x2 lo = lds_tr_read(Gb + colG[i]), hi = lds_tr_read(Gb + colG[i] + 16 * RBG);
                gf[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x2 lo = lds_tr_read(Xb + colX[j]), hi = lds_tr_read(Xb + colX[j] + 16 * RBX);
                xf[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[i]),
                                                                        __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[i]),
                                                                      __builtin_bit_cast(bf16x8, ones), accb[i], 0, 0, 0);
            }
        }
    };

    if (m_begin < m_end) {
        issue_dma(0, m_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int it = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += PK, ++it) {
            const int cur = it & 1;
            if (m0 + PK < m_end && p.probe != 2) issue_dma(cur ^ 1, m0 + PK);
            if (p.probe != 1) compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    float* dst = p.partial + (int64_t)slice * p.cout_pad * p.kcols_pad;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int co = co_tile * BCO + wm * (BCO / 2) + i * 16 + (lane >> 4) * 4;
            int kc = k_tile * WG_TILE + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(int64_t)(co + e) * p.kcols_pad + kc] = acc[i][j][e];
        }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            int co = co_tile * BCO + wm * (BCO / 2) + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (co + e < p.Cout) atomicAdd(p.dbias + co + e, accb[i][e]);
        }
    }
#endif
}

// v4 ("ring"): the v3 kernel is bound by the global->LDS stream (DIN_WGRAD_PROBE=1: streaming alone takes 75-87 % of its time, at
// ~9-10 TB/s of LDS-DMA traffic), so the lever is bytes per FLOP: BCO x BK = {128,192,256} x 256 tiles (per-wave (BCO/2) x 128) move
// 1.3-2.3x fewer bytes than BCO x 128.  One workgroup per CU (accumulators fill the register file), so latency is hidden inside
// the wave: a 4-stage ring of 32-pixel stages with the LDS-DMA issued THREE stages ahead (hand-counted vmcnt), and the transpose
// reads of stage s in flight while the MFMAs of stage s-1 run (register double buffer).  One s_barrier per stage.
template <int BCO, int BK>
__global__ __launch_bounds__(512, (4 * (((32 * (BCO / 8) + 511) / 512) * 8192 + 32 * BK * 2) <= 80 * 1024) ? 2 : 1) void conv_wgrad_ring_kernel(WgradK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 8 waves (2 x 4: filter half wm, k-column quarter wn), two per SIMD: one wave's barrier / vmcnt / transpose-read waits hide
    // behind the other's MFMAs (with one wave per SIMD they are all serial; cf. profiles/r01_halo_probe.txt)
    constexpr int PK = 32, NS = 4, NWV = 8, NTH = 64 * NWV;
    constexpr int CG = BCO / 8, CX = BK / 8;                      // 16-byte chunks per G / X row
    constexpr int RBG = BCO * 2, RBX = BK * 2;                    // row bytes (unpadded)
    constexpr int TI = BCO / 32, XJ = BK / 64;                    // 16-row / 16-column MFMA tiles per wave (wave tile = BCO/2 x BK/4)
    constexpr int GP = (PK * CG + NTH - 1) / NTH, XP = PK * CX / NTH;
    static_assert(PK * CX % NTH == 0, "whole X DMA transfers per thread");
    // every wave issues the same number of transfers (the vmcnt bookkeeping is a compile-time constant): when the G tile is not a
    // whole number of 4-KiB rounds (BCO = 160) the surplus transfers fetch nothing and land in a pad behind the G tile
    constexpr int OPG = GP * 1024 * NWV, OPX = PK * RBX, STAGE = OPG + OPX;
    static_assert(OPG >= PK * RBG, "G region");
    constexpr int RPT = 64 / CX;                                  // X rows per wave-level transfer
    constexpr int NDMA = GP + XP;                                 // DMA instructions per stage per wave
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    int bx_, by_;
    xcd_block(bx_, by_);
    const int k_tile = bx_ % p.n_k_tiles, co_tile = bx_ / p.n_k_tiles;
    const int slice = by_;
    const int m_begin = slice * p.m_per_slice;                   // multiple of PK
    int m_end = m_begin + p.m_per_slice;
    if (m_end > p.M) m_end = p.M;

    const int ohw = p.OH * p.OW;
    const int n_first = m_begin / ohw;
    const long long img_bytes = (long long)p.H * p.W * p.ldi * 2ll;
    const long long x_off = (long long)n_first * img_bytes;
    long long x_rem = (long long)p.NB * img_bytes - x_off;
    if (x_rem > 0x7fffffffll) x_rem = 0x7fffffffll;
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in)) + x_off, 0, (int)x_rem, 0x00020000);
    const long long g_off = (long long)m_begin * p.ldo * 2ll;
    long long g_rem = (long long)p.M * p.ldo * 2ll - g_off;
    if (g_rem > 0x7fffffffll) g_rem = 0x7fffffffll;
    __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.g)) + g_off, 0, (int)g_rem, 0x00020000);

    // ---- G DMA plan: transfer t covers chunk ids [(wid + 4 t) * 64, +64): id -> (row id / CG, slot id % CG); the lane fetches the
    //      logical chunk (slot - 2*(row & 7)) mod CG of that row (rotation against transpose-read bank conflicts) --------------
    unsigned voffG[GP];
#pragma unroll
    for (int t = 0; t < GP; ++t) {
        const int id = (wid + NWV * t) * 64 + lane;
        const int grow = id / CG, slot = id - grow * CG;
        int gc = slot - 2 * (grow & 7);
        gc += gc < 0 ? CG : 0;
        gc += gc < 0 ? CG : 0;                                      // 2*(row&7) <= 14 exceeds CG = 8 / 12 once more
        const int gco = co_tile * BCO + gc * 8;
        voffG[t] = (grow < PK && gco + 7 < p.Cout) ? (unsigned)((grow * p.ldo + p.cooff + gco) * 2) : OOB;   // Cout % 8 == 0 enforced
    }
    // ---- X DMA plan: transfer t covers rows (wid + 4 t) * RPT + lane / CX; (row & 7) is the same for all t -----------------------
    const int xrow0 = wid * RPT + lane / CX;                       // rows xrow0 + NWV RPT t
    int xc = (lane % CX) - 2 * (xrow0 & 7);
    xc += xc < 0 ? CX : 0;
    const int kcol = k_tile * BK + xc * 8;
    const bool kok = kcol < p.kcols;
    const int tap = kok ? kcol / p.cin_pad : 0;
    const int ci = kcol - tap * p.cin_pad;
    const int tr_ = tap / p.kw, ts_ = tap - tr_ * p.kw;
    const bool ci_ok = kok && ci + 7 < p.Cin;
    const int dy0 = -p.ph + tr_ * p.dh, dx0 = -p.pw + ts_ * p.dw;   // iy = oy*sh + dy0, ix = ox*sw + dx0
    const int step_bytes = p.sw * p.ldi * 2;
    int px[XP], py[XP], rowoff[XP];
#pragma unroll
    for (int t = 0; t < XP; ++t) {
        int m = m_begin + xrow0 + NWV * RPT * t;
        int n = m / ohw;
        int rem = m - n * ohw;
        py[t] = rem / p.OW; px[t] = rem - py[t] * p.OW;
        rowoff[t] = (((n - n_first) * p.H + py[t] * p.sh + dy0) * p.W + dx0) * p.ldi * 2 + (p.cioff + ci) * 2;
    }
    const int row_jump = p.sh * p.W * p.ldi * 2;
    const int img_jump = (p.H - p.OH * p.sh) * p.W * p.ldi * 2;

    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t ldsW = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(wid * 1024));
    auto issue_dma = [&](int buf, int m0) {                         // called with consecutive m0 (the X cursors advance by PK)
        const uint32_t Gd = ldsW + (uint32_t)(buf * STAGE), Xd = Gd + (uint32_t)OPG;
        const int soffG = (m0 - m_begin) * p.ldo * 2;
#pragma unroll
        for (int t = 0; t < GP; ++t) lds_dma16(Gd + (uint32_t)(t * 1024 * NWV), rsG, (int)voffG[t], soffG);   // rows past M: out of range -> zeros
#pragma unroll
        for (int t = 0; t < XP; ++t) {
            const int iy = py[t] * p.sh + dy0, ix = px[t] * p.sw + dx0;
            const bool ok = ci_ok && (m0 + xrow0 + NWV * RPT * t < m_end) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            lds_dma16(Xd + (uint32_t)(t * 1024 * NWV), rsX, ok ? rowoff[t] + px[t] * step_bytes : (int)OOB, 0);
            px[t] += PK;
            while (px[t] >= p.OW) {
                px[t] -= p.OW; rowoff[t] += row_jump;
                if (++py[t] == p.OH) { py[t] = 0; rowoff[t] += img_jump; }
            }
        }
    };

    f32x4 acc[TI][XJ], accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < XJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = p.dbias != nullptr && k_tile == 0 && wn == 0;       // wave-uniform
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    const int li = lane & 15, lg = lane >> 4;
    const int prow = 4 * lg + (li >> 2);
    const int rot = 2 * (prow & 7);
    uint32_t colG[TI], colX[XJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        int ch = (wm * (BCO / 2) + i * 16) / 8 + rot;                         // even
        ch -= ch >= CG ? CG : 0;
        ch -= ch >= CG ? CG : 0;
        colG[i] = (uint32_t)(prow * RBG + (ch + ((li & 3) >> 1)) * 16 + (li & 1) * 8);
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        int ch = (wn * (BK / 4) + j * 16) / 8 + rot;
        ch -= ch >= CX ? CX : 0;
        colX[j] = (uint32_t)(prow * RBX + (ch + ((li & 3) >> 1)) * 16 + (li & 1) * 8);
    }
    u32x4 gf[TI], xf[XJ];
    auto load_frags = [&](int buf) {
        const uint32_t Gb = lds_base + (uint32_t)(buf * STAGE), Xb = Gb + (uint32_t)OPG;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            u32x2 lo = lds_tr_read(Gb + colG[i]), hi = lds_tr_read(Gb + colG[i] + 16 * RBG);
            gf[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            u32x2 lo = lds_tr_read(Xb + colX[j]), hi = lds_tr_read(Xb + colX[j] + 16 * RBX);
            xf[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };

    const int nst = (m_end - m_begin + PK - 1) / PK;                          // stages of this slice (>= 0)
    if (nst > 0) {
        // prologue: stages 0..2 in flight
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (s0 < nst) issue_dma(s0, m_begin + s0 * PK);
        // iteration s: [stage s landed] barrier, DMA stage s+3, transpose reads + MFMAs of stage s (the SIMD's other wave overlaps)
        for (int s2 = 0; s2 < nst; ++s2) {
            // stages s2+1, s2+2 may stay in flight (issued after stage s2); near the end fewer are outstanding -> drain
            if (s2 + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NDMA) : "memory");
            else if (s2 + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s2 + NS - 1 < nst) issue_dma((s2 + NS - 1) & (NS - 1), m_begin + (s2 + NS - 1) * PK);
            load_frags(s2 & (NS - 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < XJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[i]), __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[i]), __builtin_bit_cast(bf16x8, ones), accb[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* dst = p.partial + (int64_t)slice * p.cout_pad * p.kcols_pad;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            int co = co_tile * BCO + wm * (BCO / 2) + i * 16 + (lane >> 4) * 4;
            int kc = k_tile * BK + wn * (BK / 4) + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(int64_t)(co + e) * p.kcols_pad + kc] = acc[i][j][e];
        }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            int co = co_tile * BCO + wm * (BCO / 2) + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (co + e < p.Cout) atomicAdd(p.dbias + co + e, accb[i][e]);
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// wgrad of the stem layers (3x3; 32 -> 32/64 channels stride 1, and the image layer <= 8 padded channels -> 32, stride 2): millions of
// pixels, a few thousand filter gradients.  The general kernel re-pulls every input pixel once per tap and every G pixel once per
// k-column tile; here each 8x32 output tile brings its G tile and its input HALO in once (LDS-DMA, out-of-image -> hardware zeros),
// all (tap, ci) columns are formed from LDS by transpose reads, and persistent workgroups keep the whole dW block
// (BN x 9*cin_pad fp32) in registers across their tiles: HBM traffic = the two tensors once.
//   k (pixel) assignment inside a 32-pixel k-step = one tile row: read rd, lane group g4, sub-row q -> x = 16 rd + 4 g4 + q; each
//   32-lane half of a ds_read_b64_tr_b16 then covers 8 consecutive pixels, conflict-free with the unit swizzles below.
//   wave w owns the 16-column tiles w, w+4, ... of the (tap, ci) axis and all BN filter rows; wave 0 also forms the bias gradient
//   (G^T x ones).  Result: one fp32 partial slab per workgroup in the layout conv_wgrad_reduce_kernel expects.
// ------------------------------------------------------------------------------------------------
// NW: waves per workgroup.  The 32 -> 64 layer (Conv2d_2b) needs 108 KiB of LDS, i.e. one workgroup per CU: with four waves that is ONE wave
// per SIMD (3.1 TB/s); eight waves split the 18 column tiles 3 / 2 per wave instead of 5 / 4 and give every SIMD two waves.
// TH_ / RING (round 6): the 32 -> 64 layer runs ONE workgroup per CU, and with two 54 KB stages only one tile's operands are in flight per CU
// while the current tile is multiplied -- the tile time was the memory latency + transfer of one stage (3.05 us, 4.1 TB/s), not max(compute,
// transfer).  <..., TH_ = 6, RING = 3>: 6 x 32 tiles (42 KB per stage: 24.6 KB of G + 17.4 KB of halo) in a THREE-slot ring with TWO stages in
// flight (126 KB), the oldest waited for with a counted vmcnt.  The halo overhead grows from 10/8 to 8/6 of the input (+2 % bytes).
template <int CPP, int BN, int ST, bool U8 = false, int NW = 4, int TH_ = 8, int RING = 2>
__global__ __launch_bounds__(64 * NW, (CPP == 4 && BN == 64) ? 1 : 2) void conv_wgrad_small_kernel(WgradK p) {
    static_assert(!U8 || CPP == 1, "uint8 frames feed the image layer only");
    static_assert(RING == 2 || (RING == 3 && !U8), "three-slot ring: LDS-DMA operands only");
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = TH_, TW = 32, NPX = TH * TW, KH = 3, KW = 3;
    constexpr int HWW = (TW - 1) * ST + KW, HWH = (TH - 1) * ST + KH, HPX = HWW * HWH, HC = HPX * CPP;
    constexpr int HBYTES = (HC * 16 + 1023) / 1024 * 1024, NSLOT_H = HBYTES / 1024, NTR_H = (NSLOT_H + NW - 1) / NW;
    constexpr int CG = BN / 8, GBYTES = NPX * CG * 16, NTR_G = GBYTES / (1024 * NW);
    constexpr int STAGE = HBYTES + GBYTES;
    constexpr int NCT = CPP == 4 ? 18 : 5, TI = BN / 16, TJ = (NCT + NW - 1) / NW;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    auto swzX = [](int hp) { return CPP == 4 ? ((hp >> 2) & 1) * 2 : 0; };
    auto swzG = [](int t) { return BN == 32 ? ((t >> 2) & 1) * 2 : ((t >> 1) & 3) * 2; };

    // ---- DMA plans (per lane, tile independent) ----------------------------------------------------------------------------
    int relH[NTR_H]; short hyv[NTR_H], hxv[NTR_H];
#pragma unroll
    for (int i = 0; i < NTR_H; ++i) {
        const int id = (wid + NW * i) * 64 + lane;
        const int hp = id / CPP, slot = id - hp * CPP;
        const int cc = slot ^ swzX(hp);
        const int hy = hp / HWW, hx = hp - hy * HWW;
        relH[i] = id < HC ? (hy * p.W + hx) * p.ldi * 2 + cc * 16 : -1;
        hyv[i] = (short)hy; hxv[i] = (short)hx;
    }
    int relG[NTR_G]; short gyv[NTR_G], gxv[NTR_G];
#pragma unroll
    for (int i = 0; i < NTR_G; ++i) {
        const int id = (wid + NW * i) * 64 + lane;
        const int t = id / CG, slot = id - t * CG;
        const int cc = slot ^ swzG(t);
        gyv[i] = (short)(t >> 5); gxv[i] = (short)(t & 31);
        relG[i] = (cc * 8 + 7 < p.Cout) ? ((t >> 5) * p.OW + (t & 31)) * p.ldo * 2 + cc * 16 : -1;     // Cout % 8 == 0
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
        const int gy0 = ty * TH * ST - p.ph, gx0 = tx * TW * ST - p.pw;
        __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.in)) + (long long)n * ximg, 0, (int)ximg, 0x00020000);
        __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.g)) + (long long)n * gimg, 0, (int)gimg, 0x00020000);
        const int baseX = (gy0 * p.W + gx0) * p.ldi * 2 + p.cioff * 2;
        const int baseG = ((ty * TH) * p.OW + tx * TW) * p.ldo * 2 + p.cooff * 2;
        const uint32_t dH = ldsW + (uint32_t)(buf * STAGE), dG = dH + (uint32_t)HBYTES;
        if (!U8) {
#pragma unroll
            for (int i = 0; i < NTR_H; ++i) {
                if (wid + NW * i < NSLOT_H) {
                    const int gy = gy0 + hyv[i], gx = gx0 + hxv[i];
                    const bool ok = relH[i] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                    lds_dma16(dH + (uint32_t)(i * 1024 * NW), rsX, ok ? baseX + relH[i] : (int)OOB, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NTR_G; ++i) {
            const bool ok = relG[i] >= 0 && ty * TH + gyv[i] < p.OH && tx * TW + gxv[i] < p.OW;
            lds_dma16(dG + (uint32_t)(i * 1024 * NW), rsG, ok ? baseG + relG[i] : (int)OOB, 0);
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
    //      (k row = i16 >> 2, columns 4*(i16&3)..+3) and receives column i16 ---------------------------------------------------
    const int i16 = lane & 15, g4 = lane >> 4;
    const int xq = g4 * 4 + (i16 >> 2);                                 // x inside the 16-pixel read (add 16 * rd)
    const int csel = (i16 & 3) >> 1, chalf = (i16 & 1) * 8;
    int tapoff[TJ];                                                     // halo pixel offset of this lane's tap for its column tiles
    int unitX[TJ];
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
        const int j = min(wid + NW * jj, NCT - 1);
        const int tap = CPP == 4 ? (j >> 1) : min(2 * j + csel, KH * KW - 1);
        const int r = tap / KW, s2 = tap - r * KW;
        tapoff[jj] = r * HWW + s2;
        unitX[jj] = j & 1;
    }

    // uint8 frames (U8): the input halo is fetched as bytes into registers where the DMA is issued, and written (normalised, bf16) to the
    // other stage's halo buffer after this tile's MFMAs; the G tile still arrives by LDS-DMA
    U8Halo<NTR_H> u8h;
    bf16_t* u8lut = reinterpret_cast<bf16_t*>(smem_raw + 2 * STAGE);                    // 512 bytes behind the two stages (host adds them)
    if (U8) { u8_lut_init(u8lut, tid); __syncthreads(); }
    auto u8_load = [&](int tile) {
        const int n = tile / tiles_img;
        const int tr = tile - n * tiles_img;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        u8h.template load<NSLOT_H>(p.u8, n, p.H, p.W, ty * TH * ST - p.ph, tx * TW * ST - p.pw, wid, hyv, hxv, relH);
    };
    // one tile's MFMAs from ring slot `slot` (shared by both ring forms)
    auto multiply = [&](int slot) {
        const uint32_t Hb = lds_base + (uint32_t)(slot * STAGE), Gb = Hb + (uint32_t)HBYTES;
        u32x4 gf[2][TI], xf[2][TJ];
        auto load_frags = [&](int ks, u32x4 (&gfr)[TI], u32x4 (&xfr)[TJ]) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                u32x2 rr[2];
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int t = ks * 32 + rd * 16 + xq;
                    const int ch = (i * 2) ^ swzG(t);
                    rr[rd] = lds_tr_read(Gb + (uint32_t)((t * CG + ch + csel) * 16 + chalf));
                }
                gfr[i] = u32x4{rr[0][0], rr[0][1], rr[1][0], rr[1][1]};
            }
#pragma unroll
            for (int jj = 0; jj < TJ; ++jj) {
                u32x2 rr[2];
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int hp = ks * ST * HWW + (rd * 16 + xq) * ST + tapoff[jj];
                    uint32_t a;
                    if (CPP == 4) a = (uint32_t)((hp * 4 + ((unitX[jj] * 2) ^ swzX(hp)) + csel) * 16 + chalf);
                    else a = (uint32_t)(hp * 16 + chalf);
                    rr[rd] = lds_tr_read(Hb + a);
                }
                xfr[jj] = u32x4{rr[0][0], rr[0][1], rr[1][0], rr[1][1]};
            }
        };
        load_frags(0, gf[0], xf[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < TH; ++ks) {
            if (ks + 1 < TH) load_frags(ks + 1, gf[(ks + 1) & 1], xf[(ks + 1) & 1]);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int jj = 0; jj < TJ; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[ks & 1][i]),
                                                                         __builtin_bit_cast(bf16x8, xf[ks & 1][jj]), acc[i][jj], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[ks & 1][i]), __builtin_bit_cast(bf16x8, ones), accb[i], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int cur = 0;
    int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if constexpr (RING == 3) {
        // transfers per wave and stage: NTR_G for the G tile + the halo slots this wave owns (wave w: slots w, w + NW, ... < NSLOT_H)
        constexpr int HREM = NSLOT_H % NW, NH_LO = NSLOT_H / NW;            // waves < HREM issue NH_LO + 1 halo transfers, the others NH_LO
        const bool more_h = HREM != 0 && wid < HREM;                        // wave-uniform
        const int stride = (int)gridDim.x;
        if (tile < ntiles) issue(0, tile);
        if (tile + stride < ntiles) issue(1, tile + stride);
        int fill = 2;                                                       // slot the next issue goes to
        for (; tile < ntiles; tile += stride) {
            // stage `cur` (this tile) must have landed; the stage of tile + stride, if it exists, stays in flight
            if (tile + stride < ntiles) {
                if (more_h) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NTR_G + NH_LO + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NTR_G + NH_LO) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                   // stage(cur) complete; everyone finished reading the slot consumed last (= fill)
            asm volatile("" ::: "memory");
            if (tile + 2 * stride < ntiles) issue(fill, tile + 2 * stride);
            multiply(cur);
            cur = cur + 1 == 3 ? 0 : cur + 1;
            fill = fill + 1 == 3 ? 0 : fill + 1;
        }
    } else {
    if (tile < ntiles) {
        issue(0, tile);
        if (U8) { u8_load(tile); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); u8h.landed(); u8h.template store<NSLOT_H>(smem_raw, u8lut, wid, lane); }
    }
    for (; tile < ntiles; tile += gridDim.x) {
        if (U8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // stage(cur) landed; everyone finished reading stage(cur^1)
        asm volatile("" ::: "memory");
        const bool have_next = tile + (int)gridDim.x < ntiles;
        if (have_next) issue(cur ^ 1, tile + gridDim.x);
        if (U8 && have_next) u8_load(tile + gridDim.x);
        multiply(cur);                                                  // (the transpose reads of k-step ks+1 are in flight while the MFMAs of k-step ks run)
        if (U8 && have_next) {                                          // requested at the top of this tile (after the next G tile's transfers)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); u8h.landed();
            u8h.template store<NSLOT_H>(smem_raw + (cur ^ 1) * STAGE, u8lut, wid, lane);
        }
        cur ^= 1;
    }
    }
    // ---- this workgroup's partial slab: [BN][NCT * 16] fp32 ------------------------------------------------------------------
    float* dst = p.partial + (int64_t)blockIdx.x * p.cout_pad * p.kcols_pad;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
            const int j = wid + NW * jj;
            if (j < NCT) {
                const int co = i * 16 + g4 * 4, kc = j * 16 + i16;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(int64_t)(co + e) * p.kcols_pad + kc] = acc[i][jj][e];
            }
        }
    if (do_bias && i16 == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int co = i * 16 + g4 * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (co + e < p.Cout) atomicAdd(p.dbias + co + e, accb[i][e]);
        }
    }
#endif
}

// tail kernel for channel counts that are not multiples of 8 (conv1: cin = 3): the round-1 32-pixel kernel
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_bf16_tail_kernel(WgradK p) {
    constexpr int PK = 32;
    constexpr int RSB = WG_TILE * 2 + 32;          // row stride in bytes (288)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int OPB = PK * RSB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int bid, slice_;
    xcd_block(bid, slice_);
    const int k_tile = bid % p.n_k_tiles;
    const int co_tile = bid / p.n_k_tiles;
    const int slice = slice_;
    const int m_begin = slice * p.m_per_slice;
    int m_end = m_begin + p.m_per_slice;
    if (m_end > p.M) m_end = p.M;
    const int cc = tid & 15, rr = tid >> 4;
    const int kcol = k_tile * WG_TILE + cc * 8;
    const bool kok = kcol < p.kcols;
    const int tap = kok ? kcol / p.cin_pad : 0;
    const int ci = kcol - tap * p.cin_pad;
    const int r = tap / p.kw, s = tap - r * p.kw;
    const bool ci_ok = kok && ci < p.Cin;
    const int gco = co_tile * WG_TILE + cc * 8;
    const bf16_t* __restrict__ inp = reinterpret_cast<const bf16_t*>(p.in);
    const bf16_t* __restrict__ gp = reinterpret_cast<const bf16_t*>(p.g);
    int pn[2], py[2], px[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m_begin + rr + 16 * i;
        int n = m / (p.OH * p.OW);
        int rem = m - n * (p.OH * p.OW);
        pn[i] = n; py[i] = rem / p.OW; px[i] = rem - py[i] * p.OW;
    }
    u32x4 xa[2], ga[2];
    auto load_global = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + rr + 16 * i;
            u32x4 xv = {0u, 0u, 0u, 0u}, gv = {0u, 0u, 0u, 0u};
            if (m < m_end) {
                int iy = py[i] * p.sh - p.ph + r * p.dh, ix = px[i] * p.sw - p.pw + s * p.dw;
                if (ci_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    const bf16_t* src = inp + (int64_t)((pn[i] * p.H + iy) * p.W + ix) * p.ldi + p.cioff + ci;
                    if (ci + 7 < p.Cin) xv = *reinterpret_cast<const u32x4*>(src);
                    else {
                        bf16_t tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        for (int e = 0; e < 8 && ci + e < p.Cin; ++e) tmp[e] = src[e];
                        xv = *reinterpret_cast<u32x4*>(tmp);
                    }
                }
                if (gco < p.Cout) {
                    const bf16_t* src = gp + (int64_t)m * p.ldo + p.cooff + gco;
                    if (gco + 7 < p.Cout) gv = *reinterpret_cast<const u32x4*>(src);
                    else {
                        bf16_t tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        for (int e = 0; e < 8 && gco + e < p.Cout; ++e) tmp[e] = src[e];
                        gv = *reinterpret_cast<u32x4*>(tmp);
                    }
                }
            }
            xa[i] = xv; ga[i] = gv;
            px[i] += PK;
            while (px[i] >= p.OW) { px[i] -= p.OW; if (++py[i] == p.OH) { py[i] = 0; ++pn[i]; } }
        }
    };
    auto store_lds = [&](int buf) {
        unsigned char* G = smem_raw + buf * 2 * OPB;
        unsigned char* X = G + OPB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(G + (rr + 16 * i) * RSB + cc * 16) = ga[i];
            *reinterpret_cast<u32x4*>(X + (rr + 16 * i) * RSB + cc * 16) = xa[i];
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lg = lane >> 4;
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t piece = (uint32_t)((4 * lg + (li >> 2)) * RSB + (li & 3) * 8);
    if (m_begin < m_end) {
        load_global(m_begin);
        store_lds(0);
        __syncthreads();
        int it = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += PK, ++it) {
            const int cur = it & 1;
            const bool more = m0 + PK < m_end;
            if (more) load_global(m0 + PK);
            const uint32_t Gb = lds_base + cur * 2 * OPB + piece;
            const uint32_t Xb = Gb + OPB;
            u32x4 gf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t a = Gb + (wm * 64 + i * 16) * 2;
                u32x2 lo = lds_tr_read(a), hi = lds_tr_read(a + 16 * RSB);
                gf[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t a = Xb + (wn * 64 + j * 16) * 2;
                u32x2 lo = lds_tr_read(a), hi = lds_tr_read(a + 16 * RSB);
                xf[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf[i]),
                                                                        __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
            if (more) store_lds(cur ^ 1);
            __syncthreads();
        }
    }
    float* dst = p.partial + (int64_t)slice * p.cout_pad * p.kcols_pad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int co = co_tile * WG_TILE + wm * 64 + i * 16 + (lane >> 4) * 4;
            int kc = k_tile * WG_TILE + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(int64_t)(co + e) * p.kcols_pad + kc] = acc[i][j][e];
        }
}

// reduce the slices, un-permute to the reference layout [cout][cin][kh][kw], apply scale, optional <w, dw_raw>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, float* __restrict__ dw,
                                                  const float* __restrict__ scale, const float* __restrict__ w,
                                                  float* __restrict__ wdot, int cout, int cin, int kh, int kw, int cin_pad,
                                                  int cout_pad, int kcols_pad, int slices, int accumulate, const int co, const int kchunk) {
    // (co, 256-column chunk) per workgroup, block (64 lanes x 4 columns, SG slice groups): every wave reads 1 KiB contiguous of one slice per
    // step as float4 (partials keep their own column order; cin_pad % 4 == 0, so a float4 never straddles a tap), the SG partial sums
    // meet in LDS in a fixed order (deterministic result).  50 MB of partials per layer: 16-byte lanes run this 14 -> ~8 us.
    __shared__ f32x4 red[16][64];
    const int taps = kh * kw;
    const int per = cin * taps;
    const int kc = (kchunk * 64 + threadIdx.x) * 4;
    const int sg = threadIdx.y, nsg = blockDim.y;
    const bool col_ok = kc < taps * cin_pad;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col_ok) {
        const f32x4* __restrict__ pp = reinterpret_cast<const f32x4*>(partial + (int64_t)co * kcols_pad + kc);
        const int64_t sstride = (int64_t)cout_pad * kcols_pad / 4;
        f32x4 v0 = v, v1 = v, v2 = v, v3 = v;
        int s = sg;
        for (; s + 3 * nsg < slices; s += 4 * nsg) {
            v0 += pp[(int64_t)s * sstride];
            v1 += pp[(int64_t)(s + nsg) * sstride];
            v2 += pp[(int64_t)(s + 2 * nsg) * sstride];
            v3 += pp[(int64_t)(s + 3 * nsg) * sstride];
        }
        for (; s < slices; s += nsg) v0 += pp[(int64_t)s * sstride];
        v = (v0 + v1) + (v2 + v3);
    }
    red[sg][threadIdx.x] = v;
    __syncthreads();
    if (sg != 0) return;
    for (int g = 1; g < nsg; ++g) v += red[g][threadIdx.x];
    float dot = 0.f;
    if (col_ok) {
        const int t = kc / cin_pad, ci0 = kc - t * cin_pad;    // t = r*kw + s
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ci = ci0 + e;
            if (ci < cin) {
                const int64_t o = (int64_t)co * per + (int64_t)ci * taps + t;
                float x = v[e];
                if (wdot) dot += x * w[o];
                if (scale) x *= scale[co];
                dw[o] = accumulate ? dw[o] + x : x;
            }
        }
    }
    if (wdot) {
        dot = wave_sum(dot);
        if (threadIdx.x == 0) atomicAdd(wdot + co, dot);
    }
}

__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                         const float* __restrict__ scale, const float* __restrict__ w,
                                         float* __restrict__ wdot, int cout, int cin, int kh, int kw, int cin_pad,
                                         int cout_pad, int kcols_pad, int slices, int accumulate) {
    wgrad_reduce_body(partial, dw, scale, w, wdot, cout, cin, kh, kw, cin_pad, cout_pad, kcols_pad, slices, accumulate, (int)blockIdx.x, (int)blockIdx.y);
}

// the slice reduces of all layers of a grouped weight-gradient launch (din_conv_wgrad_group) as ONE launch: block l of item g
// (first[g] <= l < first[g + 1]) is (filter l' / kchunks, 256-column chunk l' % kchunks) of that layer
struct WgradReduceItem { const float* partial; float* dw; const float* scale; const float* w; float* wdot;
                         int cout, cin, kh, kw, cin_pad, cout_pad, kcols_pad, slices, accumulate, kchunks; };
struct WgradReduceGroupK { WgradReduceItem it[din_wgrad::WGRAD_GROUP_MAX]; int first[din_wgrad::WGRAD_GROUP_MAX + 1]; int n; };
__global__ void conv_wgrad_reduce_group_kernel(WgradReduceGroupK grp) {
    const int l = (int)blockIdx.x;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < din_wgrad::WGRAD_GROUP_MAX; ++i) gi += (i < grp.n && l >= grp.first[i]) ? 1 : 0;
    const WgradReduceItem it = grp.it[gi];
    const int local = l - grp.first[gi], co = local / it.kchunks;
    wgrad_reduce_body(it.partial, it.dw, it.scale, it.w, it.wdot, it.cout, it.cin, it.kh, it.kw, it.cin_pad, it.cout_pad, it.kcols_pad, it.slices,
                      it.accumulate, co, local - co * it.kchunks);
}

// column sums of G [M][cout] (pixel stride ld, offset coff) -> dbias[cout] (atomic accumulate; caller zeroes)
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ g, float* __restrict__ out, int64_t M, int cout, int ld, int coff,
                              int64_t rows_per_block) {
    // block handles rows [b*rpb, (b+1)*rpb); thread t handles columns t, t+256, ...
    int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    for (int c = threadIdx.x; c < cout; c += blockDim.x) {
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) s += Elem<T>::ld(g + r * ld + coff + c);
        atomicAdd(out + c, s);
    }
}

// vectorised form: thread = (row lane r, 16-byte channel chunk c); four rows in flight per thread, LDS cross-row reduce, one atomic
// per column per block
template <typename T, int U = 4>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ g, float* __restrict__ out, int64_t M, int cout, int ld,
                                                         int coff, int64_t rows_per_block) {
    constexpr int EPC = 16 / sizeof(T);
    __shared__ float red[256][EPC + 1];
    const int ncg = cout / EPC;                      // <= 256
    const int rows_pp = 256 / ncg;
    const int r = threadIdx.x / ncg, c = threadIdx.x - r * ncg;
    int64_t r0 = (int64_t)xcd_remap((int)blockIdx.x, (int)gridDim.x) * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    auto add = [&](const u32x4& v) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += __uint_as_float(v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(v[e] << 16); acc[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u); }
        }
    };
    if (r < rows_pp) {
        const T* base = g + coff + c * EPC;
        int64_t row = r0 + r;
        for (; row + (U - 1) * rows_pp < r1; row += U * rows_pp) {         // U independent 16-byte loads in flight per thread
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u32x4*>(base + (row + u * rows_pp) * ld);
#pragma unroll
            for (int u = 0; u < U; ++u) add(v[u]);
        }
        for (; row < r1; row += rows_pp) add(*reinterpret_cast<const u32x4*>(base + row * ld));
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    for (int idx = threadIdx.x; idx < ncg * EPC; idx += 256) {
        const int cc = idx / EPC, e = idx - cc * EPC;
        float t = 0.f;
        for (int k = 0; k < rows_pp; ++k) t += red[k * ncg + cc][e];
        atomicAdd(out + idx, t);
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                               float* scale, float* shift, int c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) {
        float s = gamma[i] / sqrtf(var[i] + eps);
        scale[i] = s;
        shift[i] = beta[i] - mean[i] * s;
    }
}
__global__ void bn_fold_bwd_kernel(const float* wdot, const float* dshift, const float* mean, const float* var, float eps,
                                   float* dgamma, float* dbeta, int c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) {
        float rstd = 1.f / sqrtf(var[i] + eps);
        dgamma[i] = (wdot[i] - dshift[i] * mean[i]) * rstd;
        dbeta[i] = dshift[i];
    }
}

// all BatchNorm layers of a backbone in one launch: ptrs[l] = {gamma, beta, mean, var}, channel l-range = [offs[l], offs[l+1])
__device__ __forceinline__ int bn_layer_of(const int32_t* __restrict__ offs, int n, int i) {
    int lo = 0, hi = n;                                       // offs[lo] <= i < offs[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}
__global__ void bn_fold_multi_kernel(const uint64_t* __restrict__ ptrs, const int32_t* __restrict__ offs, int n, int total, float eps,
                                     float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int l = bn_layer_of(offs, n, i), c = i - offs[l];
    const float* gamma = reinterpret_cast<const float*>(ptrs[4 * l + 0]);
    const float* beta = reinterpret_cast<const float*>(ptrs[4 * l + 1]);
    const float* mean = reinterpret_cast<const float*>(ptrs[4 * l + 2]);
    const float* var = reinterpret_cast<const float*>(ptrs[4 * l + 3]);
    const float s = gamma[c] / sqrtf(var[c] + eps);
    scale[i] = s;
    shift[i] = beta[c] - mean[c] * s;
}
__global__ void bn_fold_bwd_multi_kernel(const uint64_t* __restrict__ ptrs, const int32_t* __restrict__ offs, int n, int total, float eps,
                                         const float* __restrict__ wdot, const float* __restrict__ dshift, float* __restrict__ dgamma,
                                         float* __restrict__ dbeta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int l = bn_layer_of(offs, n, i), c = i - offs[l];
    const float* mean = reinterpret_cast<const float*>(ptrs[4 * l + 2]);
    const float* var = reinterpret_cast<const float*>(ptrs[4 * l + 3]);
    const float rstd = 1.f / sqrtf(var[c] + eps);
    dgamma[i] = (wdot[i] - dshift[i] * mean[c]) * rstd;
    dbeta[i] = dshift[i];
}

// ---- host-side planning ----------------------------------------------------------------------------
inline int epc_of(int dtype) { return dtype == DIN_F32 ? 4 : 8; }
inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

struct GatherPlan { int bm, bn, n_co_tiles, n_px_tiles, cpt, Q, nk, splitk, ks_per_split, cout_pad; int64_t ws_bytes; };

// geometry of a gather launch whose reduction runs over `cred` channels x taps and produces `cprod` channels
GatherPlan plan_gather(int M, int cred, int cprod, int taps, int dtype, bool strided_out = false) {
    GatherPlan g;
    int epc = epc_of(dtype);
    g.cpt = pad_to(cred, epc) / epc;
    g.Q = taps * g.cpt;
    g.nk = (g.Q + KC - 1) / KC;
    g.bn = cprod <= 64 ? 64 : 128;
    g.bm = 128;
    if (cprod > 64 && !(DIN_OPT("DIN_CONV_BN") && atoi(DIN_OPT("DIN_CONV_BN")) == 128)) {
        // filter-tile width in {96,128,160,192}: least padded filters, ties to the wider tile (fewer re-reads of the pixel tile).
        // Inception's 96/160/192/288/384-filter layers otherwise waste 25-37 % of a 128-wide tile.
        // fewest filter tiles first (each tile re-reads the pixel tile), then least padding
        int best = 128, best_tiles = (cprod + 127) / 128, best_pad = best_tiles * 128;
        const int cands[3] = {96, 160, 192};
        for (int ci = 0; ci < 3; ++ci) {
            int bnc = cands[ci], tl = (cprod + bnc - 1) / bnc, pad = tl * bnc;
            if (tl < best_tiles || (tl == best_tiles && pad < best_pad)) { best = bnc; best_tiles = tl; best_pad = pad; }
        }
        g.bn = best;
        if (const char* fb = DIN_OPT("DIN_CONV_BN")) { const int v = atoi(fb); if (v == 96 || v == 128 || v == 160 || v == 192 || v == 256) g.bn = v; }   // tuning / test override
        // parity classes of a strided dgrad write every other pixel of dX: each tile's epilogue is a scattered write, and more, narrower
        // tiles per CU overlap it better -- 3 x 96 beats 2 x 160 on the 288-channel stride-2 dgrad (1642 -> 1427 us)
        if (strided_out && g.bn == 160 && cprod % 96 == 0) g.bn = 96;
    }
    // Tile choice (measured on MI355X, tools/conv_bench.py).  The L2->CU operand stream limits the 128x128
    // tile to ~770 TFLOP/s (64 FLOP per byte pulled from L2):
    //   64-filter launches over many pixels              -> 256x64  (256 threads, 2 workgroups / CU: 2x work per barrier)
    //   bf16, filters fill 256-wide tiles, long reduction -> 256x256 (512 threads, 128 FLOP/B; +12..16 % on VGG conv3/conv4)
    //   everything else                                   -> 128x128 / 128x64 at 2 workgroups / CU (a 256x128 tile at one
    //                                                        workgroup / CU measured 8..25 % SLOWER: no prologue/epilogue overlap)
    g.nk = (taps * g.cpt + KC - 1) / KC;
    const int64_t tiles256 = ((int64_t)M + 255) / 256;
    const char* force = DIN_OPT("DIN_CONV_TILE");
    if (g.bn == 64) {
        // fp32: 256x64 (4 waves, 2 workgroups / CU).  bf16: 128x64 on 8 waves measured +6..14 % on the 1x1 / dgrad launches, -2.5 % on the
        // 5x5 forward (DIN_CONV_TILE=256 restores 256x64)
        if (M >= 256 * 1024 && (dtype == DIN_F32 || (force && atoi(force) == 256))) g.bm = 256;
    } else if (!(force && atoi(force) == 128)) {
        const int nco256 = (cprod + 255) / 256;
        if (dtype == DIN_BF16 && cprod >= 224 && nco256 * 256 * 100 <= cprod * 115 && g.nk >= 24 && tiles256 * nco256 >= 768) { g.bm = 256; g.bn = 256; }
        else if (dtype == DIN_BF16 && force && atoi(force) == 256 && tiles256 >= 512) g.bm = 256;      // experiment: 256 x {96..192}
    }
    g.cout_pad = pad_to(cprod, 128);
    g.n_co_tiles = (cprod + g.bn - 1) / g.bn;
    g.n_px_tiles = (M + g.bm - 1) / g.bm;
    // split-K only when the launch cannot fill the chip and the reduction is long
    int tiles = g.n_co_tiles * g.n_px_tiles;
    g.splitk = 1;
    if (tiles < 192 && g.nk >= 8) {
        int want = (512 + tiles - 1) / tiles;
        int maxs = g.nk / 2;
        g.splitk = want < maxs ? want : maxs;
        if (g.splitk < 1) g.splitk = 1;
        if (g.splitk > 64) g.splitk = 64;
    }
    g.ks_per_split = (g.nk + g.splitk - 1) / g.splitk;
    if (g.ks_per_split < 1) g.ks_per_split = 1;                    // nk == 0: launch still runs its epilogue (zeros / accumulate)
    g.splitk = g.nk > 0 ? (g.nk + g.ks_per_split - 1) / g.ks_per_split : 1;
    g.ws_bytes = g.splitk > 1 ? (int64_t)g.splitk * M * (g.n_co_tiles * g.bn) * 4 : 0;
    return g;
}

struct WgradPlan { int cin_pad, kcols, kcols_pad, cout_pad, n_co_tiles, n_k_tiles, slices, m_per_slice, bco, v2, small, ring, bk, pipe, atomic; int64_t ws_bytes; };
constexpr int WGRAD_SMALL_GRID = 512;
constexpr int WGRAD_HALO_GRID = 128;       // persistent workgroups per filter-row class of conv_wgrad_halo_kernel (2 classes x 128 = one per CU)
WgradPlan plan_wgrad(const din_conv_desc* d) {
    WgradPlan w;
    w.pipe = w.atomic = 0;
    int epc = epc_of(d->dtype);
    // bf16 v2 kernel needs whole 16-byte channel chunks on both operands; otherwise the tail kernel (conv1: cin = 3)
    w.v2 = d->dtype == DIN_BF16 && d->cin % 8 == 0 && d->cout % 8 == 0 && d->ldi % 8 == 0 && d->cioff % 8 == 0;
    w.bco = 128;
    if (w.v2) {
        // filter-tile width in {64,96,128,160}: least padding, ties to the wider tile (192 -> 2 x 96, 288 -> 3 x 96, 384 -> 3 x 128)
        if (d->cout <= 64) w.bco = 64;
        else if (!(DIN_OPT("DIN_CONV_BN") && atoi(DIN_OPT("DIN_CONV_BN")) == 128)) {
            // fewest filter tiles first; on a tie keep 128 when cout > 128 (measured: 192 as 2 x 96 is slower than 2 x 128),
            // otherwise the least padded width
            int best = 128, best_tiles = (d->cout + 127) / 128, best_pad = best_tiles * 128;
            const int cands[2] = {96, 160};
            for (int ci = 0; ci < 2; ++ci) {
                int bc = cands[ci], tl = (d->cout + bc - 1) / bc, pad = tl * bc;
                if (tl < best_tiles || (tl == best_tiles && d->cout <= 128 && pad < best_pad)) { best = bc; best_tiles = tl; best_pad = pad; }
            }
            w.bco = best;
        }
    }
    // stem layers (conv_wgrad_small_kernel): small = 1: 32 -> <=32, 2: 32 -> <=64 (stride 1), 3: image layer (<= 8 channels, stride 2)
    w.small = 0;
    {
        const char* sv = DIN_OPT("DIN_CONV_SMALL");
        const bool want = sv ? atoi(sv) != 0 : true;
        const int64_t M = (int64_t)d->nb * d->oh * d->ow;
        const bool common = want && d->dtype == DIN_BF16 && d->kh == 3 && d->kw == 3 && d->dh == 1 && d->dw == 1 && d->cout % 8 == 0 &&
                            d->ldi % 8 == 0 && d->cioff % 8 == 0 && d->ldo % 8 == 0 && d->cooff % 8 == 0 && M >= 256 * 1024 &&
                            (long long)d->h * d->w * d->ldi * 2 < 0x7fffffffll && (long long)d->oh * d->ow * d->ldo * 2 < 0x7fffffffll;
        if (common && d->sh == 1 && d->sw == 1 && d->cin == 32 && d->cout <= 64) w.small = d->cout <= 32 ? 1 : 2;
        else if (common && d->sh == 2 && d->sw == 2 && d->cin <= 8 && d->ldi >= d->cioff + 8 && d->cout <= 32) w.small = 3;
    }
    // narrow mid-network layers (conv_wgrad_halo.hip): small = 4 -- dW stationary in registers, halo tiles, two filter-row classes
    if (!w.small) {
        const char* hv = DIN_OPT("DIN_WGRAD_HALO");
        const int64_t M = (int64_t)d->nb * d->oh * d->ow;
        int bnt = 0;
        const int hmode = hv ? atoi(hv) : 1;                      // 0: off, 1: launches of >= 128K pixels (12 frames of 87x157 measured +3..34 %), 2: any size (tests)
        if (hmode && d->dtype == DIN_BF16 && d->sh == 1 && d->sw == 1 && d->dh == 1 && d->dw == 1 &&
            din_wgrad::wgrad_halo_shape(d->cin, d->cout, d->kh, d->kw, &bnt) && d->ldi % 8 == 0 && d->cioff % 8 == 0 && d->ldo % 8 == 0 &&
            d->cooff % 8 == 0 && (M >= 128 * 1024 || hmode == 2) && d->ow >= 32 && (long long)d->h * d->w * d->ldi * 2 < 0x7fffffffll &&
            (long long)d->oh * d->ow * d->ldo * 2 < 0x7fffffffll) {
            w.small = 4; w.v2 = 0; w.bco = bnt;
            w.cin_pad = d->cin;
            w.kcols = w.kcols_pad = d->kh * d->kw * d->cin;
            w.cout_pad = d->cout; w.n_co_tiles = 2; w.n_k_tiles = 1;
            w.slices = WGRAD_HALO_GRID; w.m_per_slice = 0;
            w.ring = 0; w.bk = 0;
            w.ws_bytes = (int64_t)w.slices * w.cout_pad * w.kcols_pad * 4;
            return w;
        }
    }
    if (w.small) {
        w.v2 = 0; w.bco = w.small == 2 ? 64 : 32;
        w.cin_pad = pad_to(d->cin, 8);
        w.kcols = 9 * w.cin_pad;
        w.kcols_pad = (w.small == 3 ? 5 : 18) * 16;
        w.cout_pad = w.bco; w.n_co_tiles = 1; w.n_k_tiles = 1;
        w.slices = WGRAD_SMALL_GRID; w.m_per_slice = 0;
        w.ws_bytes = (int64_t)w.slices * w.cout_pad * w.kcols_pad * 4;
        return w;
    }
    w.cin_pad = pad_to(d->cin, epc);
    w.kcols = d->kh * d->kw * w.cin_pad;
    // ring kernel (BCO x 256 tiles, one workgroup per CU): wide filter banks with enough k columns -- fewest filter tiles, then least padding
    w.ring = 0;
    {
        const char* rv = DIN_OPT("DIN_WGRAD_RING");
        const int mode = rv ? atoi(rv) : 1;
        if (w.v2 && mode && d->cout >= (mode == 2 ? 64 : 112) && w.kcols >= 256) {
            w.ring = 1;
            int best = 128, best_tiles = (d->cout + 127) / 128, best_pad = best_tiles * 128;
            const int cands[2] = {160, 192};
            for (int ci = 0; ci < 2; ++ci) {
                int bc = cands[ci], tl = (d->cout + bc - 1) / bc, pad = tl * bc;
                if (tl < best_tiles || (tl == best_tiles && pad < best_pad)) { best = bc; best_tiles = tl; best_pad = pad; }
            }
            // measured (profiles/r01_wgrad_ring.txt): the ring wins where the 128-row tiles pad badly (cout 192 -> 2 x 128 wastes a
            // quarter of the MFMAs); at equal tile height the two-workgroups-per-CU v3 kernel is faster
            // (8-wave ring: +25..40 % on 192-row banks, +5..11 % on exact 128 / 256-row banks, behind v3 when rows or k columns pad)
            const int kpad = pad_to(w.kcols, 256);
            const char* pe = DIN_OPT("DIN_WGRAD_PIPE");
            const int pipe_mode = pe ? atoi(pe) : 1;
            if (pipe_mode == 1) {
                // wide banks run the pipelined kernel with rows padded to the next of {128, 192, 256} whatever their k-column padding:
                // measured against the round-1 choice below (DIN_WGRAD_PIPE=3; profiles/r02_wgrad_ring_vs_pipe_vs_atomic.txt) it is
                // 11-17 % faster on the 112 / 128 / 256-row banks the k-padding rule used to send to the two-workgroup kernels
                int pb = 128, pt = (d->cout + 127) / 128, pp = pt * 128;
                const int pc[2] = {192, 256};
                for (int ci = 0; ci < 2; ++ci) {
                    int bc = pc[ci], tl = (d->cout + bc - 1) / bc, pad = tl * bc;
                    if (tl < pt || (tl == pt && pad < pp)) { pb = bc; pt = tl; pp = pad; }
                }
                // (row padding above 15 % -- the 160-row banks as 192 -- measured +5 % only: those stay on the round-1 choice below)
                // allowed row padding in percent: 20 admits the 160-row banks of Mixed_6c / 6d (160 -> 192 rows: 168 -> 155 us per launch
                // against conv_wgrad_bf16_kernel<160>, steady-state clocks; profiles/r03_power_clocks.txt).  DIN_WGRAD_PIPE_PAD: tuning aid
                const char* ppv = DIN_OPT("DIN_WGRAD_PIPE_PAD");
                const int pad_pct = ppv ? atoi(ppv) : 20;
                if (pp * 100 <= d->cout * (100 + pad_pct)) w.bco = pb;
                else if (best_pad * 100 <= d->cout * 105 && (best == 192 || kpad * 100 <= w.kcols * 112)) w.bco = best;
                else w.ring = 0;
            } else if ((best_pad * 100 <= d->cout * 105 && (best == 192 || kpad * 100 <= w.kcols * 112)) || mode == 2) w.bco = best;
            else w.ring = 0;
        }
    }
    int ring_bk = 256;
    {
        const char* rv = DIN_OPT("DIN_WGRAD_RING");
        const int mode = rv ? atoi(rv) : 1;
        // 64 / 96-row banks: the 8-wave ring with 128 k columns (two workgroups per CU) beats v3 by ~20 %; 128 rows: equal, 160: behind
        if (!w.ring && w.v2 && mode != 0 && (mode == 3 || w.bco == 64 || w.bco == 96)) { w.ring = 1; ring_bk = 128; }
    }
    const int bk = w.ring ? ring_bk : WG_TILE;
    {   // BCO x 256 ring tiles whose wave tile is whole 32x32 MFMA tiles run the software-pipelined kernel (conv_wgrad_pipe.hip)
        const char* pe = DIN_OPT("DIN_WGRAD_PIPE");          // (read per call: the tests switch them inside one process)
        const char* ae = DIN_OPT("DIN_WGRAD_ATOMIC");
        const int pipe_env = pe ? atoi(pe) : 1, atomic_env = ae ? atoi(ae) : 0;   // 0: ring kernel, 1: pipe (wide choice), 3: pipe (round-1 tile choice)
        if (w.ring && ring_bk == 256 && (w.bco == 128 || w.bco == 192 || w.bco == 256) && pipe_env) { w.pipe = 1; w.atomic = atomic_env; }
    }
    int pk = d->dtype == DIN_F32 ? 16 : (w.ring ? 32 : (w.v2 ? 64 : 32));
    w.bk = bk;
    w.kcols_pad = pad_to(w.kcols, bk);
    w.n_co_tiles = (d->cout + w.bco - 1) / w.bco;
    w.cout_pad = pad_to(w.n_co_tiles * w.bco, WG_TILE);            // partial-sum rows cover every filter tile
    w.n_k_tiles = w.kcols_pad / bk;
    int M = d->nb * d->oh * d->ow;
    int tiles = w.n_co_tiles * w.n_k_tiles;
    // v2/v3 kernels: ~4 workgroups per CU; short reductions (small per-GPU batch) take 2 -- every workgroup writes a full partial tile, so
    // halving them halves the partial traffic (4-clip step 12.47 -> 12.14 ms).  The ring kernel sets its own count below.
    const int want_env = DIN_OPT("DIN_WGRAD_BLOCKS") ? atoi(DIN_OPT("DIN_WGRAD_BLOCKS")) : 0;
    const int want_total = want_env > 0 ? want_env : (M < 128 * 1024 ? 512 : 1024);
    int want = (want_total + tiles - 1) / tiles;
    if (w.ring) {                                      // one resident workgroup per CU: a single full round (or two for long slices)
        const int rounds = (int64_t)M * tiles >= (int64_t)256 * 64 * 1024 ? 2 : 1;
        want = 256 * rounds * (ring_bk == 128 ? 2 : 1) / tiles;
        if (want < 1) want = 1;
    }
    int64_t max_by_ws = ((int64_t)1 << 30) / ((int64_t)w.cout_pad * w.kcols_pad * 4);   // keep workspace <= 1 GiB
    if (max_by_ws < 1) max_by_ws = 1;
    if (want > max_by_ws) want = (int)max_by_ws;
    int mps = (M + want - 1) / want;
    mps = pad_to(mps < pk ? pk : mps, pk);
    w.m_per_slice = mps;
    w.slices = (M + mps - 1) / mps;
    w.ws_bytes = (int64_t)w.slices * w.cout_pad * w.kcols_pad * 4;
    if (w.pipe) w.ws_bytes += (int64_t)w.slices * w.n_co_tiles * 8 * 4 + 64;      // pacing words
    return w;
}

int check_desc(const din_conv_desc* d) {
    DIN_REQUIRE(d != nullptr, "conv: null descriptor");
    DIN_REQUIRE(d->dtype == DIN_F32 || d->dtype == DIN_BF16, "conv: bad dtype %d", d->dtype);
    int epc = epc_of(d->dtype);
    DIN_REQUIRE(d->nb > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "conv: empty tensor");
    DIN_REQUIRE(d->ldi % epc == 0 && d->cioff % epc == 0, "conv: input pixel stride/offset must be multiples of %d", epc);
    DIN_REQUIRE(d->ldo % 4 == 0 && d->cooff % 4 == 0, "conv: output pixel stride/offset must be multiples of 4");
    // operands are read in whole 16-byte chunks: a channel count that is not a chunk multiple must be followed by
    // finite (zero) padding inside the pixel stride -- the packed filters hold zeros there
    DIN_REQUIRE(d->ldi >= d->cioff + pad_to(d->cin, epc), "conv: ldi %d too small for cin %d (+pad)", d->ldi, d->cin);
    DIN_REQUIRE(d->ldo >= d->cooff + d->cout, "conv: ldo too small");
    int eoh = (d->h + 2 * d->ph - d->dh * (d->kh - 1) - 1) / d->sh + 1;
    int eow = (d->w + 2 * d->pw - d->dw * (d->kw - 1) - 1) / d->sw + 1;
    DIN_REQUIRE(eoh == d->oh && eow == d->ow, "conv: output size %dx%d inconsistent (expected %dx%d)", d->oh, d->ow, eoh, eow);
    DIN_REQUIRE((int64_t)d->nb * d->h * d->w < (1ll << 31) && (int64_t)d->nb * d->oh * d->ow < (1ll << 31), "conv: too many pixels");
    DIN_REQUIRE(d->in_u8 == 0 || d->in_u8 == 1, "conv: in_u8 must be 0 or 1");
    return DIN_OK;
}


// halo kernel eligibility / shape (shared by run_gather and din_conv_kernel_tile).  Returns the filter-tile width (0: not eligible).
// hipFuncSetAttribute is a slow host call: raise a kernel's dynamic-LDS limit once per (thread, kernel), not per launch
template <typename K>
static void raise_lds_limit(K kern, size_t lds) { din_raise_lds(reinterpret_cast<const void*>(kern), lds); }

struct HaloPlan { int bn, th, tw, nsw, nwv, n_co_tiles; size_t lds; };
static bool plan_halo(int dtype, int kh, int kw, int cred, int cprod, int oh, int ow, int64_t M, HaloPlan& hp) {
    const char* hv = DIN_OPT("DIN_CONV_HALO");
    if (hv && atoi(hv) == 0) return false;
    if (dtype != DIN_BF16 || cred < 32 || cred % 8 != 0 || cprod % 8 != 0 || cprod < 40 || M < 64 * 1024) return false;
    const bool k33 = kh == 3 && kw == 3, k17 = kh == 1 && kw == 7, k71 = kh == 7 && kw == 1;
    if (!(k33 || k17 || k71)) return false;
    // filter tiles of 64 or 96 rows: fewest tiles, then least padding
    const int t96 = (cprod + 95) / 96, t64 = (cprod + 63) / 64;
    hp.bn = (t96 < t64 || (t96 == t64 && t96 * 96 <= t64 * 64)) ? 96 : 64;
    hp.n_co_tiles = hp.bn == 96 ? t96 : t64;
    const bool bn80 = k33 && cprod > 64 && cprod <= 80;   // 48 + 32 rows (16-wave kernel only, below)
    if (hp.n_co_tiles * hp.bn * 100 > cprod * 125) return false;
    // measured (profiles/r01_halo_probe.txt): wins 15-22 % on 3x3 layers whose filters fit ONE tile (no halo re-read per filter tile);
    // loses against the 128x192 / 128x160 gather tiles on the wide 192-filter and 7-tap layers.  DIN_CONV_HALO=2 forces it everywhere.
    if (!(hv && atoi(hv) == 2) && !(k33 && hp.n_co_tiles == 1)) return false;
    hp.th = k33 ? 8 : 16; hp.tw = k33 ? 32 : 16;
    hp.nwv = 8;
    // padded-area waste of the tile grid must stay moderate
    const int64_t padded = (int64_t)((oh + hp.th - 1) / hp.th * hp.th) * ((ow + hp.tw - 1) / hp.tw * hp.tw);
    if (padded * 100 > (int64_t)oh * ow * 118) return false;
    const int hpx = (hp.th + kh - 1) * (hp.tw + kw - 1);
    const size_t hbytes = (size_t)((hpx * 10 + 511) / 512) * 8192, wbytes = (size_t)((hp.bn * 10 + 511) / 512) * 8192;
    hp.nsw = 3;
    hp.lds = 2 * hbytes + 3 * wbytes;
    {   // sixteen waves (3x3 tiles only): transfers cover 16 KiB, the filter ring shrinks to two slots to stay inside 160 KiB
        const char* wv = DIN_OPT("DIN_HALO_WAVES");
        const int want = wv ? atoi(wv) : 16;
        const size_t hb16 = (size_t)((hpx * 10 + 1023) / 1024) * 16384, wb16 = (size_t)((hp.bn * 10 + 1023) / 1024) * 16384;
        if (want == 16 && k33 && 2 * hb16 + 2 * wb16 <= 160 * 1024) {
            hp.nwv = 16; hp.nsw = 2; hp.lds = 2 * hb16 + 2 * wb16;
            if (bn80) hp.bn = 80;                               // (same 16 KiB slab slots: 80 x 10 chunks <= one transfer per wave)
        }
    }
    return hp.lds <= 160 * 1024;
}

template <typename T, int BMT, int BN, int WM, int WN, int KCS, int NS, bool FASTK = false, bool XSRC = false, bool LANEK = false>
void launch_fast(const ConvK& k, dim3 grid, hipStream_t st) {
    constexpr int LR_ = 64 * WM * WN / KCS, BNP_ = (BN + LR_ - 1) / LR_ * LR_;       // filter rows padded to whole loader passes
    size_t stage = (size_t)NS * (BMT + BNP_) * KCS * 16 + (k.remap ? 128 : 0);   // stage ring (+ remap table)
    // a single k-step (1x1 layers with <= 64 input channels: Conv2d_3b, the 64-channel dgrads) only ever touches ring stage 0: ask for one
    // stage, so that more of these memory-bound workgroups are resident per CU and their loads / stores overlap (DIN_CONV_ONESTAGE=0: off)
    const bool one_stage_ok = !(DIN_OPT("DIN_CONV_ONESTAGE") && atoi(DIN_OPT("DIN_CONV_ONESTAGE")) == 0);
    if (one_stage_ok && !k.remap && k.xsteps == 0 && k.ks_per_split * (8 / KCS) <= 1) stage = (size_t)(BMT + BNP_) * KCS * 16;
    size_t epi = (size_t)BMT * (BN * sizeof(T) + 16);
    size_t lds = stage > epi ? stage : epi;
    auto kern = conv_gather_fast_kernel<T, BMT, BN, WM, WN, KCS, NS, false, FASTK, XSRC, LANEK>;
    if (lds > 65536) raise_lds_limit(kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, st, k);
}

// the per-lane k-walk (LANEK) serves: bf16 8-wave 128-pixel tiles, single source, no tap remap, tap-major k-order, reduction channels that
// are NOT whole k-steps per tap but at least one k-step wide (so a k-step crosses at most one tap boundary)
static bool gather_lanek(const ConvK& k) {
    const char* lv = DIN_OPT("DIN_CONV_LANEK");
    const char* fv = DIN_OPT("DIN_CONV_FASTK");
    // measured (tools/ab_lanek.sh, profiles/r06_lanek.txt): forward launches +4..7 % (Conv2d_4a 1929 -> 1858 us, the 160-channel 7-tap layers
    // 165 -> 155 us); data gradients (ReLU mask / accumulate operands in the epilogue, the register file full) 1-2 % SLOWER: forward only
    // unless DIN_CONV_LANEK=2
    const int mode = lv ? atoi(lv) : 1;
    if (mode == 1 && (k.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM))) return false;
    return mode != 0 && (fv ? atoi(fv) != 0 : true) && !k.remap && k.nsrc == 0 && !k.korder && k.xsteps == 0 &&
           (k.cpt % 8) != 0 && k.cpt >= 8 && k.kh * k.kw > 1 && k.kh * k.kw <= 31;
}

// 8-wave 128 x BN tile: the scalar-walk specialisation (FASTK) whenever the launch qualifies (bf16, whole k-steps per tap, no tap remap,
// taps-inside-chunks k-order or a single tap); DIN_CONV_FASTK=0 keeps the general loop
template <typename T, int BN>
void launch_wave8(const ConvK& k, dim3 grid, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        const char* fv = DIN_OPT("DIN_CONV_FASTK");
        const bool want = fv ? atoi(fv) != 0 : true;
        const bool fastk = want && !k.remap && k.nsrc == 0 && (k.cpt % 8) == 0 && (k.korder || k.kh * k.kw == 1);
        if constexpr (BN == 192) {
            // wave grid 2 x 4 (64 pixels x 48 filters per wave: 4 + 3 fragments per 12 MFMAs) instead of 4 x 2 (32 x 96: 2 + 6): an eighth
            // fewer LDS fragment reads for the same tile (experiment switch DIN_CONV_WAVEGRID=24)
            const bool grid24 = DIN_OPT("DIN_CONV_WAVEGRID") && atoi(DIN_OPT("DIN_CONV_WAVEGRID")) == 24;
            if (grid24) {
                if (fastk) launch_fast<T, 128, BN, 2, 4, 8, 2, true>(k, grid, st);
                else launch_fast<T, 128, BN, 2, 4, 8, 2>(k, grid, st);
                return;
            }
        }
#ifdef DIN_EXPERIMENTS
        if constexpr (BN >= 128) {
            const bool wg3 = DIN_OPT("DIN_CONV_WG3") && atoi(DIN_OPT("DIN_CONV_WG3")) == 1;
            if (fastk && wg3) { launch_fast<T, 128, BN, 2, 2, 4, 2, true>(k, grid, st); return; }
        }
#endif
        if (fastk) {
            launch_fast<T, 128, BN, 4, 2, 8, 2, true>(k, grid, st);
            return;
        }
        if (gather_lanek(k)) {
            launch_fast<T, 128, BN, 4, 2, 8, 2, true, false, true>(k, grid, st);
            return;
        }
    }
    if constexpr (sizeof(T) == 2 && (BN == 192 || BN == 160)) {
        // experiment switch DIN_CONV_RING=3: three 32-deep stages (two in flight, one counted vmcnt per barrier) instead of two 64-deep ones
        // (one in flight, vmcnt(0)) for the general loop's 8-wave tiles -- same LDS budget (72 vs 80 KiB per workgroup), half the MFMAs per barrier
        #ifdef DIN_EXPERIMENTS
        const bool ring3 = DIN_OPT("DIN_CONV_RING") && atoi(DIN_OPT("DIN_CONV_RING")) == 3;
#else
        constexpr bool ring3 = false;
#endif
        if (ring3 && !k.remap) { launch_fast<T, 128, BN, 4, 2, 4, 3>(k, grid, st); return; }
    }
    launch_fast<T, 128, BN, 4, 2, 8, 2>(k, grid, st);
}

template <typename T, int BN>
void launch_fast_multi(const ConvK& k, dim3 grid, hipStream_t st) {
    size_t epi = (size_t)128 * (BN * sizeof(T) + 16);
    const char* pv = DIN_OPT("DIN_CONV_PIPE");
    if (sizeof(T) == 2 && (BN % 64 == 0 || (pv && atoi(pv) == 8)) && !(pv && atoi(pv) == 4)) {   // 8 waves (bf16) where the filter tile is whole 64-row loader passes
        size_t stage8 = (size_t)2 * (128 + (BN + 63) / 64 * 64) * 8 * 16;
        size_t lds8 = stage8 > epi ? stage8 : epi;
        auto kern = conv_gather_fast_kernel<T, 128, BN, 4, 2, 8, 2, true>;
        if (lds8 > 65536) raise_lds_limit(kern, lds8);
        hipLaunchKernelGGL(kern, grid, dim3(512), lds8, st, k);
        return;
    }
    size_t stage = (size_t)2 * (128 + BN) * 8 * 16;
    size_t lds = stage > epi ? stage : epi;
    auto kern = conv_gather_fast_kernel<T, 128, BN, 2, 2, 8, 2, true>;
    if (lds > 65536) raise_lds_limit(kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k);
}

template <typename T>
void launch_gather(const ConvK& k, int n_px_tiles, int bm, int bn, hipStream_t st) {
    if (k.nsrc > 0) {
        dim3 mgrid(n_px_tiles * k.n_co_tiles, 1);
        if (bn == 64) launch_fast_multi<T, 64>(k, mgrid, st);
        else if (bn == 96) launch_fast_multi<T, 96>(k, mgrid, st);
        else if (bn == 160) launch_fast_multi<T, 160>(k, mgrid, st);
        else if (bn == 192) launch_fast_multi<T, 192>(k, mgrid, st);
        else launch_fast_multi<T, 128>(k, mgrid, st);
        return;
    }
    const bool fast = k.divy == 1 && k.divx == 1 && k.kh * k.kw <= 32;
    dim3 grid(n_px_tiles * k.n_co_tiles, k.splitk);
    if (!fast) {
        size_t lds = 2 * (BM + 128) * KC * 16;
        if (bn == 64) hipLaunchKernelGGL((conv_gather_generic_kernel<T, 64>), grid, dim3(NTHREADS), lds, st, k);
        else hipLaunchKernelGGL((conv_gather_generic_kernel<T, 128>), grid, dim3(NTHREADS), lds, st, k);
        return;
    }
    // ring geometry per tile, from A/B runs of tools/conv_bench.py (DIN_CONV_PIPE=0/1 switches the alternatives):
    //   256x64  : 4 stages x 4 chunks (80 KiB)  -- short-K, latency-bound launches gain 9 % from the deeper ring
    //   others  : 2 stages x 8 chunks           -- MFMA-dense tiles lose 8-10 % when the stage (and the barrier interval) is halved
    const char* pv = DIN_OPT("DIN_CONV_PIPE");
    const int pipe = pv ? atoi(pv) : -1;
    if (bm == 256 && bn == 64) { if (pipe != 0) launch_fast<T, 256, 64, 4, 1, 4, 4>(k, grid, st); else launch_fast<T, 256, 64, 4, 1, 8, 2>(k, grid, st); }
    else if (bm == 256 && bn == 256) {
        if constexpr (sizeof(T) == 2) { if (pipe == 1) launch_fast<T, 256, 256, 4, 2, 4, 4>(k, grid, st); else launch_fast<T, 256, 256, 4, 2, 8, 2>(k, grid, st); }
    }
    else if (bm == 256) {
        if constexpr (sizeof(T) == 2) {
            if (bn == 96) launch_fast<T, 256, 96, 2, 2, 8, 2>(k, grid, st);
#ifdef DIN_EXPERIMENTS
            // experiment (round 4, DIN_CONV_W16=1 with DIN_CONV_TILE=256): the 256-pixel sixteen-wave tile for the 160- / 128-filter 7-tap layers
            // (Mixed_6b-6d: 98 / 85 instead of 71 / 64 FLOP per staged byte)
            else if ((bn == 160 || bn == 128) && DIN_OPT("DIN_CONV_W16") && atoi(DIN_OPT("DIN_CONV_W16")) == 1) {
                const char* fv = DIN_OPT("DIN_CONV_FASTK");
                const bool fastk = (fv ? atoi(fv) != 0 : true) && !k.remap && k.nsrc == 0 && (k.cpt % 8) == 0 && (k.korder || k.kh * k.kw == 1);
                if (bn == 160) { if (fastk) launch_fast<T, 256, 160, 8, 2, 8, 2, true>(k, grid, st); else launch_fast<T, 256, 160, 8, 2, 8, 2>(k, grid, st); }
                else { if (fastk) launch_fast<T, 256, 128, 8, 2, 8, 2, true>(k, grid, st); else launch_fast<T, 256, 128, 8, 2, 8, 2>(k, grid, st); }
            }
#endif
            else if (bn == 160) launch_fast<T, 256, 160, 2, 2, 8, 2>(k, grid, st);
            else if (bn == 192) {
                // experiment (DIN_CONV_W16=1 with DIN_CONV_TILE=256): sixteen waves as 8 x 2 on the 256 x 192 tile -- the 128 x 192 kernel's wave
                // tile and four waves per SIMD, but ONE filter stage per 256 pixels: 64 instead of 80 LDS-DMA transfers per 256-pixel k-step
                #ifdef DIN_EXPERIMENTS
                const bool w16 = DIN_OPT("DIN_CONV_W16") && atoi(DIN_OPT("DIN_CONV_W16")) == 1;
#else
                constexpr bool w16 = false;
#endif
                const char* fv = DIN_OPT("DIN_CONV_FASTK");
                const bool fastk = (fv ? atoi(fv) != 0 : true) && !k.remap && k.nsrc == 0 && (k.cpt % 8) == 0 && (k.korder || k.kh * k.kw == 1);
                if (w16 && fastk) launch_fast<T, 256, 192, 8, 2, 8, 2, true>(k, grid, st);
                else if (w16) launch_fast<T, 256, 192, 8, 2, 8, 2>(k, grid, st);
                else launch_fast<T, 256, 192, 4, 2, 8, 2>(k, grid, st);
            }
            else { if (pipe == 1) launch_fast<T, 256, 128, 4, 2, 4, 4>(k, grid, st); else launch_fast<T, 256, 128, 4, 2, 8, 2>(k, grid, st); }
        }
    }
    else if (bn == 64) {
        if (pipe == 1) launch_fast<T, 128, 64, 2, 2, 8, 3>(k, grid, st);
        else if (pipe != 4 && sizeof(T) == 2) launch_wave8<T, 64>(k, grid, st);
        else launch_fast<T, 128, 64, 2, 2, 8, 2>(k, grid, st);
    }
    // 128 x 96: four waves (eight measured 5-14 % slower on Conv2d_3b) -- except the parity classes of a strided dgrad, whose scattered,
    // epilogue-bound tiles gain 5 % from eight waves (Mixed_6a.branch3x3 dgrad 1379 -> 1312 us)
    else if (bn == 96) {
        if constexpr (sizeof(T) == 2) { if (k.xsteps > 0) { launch_fast<T, 128, 96, 4, 2, 8, 2, false, true>(k, grid, st); return; } }
        if ((pipe == 8 || (k.remap && pipe != 4)) && sizeof(T) == 2) launch_fast<T, 128, 96, 4, 2, 8, 2>(k, grid, st); else launch_fast<T, 128, 96, 2, 2, 8, 2>(k, grid, st);
    }
    else if (bn == 160) { if (pipe != 4 && sizeof(T) == 2) launch_wave8<T, 160>(k, grid, st); else launch_fast<T, 128, 160, 2, 2, 8, 2>(k, grid, st); }
    // 128 x {128,160,192}: 8 waves (4 x 2, four per SIMD at two workgroups per CU) -- same LDS ring, more waves to hide the stage waits:
    // +8..12 % on the 7-tap layers, +24 % on thin-K dgrads (bf16 only; DIN_CONV_PIPE=4 restores the 4-wave form)
    else if (bn == 192) { if (pipe != 4 && sizeof(T) == 2) launch_wave8<T, 192>(k, grid, st); else launch_fast<T, 128, 192, 2, 2, 8, 2>(k, grid, st); }
    else { if (pipe == 1) launch_fast<T, 128, 128, 2, 2, 4, 4>(k, grid, st); else if (pipe != 4 && sizeof(T) == 2) launch_wave8<T, 128>(k, grid, st); else launch_fast<T, 128, 128, 2, 2, 8, 2>(k, grid, st); }
}

// 256-pixel software-pipelined tiles (conv_gather_pipe.hip): bf16 launches with whole 32-channel blocks per tap whose filter tile the
// planner set to 128 / 192 / 256 and that still give every CU at least two tiles.  OPT-IN (DIN_GATHER_PIPE=1; 2 = also on small launches,
// used by the tests): measured against the 128-pixel 8-wave kernels at two workgroups per CU it is +5 % on the 7x1 forward but -2..-10 %
// on 1x1 / 1x7 forwards and on every dgrad (one workgroup per CU: no prologue / epilogue overlap; 64-byte instead of 128-byte gather
// segments per pixel), 520 vs 523 clips/s end to end (profiles/r02_gather_pipe_experiment.txt).
bool want_gather_pipe(int dtype, int64_t M, int cred, int taps, int bn, int splitk, int n_co_tiles) {
    const char* ev = DIN_OPT("DIN_GATHER_PIPE");
    const int mode = ev ? atoi(ev) : 0;
    if (!mode || dtype != DIN_BF16 || cred % 32 != 0 || taps > 32 || taps < 1 || splitk != 1 || !din_gather::gather_pipe_tile_ok(bn)) return false;
    const int64_t tiles = (M + 255) / 256 * n_co_tiles;
    return tiles >= (mode == 2 ? 1 : 512);
}

int run_gather(ConvK& k, GatherPlan g, int dtype, void* workspace, int64_t ws_bytes, hipStream_t st, const char* what) {
    const bool fast = k.divy == 1 && k.divx == 1 && k.kh * k.kw <= 32;
    if (!fast && g.bn != 64 && g.bn != 128) { g.bn = 128; g.n_co_tiles = (k.Cout + 127) / 128; }
    if (g.bm == 256 && (!fast || k.remap || g.splitk > 1)) {
        g.bm = 128; if (g.bn == 256) g.bn = 128;
        g.n_px_tiles = (k.M + 127) / 128; g.n_co_tiles = (k.Cout + g.bn - 1) / g.bn;
    }
    k.cpt = g.cpt; k.Q = g.Q; k.nk = g.nk;
    if (!k.remap) k.wld = g.nk * KC;
    {
        const char* ko = DIN_OPT("DIN_CONV_KORDER");
        const bool want = ko ? atoi(ko) != 0 : true;
        k.korder = (want && fast && !k.remap && g.splitk == 1 && (g.cpt % KC) == 0 && k.kh * k.kw > 1) ? 1 : 0;
    }
    k.splitk = g.splitk; k.ks_per_split = g.ks_per_split; k.n_co_tiles = g.n_co_tiles;
    if (const char* eb = DIN_OPT("DIN_CONV_EPI_BATCH")) { if (atoi(eb) == 0) k.flags |= 0x100; }
    // timing experiments only (results are WRONG): DIN_GATHER_KNOCK bit 0 = the pixel-tile transfers of the scalar-walk loop fetch nothing
    // (all lanes out of range: issued, landed as zeros, no cache / HBM access), bit 1 = the same for the filter tile, bit 2 = no MFMA
    // -- only in -DDIN_EXPERIMENTS builds (conv_gather.h)
#ifdef DIN_EXPERIMENTS
    if (const char* kn = DIN_OPT("DIN_GATHER_KNOCK")) k.flags |= (atoi(kn) & 7) << 9;
#endif
    if (DIN_OPT("DIN_DEBUG_PLAN"))
        fprintf(stderr, "[din] %s M=%d NB=%d HxW=%dx%d Cin=%d Cout=%d k=%dx%d ay=%d cy=%d tile=%dx%d splitk=%d korder=%d remap=%d flags=%d dtype=%d\n", what,
                k.M, k.NB, k.H, k.W, k.Cin, k.Cout, k.kh, k.kw, k.ay, k.cy, g.bm, g.bn, g.splitk, k.korder, k.remap, k.flags, dtype);
    if (k.csplit > 0 && (!fast || g.splitk > 1)) DIN_FAIL(DIN_E_ARG, "%s: two destinations need the staged epilogue of the buffer-addressed kernel", what);
    if (g.splitk > 1) {
        if (ws_bytes < g.ws_bytes || workspace == nullptr)
            DIN_FAIL(DIN_E_WORKSPACE, "%s: workspace %lld < %lld bytes", what, (long long)ws_bytes, (long long)g.ws_bytes);
        k.partial = reinterpret_cast<float*>(workspace);
    }
    if (fast && g.splitk == 1 && k.nsrc == 0 && k.xsteps == 0 && din_gather::conv1x1_regw_eligible(k, dtype)) {
        // 1x1 layers with a 640..768-channel reduction over a large map (the Mixed_6 block entries): filters resident in registers (conv_regw.hip)
        if (din_gather::launch_conv1x1_regw(k, st)) DIN_FAIL(DIN_E_LAUNCH, "%s: conv1x1_regw launch failed", what);
        return DIN_OK;
    }
    if (fast && g.splitk == 1 && k.nsrc == 0 && k.xsteps == 0 && din_gather::conv1x1_stream_eligible(k, dtype)) {
        // 1x1 layers with a short reduction over a large map: persistent streaming kernel (conv_stream.hip)
        if (din_gather::launch_conv1x1_stream(k, st)) DIN_FAIL(DIN_E_LAUNCH, "%s: conv1x1_stream launch failed", what);
        return DIN_OK;
    }
    {
        // mid-network multi-tap layers: halo tiles + filter-slab ring (conv_halo_kernel); the stem shapes keep their own kernel below
        HaloPlan hp;
        const bool stem_shape = k.kh == 3 && k.kw == 3 && (k.Cin == 32 || k.Cin == 64) && k.Cout <= 64 && !(k.Cin == 64 && k.Cout > 32) &&
                                (int64_t)k.M >= 256 * 1024;
        if (fast && !k.remap && k.nsrc == 0 && k.csplit == 0 && g.splitk == 1 && !stem_shape && k.ay == 1 && k.ax == 1 && (k.cy == 1 || k.cy == -1) &&
            (k.cx == 1 || k.cx == -1) && k.cy == k.cx && k.out_sy == 0 && k.ldi % 8 == 0 && k.cioff % 8 == 0 && k.ldo % 4 == 0 && k.cooff % 4 == 0 &&
            (!(k.flags & DIN_CONV_MASK) || (k.ldm % 4 == 0 && k.moff % 4 == 0)) &&
            (long long)k.H * k.W * k.ldi * 2 < 0x7fffffffll && (long long)k.OH * k.OW * k.ldo * 2 < 0x7fffffffll &&
            (long long)k.OH * k.OW * (k.ldm > 0 ? k.ldm : 1) * 2 < 0x7fffffffll &&
            plan_halo(dtype, k.kh, k.kw, k.Cin, k.Cout, k.OH, k.OW, k.M, hp)) {
            k.n_co_tiles = hp.n_co_tiles;
            const int tiles = ((k.OH + hp.th - 1) / hp.th) * ((k.OW + hp.tw - 1) / hp.tw) * k.NB * hp.n_co_tiles;
            dim3 grid(tiles < 256 ? tiles : 256);
            auto launch = [&](auto kern) {
                raise_lds_limit(kern, hp.lds);
                hipLaunchKernelGGL(kern, grid, dim3(512), hp.lds, st, k);
            };
            bool done = true;
            if (k.kh == 3 && k.kw == 3 && hp.nwv == 16) {
                auto launch16 = [&](auto kern) {
                    raise_lds_limit(kern, hp.lds);
                    hipLaunchKernelGGL(kern, grid, dim3(1024), hp.lds, st, k);
                };
                if (hp.bn == 64) launch16(conv_halo_kernel<64, 3, 3, 8, 32, 2, 16>);
                else if (hp.bn == 80) launch16(conv_halo_kernel<80, 3, 3, 8, 32, 2, 16>);
                else launch16(conv_halo_kernel<96, 3, 3, 8, 32, 2, 16>);
            }
            else if (k.kh == 3 && k.kw == 3) { if (hp.bn == 64) launch(conv_halo_kernel<64, 3, 3, 8, 32, 3>); else launch(conv_halo_kernel<96, 3, 3, 8, 32, 3>); }
            else if (k.kh == 1 && k.kw == 7) { if (hp.bn == 64) launch(conv_halo_kernel<64, 1, 7, 16, 16, 3>); else launch(conv_halo_kernel<96, 1, 7, 16, 16, 3>); }
            else if (k.kh == 7 && k.kw == 1) { if (hp.bn == 64) launch(conv_halo_kernel<64, 7, 1, 16, 16, 3>); else launch(conv_halo_kernel<96, 7, 1, 16, 16, 3>); }
            else done = false;
            if (done) { DIN_CHECK_LAUNCH(what); return DIN_OK; }
        }
    }
    {
        // stem layers: stationary filters + halo tiles (conv_small_kernel)
        const char* sv = DIN_OPT("DIN_CONV_SMALL");
        const bool want = sv ? atoi(sv) != 0 : true;
        const bool common = want && dtype == DIN_BF16 && fast && !k.remap && k.nsrc == 0 && k.csplit == 0 && g.splitk == 1 && k.kh == 3 && k.kw == 3 &&
                            k.Cout <= 64 && k.Cout % 8 == 0 && k.cooff % 8 == 0 && k.ldo % 8 == 0 && k.ldi % 8 == 0 && k.cioff % 8 == 0 &&
                            (!(k.flags & DIN_CONV_MASK) || (k.ldm % 8 == 0 && k.moff % 8 == 0)) &&
                            (long long)k.H * k.W * k.ldi * 2 < 0x7fffffffll && (long long)k.OH * k.OW * k.ldo * 2 < 0x7fffffffll &&
                            (long long)k.OH * k.OW * (k.ldm > 0 ? k.ldm : 1) * 2 < 0x7fffffffll && k.out_sy == 0 && (int64_t)k.M >= 256 * 1024;
        // 32/64-channel 3x3 stride-1 layers (fwd and dgrad) ...
        const bool stem = common && (g.cpt == 4 || g.cpt == 8) && g.cpt * 8 == k.Cin && k.ay == 1 && k.ax == 1 &&
                          (k.cy == 1 || k.cy == -1) && (k.cx == 1 || k.cx == -1) && !(g.cpt == 8 && k.Cout > 32);
        // ... and the image layer: <= 8 (zero-padded) channels per pixel, stride 2, forward only
        const bool image = common && g.cpt == 1 && k.Cin <= 8 && k.ldi >= 8 && k.Cout <= 32 && k.ay == 2 && k.ax == 2 && k.cy == 1 && k.cx == 1 &&
                           k.wld >= 12;
        if (stem || image) {
            const int bnS = k.Cout <= 32 ? 32 : 64;
            const int st_ = image ? 2 : 1;
            const int hpx = (7 * st_ + 3) * (31 * st_ + 3);
            const int hbytes = (hpx * g.cpt * 16 + 1023) / 1024 * 1024;
            // 128-byte pixels (the 64-channel dgrad of Conv2d_2b): one halo buffer (80 KiB) or -- DIN_CONV_SMALL_NBUF8=2 -- two (124 KiB, the next
            // halo in flight under this tile's MFMAs like the 64-byte-pixel variants); one workgroup per CU either way
            const char* nb8 = DIN_OPT("DIN_CONV_SMALL_NBUF8");
            const bool two8 = g.cpt == 8 && (nb8 ? atoi(nb8) == 2 : false);
            const int nbuf = (g.cpt == 8 && !two8) ? 1 : 2;
            const size_t lds = (size_t)(image ? 3 * bnS * 64 : 9 * bnS * g.cpt * 16) + (size_t)nbuf * hbytes + (k.u8 ? 512 : 0);   // (+ the uint8 -> bf16 table)
            dim3 grid(512);
            // dgrad launches with a mask / accumulate operand: the variant that requests them a tile phase early (DIN_CONV_SMALL_EPI=0: in the store loop)
            const bool epi = (k.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM)) && !(DIN_OPT("DIN_CONV_SMALL_EPI") && atoi(DIN_OPT("DIN_CONV_SMALL_EPI")) == 0);
            // the image layer's 42 KiB workgroups fit three to a CU: 768 persistent workgroups measured 690 -> 618 us on the 96 frames
            // (1024: no better, four do not fit); DIN_CONV_IMAGE_GRID overrides
            if (image) { const char* gv = DIN_OPT("DIN_CONV_IMAGE_GRID"); grid.x = gv && atoi(gv) > 0 ? atoi(gv) : 768; }
            auto launch = [&](auto kern) {
                if (lds > 65536) raise_lds_limit(kern, lds);
                hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, st, k);
            };
            if (image && k.u8) launch(conv_small_kernel<1, 32, 2, 3, 3, 2, true>);
            else if (image) launch(conv_small_kernel<1, 32, 2, 3, 3, 2>);
            else if (g.cpt == 4 && bnS == 32) {                                                // (eight waves measured slower here: 706 -> 765 us)
                if (epi) launch(conv_small_kernel<4, 32, 2, 3, 3, 1, false, 4, true>); else launch(conv_small_kernel<4, 32, 2, 3, 3, 1>);
            }
            else {
                // the two 80 KiB variants (one workgroup per CU) run on eight waves; DIN_CONV_SMALL_WAVES=4 restores four
                const bool w8 = !(DIN_OPT("DIN_CONV_SMALL_WAVES") && atoi(DIN_OPT("DIN_CONV_SMALL_WAVES")) == 4);
                auto launch8 = [&](auto kern) {
                    if (lds > 65536) raise_lds_limit(kern, lds);
                    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, k);
                };
                if (g.cpt == 4) { if (w8) launch8(conv_small_kernel<4, 64, 2, 3, 3, 1, false, 8>); else launch(conv_small_kernel<4, 64, 2, 3, 3, 1>); }
                else if (two8 && w8) { if (epi) launch8(conv_small_kernel<8, 32, 2, 3, 3, 1, false, 8, true>); else launch8(conv_small_kernel<8, 32, 2, 3, 3, 1, false, 8>); }
                else if (epi && w8) launch8(conv_small_kernel<8, 32, 1, 3, 3, 1, false, 8, true>);
                else { if (w8) launch8(conv_small_kernel<8, 32, 1, 3, 3, 1, false, 8>); else launch(conv_small_kernel<8, 32, 1, 3, 3, 1>); }
            }
            DIN_CHECK_LAUNCH(what);
            return DIN_OK;
        }
    }
    DIN_REQUIRE(!k.u8, "%s: in_u8 is only served by the image-layer kernel (see din_conv_accepts_u8)", what);
    if (fast && !k.remap && k.nsrc == 0 && want_gather_pipe(dtype, k.M, k.Cin, k.kh * k.kw, g.bn, g.splitk, (k.Cout + g.bn - 1) / g.bn) &&
        g.cpt % 4 == 0 && k.Cout % 8 == 0 && k.cooff % 8 == 0 && k.ldo % 8 == 0 &&
        (!(k.flags & DIN_CONV_MASK) || (k.ldm % 8 == 0 && k.moff % 8 == 0)) && (k.csplit == 0 || (k.csplit % 8 == 0 && k.ldo2 % 8 == 0 && k.cooff2 % 8 == 0))) {
        k.n_co_tiles = (k.Cout + g.bn - 1) / g.bn;
        if (int e = din_gather::launch_gather_pipe(k, g.bn, (k.M + 255) / 256, st)) return e;
        DIN_CHECK_LAUNCH(what);
        return DIN_OK;
    }
    if (dtype == DIN_F32) launch_gather<float>(k, g.n_px_tiles, g.bm, g.bn, st);
    else launch_gather<bf16_t>(k, g.n_px_tiles, g.bm, g.bn, st);
    DIN_CHECK_LAUNCH(what);
    if (g.splitk > 1) {
        int64_t total = (int64_t)k.M * k.Cout;
        int cpad = g.n_co_tiles * g.bn;
        if (dtype == DIN_F32) hipLaunchKernelGGL(conv_splitk_finish_kernel<float>, dim3(grid_1d(total, 256)), dim3(256), 0, st, k, cpad);
        else hipLaunchKernelGGL(conv_splitk_finish_kernel<bf16_t>, dim3(grid_1d(total, 256)), dim3(256), 0, st, k, cpad);
        DIN_CHECK_LAUNCH(what);
    }
    return DIN_OK;
}

// column sums of a pixel-major tensor view -> out[c] (zeroed here, fp32 atomics across row slabs)
static int launch_colsum(int dtype, const void* g, float* out, int64_t M, int c, int ld, int coff, hipStream_t st) {
    hipMemsetAsync(out, 0, sizeof(float) * c, st);
    const int epc = dtype == DIN_F32 ? 4 : 8;
    if (c % epc == 0 && c / epc <= 256 && ld % epc == 0 && coff % epc == 0) {
        // 256 workgroups (one per CU), eight 16-byte loads in flight per thread, each streaming a contiguous slab of rows (XCD-contiguous order).
        // Every workgroup ends with `c` float atomics on the SAME few cache lines, which L2 serialises at ~44 ns per workgroup: the kernel's time
        // grew with its workgroup count (1024: 45 us, 2048: 64 us, 4096: 110 us on the 192-channel maps; 256: 30 us -- tools/colsum_probe.py;
        // the seven launches of the default step 335 -> 248 us).  DIN_COLSUM_WGS / DIN_COLSUM_UNROLL: tuning aids
        const int wgs = DIN_OPT("DIN_COLSUM_WGS") ? atoi(DIN_OPT("DIN_COLSUM_WGS")) : 256;
        const int unr = DIN_OPT("DIN_COLSUM_UNROLL") ? atoi(DIN_OPT("DIN_COLSUM_UNROLL")) : 8;
        int64_t rpb = ceil_div64(M, wgs > 0 ? wgs : 1024);
        if (rpb < 64) rpb = 64;
        int blocks = (int)ceil_div64(M, rpb);
        if (dtype == DIN_F32)
            hipLaunchKernelGGL(colsum_vec_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)g, out, M, c, ld, coff, rpb);
        else if (unr == 8)
            hipLaunchKernelGGL((colsum_vec_kernel<bf16_t, 8>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, out, M, c, ld, coff, rpb);
        else if (unr == 16)
            hipLaunchKernelGGL((colsum_vec_kernel<bf16_t, 16>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, out, M, c, ld, coff, rpb);
        else
            hipLaunchKernelGGL(colsum_vec_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, out, M, c, ld, coff, rpb);
    } else {
        int64_t rpb = 512;
        int blocks = (int)ceil_div64(M, rpb);
        if (dtype == DIN_F32)
            hipLaunchKernelGGL(colsum_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)g, out, M, c, ld, coff, rpb);
        else
            hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, out, M, c, ld, coff, rpb);
    }
    DIN_CHECK_LAUNCH("colsum");
    return DIN_OK;
}

}  // namespace

extern "C" {

int64_t din_conv_packed_elems(const din_conv_desc* d, int transposed) {
    if (!d) return 0;
    int epc = epc_of(d->dtype);
    int cred = transposed ? d->cout : d->cin, cprod = transposed ? d->cin : d->cout;
    int cpt = pad_to(cred, epc) / epc;
    int nk = (d->kh * d->kw * cpt + KC - 1) / KC;
    return (int64_t)pad_to(cprod, 256) * nk * KC * epc;
}

int din_conv_pack_weights(const din_conv_desc* d, const float* w, const float* scale, void* wpk, int transposed, void* stream) {
    DIN_REQUIRE(d && w && wpk, "conv_pack: null pointer");
    int epc = epc_of(d->dtype);
    int cred = transposed ? d->cout : d->cin, cprod = transposed ? d->cin : d->cout;
    int inner_pad = pad_to(cred, epc);
    int cpt = inner_pad / epc;
    int nk = (d->kh * d->kw * cpt + KC - 1) / KC;
    int kelems = nk * KC * epc;
    int rows_pad = pad_to(cprod, 256);
    int64_t total = (int64_t)rows_pad * kelems;
    hipStream_t st = as_stream(stream);
    if (d->kh == 1 && d->kw == 1 && total >= (1 << 20) && !(DIN_OPT("DIN_PACK_TILES") && atoi(DIN_OPT("DIN_PACK_TILES")) == 0)) {
        // (tap-major k = channel index for a single tap; the padded tail of every row / the padded rows are written as zeros)
        dim3 grid((kelems + 63) / 64, (rows_pad + 63) / 64);
        if (d->dtype == DIN_F32) hipLaunchKernelGGL(conv_pack_1x1_kernel<float>, grid, dim3(256), 0, st, w, scale, (float*)wpk, d->cout, d->cin, rows_pad, kelems, transposed);
        else hipLaunchKernelGGL(conv_pack_1x1_kernel<bf16_t>, grid, dim3(256), 0, st, w, scale, (bf16_t*)wpk, d->cout, d->cin, rows_pad, kelems, transposed);
        DIN_CHECK_LAUNCH("conv_pack(1x1)");
        return DIN_OK;
    }
    if (d->dtype == DIN_F32)
        hipLaunchKernelGGL(conv_pack_kernel<float>, dim3(grid_1d(total, 256)), dim3(256), 0, st, w, scale, (float*)wpk,
                           d->cout, d->cin, d->kh, d->kw, cprod, rows_pad, cred, inner_pad, kelems, transposed);
    else
        hipLaunchKernelGGL(conv_pack_kernel<bf16_t>, dim3(grid_1d(total, 256)), dim3(256), 0, st, w, scale, (bf16_t*)wpk,
                           d->cout, d->cin, d->kh, d->kw, cprod, rows_pad, cred, inner_pad, kelems, transposed);
    DIN_CHECK_LAUNCH("conv_pack");
    return DIN_OK;
}

int din_conv_pack_desc(const din_conv_desc* d, const float* w, const float* scale, void* wpk, int transposed, din_pack_desc* out) {
    DIN_REQUIRE(d && w && wpk && out, "conv_pack_desc: null pointer");
    const int epc = epc_of(d->dtype);
    const int cred = transposed ? d->cout : d->cin, cprod = transposed ? d->cin : d->cout;
    const int inner_pad = pad_to(cred, epc), cpt = inner_pad / epc;
    const int nk = (d->kh * d->kw * cpt + KC - 1) / KC;
    out->w = (uint64_t)(uintptr_t)w; out->scale = (uint64_t)(uintptr_t)scale; out->out = (uint64_t)(uintptr_t)wpk;
    out->cout = d->cout; out->cin = d->cin; out->kh = d->kh; out->kw = d->kw;
    out->rows = cprod; out->rows_pad = pad_to(cprod, 256); out->inner = cred; out->inner_pad = inner_pad; out->kelems = nk * KC * epc;
    out->transposed = transposed; out->dtype = d->dtype;
    return DIN_OK;
}

int din_conv_pack_multi(const din_pack_desc* table, const int32_t* layer_of, const int32_t* chunk_index, int nblocks, int chunk_elems,
                        void* stream) {
    DIN_REQUIRE(table && layer_of && chunk_index && nblocks >= 0 && chunk_elems > 0, "conv_pack_multi: bad argument");
    if (nblocks == 0) return DIN_OK;
    hipLaunchKernelGGL(conv_pack_multi_kernel, dim3(nblocks), dim3(256), 0, as_stream(stream), table, layer_of, chunk_index, chunk_elems);
    DIN_CHECK_LAUNCH("conv_pack_multi");
    return DIN_OK;
}

int din_conv_kernel_tile(const din_conv_desc* d, int which, int32_t* bm, int32_t* bn) {
    DIN_REQUIRE(d && bm && bn && which >= 0 && which <= 2, "conv_kernel_tile: bad argument");
    if (which == 2) { WgradPlan wp = plan_wgrad(d); if (wp.small == 4) { *bm = 3; *bn = wp.bco; return DIN_OK; } *bm = wp.small ? 0 : wp.bco; *bn = wp.small ? wp.bco : (wp.pipe ? 2000 + wp.bk : wp.ring ? 1000 + wp.bk : WG_TILE); return DIN_OK; }
    const bool strided = which == 1 && (d->sh > 1 || d->sw > 1);
    GatherPlan g = which == 0 ? plan_gather(d->nb * d->oh * d->ow, d->cin, d->cout, d->kh * d->kw, d->dtype)
                              : plan_gather(d->nb * (strided ? (d->h + d->sh - 1) / d->sh * ((d->w + d->sw - 1) / d->sw) : d->h * d->w),
                                            d->cout, d->cin, d->kh * d->kw, d->dtype, strided);
    if (g.bm == 256 && (strided || g.splitk > 1)) { g.bm = 128; if (g.bn == 256) g.bn = 128; }
    *bm = g.bm; *bn = g.bn;
    {   // conv_gather_pipe_kernel<BN>: bm = 2 (the halo / stem kernels below still take precedence, as in run_gather)
        const int cred = which == 0 ? d->cin : d->cout, cprod = which == 0 ? d->cout : d->cin;
        const int64_t M = which == 0 ? (int64_t)d->nb * d->oh * d->ow : (int64_t)d->nb * d->h * d->w;
        if (!strided && d->dh == 1 && d->dw == 1 && want_gather_pipe(d->dtype, M, cred, d->kh * d->kw, g.bn, g.splitk, (cprod + g.bn - 1) / g.bn) &&
            cprod % 8 == 0 && (which == 0 ? d->cooff % 8 == 0 && d->ldo % 8 == 0 : d->cioff % 8 == 0 && d->ldi % 8 == 0)) *bm = 2;
    }
    {   // mid-network multi-tap layers run conv_halo_kernel: bm = 1
        HaloPlan hp;
        const int cred = which == 0 ? d->cin : d->cout, cprod = which == 0 ? d->cout : d->cin;
        const int64_t M = which == 0 ? (int64_t)d->nb * d->oh * d->ow : (int64_t)d->nb * d->h * d->w;
        const int oh_ = which == 0 ? d->oh : d->h, ow_ = which == 0 ? d->ow : d->w;
        if (d->sh == 1 && d->sw == 1 && d->dh == 1 && d->dw == 1 && g.splitk == 1 && plan_halo(d->dtype, d->kh, d->kw, cred, cprod, oh_, ow_, M, hp)) { *bm = 1; *bn = hp.bn; }
    }
    {   // 1x1 layers with a short reduction over a large map run conv1x1_stream_kernel (conv_stream.hip): bm = 4
        // (same conditions as din_gather::conv1x1_stream_eligible, evaluated on the descriptor)
        const char* sv = DIN_OPT("DIN_CONV_STREAM");
        const int mode = sv ? atoi(sv) : 1;
        const int cred = which == 0 ? d->cin : d->cout, cprod = which == 0 ? d->cout : d->cin;
        const int64_t M = which == 0 ? (int64_t)d->nb * d->oh * d->ow : (int64_t)d->nb * d->h * d->w;
        const int ldr = which == 0 ? d->ldi : d->ldo, offr = which == 0 ? d->cioff : d->cooff;
        const int ldp = which == 0 ? d->ldo : d->ldi, offp = which == 0 ? d->cooff : d->cioff;
        const int blocks = (pad_to(cred, 8) / 8 + 7) / 8;
        if (mode && d->dtype == DIN_BF16 && d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && !d->in_u8 &&
            g.splitk == 1 && cprod % 8 == 0 && ldp % 8 == 0 && offp % 8 == 0 && ldr % 8 == 0 && offr % 8 == 0 &&
            M * ldr * 2 < 0x7fffffffll && M * ldp * 2 < 0x7fffffffll && pad_to(cprod, din_gather::conv1x1_stream_tile(cprod)) * 4 <= 2048 &&
            (mode == 2 || (blocks <= 6 && M >= (DIN_OPT("DIN_CONV_STREAM_MINPIX") ? atoll(DIN_OPT("DIN_CONV_STREAM_MINPIX")) : 256 * 1024) && (cprod <= 96 || (cprod <= 192 && which == 0))))) { *bm = 4; *bn = din_gather::conv1x1_stream_tile(cprod); }
    }
    {   // 1x1 layers with a 640..768-channel reduction over a large map run conv1x1_regw_kernel (conv_regw.hip, filters resident in registers):
        // bm = 5, bn = 192 | 128 filters per class (same conditions as din_gather::conv1x1_regw_eligible, evaluated on the descriptor; single destination, no accumulate)
        const char* rv = DIN_OPT("DIN_CONV_REGW");
        const int mode = rv ? atoi(rv) : 1;
        const int cred = which == 0 ? d->cin : d->cout, cprod = which == 0 ? d->cout : d->cin;
        const int64_t M = which == 0 ? (int64_t)d->nb * d->oh * d->ow : (int64_t)d->nb * d->h * d->w;
        const int ldr = which == 0 ? d->ldi : d->ldo, offr = which == 0 ? d->cioff : d->cooff;
        const int ldp = which == 0 ? d->ldo : d->ldi, offp = which == 0 ? d->cooff : d->cioff;
        const int nks = cred % 8 == 0 ? (cred + 63) / 64 * 2 : 0;      // (a multi-source launch pads EACH source to whole stages: din_conv1x1_dgrad_multi decides itself)
        const bool shortk = nks == 6 || nks == 8 || nks == 10;          // Mixed_5: classes of 128 filters, two workgroups per CU
        const char* sk = DIN_OPT("DIN_CONV_REGW_SHORT");
        const char* mp = DIN_OPT("DIN_CONV_REGW_MINPIX");
        if (mode && d->dtype == DIN_BF16 && d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && !d->in_u8 &&
            g.splitk == 1 && (shortk || nks == 20 || nks == 24) && cprod <= (shortk ? 512 : 768) && cprod % 8 == 0 && ldp % 8 == 0 && offp % 8 == 0 &&
            ldr % 8 == 0 && offr % 8 == 0 && M * ldr * 2 < 0x7fffffffll &&
            (mode == 2 || (shortk ? (sk ? atoi(sk) != 0 : true) && M >= (mp ? atoll(mp) : 128 * 1024) && cprod > 96 : M >= (mp ? atoll(mp) : 64 * 1024)))) { *bm = 5; *bn = shortk ? 128 : 192; }
    }
    {   // stem layers run conv_small_kernel (same conditions as run_gather, for tensors with 16-byte aligned channel offsets): bm = 0
        const int cred = which == 0 ? d->cin : d->cout, cprod = which == 0 ? d->cout : d->cin;
        const int64_t M = which == 0 ? (int64_t)d->nb * d->oh * d->ow : (int64_t)d->nb * d->h * d->w;
        const char* sv = DIN_OPT("DIN_CONV_SMALL");
        if ((sv ? atoi(sv) != 0 : true) && d->dtype == DIN_BF16 && d->kh == 3 && d->kw == 3 && d->sh == 1 && d->sw == 1 && d->dh == 1 &&
            d->dw == 1 && (cred == 32 || cred == 64) && cprod <= 64 && cprod % 8 == 0 && !(cred == 64 && cprod > 32) && g.splitk == 1 &&
            M >= 256 * 1024) { *bm = 0; *bn = cprod <= 32 ? 32 : 64; }
        if ((sv ? atoi(sv) != 0 : true) && which == 0 && d->dtype == DIN_BF16 && d->kh == 3 && d->kw == 3 && d->sh == 2 && d->sw == 2 &&
            d->dh == 1 && d->dw == 1 && d->cin <= 8 && d->cout <= 32 && d->cout % 8 == 0 && g.splitk == 1 && M >= 256 * 1024) { *bm = 0; *bn = 32; }
    }
    return DIN_OK;
}

int din_conv_kernel_variant(const din_conv_desc* d, int which, int32_t* flags) {
    DIN_REQUIRE(d && flags && which >= 0 && which <= 1, "conv_kernel_variant: bad argument");
    int32_t bm = 0, bn = 0;
    if (int e = din_conv_kernel_tile(d, which, &bm, &bn)) return e;
    *flags = 0;
    if (bm != 128 || d->dtype != DIN_BF16) return DIN_OK;          // the 8-wave / FASTK instantiations exist for bf16 128 x BN tiles only
    const char* pv = DIN_OPT("DIN_CONV_PIPE");
    const int pipe = pv ? atoi(pv) : -1;
    const bool strided = which == 1 && (d->sh > 1 || d->sw > 1);
    const bool wave8 = ((bn == 64 || bn == 128 || bn == 160 || bn == 192) && pipe != 4 && pipe != 1) ||
                       (bn == 96 && (pipe == 8 || (strided && pipe != 4)));       // (the parity classes of a strided dgrad: launch_gather)
    if (wave8) *flags |= 2;
    const int ntaps = d->kh * d->kw, cred = which == 0 ? d->cin : d->cout;
    const int cpt = pad_to(cred, 8) / 8;
    GatherPlan g = which == 0 ? plan_gather(d->nb * d->oh * d->ow, d->cin, d->cout, ntaps, d->dtype)
                              : plan_gather(d->nb * (strided ? (d->h + d->sh - 1) / d->sh * ((d->w + d->sw - 1) / d->sw) : d->h * d->w), d->cout, d->cin, ntaps, d->dtype, strided);
    const char* ko = DIN_OPT("DIN_CONV_KORDER");
    const char* fv = DIN_OPT("DIN_CONV_FASTK");
    const bool fast = ntaps <= 32 && (which == 0 || (d->sh == 1 && d->sw == 1));      // stride-1 gather: divy == divx == 1, no tap remap
    const bool korder = (ko ? atoi(ko) != 0 : true) && fast && g.splitk == 1 && cpt % KC == 0 && ntaps > 1;
    if (wave8 && fast && cpt % KC == 0 && (korder || ntaps == 1) && (fv ? atoi(fv) != 0 : true)) *flags |= 1;
    const char* lv = DIN_OPT("DIN_CONV_LANEK");                  // bit 2: the per-lane k-walk instantiation (gather_lanek; bit 0 is set with it: FASTK = true)
    const int lmode = lv ? atoi(lv) : 1;                          // (1: forward launches only -- a data gradient carries the mask / accumulate flags)
    if (wave8 && bn != 96 && fast && !strided && cpt % KC != 0 && cpt >= 8 && ntaps > 1 && ntaps <= 31 && (fv ? atoi(fv) != 0 : true) &&
        (lmode == 2 || (lmode == 1 && which == 0))) *flags |= 1 | 4;
    return DIN_OK;
}

int64_t din_conv_workspace_bytes(const din_conv_desc* d, int which) {
    if (!d) return 0;
    if (which == 0) return plan_gather(d->nb * d->oh * d->ow, d->cin, d->cout, d->kh * d->kw, d->dtype).ws_bytes;
    if (which == 1) {
        if ((d->sh > 1 || d->sw > 1) && d->dh == 1 && d->dw == 1 && d->kh * d->kw <= 32) {
            int64_t mx = 0;
            for (int py = 0; py < d->sh; ++py)
                for (int px = 0; px < d->sw; ++px) {
                    const int Ha = (d->h - py + d->sh - 1) / d->sh, Wa = (d->w - px + d->sw - 1) / d->sw;
                    if (Ha <= 0 || Wa <= 0) continue;
                    const int r0c = (py + d->ph) % d->sh, s0c = (px + d->pw) % d->sw;
                    const int khs = r0c < d->kh ? (d->kh - r0c + d->sh - 1) / d->sh : 0;
                    const int kws = s0c < d->kw ? (d->kw - s0c + d->sw - 1) / d->sw : 0;
                    int64_t b = plan_gather(d->nb * Ha * Wa, d->cout, d->cin, khs * kws, d->dtype, true).ws_bytes;
                    if (b > mx) mx = b;
                }
            return mx;
        }
        return plan_gather(d->nb * d->h * d->w, d->cout, d->cin, d->kh * d->kw, d->dtype).ws_bytes;
    }
    return plan_wgrad(d).ws_bytes;
}

int din_conv_fwd(const din_conv_desc* d, const void* in, const void* wpk, const float* bias, void* out, int flags,
                 void* workspace, int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    DIN_REQUIRE(in && wpk && out, "conv_fwd: null pointer");
    DIN_REQUIRE(!(flags & DIN_CONV_BIAS) || bias, "conv_fwd: BIAS flag without bias");
    DIN_REQUIRE(!(flags & (DIN_CONV_ACCUM | DIN_CONV_MASK)), "conv_fwd: ACCUM/MASK are dgrad-only flags");
    ConvK k{};
    k.in = in; k.w = wpk; k.out = out; k.bias = bias; k.mask = nullptr; k.partial = nullptr;
    k.NB = d->nb; k.H = d->h; k.W = d->w; k.Cin = d->cin; k.ldi = d->ldi; k.cioff = d->cioff;
    k.OH = d->oh; k.OW = d->ow; k.Cout = d->cout; k.ldo = d->ldo; k.cooff = d->cooff;
    k.kh = d->kh; k.kw = d->kw;
    k.ay = d->sh; k.by = -d->ph; k.cy = d->dh; k.divy = 1;
    k.ax = d->sw; k.bx = -d->pw; k.cx = d->dw; k.divx = 1;
    k.M = d->nb * d->oh * d->ow; k.flags = flags; k.ldm = 0; k.moff = 0;
    k.in_bytes = (long long)d->nb * d->h * d->w * d->ldi * (d->dtype == DIN_F32 ? 4 : 2);
    k.w_bytes = din_conv_packed_elems(d, 0) * (d->dtype == DIN_F32 ? 4 : 2);
    if (d->in_u8) {                                   // raw uint8 frames: plan as the 8-channel prepared tensor the image layer would read
        DIN_REQUIRE(din_conv_accepts_u8(d), "conv_fwd: in_u8 on a layer din_conv_accepts_u8() rejects");
        k.u8 = reinterpret_cast<const unsigned char*>(in);
        k.ldi = 8; k.cioff = 0; k.in_bytes = (long long)d->nb * d->h * d->w * 16;
    }
    GatherPlan g = plan_gather(k.M, d->cin, d->cout, d->kh * d->kw, d->dtype);
    return run_gather(k, g, d->dtype, workspace, workspace_bytes, as_stream(stream), "conv_fwd");
}

int din_conv_accepts_u8(const din_conv_desc* d) {
    if (!d || d->dtype != DIN_BF16 || d->kh != 3 || d->kw != 3 || d->sh != 2 || d->sw != 2 || d->dh != 1 || d->dw != 1) return 0;
    if (d->cin != 3 || d->cout > 32 || d->cout % 8 != 0 || d->ldo % 8 != 0 || d->cooff % 8 != 0) return 0;
    if ((int64_t)d->nb * d->oh * d->ow < 256 * 1024) return 0;
    if ((long long)d->h * d->w * 16 >= 0x7fffffffll || (long long)d->oh * d->ow * d->ldo * 2 >= 0x7fffffffll) return 0;
    const char* sv = DIN_OPT("DIN_CONV_SMALL");
    if (sv && atoi(sv) == 0) return 0;
    const char* uv = DIN_OPT("DIN_CONV_U8");
    if (uv && atoi(uv) == 0) return 0;
    din_conv_desc t = *d;
    t.in_u8 = 0; t.ldi = 8; t.cioff = 0;             // the plan of the prepared-tensor form must pick the image-layer wgrad kernel
    return plan_wgrad(&t).small == 3 ? 1 : 0;
}

int din_conv_fwd2(const din_conv_desc* d, const void* in, const void* wpk, const float* bias, void* out, void* out2, int ldo2, int cooff2,
                  int csplit, int craw, int flags, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    DIN_REQUIRE(in && wpk && out && out2, "conv_fwd2: null pointer");
    DIN_REQUIRE(!d->in_u8, "conv_fwd2: in_u8 is a din_conv_fwd / din_conv_wgrad option");
    DIN_REQUIRE(!(flags & DIN_CONV_BIAS) || bias, "conv_fwd2: BIAS flag without bias");
    DIN_REQUIRE(!(flags & (DIN_CONV_ACCUM | DIN_CONV_MASK)), "conv_fwd2: ACCUM/MASK are dgrad-only flags");
    const int epc = epc_of(d->dtype);
    DIN_REQUIRE(d->kh * d->kw <= 32 && d->dh == 1 && d->dw == 1, "conv_fwd2: needs the buffer-addressed gather kernel (<= 32 taps)");
    DIN_REQUIRE(csplit > 0 && csplit < d->cout && csplit % epc == 0 && d->cout % epc == 0 && d->ldo % epc == 0 && d->cooff % epc == 0 &&
                ldo2 % epc == 0 && cooff2 % epc == 0 && ldo2 >= cooff2 + (d->cout - csplit) && d->ldo >= d->cooff + csplit,
                "conv_fwd2: split / strides / offsets must be multiples of %d and the destinations must hold their channel ranges", epc);
    ConvK k{};
    k.in = in; k.w = wpk; k.out = out; k.bias = bias; k.mask = nullptr; k.partial = nullptr;
    DIN_REQUIRE(craw == 0 || (craw >= csplit && craw < d->cout && craw % epc == 0), "conv_fwd2: craw must be 0 or a multiple of %d in [csplit, cout)", epc);
    k.out2 = out2; k.ldo2 = ldo2; k.cooff2 = cooff2; k.csplit = csplit; k.craw = craw;
    k.NB = d->nb; k.H = d->h; k.W = d->w; k.Cin = d->cin; k.ldi = d->ldi; k.cioff = d->cioff;
    k.OH = d->oh; k.OW = d->ow; k.Cout = d->cout; k.ldo = d->ldo; k.cooff = d->cooff;
    k.kh = d->kh; k.kw = d->kw;
    k.ay = d->sh; k.by = -d->ph; k.cy = d->dh; k.divy = 1;
    k.ax = d->sw; k.bx = -d->pw; k.cx = d->dw; k.divx = 1;
    k.M = d->nb * d->oh * d->ow; k.flags = flags; k.ldm = 0; k.moff = 0;
    k.in_bytes = (long long)d->nb * d->h * d->w * d->ldi * (d->dtype == DIN_F32 ? 4 : 2);
    k.w_bytes = din_conv_packed_elems(d, 0) * (d->dtype == DIN_F32 ? 4 : 2);
    GatherPlan g = plan_gather(k.M, d->cin, d->cout, d->kh * d->kw, d->dtype);
    if (g.splitk > 1) DIN_FAIL(DIN_E_ARG, "conv_fwd2: this shape runs split-K (%d pixels): launch the sibling convs separately", k.M);
    return run_gather(k, g, d->dtype, workspace, workspace_bytes, as_stream(stream), "conv_fwd2");
}

static int conv_dgrad_impl(const din_conv_desc* d, const void* dout, const void* wpk_t, void* din_, const void* mask, int ldm,
                           int moff, int flags, void* workspace, int64_t workspace_bytes, void* stream, const din_conv_src* x);

int din_conv_dgrad(const din_conv_desc* d, const void* dout, const void* wpk_t, void* din_, const void* mask, int ldm,
                   int moff, int flags, void* workspace, int64_t workspace_bytes, void* stream) {
    return conv_dgrad_impl(d, dout, wpk_t, din_, mask, ldm, moff, flags, workspace, workspace_bytes, stream, nullptr);
}

// whether the parity-class launches of this strided dgrad can carry an extra 1x1 source (conv_gather_fast_kernel<..., XSRC>: bf16, 128 x 96
// tiles, no split-K)
static bool dgrad_x_fused(const din_conv_desc* d) {
    if (DIN_OPT("DIN_DGRAD_X") && atoi(DIN_OPT("DIN_DGRAD_X")) == 0) return false;
    if (d->dtype != DIN_BF16 || !(d->sh > 1 || d->sw > 1) || d->dh != 1 || d->dw != 1 || d->kh * d->kw > 32) return false;
    for (int py = 0; py < d->sh; ++py)
        for (int px = 0; px < d->sw; ++px) {
            const int Ha = (d->h - py + d->sh - 1) / d->sh, Wa = (d->w - px + d->sw - 1) / d->sw;
            if (Ha <= 0 || Wa <= 0) continue;
            const int r0c = (py + d->ph) % d->sh, s0c = (px + d->pw) % d->sw;
            const int khs = r0c < d->kh ? (d->kh - r0c + d->sh - 1) / d->sh : 0;
            const int kws = s0c < d->kw ? (d->kw - s0c + d->sw - 1) / d->sw : 0;
            const GatherPlan g = plan_gather(d->nb * Ha * Wa, d->cout, d->cin, khs * kws, d->dtype, true);
            if (g.bn != 96 || g.splitk != 1) return false;
        }
    return true;
}

int din_conv_dgrad_x_fused(const din_conv_desc* d) { return (d && check_desc(d) == DIN_OK && dgrad_x_fused(d)) ? 1 : 0; }

int din_conv_dgrad_x(const din_conv_desc* d, const void* dout, const void* wpk_t, void* din_, const void* mask, int ldm, int moff, int flags,
                     const din_conv_src* x, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    DIN_REQUIRE(x && x->dout && x->wpk_t && x->cout > 0, "conv_dgrad_x: null extra source");
    const int epc = epc_of(d->dtype);
    DIN_REQUIRE(x->ldo % epc == 0 && x->cooff % epc == 0 && x->ldo >= x->cooff + pad_to(x->cout, epc),
                "conv_dgrad_x: extra source stride/offset must be multiples of %d and cover cout (+zero pad)", epc);
    if (dgrad_x_fused(d)) return conv_dgrad_impl(d, dout, wpk_t, din_, mask, ldm, moff, flags, workspace, workspace_bytes, stream, x);
    // not a shape the fused kernel serves: the two launches it replaces (the second accumulates; the mask is linear)
    if (int e = conv_dgrad_impl(d, dout, wpk_t, din_, mask, ldm, moff, flags, workspace, workspace_bytes, stream, nullptr)) return e;
    return din_conv1x1_dgrad_multi(1, x, d->dtype, d->nb, d->h, d->w, d->cin, d->ldi, d->cioff, din_, mask, ldm, moff, flags | DIN_CONV_ACCUM, stream);
}

static int conv_dgrad_impl(const din_conv_desc* d, const void* dout, const void* wpk_t, void* din_, const void* mask, int ldm,
                           int moff, int flags, void* workspace, int64_t workspace_bytes, void* stream, const din_conv_src* x) {
    if (int e = check_desc(d)) return e;
    DIN_REQUIRE(dout && wpk_t && din_, "conv_dgrad: null pointer");
    DIN_REQUIRE(!d->in_u8, "conv_dgrad: in_u8 is a din_conv_fwd / din_conv_wgrad option");
    DIN_REQUIRE(!(flags & (DIN_CONV_BIAS | DIN_CONV_RELU)), "conv_dgrad: BIAS/RELU are fwd-only flags");
    DIN_REQUIRE(!(flags & DIN_CONV_MASK) || mask, "conv_dgrad: MASK flag without mask");
    int epc = epc_of(d->dtype);
    DIN_REQUIRE(d->ldo % epc == 0 && d->cooff % epc == 0 && d->ldo >= d->cooff + pad_to(d->cout, epc),
                "conv_dgrad: dout stride/offset must be multiples of %d and cover cout (+zero pad)", epc);
    DIN_REQUIRE(d->ldi % 4 == 0 && d->cioff % 4 == 0, "conv_dgrad: din stride/offset must be multiples of 4");
    // the "input" of the gather is dout (geometry oh x ow x cout), the "output" is din (h x w x cin)
    ConvK k{};
    k.in = dout; k.w = wpk_t; k.out = din_; k.bias = nullptr; k.mask = mask; k.partial = nullptr;
    k.NB = d->nb; k.H = d->oh; k.W = d->ow; k.Cin = d->cout; k.ldi = d->ldo; k.cioff = d->cooff;
    k.OH = d->h; k.OW = d->w; k.Cout = d->cin; k.ldo = d->ldi; k.cooff = d->cioff;
    k.kh = d->kh; k.kw = d->kw;
    k.flags = flags; k.ldm = ldm; k.moff = moff;
    k.in_bytes = (long long)d->nb * d->oh * d->ow * d->ldo * (d->dtype == DIN_F32 ? 4 : 2);
    k.w_bytes = din_conv_packed_elems(d, 1) * (d->dtype == DIN_F32 ? 4 : 2);
    const bool strided = d->sh > 1 || d->sw > 1;
    if (!strided || d->dh != 1 || d->dw != 1 || d->kh * d->kw > 32) {
        // y_in = oy*sh - ph + r*dh  =>  oy = (y_in + ph - r*dh) / sh
        k.ay = 1; k.by = d->ph; k.cy = -d->dh; k.divy = d->sh;
        k.ax = 1; k.bx = d->pw; k.cx = -d->dw; k.divx = d->sw;
        k.M = d->nb * d->h * d->w;
        GatherPlan g = plan_gather(k.M, d->cout, d->cin, d->kh * d->kw, d->dtype);
        return run_gather(k, g, d->dtype, workspace, workspace_bytes, as_stream(stream), "conv_dgrad");
    }
    // Strided dgrad: decompose by output parity (py,px).  Class (py,px) only sees the taps r = r0 + sh*r', s = s0 + sw*s'
    // with r0 = (py+ph) % sh: a stride-1 gather oy = a + (py+ph-r0)/sh - r' over the sub-grid y = sh*a + py -- no structurally
    // zero taps are multiplied, and it runs on the fast (buffer-addressed) kernel.
    const int epc2 = epc_of(d->dtype);
    const int cpt_full = pad_to(d->cout, epc2) / epc2;
    const int wld_full = (d->kh * d->kw * cpt_full + KC - 1) / KC * KC;
    for (int py = 0; py < d->sh; ++py)
        for (int px = 0; px < d->sw; ++px) {
            const int Ha = (d->h - py + d->sh - 1) / d->sh, Wa = (d->w - px + d->sw - 1) / d->sw;
            if (Ha <= 0 || Wa <= 0) continue;
            const int r0c = (py + d->ph) % d->sh, s0c = (px + d->pw) % d->sw;
            const int khs = r0c < d->kh ? (d->kh - r0c + d->sh - 1) / d->sh : 0;
            const int kws = s0c < d->kw ? (d->kw - s0c + d->sw - 1) / d->sw : 0;
            ConvK c = k;
            c.kh = khs > 0 && kws > 0 ? khs : 0; c.kw = khs > 0 && kws > 0 ? kws : 1;
            c.ay = 1; c.by = (py + d->ph - r0c) / d->sh; c.cy = -1; c.divy = 1;
            c.ax = 1; c.bx = (px + d->pw - s0c) / d->sw; c.cx = -1; c.divx = 1;
            c.OH = Ha; c.OW = Wa; c.M = d->nb * Ha * Wa;
            c.out_sy = d->sh; c.out_sx = d->sw; c.out_y0 = py; c.out_x0 = px; c.out_H = d->h; c.out_W = d->w;
            c.remap = 1; c.wld = wld_full;
            for (int rr = 0; rr < khs; ++rr)
                for (int ss = 0; ss < kws; ++ss) c.wtap[rr * kws + ss] = (unsigned char)((r0c + d->sh * rr) * d->kw + (s0c + d->sw * ss));
            if (x) {                                   // extra 1x1 source at the output pixel (din_conv_dgrad_x)
                ConvK::Src& o = c.src[0];
                o.in = x->dout; o.w = x->wpk_t; o.ld = x->ldo; o.coff = x->cooff;
                o.cpt = pad_to(x->cout, epc2) / epc2;
                o.wld = (o.cpt + KC - 1) / KC * KC;
                o.in_bytes = (long long)d->nb * d->h * d->w * x->ldo * 2;
                o.w_bytes = (long long)pad_to(d->cin, 256) * o.wld * 16;
                c.xsteps = (o.cpt + KC - 1) / KC;
            }
            GatherPlan g = plan_gather(c.M, d->cout, d->cin, c.kh * c.kw, d->dtype, true);
            if (int e = run_gather(c, g, d->dtype, workspace, workspace_bytes, as_stream(stream), "conv_dgrad(strided)")) return e;
        }
    return DIN_OK;
}

int din_conv1x1_dgrad_multi(int nsrc, const din_conv_src* srcs, int dtype, int nb, int h, int w, int cin, int ldi, int cioff,
                            void* din_, const void* mask, int ldm, int moff, int flags, void* stream) {
    DIN_REQUIRE(nsrc >= 1 && nsrc <= 4 && srcs && din_, "conv1x1_dgrad_multi: 1..4 sources");
    DIN_REQUIRE(dtype == DIN_F32 || dtype == DIN_BF16, "conv1x1_dgrad_multi: bad dtype");
    DIN_REQUIRE(!(flags & (DIN_CONV_BIAS | DIN_CONV_RELU)), "conv1x1_dgrad_multi: BIAS/RELU are fwd-only flags");
    DIN_REQUIRE(!(flags & DIN_CONV_MASK) || mask, "conv1x1_dgrad_multi: MASK flag without mask");
    const int epc = epc_of(dtype), esz = dtype == DIN_F32 ? 4 : 2;
    DIN_REQUIRE(nb > 0 && h > 0 && w > 0 && cin > 0 && ldi % 4 == 0 && cioff % 4 == 0 && ldi >= cioff + cin, "conv1x1_dgrad_multi: bad output geometry");
    DIN_REQUIRE((int64_t)nb * h * w < (1ll << 31), "conv1x1_dgrad_multi: too many pixels");
    ConvK k{};
    k.out = din_; k.mask = mask; k.bias = nullptr; k.partial = nullptr;
    k.NB = nb; k.H = h; k.W = w; k.OH = h; k.OW = w; k.Cout = cin; k.ldo = ldi; k.cooff = cioff;
    k.kh = k.kw = 1; k.ay = k.ax = 1; k.by = k.bx = 0; k.cy = k.cx = 1; k.divy = k.divx = 1;
    k.M = nb * h * w; k.flags = flags; k.ldm = ldm; k.moff = moff;
    int steps = 0;
    for (int b = 0; b < nsrc; ++b) {
        const din_conv_src& sc = srcs[b];
        DIN_REQUIRE(sc.dout && sc.wpk_t && sc.cout > 0, "conv1x1_dgrad_multi: null source %d", b);
        DIN_REQUIRE(sc.ldo % epc == 0 && sc.cooff % epc == 0 && sc.ldo >= sc.cooff + pad_to(sc.cout, epc),
                    "conv1x1_dgrad_multi: source %d stride/offset must be multiples of %d and cover cout (+zero pad)", b, epc);
        ConvK::Src& o = k.src[b];
        o.in = sc.dout; o.w = sc.wpk_t; o.ld = sc.ldo; o.coff = sc.cooff;
        o.cpt = pad_to(sc.cout, epc) / epc;
        o.wld = (o.cpt + KC - 1) / KC * KC;
        o.in_bytes = (long long)nb * h * w * sc.ldo * esz;
        o.w_bytes = (long long)pad_to(cin, 256) * o.wld * 16;
        steps += (o.cpt + KC - 1) / KC;
    }
    k.nsrc = nsrc;
    k.in = srcs[0].dout; k.w = srcs[0].wpk_t; k.in_bytes = k.src[0].in_bytes; k.w_bytes = k.src[0].w_bytes;
    k.Cin = srcs[0].cout; k.ldi = srcs[0].ldo; k.cioff = srcs[0].cooff;
    GatherPlan g = plan_gather(k.M, 64, cin, 1, dtype);
    if (g.bn == 256) g.bn = 128;
    g.bm = 128; g.n_px_tiles = (k.M + 127) / 128; g.n_co_tiles = (cin + g.bn - 1) / g.bn;
    k.cpt = KC; k.Q = steps * KC; k.nk = steps; k.wld = k.src[0].wld;
    k.splitk = 1; k.ks_per_split = steps; k.n_co_tiles = g.n_co_tiles; k.remap = 0; k.korder = 0;
    if (DIN_OPT("DIN_DEBUG_PLAN"))
        fprintf(stderr, "[din] conv1x1_dgrad_multi M=%d nsrc=%d couts=%d,%d,%d,%d ld=%d,%d,%d,%d coff=%d,%d,%d,%d cin=%d flags=%d regw=%d\n", k.M, nsrc, srcs[0].cout,
                nsrc > 1 ? srcs[1].cout : 0, nsrc > 2 ? srcs[2].cout : 0, nsrc > 3 ? srcs[3].cout : 0, srcs[0].ldo, nsrc > 1 ? srcs[1].ldo : 0,
                nsrc > 2 ? srcs[2].ldo : 0, nsrc > 3 ? srcs[3].ldo : 0, srcs[0].cooff, nsrc > 1 ? srcs[1].cooff : 0, nsrc > 2 ? srcs[2].cooff : 0,
                nsrc > 3 ? srcs[3].cooff : 0, cin, flags, (int)din_gather::conv1x1_regw_eligible(k, dtype));
    if (din_gather::conv1x1_regw_eligible(k, dtype)) {
        if (din_gather::launch_conv1x1_regw(k, as_stream(stream))) DIN_FAIL(DIN_E_LAUNCH, "conv1x1_dgrad_multi: conv1x1_regw launch failed");
        return DIN_OK;
    }
    if (din_gather::conv1x1_stream_eligible(k, dtype)) {
        if (din_gather::launch_conv1x1_stream(k, as_stream(stream))) DIN_FAIL(DIN_E_LAUNCH, "conv1x1_dgrad_multi: conv1x1_stream launch failed");
        return DIN_OK;
    }
    if (dtype == DIN_F32) launch_gather<float>(k, g.n_px_tiles, 128, g.bn, as_stream(stream));
    else launch_gather<bf16_t>(k, g.n_px_tiles, 128, g.bn, as_stream(stream));
    DIN_CHECK_LAUNCH("conv1x1_dgrad_multi");
    return DIN_OK;
}

int din_conv_wgrad(const din_conv_desc* d, const void* in, const void* dout, float* dw, float* dbias, const float* scale,
                   const float* w, float* wdot, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    DIN_REQUIRE(in && dout && dw, "conv_wgrad: null pointer");
    DIN_REQUIRE(!wdot || w, "conv_wgrad: wdot needs w");
    DIN_REQUIRE(!d->in_u8 || din_conv_accepts_u8(d), "conv_wgrad: in_u8 on a layer din_conv_accepts_u8() rejects");
    hipStream_t st = as_stream(stream);
    const bool prezeroed = (accumulate & 2) != 0;            // dbias / wdot were zeroed by the caller (one memset for a whole backbone)
    accumulate &= 1;
    WgradPlan wp = plan_wgrad(d);
    if (workspace_bytes < wp.ws_bytes || !workspace)
        DIN_FAIL(DIN_E_WORKSPACE, "conv_wgrad: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)wp.ws_bytes);
    WgradK k{};
    k.in = in; k.g = dout; k.partial = reinterpret_cast<float*>(workspace); k.dbias = nullptr;
    k.NB = d->nb; k.H = d->h; k.W = d->w; k.Cin = d->cin; k.ldi = d->ldi; k.cioff = d->cioff;
    k.OH = d->oh; k.OW = d->ow; k.Cout = d->cout; k.ldo = d->ldo; k.cooff = d->cooff;
    k.kh = d->kh; k.kw = d->kw; k.sh = d->sh; k.sw = d->sw; k.ph = d->ph; k.pw = d->pw; k.dh = d->dh; k.dw = d->dw;
    k.cin_pad = wp.cin_pad; k.kcols = wp.kcols; k.kcols_pad = wp.kcols_pad; k.cout_pad = wp.cout_pad;
    k.M = d->nb * d->oh * d->ow; k.n_co_tiles = wp.n_co_tiles; k.n_k_tiles = wp.n_k_tiles;
    k.slices = wp.slices; k.m_per_slice = wp.m_per_slice;
#ifdef DIN_EXPERIMENTS
    { const char* pv = DIN_OPT("DIN_WGRAD_PROBE"); k.probe = pv ? atoi(pv) : 0; }      // timing probe: results are WRONG when set
#else
    k.probe = 0;
#endif
    dim3 grid(wp.n_co_tiles * wp.n_k_tiles, wp.slices);
    bool bias_fused = false;
    if (d->dtype == DIN_F32) {
        hipLaunchKernelGGL(conv_wgrad_f32_kernel, grid, dim3(NTHREADS), 0, st, k);
    } else {
        int epc = 8;
        DIN_REQUIRE(d->ldo % epc == 0 && d->cooff % epc == 0, "conv_wgrad: bf16 dout stride/offset must be multiples of 8");
        if (wp.small == 4) {
            if (dbias) {
                if (!prezeroed && hipMemsetAsync(dbias, 0, sizeof(float) * d->cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad: memset");
                k.dbias = dbias;
                bias_fused = true;
            }
            if (int e = din_wgrad::launch_wgrad_halo(k, WGRAD_HALO_GRID, st)) return e;
        } else if (wp.small) {
            if (dbias) {
                if (!prezeroed && hipMemsetAsync(dbias, 0, sizeof(float) * d->cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad: memset");
                k.dbias = dbias;
                bias_fused = true;
            }
            const int st_ = wp.small == 3 ? 2 : 1, cpp = wp.small == 3 ? 1 : 4;
            const int hbytes = ((7 * st_ + 3) * (31 * st_ + 3) * cpp * 16 + 1023) / 1024 * 1024;
            const size_t lds = 2 * ((size_t)hbytes + 256 * (size_t)wp.bco * 2) + (d->in_u8 ? 512 : 0);
            auto launch = [&](auto kern) {
                if (lds > 65536) raise_lds_limit(kern, lds);
                hipLaunchKernelGGL(kern, dim3(WGRAD_SMALL_GRID), dim3(NTHREADS), lds, st, k);
            };
            if (wp.small == 1) launch(conv_wgrad_small_kernel<4, 32, 1>);
            else if (wp.small == 2 && !(DIN_OPT("DIN_WGRAD_SMALL_WAVES") && atoi(DIN_OPT("DIN_WGRAD_SMALL_WAVES")) == 4)) {
                const char* rg = DIN_OPT("DIN_WGRAD_SMALL_RING");
                if ((rg ? atoi(rg) : 3) == 3) {                       // 6 x 32 tiles, three-slot ring, two stages in flight (126 KB)
                    const size_t hb6 = ((size_t)(5 + 3) * (31 + 3) * 4 * 16 + 1023) / 1024 * 1024, lds3 = 3 * (hb6 + 192 * (size_t)wp.bco * 2);
                    raise_lds_limit(conv_wgrad_small_kernel<4, 64, 1, false, 8, 6, 3>, lds3);
                    hipLaunchKernelGGL((conv_wgrad_small_kernel<4, 64, 1, false, 8, 6, 3>), dim3(WGRAD_SMALL_GRID), dim3(512), lds3, st, k);
                } else {
                    if (lds > 65536) raise_lds_limit(conv_wgrad_small_kernel<4, 64, 1, false, 8>, lds);
                    hipLaunchKernelGGL((conv_wgrad_small_kernel<4, 64, 1, false, 8>), dim3(WGRAD_SMALL_GRID), dim3(512), lds, st, k);
                }
            }
            else if (wp.small == 2) launch(conv_wgrad_small_kernel<4, 64, 1>);
            else if (d->in_u8) { k.u8 = reinterpret_cast<const unsigned char*>(in); launch(conv_wgrad_small_kernel<1, 32, 2, true>); }
            else launch(conv_wgrad_small_kernel<1, 32, 2>);
        } else if (wp.pipe) {
            if (dbias) {
                if (!prezeroed && hipMemsetAsync(dbias, 0, sizeof(float) * d->cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad: memset");
                k.dbias = dbias;
                bias_fused = true;
            }
            k.atomic = wp.atomic;
            if (wp.atomic && hipMemsetAsync(k.partial, 0, sizeof(float) * (size_t)wp.cout_pad * wp.kcols_pad, st) != hipSuccess)
                DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad: memset");
            {   // sibling pacing words behind the partial tiles (workspace sized for them in plan_wgrad)
                const char* pe = DIN_OPT("DIN_WGRAD_PACE");
                const int want = pe ? atoi(pe) : 1;
                const size_t words = (size_t)wp.slices * wp.n_co_tiles * 8;                 // rows of 8 words (one s_load_dwordx8)
                // measured (tools/pace_experiment.sh, profiles/r02_wgrad_pacing.txt): Conv2d_4a (3 k tiles) 6.67 -> 3.12 GB fetched per launch at
                // unchanged time; with 6+ siblings the naps cost 4-10 % and the L2 hit rate was 74 % anyway -> default: up to 3 siblings
                if (want && wp.n_k_tiles >= 2 && wp.n_k_tiles <= (want >= 2 ? 8 : 3) && !wp.atomic && wp.m_per_slice / 32 < (1 << 20) - 1) {
                    k.pace = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + (((size_t)wp.slices * wp.cout_pad * wp.kcols_pad * 4 + 31) & ~(size_t)31));
                    // no memset: every launch tags its words (1..1023 << 20, never 0); words of older launches or stale workspace contents
                    // are out of range for this tag (an alias once in 1023 launches costs one bounded spin, never correctness)
                    static std::atomic<unsigned> pace_epoch{0};
                    k.pace_base = (int)(((pace_epoch.fetch_add(1) % 1023u) + 1u) << 20);
                    (void)words;
                }
            }
            // one slice of a 1x1 layer, nothing for the reduce launch to do (no scale, no <w, dW>, no accumulate, no channel padding): straight into dW
            if (wp.slices == 1 && !wp.atomic && d->kh * d->kw == 1 && !scale && !wdot && !accumulate && wp.cin_pad == d->cin &&
                !(DIN_OPT("DIN_WGRAD_DIRECT") && atoi(DIN_OPT("DIN_WGRAD_DIRECT")) == 0)) k.direct = dw;
            if (int e = din_wgrad::launch_wgrad_pipe(k, wp.bco, wp.bk, grid, st)) return e;
        } else if (wp.ring) {
            if (dbias) {
                if (!prezeroed && hipMemsetAsync(dbias, 0, sizeof(float) * d->cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad: memset");
                k.dbias = dbias;
                bias_fused = true;
            }
            const size_t lds = 4 * ((size_t)((32 * wp.bco / 8 + 511) / 512) * 8192 + 32 * (size_t)wp.bk * 2);   // four 32-pixel stages (G tile in 8-KiB rounds)
            auto launch = [&](auto kern) {
                raise_lds_limit(kern, lds);
                hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, k);
            };
            if (wp.bk == 128) {
                if (wp.bco == 64) launch(conv_wgrad_ring_kernel<64, 128>);
                else if (wp.bco == 96) launch(conv_wgrad_ring_kernel<96, 128>);
                else if (wp.bco == 160) launch(conv_wgrad_ring_kernel<160, 128>);
                else launch(conv_wgrad_ring_kernel<128, 128>);
            }
            else if (wp.bco == 128) launch(conv_wgrad_ring_kernel<128, 256>);
            else if (wp.bco == 160) launch(conv_wgrad_ring_kernel<160, 256>);
            else launch(conv_wgrad_ring_kernel<192, 256>);
        } else if (wp.v2) {
            if (dbias) {
                if (!prezeroed && hipMemsetAsync(dbias, 0, sizeof(float) * d->cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad: memset");
                k.dbias = dbias;
                bias_fused = true;
            }
            size_t lds = 2 * 64 * ((size_t)(wp.bco * 2) + (WG_TILE * 2));        // two unpadded stages
            auto launch = [&](auto kern) {
                if (lds > 65536) raise_lds_limit(kern, lds);
                hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, st, k);
            };
            if (wp.bco == 64) launch(conv_wgrad_bf16_kernel<64>);
            else if (wp.bco == 96) launch(conv_wgrad_bf16_kernel<96>);
            else if (wp.bco == 160) launch(conv_wgrad_bf16_kernel<160>);
            else launch(conv_wgrad_bf16_kernel<128>);
        } else {
            size_t lds = 2 * 2 * 32 * (WG_TILE * 2 + 32);
            hipLaunchKernelGGL(conv_wgrad_bf16_tail_kernel, grid, dim3(NTHREADS), lds, st, k);
        }
    }
    DIN_CHECK_LAUNCH("conv_wgrad");
    if (wdot && !prezeroed && hipMemsetAsync(wdot, 0, sizeof(float) * d->cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad: memset");
    if (k.direct == nullptr) {
        int kc_total = d->kh * d->kw * wp.cin_pad;
        dim3 rgrid(d->cout, (kc_total + 255) / 256);
        const int rslices = (wp.pipe && wp.atomic) ? 1 : wp.slices;
        const int nsg = rslices >= 64 ? 16 : rslices >= 8 ? 4 : 1;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, rgrid, dim3(64, nsg), 0, st, k.partial, dw, scale, w, wdot,
                           d->cout, d->cin, d->kh, d->kw, wp.cin_pad, wp.cout_pad, wp.kcols_pad, (wp.pipe && wp.atomic) ? 1 : wp.slices, accumulate);
        DIN_CHECK_LAUNCH("conv_wgrad_reduce");
    }
    if (dbias && !bias_fused) {
        if (int e = launch_colsum(d->dtype, dout, dbias, k.M, d->cout, d->ldo, d->cooff, st)) return e;
    }
    return DIN_OK;
}

// ---- din_conv_wgrad_group: the weight gradients of several LAYERS in one launch of the pipelined kernel (conv_wgrad.h: WgradGroupK) -------
// Plan: every item keeps the tile geometry plan_wgrad gives it alone; what changes is the pixel slicing.  One common slice length mps
// (whole 32-pixel stages) is chosen so that the items' tiles x slices fill the chip's CUs ONCE: sum_g tiles_g * ceil(M_g / mps) <= CUs.
struct WgradGroupPlan { WgradPlan wp[din_wgrad::WGRAD_GROUP_MAX]; int slices[din_wgrad::WGRAD_GROUP_MAX]; int64_t part_off[din_wgrad::WGRAD_GROUP_MAX], pace_off[din_wgrad::WGRAD_GROUP_MAX]; int mps, bco, wide; int64_t ws_bytes; };

static int wgrad_group_key(const din_conv_desc* d, WgradPlan* out) {
    const char* gv = DIN_OPT("DIN_WGRAD_GROUP");
    if (gv && atoi(gv) == 0) return 0;
    if (!d || d->dtype != DIN_BF16 || d->in_u8 || d->nb <= 0 || d->oh <= 0 || d->ow <= 0) return 0;
    const char* wv = DIN_OPT("DIN_WGRAD_PIPE_WAVES");
    if (wv && atoi(wv) != 16) return 0;                           // (the group kernel is instantiated for the shipped sixteen-wave grid)
    if (d->ldo % 8 != 0 || d->cooff % 8 != 0) return 0;
    const WgradPlan wp = plan_wgrad(d);
    if (!wp.pipe || wp.atomic || wp.slices < 2) return 0;         // one slice: nothing to reduce, the single launch writes dW directly
    if (out) *out = wp;
    return wp.bco * 2 + (d->ow >= 32 ? 1 : 0);
}

static int wgrad_group_cus() {
    static std::atomic<int> cached{0};
    int c = cached.load(std::memory_order_relaxed);
    if (!c) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        c = n > 0 ? n : 256;
        cached.store(c, std::memory_order_relaxed);
    }
    return c;
}

static bool plan_wgrad_group(int n, const din_conv_wgrad_item* items, WgradGroupPlan& gp) {
    if (!items || n < 2 || n > din_wgrad::WGRAD_GROUP_MAX) return false;
    int key0 = 0;
    int64_t cost = 0;
    for (int g = 0; g < n; ++g) {
        const int key = wgrad_group_key(&items[g].desc, &gp.wp[g]);
        if (!key || (g && key != key0)) return false;
        key0 = key;
        cost += (int64_t)gp.wp[g].n_co_tiles * gp.wp[g].n_k_tiles * ((int64_t)items[g].desc.nb * items[g].desc.oh * items[g].desc.ow);
    }
    gp.bco = key0 / 2; gp.wide = key0 & 1;
    const int budget = wgrad_group_cus();
    int64_t mps = (cost + budget - 1) / budget;
    mps = (mps + 31) / 32 * 32;
    if (mps < 32) mps = 32;
    for (int iter = 0; iter < 4096; ++iter) {
        int64_t wgs = 0, next = INT64_MAX;
        for (int g = 0; g < n; ++g) {
            const int64_t M = (int64_t)items[g].desc.nb * items[g].desc.oh * items[g].desc.ow, sl = (M + mps - 1) / mps;
            wgs += sl * gp.wp[g].n_co_tiles * gp.wp[g].n_k_tiles;
            if (sl > 1) {                                           // the smallest slice length that takes one slice off this item
                int64_t m2 = ((M + sl - 2) / (sl - 1) + 31) / 32 * 32;
                if (m2 <= mps) m2 = mps + 32;
                if (m2 < next) next = m2;
            }
        }
        if (wgs <= budget || next == INT64_MAX) break;
        mps = next;
    }
    if (mps >= (1ll << 30)) return false;
    gp.mps = (int)mps;
    int64_t off = 0;
    for (int g = 0; g < n; ++g) {
        const int64_t M = (int64_t)items[g].desc.nb * items[g].desc.oh * items[g].desc.ow;
        gp.slices[g] = (int)((M + mps - 1) / mps);
        gp.part_off[g] = off;
        off += ((int64_t)gp.slices[g] * gp.wp[g].cout_pad * gp.wp[g].kcols_pad * 4 + 255) / 256 * 256;
    }
    for (int g = 0; g < n; ++g) {                                   // sibling-pacing words (WgradK::pace) behind the partial tiles
        gp.pace_off[g] = off;
        off += ((int64_t)gp.slices[g] * gp.wp[g].n_co_tiles * 8 * 4 + 255) / 256 * 256;
    }
    gp.ws_bytes = off;
    return true;
}

int din_conv_wgrad_group_key(const din_conv_desc* d) { return wgrad_group_key(d, nullptr); }

int64_t din_conv_wgrad_group_workspace(int n, const din_conv_wgrad_item* items) {
    WgradGroupPlan gp;
    return plan_wgrad_group(n, items, gp) ? gp.ws_bytes : 0;
}

int din_conv_wgrad_group(int n, const din_conv_wgrad_item* items, void* workspace, int64_t workspace_bytes, void* stream) {
    DIN_REQUIRE(items && n >= 1, "conv_wgrad_group: no items");
    WgradGroupPlan gp;
    if (!plan_wgrad_group(n, items, gp)) {                          // not a group this launch serves: layer by layer
        for (int g = 0; g < n; ++g) {
            const din_conv_wgrad_item& it = items[g];
            const int64_t need = din_conv_workspace_bytes(&it.desc, 2);
            if (need > workspace_bytes) DIN_FAIL(DIN_E_WORKSPACE, "conv_wgrad_group: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
            if (int e = din_conv_wgrad(&it.desc, it.in, it.dout, it.dw, it.dbias, it.scale, it.w, it.wdot, it.accumulate, workspace, workspace_bytes, stream)) return e;
        }
        return DIN_OK;
    }
    if (!workspace || workspace_bytes < gp.ws_bytes)
        DIN_FAIL(DIN_E_WORKSPACE, "conv_wgrad_group: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)gp.ws_bytes);
    hipStream_t st = as_stream(stream);
    din_wgrad::WgradGroupK G{};
    G.n = n;
    int first = 0;
    for (int g = 0; g < n; ++g) {
        const din_conv_wgrad_item& it = items[g];
        const din_conv_desc* d = &it.desc;
        if (int e = check_desc(d)) return e;
        DIN_REQUIRE(it.in && it.dout && it.dw, "conv_wgrad_group: null pointer in item %d", g);
        DIN_REQUIRE(!it.wdot || it.w, "conv_wgrad_group: wdot needs w");
        const WgradPlan& wp = gp.wp[g];
        const bool prezeroed = (it.accumulate & 2) != 0;
        din_wgrad::WgradK& k = G.k[g];
        k.in = it.in; k.g = it.dout; k.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + gp.part_off[g]); k.dbias = nullptr;
        k.NB = d->nb; k.H = d->h; k.W = d->w; k.Cin = d->cin; k.ldi = d->ldi; k.cioff = d->cioff;
        k.OH = d->oh; k.OW = d->ow; k.Cout = d->cout; k.ldo = d->ldo; k.cooff = d->cooff;
        k.kh = d->kh; k.kw = d->kw; k.sh = d->sh; k.sw = d->sw; k.ph = d->ph; k.pw = d->pw; k.dh = d->dh; k.dw = d->dw;
        k.cin_pad = wp.cin_pad; k.kcols = wp.kcols; k.kcols_pad = wp.kcols_pad; k.cout_pad = wp.cout_pad;
        k.M = d->nb * d->oh * d->ow; k.n_co_tiles = wp.n_co_tiles; k.n_k_tiles = wp.n_k_tiles;
        k.slices = gp.slices[g]; k.m_per_slice = gp.mps;
        if (it.dbias) {
            if (!prezeroed && hipMemsetAsync(it.dbias, 0, sizeof(float) * d->cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad_group: memset");
            k.dbias = it.dbias;
        }
        if (it.wdot && !prezeroed && hipMemsetAsync(it.wdot, 0, sizeof(float) * d->cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv_wgrad_group: memset");
        {   // the k-tile siblings of one (filter tile, pixel slice) stream the same dY rows.  Pacing them as the single-layer launch does was
            // measured inside groups (tools/ab_group_pace.sh, profiles/r06_group_pace.txt): HBM fetch 1617 -> 1549 MB per launch, but the naps cost
            // time -- 32 clips 656.3 -> 654.7 clips/s, 4 clips 9.04 -> 9.15 ms: OFF unless DIN_WGRAD_GROUP_PACE=1 (2: up to 8 siblings)
            const char* pe = DIN_OPT("DIN_WGRAD_GROUP_PACE");
            const int want = pe ? atoi(pe) : 0;
            if (want && wp.n_k_tiles >= 2 && wp.n_k_tiles <= (want >= 2 ? 8 : 3) && gp.mps / 32 < (1 << 20) - 1) {
                static std::atomic<unsigned> group_pace_epoch{0};
                k.pace = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + gp.pace_off[g]);
                k.pace_base = (int)(((group_pace_epoch.fetch_add(1) % 1023u) + 1u) << 20);
            }
        }
        G.first[g] = first;
        first += wp.n_co_tiles * wp.n_k_tiles * gp.slices[g];
    }
    for (int g = n; g <= din_wgrad::WGRAD_GROUP_MAX; ++g) G.first[g] = first;
    if (int e = din_wgrad::launch_wgrad_pipe_group(G, gp.bco, gp.wide != 0, st)) return e;
    DIN_CHECK_LAUNCH("conv_wgrad_group");
    WgradReduceGroupK R{};
    R.n = n;
    int rfirst = 0, max_slices = 1;
    for (int g = 0; g < n; ++g) {
        const din_conv_wgrad_item& it = items[g];
        const din_conv_desc* d = &it.desc;
        const WgradPlan& wp = gp.wp[g];
        WgradReduceItem& r = R.it[g];
        r.partial = G.k[g].partial; r.dw = it.dw; r.scale = it.scale; r.w = it.w; r.wdot = it.wdot;
        r.cout = d->cout; r.cin = d->cin; r.kh = d->kh; r.kw = d->kw; r.cin_pad = wp.cin_pad; r.cout_pad = wp.cout_pad; r.kcols_pad = wp.kcols_pad;
        r.slices = gp.slices[g]; r.accumulate = it.accumulate & 1; r.kchunks = (d->kh * d->kw * wp.cin_pad + 255) / 256;
        R.first[g] = rfirst;
        rfirst += d->cout * r.kchunks;
        if (gp.slices[g] > max_slices) max_slices = gp.slices[g];
    }
    for (int g = n; g <= din_wgrad::WGRAD_GROUP_MAX; ++g) R.first[g] = rfirst;
    const int nsg = max_slices >= 64 ? 16 : max_slices >= 8 ? 4 : 1;
    hipLaunchKernelGGL(conv_wgrad_reduce_group_kernel, dim3(rfirst), dim3(64, nsg), 0, st, R);
    DIN_CHECK_LAUNCH("conv_wgrad_group reduce");
    return DIN_OK;
}

static bool wgrad_multi_plan(int nsrc, const din_conv_wsrc* srcs, int dtype, int64_t pixels, int cin, din_wgrad::Wg1x1K* k) {
    const char* ev = DIN_OPT("DIN_WGRAD_1X1_MULTI");
    const int mode = ev ? atoi(ev) : 1;                        // 0: off, 1: launches of >= 128K pixels, 2: any size (tests)
    if (!mode || dtype != DIN_BF16 || !srcs || nsrc < 2 || nsrc > 4 || (pixels < 128 * 1024 && mode != 2) || pixels <= 0) return false;
    int couts[4];
    for (int s = 0; s < nsrc; ++s) {
        couts[s] = srcs[s].cout;
        if (srcs[s].ldo % 8 != 0 || srcs[s].cooff % 8 != 0 || srcs[s].ldo < srcs[s].cooff + srcs[s].cout || pixels >= 0x7fffffffll / 64) return false;
    }
    return din_wgrad::plan_wgrad_1x1_multi(nsrc, couts, cin, k);
}

int64_t din_conv1x1_wgrad_multi_workspace(int nsrc, const din_conv_wsrc* srcs, int dtype, int64_t pixels, int cin) {
    din_wgrad::Wg1x1K k{};
    if (!wgrad_multi_plan(nsrc, srcs, dtype, pixels, cin, &k)) return 0;
    return (int64_t)WGRAD_HALO_GRID * k.rows_pad * cin * 4;
}

int din_conv1x1_wgrad_multi(int nsrc, const din_conv_wsrc* srcs, int dtype, int64_t pixels, int cin, int ldi, int cioff, const void* in,
                            int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    din_wgrad::Wg1x1K k{};
    DIN_REQUIRE(in && workspace, "conv1x1_wgrad_multi: null pointer");
    DIN_REQUIRE(wgrad_multi_plan(nsrc, srcs, dtype, pixels, cin, &k), "conv1x1_wgrad_multi: this group does not fit the kernel "
                "(din_conv1x1_wgrad_multi_workspace returns 0 for it: run din_conv_wgrad per layer)");
    DIN_REQUIRE(ldi % 8 == 0 && cioff % 8 == 0 && ldi >= cioff + cin, "conv1x1_wgrad_multi: bad input view");
    const int64_t need = (int64_t)WGRAD_HALO_GRID * k.rows_pad * cin * 4;
    if (workspace_bytes < need) DIN_FAIL(DIN_E_WORKSPACE, "conv1x1_wgrad_multi: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    hipStream_t st = as_stream(stream);
    const bool prezeroed = (accumulate & 2) != 0;
    accumulate &= 1;
    k.x = in; k.partial = reinterpret_cast<float*>(workspace);
    k.M = (int)pixels; k.Cin = cin; k.ldi = ldi; k.cioff = cioff; k.nsrc = nsrc;
    for (int s = 0; s < nsrc; ++s) {
        DIN_REQUIRE(srcs[s].dout && srcs[s].dw && (!srcs[s].wdot || srcs[s].w), "conv1x1_wgrad_multi: null pointer in source %d", s);
        k.src[s].g = srcs[s].dout; k.src[s].dbias = srcs[s].dbias; k.src[s].cout = srcs[s].cout; k.src[s].ld = srcs[s].ldo; k.src[s].coff = srcs[s].cooff;
        if (!prezeroed) {
            if (srcs[s].dbias && hipMemsetAsync(srcs[s].dbias, 0, sizeof(float) * srcs[s].cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv1x1_wgrad_multi: memset");
            if (srcs[s].wdot && hipMemsetAsync(srcs[s].wdot, 0, sizeof(float) * srcs[s].cout, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "conv1x1_wgrad_multi: memset");
        }
    }
    if (int e = din_wgrad::launch_wgrad_1x1_multi(k, WGRAD_HALO_GRID, st)) return e;
    DIN_CHECK_LAUNCH("conv1x1_wgrad_multi");
    for (int s = 0; s < nsrc; ++s) {
        dim3 rgrid(srcs[s].cout, (cin + 255) / 256);
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, rgrid, dim3(64, 16), 0, st, k.partial + (int64_t)k.src[s].row0 * cin, srcs[s].dw, srcs[s].scale,
                           srcs[s].w, srcs[s].wdot, srcs[s].cout, cin, 1, 1, cin, k.rows_pad, cin, WGRAD_HALO_GRID, accumulate);
        DIN_CHECK_LAUNCH("conv1x1_wgrad_multi reduce");
    }
    return DIN_OK;
}

int din_colsum(const void* g, int dtype, int64_t rows, int c, int ld, int coff, float* out, void* stream) {
    DIN_REQUIRE(g && out && rows > 0 && c > 0 && ld >= coff + c && coff >= 0, "colsum: bad argument");
    DIN_REQUIRE(dtype == DIN_F32 || dtype == DIN_BF16, "colsum: bad dtype");
    return launch_colsum(dtype, g, out, rows, c, ld, coff, as_stream(stream));
}

int din_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                float* shift, int c, void* stream) {
    DIN_REQUIRE(gamma && beta && mean && var && scale && shift && c > 0, "bn_fold: bad argument");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), gamma, beta, mean, var, eps, scale, shift, c);
    DIN_CHECK_LAUNCH("bn_fold");
    return DIN_OK;
}
int din_bn_fold_bwd(const float* wdot, const float* dshift, const float* mean, const float* var, float eps, float* dgamma,
                    float* dbeta, int c, void* stream) {
    DIN_REQUIRE(wdot && dshift && mean && var && dgamma && dbeta && c > 0, "bn_fold_bwd: bad argument");
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), wdot, dshift, mean, var, eps, dgamma, dbeta, c);
    DIN_CHECK_LAUNCH("bn_fold_bwd");
    return DIN_OK;
}

int din_bn_fold_multi(const uint64_t* ptrs, const int32_t* offs, int n, int total, float eps, float* scale, float* shift, void* stream) {
    DIN_REQUIRE(ptrs && offs && scale && shift && n > 0 && total > 0, "bn_fold_multi: bad argument");
    hipLaunchKernelGGL(bn_fold_multi_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), ptrs, offs, n, total, eps, scale, shift);
    DIN_CHECK_LAUNCH("bn_fold_multi");
    return DIN_OK;
}
int din_bn_fold_bwd_multi(const uint64_t* ptrs, const int32_t* offs, int n, int total, float eps, const float* wdot, const float* dshift,
                          float* dgamma, float* dbeta, void* stream) {
    DIN_REQUIRE(ptrs && offs && wdot && dshift && dgamma && dbeta && n > 0 && total > 0, "bn_fold_bwd_multi: bad argument");
    hipLaunchKernelGGL(bn_fold_bwd_multi_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), ptrs, offs, n, total, eps, wdot,
                       dshift, dgamma, dbeta);
    DIN_CHECK_LAUNCH("bn_fold_bwd_multi");
    return DIN_OK;
}

}  // extern "C"
