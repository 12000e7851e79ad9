// 1x1 convolutions with a short reduction (<= 384 channels over hundreds of thousands of pixels) for gfx950: persistent workgroups that stream
// (pixel tile, filter tile, 64-channel block) steps through one LDS ring WITHOUT draining it between tiles.
//
//   D[pix][co] = sum_k X[pix][k] * Wpk[co][k]      X = one NHWC channel view, or the concatenation of up to four (multi-source dgrad)
//   (torch.nn.Conv2d 1x1 reached from the reference at backbone/backbone.py:44-99 -- the Inception 35x35 blocks' branch heads, Conv2d_3b --
//    and its autograd for the data gradient)
//
// Why: the 128-pixel implicit-GEMM kernel (conv_igemm.hip) pays its whole pipeline fill, the stage waits of a handful of k-steps and the staged
// epilogue once per tile; with K = 64 .. 288 that is ~12 us per tile whatever the tile does (Mixed_5b's one 192-wide filter tile runs at the HBM
// rate, Mixed_5c / 5d with two filter tiles take twice / 2.6x as long for the same bytes; profiles/r03_inv3_bf16_per_layer.txt).  Here
//   * one 8-wave workgroup per CU walks items (128 pixels x BN filters; filter tile fastest so that the tiles sharing a pixel tile run on the
//     same XCD at the same time and the second and third read of the pixels is an L2 hit);
//   * a step = one 64-channel block of the item: 128 pixel rows + BN filter rows, 160-byte row pitch (8 data chunks + 2 pad chunks: any 16
//     consecutive rows x one chunk are conflict-free for ds_read_b128), brought in by LDS-DMA as ONE lane-linear image of 35 (BN = 96) or 30
//     (BN = 64) 1-KiB wave transfers; every wave issues the same number of transfers per step (surplus ones fetch nothing into a dump area);
//   * the ring holds 4 steps and runs 3 steps ahead ACROSS items: the first block of the next item is in flight while this item's last block is
//     multiplied and stored;
//   * in-order completion is counted at run time: a per-wave count of issued vector-memory operations and its value when each slab (and the
//     epilogue operands) was issued; "slab landed" = s_waitcnt vmcnt(count now - count then) -- one uniform switch per step instead of a table of
//     hand-derived immediates per (ring depth, epilogue kind);
//   * epilogue: bias (from LDS) and ReLU on the accumulators, the tile staged in the ring slot just consumed and written as 16-byte chunks of whole
//     pixel rows (always issued: out-of-range -> dropped); ReLU-backward mask and accumulate operands are requested at the item's FIRST step
//     through inline asm (the compiler must not wait for them where they are issued -- its wait would also drain the ring); second destination /
//     raw channels of the fused sibling launches.
// bf16 only.  Host: conv1x1_stream_eligible() / launch_conv1x1_stream() (called from conv_igemm.hip).
#include "conv_gather.h"
#include "conv_wgrad.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace din_gather {
namespace {

using din_wgrad::lds_dma16;

// 8-byte buffer load the compiler does not track (no s_waitcnt where it is issued or first used: the caller waits by count)
__device__ __forceinline__ u32x2 buffer_load_b64_untracked(__amdgpu_buffer_rsrc_t rs, int voff) {
    u32x2 v;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}
__device__ __forceinline__ u32x4 buffer_load_b128_untracked(__amdgpu_buffer_rsrc_t rs, int voff) {
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// what one 64-channel block of the reduction reads (multi-source launches keep one entry per block in LDS)
struct BlockDesc { uint32_t a_lo, a_hi; int a_bytes, ld2, abase; uint32_t w_lo, w_hi; int w_bytes, wld16, wbase, nch, pad; };
static_assert(sizeof(BlockDesc) == 48, "three 16-byte reads");

// MULTI: reduction over the concatenation of p.src[0 .. nsrc) (fused dgrad).  EPI: ReLU-backward mask and / or accumulate operand.
// SPLIT: second destination (fused sibling forward launches).
// NSW: ring slots (4 for the 64- / 96-filter tiles; the 192-filter tile's 50-KiB slabs leave room for 3).
template <int BN, int NSW, bool MULTI, bool EPI, bool SPLIT>
__global__ __launch_bounds__(512, 1) void conv1x1_stream_kernel(ConvK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NWV = 8, BM = 128, TI = BN / 32, TJ = 2, PCH = 10, PB = PCH * 16, LOOK = NSW - 1;
    static_assert(NSW == 3 || NSW == 4, "wait counts are written for a ring that runs 2 or 3 steps ahead");
    constexpr int NXA = BM * PCH / 64, NXF = (BM + BN) * PCH / 64;      // wave transfers of the pixel rows / of the whole slab (20, 35 | 30)
    constexpr int NTR = (NXF + NWV - 1) / NWV;                          // per wave per step
    constexpr int SLOT = NXF * 1024, DUMP = NSW * SLOT, BIAS = DUMP + 1024, TABLE = BIAS + 2048;   // (dump: 1 KiB of zeros, shared)
    constexpr int CPR = BN / 8, CH = BM * CPR / 512, CP = BN * 2 + 16;  // staged epilogue: 16-byte chunks per pixel row / per thread, LDS row pitch
    constexpr int NE = EPI ? 2 * CH : 0;                                // epilogue operand loads per item (mask AND old are always issued)
    constexpr int NSTO = SPLIT ? 2 * CH : CH;                           // stores per item
    static_assert(BM * CPR % 512 == 0 && BM * CP <= ((BM + BN) * 10 / 64) * 1024, "the output tile is staged in the ring slot just consumed");
    constexpr unsigned OOB = 0x80000000u;
    static_assert(BM * PCH % 64 == 0 && (BM + BN) * PCH % 64 == 0 && BN % 32 == 0, "slab = whole wave transfers");
    static_assert(!(SPLIT && (EPI || MULTI)), "two destinations: forward launches only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = sgpr(tid >> 6);
    const int frow = lane & 15, g4 = lane >> 4;
    const int wp = wid & 3, wc = wid >> 2;                              // pixel group (32 pixels), filter half
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;

    // ---- the reduction's blocks ----------------------------------------------------------------------------------------------------------------
    int nblk;
    BlockDesc* table = reinterpret_cast<BlockDesc*>(smem_raw + TABLE);
    if (MULTI) {
        nblk = 0;
        for (int s = 0; s < p.nsrc; ++s) nblk += (p.src[s].cpt + 7) >> 3;
        if (tid == 0) {
            int b = 0;
            for (int s = 0; s < p.nsrc; ++s) {
                const ConvK::Src& q = p.src[s];
                for (int lb = 0; lb * 8 < q.cpt; ++lb, ++b) {
                    BlockDesc e;
                    e.a_lo = (uint32_t)(uintptr_t)q.in; e.a_hi = (uint32_t)((uintptr_t)q.in >> 32); e.a_bytes = (int)q.in_bytes;
                    e.ld2 = q.ld * 2; e.abase = (q.coff + lb * 64) * 2;
                    e.w_lo = (uint32_t)(uintptr_t)q.w; e.w_hi = (uint32_t)((uintptr_t)q.w >> 32); e.w_bytes = (int)q.w_bytes;
                    e.wld16 = q.wld * 16; e.wbase = lb * 128; e.nch = q.cpt - lb * 8; e.pad = 0;
                    table[b] = e;
                }
            }
        }
    } else nblk = (p.cpt + 7) >> 3;
    nblk = sgpr(nblk);
    auto block = [&](int b) {
        BlockDesc e;
        if (MULTI) {
            const u32x4* t = reinterpret_cast<const u32x4*>(table + b);
            const u32x4 x0 = t[0], x1 = t[1], x2 = t[2];
            e.a_lo = (uint32_t)sgpr((int)x0[0]); e.a_hi = (uint32_t)sgpr((int)x0[1]); e.a_bytes = sgpr((int)x0[2]); e.ld2 = sgpr((int)x0[3]);
            e.abase = sgpr((int)x1[0]); e.w_lo = (uint32_t)sgpr((int)x1[1]); e.w_hi = (uint32_t)sgpr((int)x1[2]); e.w_bytes = sgpr((int)x1[3]);
            e.wld16 = sgpr((int)x2[0]); e.wbase = sgpr((int)x2[1]); e.nch = sgpr((int)x2[2]); e.pad = 0;
        } else {
            e.a_lo = (uint32_t)(uintptr_t)p.in; e.a_hi = (uint32_t)((uintptr_t)p.in >> 32); e.a_bytes = (int)p.in_bytes;
            e.ld2 = p.ldi * 2; e.abase = (p.cioff + b * 64) * 2;
            e.w_lo = (uint32_t)(uintptr_t)p.w; e.w_hi = (uint32_t)((uintptr_t)p.w >> 32); e.w_bytes = (int)p.w_bytes;
            e.wld16 = p.wld * 16; e.wbase = b * 128; e.nch = p.cpt - b * 8; e.pad = 0;
        }
        return e;
    };
    // ---- bias into LDS (forward launches) ---------------------------------------------------------------------------------------------------
    float* bias_l = reinterpret_cast<float*>(smem_raw + BIAS);
    if (p.flags & DIN_CONV_BIAS) {
        const int cpad = p.n_co_tiles * BN;
        for (int c = tid; c < cpad; c += 512) bias_l[c] = (c < p.Cout && (p.craw <= 0 || c < p.craw)) ? p.bias[c] : 0.f;
    }
    // ---- per-lane transfer plan (item independent): transfer i of this wave covers slab chunk ids [(wid + 8 i) * 64, +64) ----------------------
    // A wave issues ~4 instructions per cycle-quartet at best (two waves per SIMD here), so the step is priced by its instruction count: everything
    // that does not change from step to step lives in registers -- per lane the row inside its operand and the chunk byte offset, per wave the
    // kind of each transfer -- and what changes is uniform: the row pitch, the chunk count of the block and ONE scalar offset per operand (pixel
    // tile * pitch + channel block), passed as the buffer instruction's soffset.  Rows beyond the tensor / the filter bank are out of range of the
    // buffer resource and arrive as zeros.
    int prw[NTR], pc16[NTR], pcc[NTR];
    int kind[NTR];                                                     // 0: pixel rows, 1: filter rows, 2: surplus (uniform per wave)
    uint32_t dst0[NTR];                                                 // LDS byte address inside ring slot 0 (or the dump area)
#pragma unroll
    for (int i = 0; i < NTR; ++i) {
        const int t = wid + NWV * i, id = t * 64 + lane;
        const int row = id / PCH;
        pcc[i] = id - row * PCH;
        pc16[i] = pcc[i] * 16;
        kind[i] = t < NXA ? 0 : t < NXF ? 1 : 2;
        prw[i] = t < NXA ? row : row - BM;
        dst0[i] = lds_base + (uint32_t)(t < NXF ? t * 1024 : DUMP);
    }
    const int n_px = (p.M + BM - 1) / BM, nco = p.n_co_tiles;
    const int G = (int)gridDim.x, Gq = G / nco, Gr = G - Gq * nco;     // item += G  <=>  (pixel tile, filter tile) += (Gq, Gr) with carry
    const bool ka = DIN_KNOCK(p.flags, 0x200), kw = DIN_KNOCK(p.flags, 0x400);    // DIN_GATHER_KNOCK (timing experiments): fetch nothing

    struct Cursor { int pt, ct, b; };                                  // pixel tile, filter tile, block of the reduction
    auto advance = [&](Cursor& c) {
        if (++c.b == nblk) {
            c.b = 0; c.pt += Gq; c.ct += Gr;
            if (c.ct >= nco) { c.ct -= nco; ++c.pt; }
        }
    };
    auto issue_slab = [&](int slot, const Cursor& c) {
        const BlockDesc e = block(c.b);
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)e.a_hi << 32) | e.a_lo), 0, ka ? 0 : e.a_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)e.w_hi << 32) | e.w_lo), 0, kw ? 0 : e.w_bytes, 0x00020000);
        const int soffA = c.pt * BM * e.ld2 + e.abase, soffW = c.ct * BN * e.wld16 + e.wbase;
        const int nchv = e.nch < 8 ? e.nch : 8;
#pragma unroll
        for (int i = 0; i < NTR; ++i) {
            const bool isA = kind[i] == 0;                              // uniform
            const int vo = prw[i] * (isA ? e.ld2 : e.wld16) + pc16[i];
            const int voff = (pcc[i] < nchv && kind[i] != 2) ? vo : (int)OOB;
            lds_dma16(dst0[i] + (uint32_t)(kind[i] != 2 ? slot * SLOT : 0), isA ? rsA : rsW, voff, isA ? soffA : soffW);
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t xbase[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) xbase[j] = (uint32_t)(((wp * TJ + j) * 16 + frow) * PB + g4 * 16);
    const uint32_t wfbase = (uint32_t)((BM + wc * (BN / 2) + frow) * PB + g4 * 16);

    const int item0 = xcd_remap((int)blockIdx.x, G);
    Cursor cur{item0 / nco, item0 % nco, 0};
    if (cur.pt >= n_px) return;
    Cursor nxt = cur;                                                   // next slab to issue
    __syncthreads();                                                    // bias / block table visible
#pragma unroll
    for (int s = 0; s < NSW - 1; ++s)
        if (nxt.pt < n_px) { issue_slab(s, nxt); advance(nxt); }
    // In-order completion by count.  Per step t the wave issues: [NE operand loads if t is the first block of its item], NTR slab transfers
    // (the slab of step t + LOOK), [NSTO stores if t is the last block].  Younger than the slab of step s when step s starts (LOOK = 3): the slabs
    // of s + 1, s + 2, the stores of the epilogues at steps s - 3, s - 2, s - 1 and the operand loads at s - 2, s - 1 (first(t) == last(t - 1));
    // LOOK = 2: the slab of s + 1, the stores at s - 2, s - 1 and the operand loads at s - 1.
    // Once a slab could not be issued (end of the walk) the counts no longer hold: wait for everything.
    bool e1 = false, e2 = false, e3 = false, tail = nxt.pt >= n_px;     // last(s - 1), last(s - 2), last(s - 3)
    u32x4 mk[CH], old[CH];                                              // this thread's chunks of the item's output tile: id = tid + 512 q
    __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((long long)p.M * p.ldo * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rsO2 = __builtin_amdgcn_make_buffer_rsrc(SPLIT ? p.out2 : p.out, 0, SPLIT ? (int)((long long)p.M * p.ldo2 * 2) : 0, 0x00020000);
    const bool want_mask = (p.flags & DIN_CONV_MASK) != 0, want_old = (p.flags & DIN_CONV_ACCUM) != 0;
    __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(want_mask ? p.mask : (const void*)p.out), 0,
                                                                   want_mask ? (int)((long long)p.M * p.ldm * 2) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsOld = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, want_old ? (int)((long long)p.M * p.ldo * 2) : 0, 0x00020000);
    for (;;) {
#pragma unroll
        for (int ws = 0; ws < NSW; ++ws) {                              // ring slot of this step (compile time)
            if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else {
#define DIN_ALLOW(E1, E2, E3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(((LOOK - 1) * NTR + NSTO * (E1 + E2 + (LOOK == 3 ? E3 : 0)) + NE * (E2 + (LOOK == 3 ? E3 : 0))) > 63 ? 63 : \
                                                                       ((LOOK - 1) * NTR + NSTO * (E1 + E2 + (LOOK == 3 ? E3 : 0)) + NE * (E2 + (LOOK == 3 ? E3 : 0)))) : "memory")
                const int code = (e1 ? 1 : 0) | (e2 ? 2 : 0) | ((e3 && LOOK == 3) ? 4 : 0);
                if (code == 0) DIN_ALLOW(0, 0, 0);
                else if (code == 1) DIN_ALLOW(1, 0, 0);
                else if (code == 2) DIN_ALLOW(0, 1, 0);
                else if (code == 4) DIN_ALLOW(0, 0, 1);
                else if (code == 3) DIN_ALLOW(1, 1, 0);
                else if (code == 5) DIN_ALLOW(1, 0, 1);
                else if (code == 6) DIN_ALLOW(0, 1, 1);
                else DIN_ALLOW(1, 1, 1);
#undef DIN_ALLOW
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int m0 = cur.pt * BM;
            const bool first = cur.b == 0, last = cur.b == nblk - 1;
            if (EPI && first) {
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int id = tid + 512 * q, row = id / CPR, c = id - row * CPR;
                    const int m = m0 + row, co = cur.ct * BN + c * 8;
                    const bool ok = m < p.M && co < p.Cout;
                    mk[q] = buffer_load_b128_untracked(rsM, ok ? (m * p.ldm + p.moff + co) * 2 : (int)OOB);               // (empty resource: zeros)
                    old[q] = buffer_load_b128_untracked(rsOld, ok ? (m * p.ldo + p.cooff + co) * 2 : (int)OOB);
                }
            }
            {   // the slab three steps ahead goes into the slot the previous step just finished with
                const int is = (ws + NSW - 1) % NSW;                    // (compile time after unrolling)
                if (nxt.pt < n_px) { issue_slab(is, nxt); advance(nxt); } else tail = true;
            }
            // ---- MFMAs of this block ----------------------------------------------------------------------------------------------------
            const int nch = MULTI ? sgpr(table[cur.b].nch) : p.cpt - cur.b * 8;
            const bool two = nch > 4;
            if (!DIN_KNOCK(p.flags, 0x800)) {                                 // (DIN_GATHER_KNOCK bit 2: no fragment reads / MFMAs)
            const unsigned char* slab = smem_raw + ws * SLOT;
            u32x4 wf[2][TI], xf[2][TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) wf[0][i] = *reinterpret_cast<const u32x4*>(slab + wfbase + i * 16 * PB);
#pragma unroll
            for (int j = 0; j < TJ; ++j) xf[0][j] = *reinterpret_cast<const u32x4*>(slab + xbase[j]);
            if (two) {
#pragma unroll
                for (int i = 0; i < TI; ++i) wf[1][i] = *reinterpret_cast<const u32x4*>(slab + wfbase + i * 16 * PB + 64);
#pragma unroll
                for (int j = 0; j < TJ; ++j) xf[1][j] = *reinterpret_cast<const u32x4*>(slab + xbase[j] + 64);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[0][i]), __builtin_bit_cast(bf16x8, xf[0][j]), acc[i][j], 0, 0, 0);
            if (two) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[1][i]), __builtin_bit_cast(bf16x8, xf[1][j]), acc[i][j], 0, 0, 0);
            }
            }
            // ---- last block of the item: the output tile is staged in the ring slot just consumed and leaves as 16-byte chunks of whole pixel
            //      rows (the accumulator layout would give 8-byte pieces of 32-byte runs: partial-line writes, ~3 TB/s) ----------------------------
            if (last) {
                __builtin_amdgcn_s_barrier();                           // every wave is done with this slot's fragments
                unsigned char* stage = smem_raw + ws * SLOT;
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
                        const int cl = wc * (BN / 2) + i * 16 + g4 * 4, co = cur.ct * BN + cl;
                        f32x4 v = acc[i][j];
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (p.flags & DIN_CONV_BIAS) v += *reinterpret_cast<const f32x4*>(bias_l + co);
                        if ((p.flags & DIN_CONV_RELU) && (p.craw <= 0 || co < p.craw)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        *reinterpret_cast<u32x2*>(stage + ((wp * TJ + j) * 16 + frow) * CP + cl * 2) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (not __syncthreads(): its fence would also wait for the ring's transfers)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (EPI) {
                    // younger than the operand loads: the slabs issued since (one per step of this item, at most the ring's LOOK in flight)
                    if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (nblk >= LOOK) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOOK * NTR) : "memory");
                    else if (nblk == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NTR) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NTR) : "memory");
#pragma unroll
                    for (int q = 0; q < CH; ++q) { asm volatile("" : "+v"(mk[q])); asm volatile("" : "+v"(old[q])); }
                }
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int id = tid + 512 * q, row = id / CPR, c = id - row * CPR;
                    const int m = m0 + row, co = cur.ct * BN + c * 8;
                    const bool ok = m < p.M && co < p.Cout;
                    u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * CP + c * 16);
                    if (EPI) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                            if (want_mask) {
                                if (!(__uint_as_float(mk[q][e] << 16) > 0.f)) lo = 0.f;
                                if (!(__uint_as_float(mk[q][e] & 0xffff0000u) > 0.f)) hi = 0.f;
                            }
                            if (want_old) { lo += __uint_as_float(old[q][e] << 16); hi += __uint_as_float(old[q][e] & 0xffff0000u); }
                            v[e] = pack_bf16x2(lo, hi);
                        }
                    }
                    if (SPLIT) {                                        // fused sibling launch: channels >= csplit go to the second tensor
                        const bool second = co >= p.csplit;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsO, (ok && !second) ? (m * p.ldo + p.cooff + co) * 2 : (int)OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsO2, (ok && second) ? (m * p.ldo2 + p.cooff2 + co - p.csplit) * 2 : (int)OOB, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsO, ok ? (m * p.ldo + p.cooff + co) * 2 : (int)OOB, 0, 0);
                    }
                }
            }
            e3 = e2; e2 = e1; e1 = last;
            advance(cur);
            if (cur.pt >= n_px) return;
        }
    }
#endif
}

template <int BN, int NSW> constexpr size_t stream_lds_bytes() { return (size_t)NSW * ((128 + BN) * 10 / 64) * 1024 + 1024 + 2048 + 1024; }

}  // namespace

// 0: never, 1: where measured to win (default), 2: every eligible launch (tests)
static int stream_mode() { const char* e = DIN_OPT("DIN_CONV_STREAM"); return e ? atoi(e) : 1; }

bool conv1x1_stream_eligible(const ConvK& k, int dtype) {
    const int mode = stream_mode();
    if (!mode || dtype != DIN_BF16 || k.kh != 1 || k.kw != 1 || k.ay != 1 || k.ax != 1 || k.by != 0 || k.bx != 0 || k.divy != 1 || k.divx != 1 ||
        k.remap || k.out_sy != 0 || k.u8 || k.M <= 0 || k.Cout % 8 != 0 || k.ldo % 8 != 0 || k.cooff % 8 != 0)
        return false;
    if ((k.flags & DIN_CONV_MASK) && (k.ldm % 8 != 0 || k.moff % 8 != 0 || (long long)k.M * k.ldm * 2 >= 0x7fffffffll)) return false;
    if ((long long)k.M * k.ldo * 2 >= 0x7fffffffll) return false;
    if (k.csplit > 0 && (k.csplit % 8 != 0 || k.ldo2 % 8 != 0 || k.cooff2 % 8 != 0 || (long long)k.M * k.ldo2 * 2 >= 0x7fffffffll ||
                         (k.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM))))
        return false;
    if (k.craw > 0 && k.craw % 4 != 0) return false;
    const int bn = conv1x1_stream_tile(k.Cout);
    if (((k.Cout + bn - 1) / bn) * bn * 4 > 2048) return false;                      // bias image in LDS
    int blocks = 0;
    const int ns = k.nsrc > 0 ? k.nsrc : 1;
    for (int s = 0; s < ns; ++s) {
        const int cpt = k.nsrc > 0 ? k.src[s].cpt : k.cpt, ld = k.nsrc > 0 ? k.src[s].ld : k.ldi, coff = k.nsrc > 0 ? k.src[s].coff : k.cioff;
        const long long ib = k.nsrc > 0 ? k.src[s].in_bytes : k.in_bytes, wb = k.nsrc > 0 ? k.src[s].w_bytes : k.w_bytes;
        if (cpt <= 0 || ld % 8 != 0 || coff % 8 != 0 || ib >= 0x7fffffffll || wb >= 0x7fffffffll || ib <= 0 || wb <= 0) return false;
        blocks += (cpt + 7) / 8;
    }
    if (blocks > 20) return false;                                                   // block table in LDS
    if (mode == 2) return true;
    // measured window (profiles/r03_conv_stream.txt): short reductions over large maps whose filters fit ONE tile -- with two or three filter
    // tiles every tile re-streams the pixels through the ring and the 128-pixel kernel is as fast or faster; the 192-filter tile (3-slot ring)
    // pays for forward launches only
    const bool plain_fwd = k.nsrc == 0 && !(k.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM));
    const char* wv = DIN_OPT("DIN_CONV_STREAM_WIDE");                                 // 0: never the 192-filter tile
    const bool wide = wv ? atoi(wv) != 0 : true;
    const char* mp = DIN_OPT("DIN_CONV_STREAM_MINPIX");
    const long long minpix = mp ? atoll(mp) : 256 * 1024;
    return blocks <= 6 && (long long)k.M >= minpix && (k.Cout <= 96 || (k.Cout <= 192 && plain_fwd && wide));
}

template <int BN, int NSW, bool MULTI, bool EPI, bool SPLIT>
static void launch_stream(const ConvK& k, dim3 grid, hipStream_t st) {
    constexpr size_t lds = stream_lds_bytes<BN, NSW>();
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = conv1x1_stream_kernel<BN, NSW, MULTI, EPI, SPLIT>;
    din_raise_lds(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, k);
}
template <int BN, int NSW>
static void launch_stream_bn(const ConvK& k, dim3 grid, hipStream_t st) {
    const bool multi = k.nsrc > 0, epi = (k.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM)) != 0, split = k.csplit > 0;
    if (split) launch_stream<BN, NSW, false, false, true>(k, grid, st);
    else if (multi) { if (epi) launch_stream<BN, NSW, true, true, false>(k, grid, st); else launch_stream<BN, NSW, true, false, false>(k, grid, st); }
    else { if (epi) launch_stream<BN, NSW, false, true, false>(k, grid, st); else launch_stream<BN, NSW, false, false, false>(k, grid, st); }
}

int conv1x1_stream_tile(int cout) {
    const char* e = DIN_OPT("DIN_CONV_STREAM_BN");                      // tuning aid: force the filter tile (64 | 96 | 192)
    if (e && (atoi(e) == 64 || atoi(e) == 96 || atoi(e) == 192)) return atoi(e);
    return cout <= 64 ? 64 : cout <= 96 ? 96 : 192;
}

int launch_conv1x1_stream(ConvK k, hipStream_t st) {
    const int bn = conv1x1_stream_tile(k.Cout);
    k.n_co_tiles = (k.Cout + bn - 1) / bn;
    const long long items = (long long)((k.M + 127) / 128) * k.n_co_tiles;
    const dim3 grid((unsigned)(items < 256 ? items : 256));
    if (bn == 64) launch_stream_bn<64, 4>(k, grid, st);
    else if (bn == 96) launch_stream_bn<96, 4>(k, grid, st);
    else launch_stream_bn<192, 3>(k, grid, st);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace din_gather
