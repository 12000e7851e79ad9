// Forward / data-gradient implicit GEMM for the wide filter banks (bf16 operands, fp32 accumulation) for gfx950 -- software-pipelined
// 256-pixel tiles.
//
//   D[co][pix] = sum_k Wpk[co][k] * im2col(X)[pix][k]        k = (r, s, ci), NHWC, ci contiguous
//   (torch.nn.Conv2d reached from the reference at backbone/backbone.py:44-99; its autograd for the data gradient)
//
// Same operands, packing and epilogue as conv_gather_fast_kernel (conv_igemm.hip) -- and the same restructuring that took the weight
// gradient from 0.28 to 0.37-0.45 of the bf16 MFMA peak (conv_wgrad_pipe.hip): the round-1 kernels run every wave through
// "barrier -> DMA issues -> fragment reads -> wait -> MFMAs" in lock-step on a double buffer.  Here
//   * a 256-pixel x BN-filter tile per 8-wave workgroup (one per CU), waves 4 (pixels) x 2 (filters): per-wave 64 x BN/2, i.e.
//     2 x BN/64 tiles of v_mfma_f32_32x32x16_bf16 -- 0.83 (BN = 192) fragment reads per MFMA instead of 1.33 on the 128 x 192 tile;
//   * 32-k stages (64 B per tile row) in a 4-slot LDS ring filled by LDS-DMA three stages ahead (hand-counted vmcnt, one barrier per
//     stage); two fragment register sets: the ds_read_b128 of half-stage h+1 fly while the MFMAs of half-stage h run, the DMA issues
//     and the scalar k-walk of stage s+3 sit between MFMAs;
//   * LDS image [row][4 chunks] with chunk' = chunk ^ ((row >> 2) & 3): each of the four 16-lane groups of a ds_read_b128 (32 rows x
//     one chunk column) then covers all 64 banks exactly once; applied on the source side of the lane-linear DMA;
//   * reduction order: taps inside 32-channel blocks (consecutive stages re-read almost the same pixels, shifted by one tap).
// Eligibility (host): bf16, whole 32-channel blocks per tap (Cin % 32 == 0), <= 32 taps, unit stride of the gather (forward of any
// stride-1 conv, dgrad of stride-1 convs), no split-K, no tap remap, no multi-source launch.
#include "conv_gather.h"
#include "conv_wgrad.h"
#include <unordered_map>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace din_gather {
namespace {

using din_wgrad::lds_dma16;
template <int V> struct IC { static constexpr int value = V; };

template <int BN>
__global__ __launch_bounds__(512, 1) void conv_gather_pipe_kernel(ConvK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef bf16_t T;
    constexpr int BM = 256, NT = 512, KC = 4, NS = 4;               // pixels per tile, threads, 16-byte chunks per stage row, ring slots
    constexpr int WM = 4, WN = 2;                                    // waves: 4 along the pixels x 2 along the filters
    constexpr int LR = NT / KC;                                      // 128 tile rows per loader pass
    constexpr int PA = BM / LR, PB = (BN + LR - 1) / LR;             // DMA transfers per thread per stage (pixel rows / filter rows)
    constexpr int ROWB = KC * 16;                                    // 64 B per tile row per stage
    constexpr int OPA = BM * ROWB, OPB = PB * LR * ROWB, STAGE = OPA + OPB;
    constexpr int TI = BN / WN / 32, TJ = BM / WM / 32;              // 32x32 MFMA tiles per wave: filters x pixels
    constexpr int NM = TI * TJ, NF = TI + TJ;
    constexpr int NDMA = PA + PB;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(BN % 64 == 0 && BN <= 256 && NM >= NF, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    int bx_, by_;
    xcd_block(bx_, by_);
    const int co_tile = bx_ % p.n_co_tiles, px_tile = bx_ / p.n_co_tiles;
    const int ntaps = p.kh * p.kw;
    const int nsteps = ntaps * (p.cpt / KC);                         // host guarantees cpt % KC == 0
    // byte offset of tap t = (r, s) relative to tap (0, 0)
    const int dA = p.cy * p.W * p.ldi * (int)sizeof(T), dB = p.cx * p.ldi * (int)sizeof(T);

    const int m_first = px_tile * BM;
    const int n_first = m_first / (p.OH * p.OW);
    const long long img_bytes = (long long)p.H * p.W * p.ldi * (long long)sizeof(T);
    const long long a_off = (long long)n_first * img_bytes;
    long long a_rem = p.in_bytes - a_off;
    if (a_rem > 0x7fffffffll) a_rem = 0x7fffffffll;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in)) + a_off, 0, (int)a_rem, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

    // ---- loader state: thread -> tile rows r0 + 128 i, slot tid % 4; the lane fetches logical chunk cq = slot ^ swizzle(row) ----------
    const int r0 = tid / KC;
    const int cq = (tid % KC) ^ ((r0 >> 2) & 3);                     // (row >> 2) & 3 is the same for r0 + 128 i
    unsigned pixq[PA], vmask[PA];
    {
        const unsigned full_row = p.kw >= 32 ? 0xffffffffu : ((1u << p.kw) - 1u);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = m_first + r0 + LR * i;
            vmask[i] = 0u; pixq[i] = 0u;
            if (m < p.M) {
                const int n = m / (p.OH * p.OW), rem = m - n * (p.OH * p.OW);
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                const int ty0 = oy * p.ay + p.by, tx0 = ox * p.ax + p.bx;
                unsigned cmask = 0u;
                for (int s2 = 0; s2 < p.kw; ++s2) {
                    const int tx = tx0 + s2 * p.cx;
                    cmask |= (tx >= 0 && tx < p.W) ? (1u << s2) : 0u;
                }
                cmask &= full_row;
                unsigned mk = 0u;
                for (int r = 0; r < p.kh; ++r) {
                    const int ty = ty0 + r * p.cy;
                    mk |= (ty >= 0 && ty < p.H) ? (cmask << (r * p.kw)) : 0u;
                }
                vmask[i] = mk;
                pixq[i] = (unsigned)((((n - n_first) * p.H + ty0) * p.W + tx0) * p.ldi * (int)sizeof(T) + p.cioff * (int)sizeof(T) + cq * 16);
            }
        }
    }
    int voffB[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i)
        voffB[i] = (r0 + LR * i < BN) ? ((co_tile * BN + r0 + LR * i) * p.wld + cq) * 16 : (int)OOB;   // rows of the next tile / beyond the bank: zeros

    // ---- scalar walk over the stages: step -> (32-channel block step / ntaps, tap step % ntaps) ---------------------------------------
    int tap = 0, tc = 0, td = 0, fa = 0, fb = 0;                     // next stage to ISSUE
    const int tap_row_wrap = dA - p.kw * dB, cpt16 = p.cpt * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t ldsW = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(wid * 1024));   // this wave's 16 rows of pass 0
    auto issue = [&](int slot, int part) {                            // part 0: pixel rows, 1: filter rows (+ advance the walk)
        const uint32_t A = ldsW + (uint32_t)(slot * STAGE), B = A + (uint32_t)OPA;
        if (part == 0) {
            const unsigned bit = 1u << tap;
#pragma unroll
            for (int i = 0; i < PA; ++i)
                lds_dma16(A + (uint32_t)(i * LR * ROWB), rsA, (int)((vmask[i] & bit) ? pixq[i] + (unsigned)td : OOB), fa);
        } else {
#pragma unroll
            for (int i = 0; i < PB; ++i) lds_dma16(B + (uint32_t)(i * LR * ROWB), rsB, voffB[i], fb);
            ++tap; ++tc; td += dB; fb += cpt16;
            if (tc == p.kw) { tc = 0; td += tap_row_wrap; }
            if (tap == ntaps) { tap = 0; td = 0; fa += KC * 16; fb = fa; }
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment addressing: lane l reads row l & 31 of a 32-row tile, chunk 2 u + (l >> 5) of half-stage u, swizzled ---------------
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 2) & 3;
    uint32_t adA[2][2], adB[2][2];                                    // [slot pair][half-stage]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t ch = (uint32_t)(((2 * u + lh) ^ sw) * 16);
        adA[0][u] = lds_base + (uint32_t)((wm * (BM / WM) + lr) * ROWB) + ch;
        adB[0][u] = lds_base + (uint32_t)(OPA + (wn * (BN / WN) + lr) * ROWB) + ch;
        adA[1][u] = adA[0][u] + 2 * STAGE;
        adB[1][u] = adB[0][u] + 2 * STAGE;
    }
    bf16x8 wf[2][TI], xf[2][TJ];                                      // two fragment sets: filters, pixels
    auto load_frag = [&](auto slot_c, int set, int u, int q) {
        constexpr int SLOT = decltype(slot_c)::value;
        const uint32_t imm = (uint32_t)((SLOT & 1) * STAGE);
        if (q < TI) wf[set][q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(adB[SLOT >> 1][u] + imm + (uint32_t)(q * 32 * ROWB)));
        else xf[set][q - TI] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(adA[SLOT >> 1][u] + imm + (uint32_t)((q - TI) * 32 * ROWB)));
    };
    auto mma = [&](int set, int m) {
        acc[m / TJ][m % TJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[set][m / TJ], xf[set][m % TJ], acc[m / TJ][m % TJ], 0, 0, 0);
    };

    auto stage = [&](auto slot_c, int s) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int NEXT = (SLOT + 1) & (NS - 1), FILL = (SLOT + NS - 1) & (NS - 1);
        const bool more = s + NS - 1 < nsteps;
#pragma unroll
        for (int q = 0; q < NF; ++q) load_frag(slot_c, 1, 1, q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mma(0, m);
            if (m == 1 || m == NM - 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue(FILL, m == 1 ? 0 : 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_assert(NM >= 4, "two DMA slots between the MFMAs of a half-stage");
        if (s + 3 < nsteps) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * NDMA) : "memory");
        else if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mma(1, m);
            if (m < NF) {
                __builtin_amdgcn_sched_barrier(0);
                load_frag(IC<NEXT>{}, 0, 0, m);                       // (behind the last stage: stale ring contents, never used)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (nsteps > 0) {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (s0 < nsteps) { issue(s0, 0); issue(s0, 1); }
        if (nsteps >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NDMA) : "memory");
        else if (nsteps == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < NF; ++q) load_frag(IC<0>{}, 0, 0, q);
        for (int s = 0; s < nsteps; s += NS) {
            stage(IC<0>{}, s);
            if (s + 1 < nsteps) stage(IC<1>{}, s + 1);
            if (s + 2 < nsteps) stage(IC<2>{}, s + 2);
            if (s + 3 < nsteps) stage(IC<3>{}, s + 3);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                                  // all waves done with the ring: the epilogue reuses it

    // ---- epilogue: bias / ReLU on the accumulators, tile -> LDS [pixel][BN] (pitch BN * 2 + 16), then the shared coalesced store -------
    // 32x32 accumulator layout: element e of lane l = filter row 8 (e >> 2) + 4 (l >> 5) + (e & 3), pixel column l & 31
    {
        constexpr int CPITCH = BN * (int)sizeof(T) + 16;
        // (BN = 256: the staged output tile is larger than the ring; the host sizes the dynamic LDS for the larger of the two)
        const int px_l = wm * (BM / WM) + lr;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co_l = wn * (BN / WN) + 32 * i + 8 * g4 + 4 * lh;   // 4 consecutive channels
                const int co = co_tile * BN + co_l;
                const bool cooked = p.craw <= 0 || co < p.craw;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if ((p.flags & DIN_CONV_BIAS) && co < p.Cout && cooked) {            // Cout % 8 == 0 on this path
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = p.bias[co + e];
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][4 * g4 + e] + bv[e];
                        if ((p.flags & DIN_CONV_RELU) && cooked) v[e] = fmaxf(v[e], 0.f);
                    }
                    unsigned char* dst = smem_raw + (px_l + 32 * j) * CPITCH + co_l * (int)sizeof(T);
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                }
            }
    }
    staged_tile_store<T, BM, BN, NT>(p, smem_raw, tid, co_tile, m_first);
#endif
}

template <typename K>
void raise_lds(K kern, size_t lds) { din_raise_lds(reinterpret_cast<const void*>(kern), lds); }

}  // namespace

bool gather_pipe_tile_ok(int bn) { return bn == 128 || bn == 192 || bn == 256; }

int launch_gather_pipe(const ConvK& k, int bn, int n_px_tiles, hipStream_t st) {
    DIN_REQUIRE(gather_pipe_tile_ok(bn), "gather pipe kernel: filter tile %d not instantiated", bn);
    DIN_REQUIRE(k.cpt % 4 == 0 && k.kh * k.kw <= 32 && k.splitk == 1 && !k.remap && k.nsrc == 0 && k.divy == 1 && k.divx == 1,
                "gather pipe kernel: launch not eligible");
    dim3 grid(n_px_tiles * k.n_co_tiles, 1);
    auto launch = [&](auto kern, size_t lds) {
        raise_lds(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, k);
    };
    auto bytes = [](int ring_rows, int bn_) { const size_t ring = 4 * (size_t)ring_rows * 64, tile = 256 * ((size_t)bn_ * 2 + 16); return ring > tile ? ring : tile; };
    if (bn == 128) launch(conv_gather_pipe_kernel<128>, bytes(256 + 128, 128));
    else if (bn == 192) launch(conv_gather_pipe_kernel<192>, bytes(256 + 256, 192));
    else launch(conv_gather_pipe_kernel<256>, bytes(256 + 256, 256));
    return DIN_OK;
}

}  // namespace din_gather
