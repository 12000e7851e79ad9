// Weight gradient of the wide filter banks (bf16 operands, fp32 accumulation) for gfx950 -- software-pipelined ring kernel.
//
//   dW[co][(r,s,ci)] = sum_pix G[pix][co] * im2col(X)[pix][(r,s,ci)]        (reference: autograd of torch.nn.Conv2d reached from
//   backbone/backbone.py:44-99; both operands are stored [pixel][channel], i.e. the reduction index is the STRIDED one on both sides)
//
// Same data movement as conv_wgrad_ring_kernel (conv_igemm.hip): BCO x 256 output tile per 8-wave workgroup, 32-pixel stages in a
// 4-slot LDS ring filled by LDS-DMA three stages ahead, operands read with the transposing ds_read_b64_tr_b16.  What changes is the
// schedule inside a wave -- the round-1 counters (profiles/r01_wgrad_ring.txt: MFMA busy 33 %, LDS conflicts 26 % of LDS cycles,
// 2.7 SALU + 2.0 VALU per MFMA) showed every wave doing "barrier -> 4 DMA issues -> 20 transpose reads -> wait -> 24 MFMAs" in
// lock-step, so the matrix pipe idled through the first three phases:
//   * v_mfma_f32_32x32x16_bf16: 12 instead of 24 MFMAs per stage, 32 cycles each -- room for ~5 other issues behind every one;
//   * two fragment register sets: the transpose reads of half-stage h+1 are in flight while the MFMAs of half-stage h run, and the
//     DMA issues of stage s+3 sit between MFMAs instead of in front of them (no LDS or DMA-issue latency on the critical path);
//   * both operand tiles use a 512-byte LDS row pitch with the 64-byte slots of row r rotated by (r & 3): every 32-lane half of a
//     transpose read then touches 4 rows x 64 B = all 64 banks exactly once (the 384-byte rows of the 192-filter G tile could not be
//     made conflict-free by rotation alone: a wrapped slot lands 128 B off its bank group);
//   * per-stage coordinate updates are branch-free when a feature-map row holds at least one stage (OW >= 32);
//   * the bias gradient (BatchNorm shift gradient) is a packed dot product against ones on the VALU (v_dot2c_f32_bf16, 4 per
//     half-stage per wave, spread over the waves of the k_tile == 0 workgroups) instead of 25 % extra MFMAs on a quarter of the waves;
//   * optionally (WgradK::atomic) the workgroups add their tiles into ONE fp32 tile buffer with native atomics, so the slice
//     partials (25-80 MB per launch whatever the batch) are neither written nor read back.
#include "conv_wgrad.h"
#ifndef DIN_PIPE_KNOCK
#define DIN_PIPE_KNOCK 0      // timing experiments only (results are wrong for != 0): 1 no stage barrier, 2 fragments read once, 3 no DMA after the prologue, 4 no MFMA
#endif
#include <unordered_map>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace din_wgrad {
namespace {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ s16x4 tr_read(uint32_t byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) s16x4*>(byte_addr));
}
#endif

template <int V> struct IC { static constexpr int value = V; };

// WN: waves along the k columns (WM = 2 along the filter rows): 8 -> 16 waves with (BCO/2) x 32 wave tiles (shipped: four waves per SIMD);
// 4 -> 8 waves with (BCO/2) x 64 wave tiles; 2 -> 4 waves with (BCO/2) x 128 wave tiles (one wave per SIMD, 7 instead of 10 fragment reads
// per 12 MFMAs).  Measured (profiles/r02_wgrad_pipe_knockout.txt): more waves win although they read more fragments.
#if defined(__HIP_DEVICE_COMPILE__)
template <int BCO, int BK, bool WIDE, int WN>
__device__ __forceinline__ void wgrad_pipe_body(const WgradK& p, const int bx_, const int by_) {
    constexpr int PK = 32, NS = 4, NWV = 2 * WN;
    constexpr int TR = 16 / NWV;                                   // wave-level transfers per operand and stage (2 tile rows each)
    constexpr int ROWB = 512;                                      // LDS row pitch of both operand tiles (bytes)
    static_assert(BK == 256 && BCO % 64 == 0 && BCO <= 256, "tile");
    constexpr int CG = BCO / 8;                                    // real 16-byte chunks of a G row (of 32 slots)
    constexpr int OPG = PK * ROWB, STAGE = 2 * OPG;                // 16 KiB per operand, 32 KiB per stage
    constexpr int TI = BCO / 64, XJ = BK / (32 * WN);              // 32x32 MFMA tiles per wave (wave tile = BCO/2 x BK/WN)
    constexpr int NM = TI * XJ;                                    // MFMAs per half-stage
    constexpr int NDMA = 2 * TR;                                   // wave-level DMAs per stage per wave: TR for G + TR for X
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: lives in an SGPR, branches on it are scalar
    const int wm = wid / WN, wn = wid % WN;
    const int k_tile = bx_ % p.n_k_tiles, co_tile = bx_ / p.n_k_tiles;
    const int slice = by_;
    const int m_begin = slice * p.m_per_slice;                     // multiple of PK
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

    // ---- DMA plan.  A wave-level transfer is 1 KiB lane-linear = 2 tile rows x 32 slots of 16 B; transfer t of wave w covers rows
    //      2 (w + NWV t) + (lane >> 5).  Slot s of row r holds the row's logical chunk (s - 4 (r & 3)) mod 32: the rotation is applied on
    //      the source side (the lane fetches the chunk that belongs at its slot) and again in the transpose-read addresses.
    const int drow = 2 * wid + (lane >> 5);                       // rows drow + 2 NWV t ((row & 3) is the same for all t)
    const int lchunk = ((lane & 31) - 4 * (drow & 3)) & 31;       // logical chunk at this lane's slot
    unsigned voffG[TR];
    {
        const int gco = co_tile * BCO + lchunk * 8;
        const bool ok = lchunk < CG && gco + 7 < p.Cout;          // Cout % 8 == 0 enforced by the host
#pragma unroll
        for (int t = 0; t < TR; ++t) voffG[t] = ok ? (unsigned)(((drow + 2 * NWV * t) * p.ldo + p.cooff + gco) * 2) : OOB;
    }
    const int kcol = k_tile * BK + lchunk * 8;
    const bool kok = kcol < p.kcols;
    const int tap = kok ? kcol / p.cin_pad : 0;
    const int ci = kcol - tap * p.cin_pad;
    const int tr_ = tap / p.kw, ts_ = tap - tr_ * p.kw;
    const bool ci_ok = kok && ci + 7 < p.Cin;                      // Cin % 8 == 0 enforced by the host
    const int dy0 = -p.ph + tr_ * p.dh, dx0 = -p.pw + ts_ * p.dw;  // iy = oy*sh + dy0, ix = ox*sw + dx0
    // per-transfer cursor of this lane's pixel row: output column px, input coordinates (ix, iy) of the lane's tap, byte offset `off` of
    // that input pixel (+ channel) from the resource base, rows left in the slice.  One stage = +32 output pixels.
    const int step32 = PK * p.sw * p.ldi * 2, dix32 = PK * p.sw;
    const int wrap_off = (p.sh * p.W - p.OW * p.sw) * p.ldi * 2;   // back to column 0 of the next output row
    const int wrap_ix = p.OW * p.sw;
    const int img_off = (p.H - p.OH * p.sh) * p.W * p.ldi * 2;     // extra when wrapping to the next image
    const int img_iy = p.OH * p.sh;
    int px[TR], py[TR], ix[TR], iy[TR], off[TR], left[TR];
#pragma unroll
    for (int t = 0; t < TR; ++t) {
        const int m = m_begin + drow + 2 * NWV * t;
        const int n = m / ohw, rem = m - n * ohw;
        py[t] = rem / p.OW; px[t] = rem - py[t] * p.OW;
        iy[t] = py[t] * p.sh + dy0; ix[t] = px[t] * p.sw + dx0;
        off[t] = (((n - n_first) * p.H + iy[t]) * p.W + ix[t]) * p.ldi * 2 + (p.cioff + ci) * 2;
        left[t] = ci_ok ? m_end - m : 0;                           // <= 0: nothing (more) to fetch for this row
    }

    const uint32_t lds_base = (uint32_t)(uintptr_t)smem_raw;
    const uint32_t ldsW = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(wid * 1024));
    auto issue_g = [&](int slot, int st) {                          // stage st -> ring slot
        const uint32_t Gd = ldsW + (uint32_t)(slot * STAGE);
        const int soffG = st * PK * p.ldo * 2;
#pragma unroll
        for (int t = 0; t < TR; ++t) lds_dma16(Gd + (uint32_t)(t * 1024 * NWV), rsG, (int)voffG[t], soffG);   // rows past M: out of range -> zeros
    };
    auto issue_x = [&](int slot, int t) {                           // called once per stage and t, stages in order (the cursor advances)
        const uint32_t Xd = ldsW + (uint32_t)(slot * STAGE + OPG + t * 1024 * NWV);
        const bool ok = (left[t] > 0) & ((unsigned)iy[t] < (unsigned)p.H) & ((unsigned)ix[t] < (unsigned)p.W);
        lds_dma16(Xd, rsX, ok ? off[t] : (int)OOB, 0);
        px[t] += PK; ix[t] += dix32; off[t] += step32; left[t] -= PK;
        if constexpr (WIDE) {                                       // OW >= 32: at most one row wrap per stage, and a wave's two rows
            if (__builtin_amdgcn_ballot_w64(px[t] >= p.OW)) {      // of a transfer are neighbours -> the wrap is (nearly) wave-uniform
                const bool wrap = px[t] >= p.OW;
                px[t] -= wrap ? p.OW : 0; ix[t] -= wrap ? wrap_ix : 0; off[t] += wrap ? wrap_off : 0;
                iy[t] += wrap ? p.sh : 0; py[t] += wrap ? 1 : 0;
                const bool wimg = py[t] == p.OH;
                py[t] = wimg ? 0 : py[t]; iy[t] -= wimg ? img_iy : 0; off[t] += wimg ? img_off : 0;
            }
        } else {
            while (px[t] >= p.OW) {
                px[t] -= p.OW; ix[t] -= wrap_ix; off[t] += wrap_off; iy[t] += p.sh;
                if (++py[t] == p.OH) { py[t] = 0; iy[t] -= img_iy; off[t] += img_off; }
            }
        }
    };

    f32x16 acc[TI][XJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < XJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // bias gradient: wave wn of a k_tile == 0 workgroup sums the G tiles wn, wn + WN, ... (< TI) of its filter half
    const bool do_bias = p.dbias != nullptr && k_tile == 0 && wn < TI;        // wave-uniform
    float bsum[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) bsum[i] = 0.f;

    // ---- transpose-read addressing for the 32x32x16 operand layout (A: row l & 31, B: column l & 31; k = 8 (l >> 5) .. +7).  Lane
    //      l = 16 q + i RECEIVES channel 16 (q & 1) + i of a 32-channel tile for the 8 pixels 8 (q >> 1) .. +7 of the half-stage and
    //      ADDRESSES pixel row 8 (q >> 1) + 4 r + (i >> 2) (r = 0, 1: the two reads), 4 channels at 4 (i & 3) of its 16-channel group
    //      (ds_read_b64_tr_b16 lane map: profiles/r01_probe_ds_read_tr_b16.txt).  (row & 3) == i >> 2 for every read.  Two address
    //      registers per tile (ring slots 0-1 / 2-3): everything else of an address is an immediate (16-bit offset field).
    const int li = lane & 15, lq = lane >> 4;
    const uint32_t lrow = lds_base + (uint32_t)((8 * (lq >> 1) + (li >> 2)) * ROWB + 32 * (lq & 1) + 8 * (li & 3));
    uint32_t colG[2][TI], colX[2][XJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        colG[0][i] = lrow + (uint32_t)((64 * (wm * TI + i) + 64 * (li >> 2)) & (ROWB - 1));
        colG[1][i] = colG[0][i] + 2 * STAGE;
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        colX[0][j] = lrow + (uint32_t)(OPG + ((64 * (wn * XJ + j) + 64 * (li >> 2)) & (ROWB - 1)));
        colX[1][j] = colX[0][j] + 2 * STAGE;
    }

    bf16x8 ga[2][TI], xb[2][XJ];                                   // two fragment sets
#if DIN_PIPE_KNOCK == 2
    bool knock_first = true;
#endif
    auto frag = [&](uint32_t addr) {                               // one operand fragment: pixels 0..3 and 4..7 of the lane's octet
        const s16x4 lo = tr_read(addr), hi = tr_read(addr + 4 * ROWB);
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // fragment q (G tiles first, then X tiles) of half-stage u of ring slot SLOT -> set
    // request order = first use by the MFMA sequence (0,0) (0,1) (1,0) (1,1) ...: G tile 0, the X tiles, then the other G tiles -- the
    // first MFMA of the next half-stage then waits for the 1st and 2nd request instead of the 1st and (TI+1)-th
    auto load_frag = [&](auto slot_c, int set, int u, int q_req) {
        constexpr int SLOT = decltype(slot_c)::value;
        const int q = q_req == 0 ? 0 : (q_req <= XJ ? TI + q_req - 1 : q_req - XJ);
        const uint32_t imm = (uint32_t)((SLOT & 1) * STAGE + u * 16 * ROWB);
#if DIN_PIPE_KNOCK == 2
        if (knock_first) {
#endif
        if (q < TI) ga[set][q] = frag(colG[SLOT >> 1][q] + imm);
        else xb[set][q - TI] = frag(colX[SLOT >> 1][q - TI] + imm);
#if DIN_PIPE_KNOCK == 2
        }
#endif
    };
    auto mma = [&](int set, int m) {                               // MFMA m of a half-stage: tile (m / XJ, m % XJ)
#if DIN_PIPE_KNOCK == 4
        asm volatile("" : "+v"(acc[m / XJ][m % XJ]) : "v"(ga[set][m / XJ]), "v"(xb[set][m % XJ]));
#else
        acc[m / XJ][m % XJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[set][m / XJ], xb[set][m % XJ], acc[m / XJ][m % XJ], 0, 0, 0);
#endif
    };
    // (inline asm on purpose: with the builtin the compiler merges the per-tile branches below into one indexed access of the fragment
    // array, which then lives in scratch memory -- and scratch traffic would also break the hand-counted vmcnt of the DMA ring)
    auto ones_dot = [&](const bf16x8& f, float& sum) {
        const u32x4 v = __builtin_bit_cast(u32x4, f);
        asm volatile("v_dot2c_f32_bf16 %0, %1, %2\n\tv_dot2c_f32_bf16 %0, %1, %3\n\tv_dot2c_f32_bf16 %0, %1, %4\n\tv_dot2c_f32_bf16 %0, %1, %5"
                     : "+v"(sum) : "s"(0x3f803f80u), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    };
    auto bias = [&](int set) {                                     // scalar branches: wn lives in an SGPR
        if (do_bias) {
            if (wn == 0 % WN) ones_dot(ga[set][0], bsum[0]);
            if constexpr (TI > 1) { if (wn == 1 % WN) ones_dot(ga[set][1], bsum[1]); }
            if constexpr (TI > 2) { if (wn == 2 % WN) ones_dot(ga[set][2], bsum[2]); }
            if constexpr (TI > 3) { if (wn == 3 % WN) ones_dot(ga[set][3], bsum[3]); }
        }
    };

    const int nst = (m_end - m_begin + PK - 1) / PK;               // stages of this slice (>= 0)
    // ---- sibling pacing (see WgradK::pace): wave 0 publishes / checks every PACE_EVERY stages, the stage barrier holds the others back.
    //      The siblings' progress words are read with a SCALAR load (glc: past the scalar cache; counted by lgkmcnt): a vector load would
    //      make the compiler drain vmcnt -- i.e. the three DMA stages in flight -- before its result can be used.  The publishing store
    //      is one more VMEM op in the stream: the hand-counted vmcnt then waits for at most one younger DMA piece too many (conservative).
    constexpr int PACE_EVERY = 4, PACE_SLACK = 3, PACE_SPINS = 6, PACE_ROW = 8;
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    int* pace_row = p.pace ? p.pace + (int64_t)(slice * p.n_co_tiles + co_tile) * PACE_ROW : nullptr;
    auto pace = [&](int s) {
        if (pace_row == nullptr || wid != 0 || (s & (PACE_EVERY - 1)) != 0) return;      // scalar conditions
        if (lane == 0) __hip_atomic_store(pace_row + k_tile, p.pace_base + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < PACE_SPINS; ++spin) {
            i32x8 v;
            asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pace_row) : "memory");
            int behind = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < PACE_ROW; ++e) {        // words outside this launch's tag range: not started / no such sibling / an older launch
                const unsigned rel = (unsigned)(v[e] - p.pace_base - 1);
                behind = (rel < (1u << 20) && (int)rel + 1 < behind) ? (int)rel + 1 : behind;
            }
            if (s + 1 - behind <= PACE_SLACK) break;                 // (own word included: behind <= s + 1 once the store landed)
            __builtin_amdgcn_s_sleep(4);
        }
    };
    // MFMA of the first half behind which X transfer t of stage s+3 is issued (G: behind MFMA 1)
    auto xslot = [](int t) { return WN == 4 ? (t == 0 ? NM / 2 : NM - 2) : WN == 2 ? 3 + 2 * t : NM - 1; };
    // One stage (ring slot SLOT).  On entry its first half-stage's fragments (set 0) are already requested.
    auto stage = [&](auto slot_c, int s) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int NEXT = (SLOT + 1) & (NS - 1), FILL = (SLOT + NS - 1) & (NS - 1);
#if DIN_PIPE_KNOCK == 3
        const bool more = false;
#else
        const bool more = s + NS - 1 < nst;
#endif
#if DIN_PIPE_KNOCK == 2
        if (s > 0) knock_first = false;
#endif
        if constexpr (SLOT == 0) pace(s);
        // ---- first half: request the second half's fragments (set 1), then the MFMAs of set 0 with the DMA issues of stage s+3
        //      between them (they go to the ring slot of stage s-1: every wave finished reading it before the last barrier)
#pragma unroll
        for (int q = 0; q < TI + XJ; ++q) load_frag(slot_c, 1, 1, q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mma(0, m);
            // DMA issues of stage s+3 behind MFMAs 1 (G) and, per X transfer, NM/2 and NM-2 (8 waves) / 3, 5, 7, 9 (4 waves)
            bool slot_here = m == 1;
#pragma unroll
            for (int t = 0; t < TR; ++t) slot_here = slot_here || m == xslot(t);
            if (slot_here) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    if (m == 1) issue_g(FILL, s + NS - 1);
#pragma unroll
                    for (int t = 0; t < TR; ++t)
                        if (m == xslot(t)) issue_x(FILL, t);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        bias(0);
        // ---- stage s+1 landed (own share) + all transpose reads of stage s returned, then the workgroup barrier: afterwards
        //      stage s+1 is complete in LDS and nobody reads this stage's ring slot any more
        if (s + 3 < nst) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * NDMA) : "memory");
        else if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if DIN_PIPE_KNOCK != 1
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        // ---- second half: MFMAs of set 1 with the requests for the next stage's first half (set 0) between them
        //      (unconditionally: behind the last stage they fetch stale ring contents that nobody uses)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mma(1, m);
            if (m < TI + XJ) {
                __builtin_amdgcn_sched_barrier(0);
                load_frag(IC<NEXT>{}, 0, 0, m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (NM < TI + XJ) {                               // (16 waves: 3 MFMAs, 4 fragments) the rest behind the last MFMA
#pragma unroll
            for (int q = NM; q < TI + XJ; ++q) load_frag(IC<NEXT>{}, 0, 0, q);
        }
        bias(1);
    };
    if (nst > 0) {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (s0 < nst) {
                issue_g(s0, s0);
#pragma unroll
                for (int t = 0; t < TR; ++t) issue_x(s0, t);
            }
        if (nst >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NDMA) : "memory");
        else if (nst == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < TI + XJ; ++q) load_frag(IC<0>{}, 0, 0, q);
        for (int s = 0; s < nst; s += NS) {
            stage(IC<0>{}, s);
            if (s + 1 < nst) stage(IC<1>{}, s + 1);
            if (s + 2 < nst) stage(IC<2>{}, s + 2);
            if (s + 3 < nst) stage(IC<3>{}, s + 3);
        }
    }

    // ---- epilogue.  32x32 accumulator layout: element e of lane l is row 8 (e >> 2) + 4 (l >> 5) + (e & 3), column l & 31
    float* dst = p.partial + (p.atomic ? (int64_t)0 : (int64_t)slice * p.cout_pad * p.kcols_pad);
    const int co0 = co_tile * BCO + wm * (BCO / 2) + 4 * (lane >> 5);
    const int kc0 = k_tile * BK + wn * (BK / WN) + (lane & 31);
    if (p.atomic) {
        if (nst > 0) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < XJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        atomicAdd(dst + (int64_t)(co0 + 32 * i + 8 * (e >> 2) + (e & 3)) * p.kcols_pad + kc0 + 32 * j, acc[i][j][e]);
        }
    } else if (p.direct != nullptr) {                               // one slice, 1x1, no scale: the tile IS the gradient
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < XJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = co0 + 32 * i + 8 * (e >> 2) + (e & 3), kc = kc0 + 32 * j;
                    if (co < p.Cout && kc < p.Cin) p.direct[(int64_t)co * p.Cin + kc] = acc[i][j][e];
                }
    } else {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < XJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    dst[(int64_t)(co0 + 32 * i + 8 * (e >> 2) + (e & 3)) * p.kcols_pad + kc0 + 32 * j] = acc[i][j][e];
    }
    if (do_bias) {
        // lanes l and l ^ 32 hold the two pixel-octets' sums of the same channel
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (i % WN != wn) continue;
            const float tsum = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int co = co_tile * BCO + wm * (BCO / 2) + 32 * i + (lane & 31);
            if (lane < 32 && co < p.Cout) atomicAdd(p.dbias + co, tsum);
        }
    }
}
#endif

template <int BCO, int BK, bool WIDE, int WN = 4>
__global__ __launch_bounds__(128 * WN, 1) void conv_wgrad_pipe_kernel(WgradK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    int bx_, by_;
    xcd_block(bx_, by_);
    wgrad_pipe_body<BCO, BK, WIDE, WN>(p, bx_, by_);
#endif
}

// several layers in one launch (conv_wgrad.h: WgradGroupK).  The item is found with a scalar scan of first[]; its argument block is read
// from the kernel-argument segment through a wave-uniform index (scalar loads), the body is the single-layer kernel's.
template <int BCO, int BK, bool WIDE, int WN>
__global__ __launch_bounds__(128 * WN, 1) void conv_wgrad_pipe_group_kernel(WgradGroupK grp) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int l = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int gi = 0;
#pragma unroll
    for (int i = 1; i < WGRAD_GROUP_MAX; ++i) gi += (i < grp.n && l >= grp.first[i]) ? 1 : 0;
    gi = __builtin_amdgcn_readfirstlane(gi);
    const WgradK p = grp.k[gi];                                    // a COPY: through a reference the stage loop re-reads fields with s_load (5 per stage)
    const int local = l - grp.first[gi], nx = p.n_co_tiles * p.n_k_tiles;
    const int by_ = local / nx, bx_ = local - by_ * nx;
    if (by_ >= p.slices) return;                                   // (an item's range is padded to whole rows of tiles only: never taken)
    wgrad_pipe_body<BCO, BK, WIDE, WN>(p, bx_, by_);
#endif
}

template <typename K>
void raise_lds(K kern, size_t lds) { din_raise_lds(reinterpret_cast<const void*>(kern), lds); }

}  // namespace

size_t wgrad_pipe_lds_bytes(int, int) { return 4 * 2 * 32 * 512; }

int launch_wgrad_pipe(const WgradK& k, int bco, int bk, dim3 grid, hipStream_t st) {
    DIN_REQUIRE(bk == 256 && (bco == 128 || bco == 192 || bco == 256), "wgrad pipe kernel: tile %dx%d not instantiated", bco, bk);
    const size_t lds = wgrad_pipe_lds_bytes(bco, bk);
    // wave grid of the workgroup: 2 x 8 (sixteen waves, four per SIMD, (BCO/2) x 32 wave tiles; default: +6 % on the dominant kernel inside
    // the training step over 2 x 4, whose 60 % fewer fragment reads do not matter), DIN_WGRAD_PIPE_WAVES=8: 2 x 4, =4: 2 x 2 (-13 %)
    const char* wv = DIN_OPT("DIN_WGRAD_PIPE_WAVES");
    const int waves = wv ? atoi(wv) : 16;
    const bool four = waves == 4 && bco <= 192, sixteen = waves == 16;   // (the four-wave 256-row tile spills: scratch traffic would break the vmcnt count)
    auto launch = [&](auto kern, int threads) {
        raise_lds(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, k);
    };
    const bool wide = k.OW >= 32;                                  // a feature-map row holds at least one 32-pixel stage
    if (sixteen) {
        if (bco == 128) { if (wide) launch(conv_wgrad_pipe_kernel<128, 256, true, 8>, 1024); else launch(conv_wgrad_pipe_kernel<128, 256, false, 8>, 1024); }
        else if (bco == 192) { if (wide) launch(conv_wgrad_pipe_kernel<192, 256, true, 8>, 1024); else launch(conv_wgrad_pipe_kernel<192, 256, false, 8>, 1024); }
        else { if (wide) launch(conv_wgrad_pipe_kernel<256, 256, true, 8>, 1024); else launch(conv_wgrad_pipe_kernel<256, 256, false, 8>, 1024); }
        return DIN_OK;
    }
    if (four) {
        if (bco == 128) { if (wide) launch(conv_wgrad_pipe_kernel<128, 256, true, 2>, 256); else launch(conv_wgrad_pipe_kernel<128, 256, false, 2>, 256); }
        else if (bco == 192) { if (wide) launch(conv_wgrad_pipe_kernel<192, 256, true, 2>, 256); else launch(conv_wgrad_pipe_kernel<192, 256, false, 2>, 256); }
        else { if (wide) launch(conv_wgrad_pipe_kernel<256, 256, true, 2>, 256); else launch(conv_wgrad_pipe_kernel<256, 256, false, 2>, 256); }
        return DIN_OK;
    }
    if (bco == 128) { if (wide) launch(conv_wgrad_pipe_kernel<128, 256, true>, 512); else launch(conv_wgrad_pipe_kernel<128, 256, false>, 512); }
    else if (bco == 192) { if (wide) launch(conv_wgrad_pipe_kernel<192, 256, true>, 512); else launch(conv_wgrad_pipe_kernel<192, 256, false>, 512); }
    else { if (wide) launch(conv_wgrad_pipe_kernel<256, 256, true>, 512); else launch(conv_wgrad_pipe_kernel<256, 256, false>, 512); }
    return DIN_OK;
}

int launch_wgrad_pipe_group(const WgradGroupK& g, int bco, bool wide, hipStream_t st) {
    DIN_REQUIRE(g.n >= 1 && g.n <= WGRAD_GROUP_MAX && (bco == 128 || bco == 192 || bco == 256), "wgrad pipe group: %d items, tile %d", g.n, bco);
    const size_t lds = wgrad_pipe_lds_bytes(bco, 256);
    const dim3 grid(g.first[g.n]);
    auto launch = [&](auto kern) {
        raise_lds(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, g);
    };
    if (bco == 128) { if (wide) launch(conv_wgrad_pipe_group_kernel<128, 256, true, 8>); else launch(conv_wgrad_pipe_group_kernel<128, 256, false, 8>); }
    else if (bco == 192) { if (wide) launch(conv_wgrad_pipe_group_kernel<192, 256, true, 8>); else launch(conv_wgrad_pipe_group_kernel<192, 256, false, 8>); }
    else { if (wide) launch(conv_wgrad_pipe_group_kernel<256, 256, true, 8>); else launch(conv_wgrad_pipe_group_kernel<256, 256, false, 8>); }
    return DIN_OK;
}

}  // namespace din_wgrad
