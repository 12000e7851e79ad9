// Rows D2-D4: fused Dynamic Relation + Dynamic Walk (infer_module/dynamic_infer_module.py:184-282, 344-404).
//
// One workgroup owns (clip b, 64-channel chunk).  The zero-padded (T+2pt) x (N+2pl) x 64 actor-feature tile of the
// clip is staged ONCE into LDS with coalesced 256-byte rows; every wave then walks actor positions with lane ==
// channel, so the k2 x 4 bilinear corner fetches are conflict-free LDS reads instead of the reference's four
// materialised [B,T,N,k2,C] torch.gather tensors.  The relation softmax over k2 and the floor/clamp/coefficients
// are computed from the fused p_conv/scale_conv prediction `pred` (produced by the MFMA conv kernel).
// No MFMA here: the gather is sparse per actor.
//
// Bit-exactness: floor/clamp corner indices are integer decisions and reproduce the reference's fp32 expression
// order (pos = (pos_0 + pos_k) + offset; lt = floor(pos); clamp) exactly; compile with -ffp-contract=off.
#include "din_common.h"

namespace {

constexpr int CH = 64;            // channels per workgroup (one lane each)
constexpr int WALK_THREADS = 1024;   // 16 waves: the T x N positions of a clip are walked 16 at a time (the kernel is latency-bound: one
                                     // workgroup per (clip, 64-channel chunk), a few dozen positions, k2 dependent steps each)
constexpr int MAXK2 = 49;         // up to 7x7 ST kernels

struct WalkK {
    const float* x; const float* pred; float* z; float* a; int32_t* idx; float* mad;
    const float* gz; float* dx; float* scratch;          // backward only
    int cp, b, t, n, c, kh, kw, ratio, scale_factor;
    int pt, pl, hp, wp, k2, ky0, kx0;
    int plain;                        // 1: no walk -- S_k is the feature AT lattice point k (plain_infer_ratio / the relation half of parallel_infer,
                                      // dynamic_infer_module.py:154-181,285-298); offsets are ignored and get no gradient
    int ihy, ihx, phy, phx;           // clamp maxima of the corner indices / of the sampling position; -1 = the padded grid's (hp-1, wpc-1), which
                                      // is what dynamic_infer_ratio uses (:216-226); parallel_infer clamps with person_mat_shape instead (:307-317)
    int ngroups;                      // backward only: position groups per (clip, channel chunk); a wave serves positions grp*W + w, + ngroups*W, ...
    const int32_t* n_per_clip;        // optional [b]: clip i is a T x n_per_clip[i] grid stored in the first columns of its T x n slab
                                      // (Dynamic_collective, infer_model.py:1286-1293); columns beyond it are zero padding
};

// actors of clip b (1..p.n)
__device__ __forceinline__ int clip_n(const WalkK& p, int b) {
    if (!p.n_per_clip) return p.n;
    const int v = p.n_per_clip[b];
    return v < 1 ? 1 : (v > p.n ? p.n : v);
}

struct Corner { int ly, ry, lx, rx; float py, px, py0, px0; };

// wpc: padded width of THIS clip's grid (n_b + 2 pl): the x clamp range; the LDS tile keeps the row pitch p.wp of the widest clip
__device__ __forceinline__ Corner corners(const WalkK& p, int wpc, int tt, int nn, int k, float oy, float ox) {
    const int r = k / p.kw, s = k - r * p.kw;
    const float base_y = (float)(p.pt + tt + p.ky0 + r * p.ratio);     // pos_0 + pos_k : exact small integers
    const float base_x = (float)(p.pl + nn + p.kx0 + s * p.ratio);
    Corner c;
    c.py0 = __fadd_rn(base_y, oy);
    c.px0 = __fadd_rn(base_x, ox);
    const float fy = floorf(c.py0), fx = floorf(c.px0);
    const float hy = (float)(p.ihy >= 0 ? p.ihy : p.hp - 1), hx = (float)(p.ihx >= 0 ? p.ihx : wpc - 1);
    c.ly = (int)fminf(fmaxf(fy, 0.f), hy);
    c.ry = (int)fminf(fmaxf(fy + 1.f, 0.f), hy);
    c.lx = (int)fminf(fmaxf(fx, 0.f), hx);
    c.rx = (int)fminf(fmaxf(fx + 1.f, 0.f), hx);
    c.py = fminf(fmaxf(c.py0, 0.f), p.phy >= 0 ? (float)p.phy : hy);
    c.px = fminf(fmaxf(c.px0, 0.f), p.phx >= 0 ? (float)p.phx : hx);
    return c;
}
__device__ __forceinline__ float coef(float p, int c) { return __fsub_rn(1.f, fabsf(__fsub_rn(p, (float)c))); }

// stage the zero-padded tile of clip b / channel chunk c0 into LDS: tile[(y*wp + x)*CH + lane]
__device__ __forceinline__ void stage_tile(const WalkK& p, const float* __restrict__ src, int b, int c0, float* tile) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool cok = c0 + lane < p.c;
    const int nb = clip_n(p, b);
    for (int cell = w; cell < p.hp * p.wp; cell += (int)blockDim.x / 64) {
        int y = cell / p.wp, xx = cell - y * p.wp;
        int tt = y - p.pt, nn = xx - p.pl;
        float v = 0.f;
        if (cok && tt >= 0 && tt < p.t && nn >= 0 && nn < nb)
            v = src[((int64_t)(b * p.t + tt) * p.n + nn) * p.c + c0 + lane];
        tile[cell * CH + lane] = v;
    }
}

// softmax over k2 of the relation logits of every position of clip b -> LDS a_s[pos*k2 + k]
__device__ __forceinline__ void stage_relation(const WalkK& p, int b, float* a_s) {
    for (int pos = threadIdx.x; pos < p.t * p.n; pos += WALK_THREADS) {
        const float* pr = p.pred + ((int64_t)b * p.t * p.n + pos) * p.cp;
        if (p.scale_factor) {
            float mx = -INFINITY;
            for (int k = 0; k < p.k2; ++k) mx = fmaxf(mx, pr[2 * p.k2 + k]);
            float sum = 0.f;
            for (int k = 0; k < p.k2; ++k) { float e = expf(pr[2 * p.k2 + k] - mx); a_s[pos * p.k2 + k] = e; sum += e; }
            for (int k = 0; k < p.k2; ++k) a_s[pos * p.k2 + k] = a_s[pos * p.k2 + k] / sum;
        } else {
            for (int k = 0; k < p.k2; ++k) a_s[pos * p.k2 + k] = 1.f / (float)p.k2;
        }
    }
}

// offsets (2*k2 per position) of clip b -> LDS off_s[pos*2*k2 + j]: the walk loops then touch global memory only for gz / z
__device__ __forceinline__ void stage_offsets(const WalkK& p, int b, float* off_s) {
    const int per = 2 * p.k2;
    for (int i = threadIdx.x; i < p.t * p.n * per; i += (int)blockDim.x) {
        const int pos = i / per, j = i - pos * per;
        off_s[i] = p.pred[((int64_t)b * p.t * p.n + pos) * p.cp + j];
    }
}

__global__ __launch_bounds__(WALK_THREADS) void din_walk_fwd_kernel(WalkK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                   // hp*wp*CH
    float* a_s = tile + p.hp * p.wp * CH;                 // t*n*k2
    float* off_s = a_s + p.t * p.n * p.k2;                // t*n*2*k2
    const int nchunks = (p.c + CH - 1) / CH;
    const int b = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks, c0 = chunk * CH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nb = clip_n(p, b), wpc = nb + 2 * p.pl;
    stage_tile(p, p.x, b, c0, tile);
    stage_relation(p, b, a_s);
    stage_offsets(p, b, off_s);
    __syncthreads();
    if (chunk == 0) {
        // saved relation weights and the (bit-exact) integer corners
        for (int i = threadIdx.x; i < p.t * p.n * p.k2; i += WALK_THREADS) {
            int pos = i / p.k2, k = i - pos * p.k2;
            int tt = pos / p.n, nn = pos - tt * p.n;
            const float* pr = off_s + pos * 2 * p.k2;
            Corner c = corners(p, wpc, tt, nn, k, p.plain ? 0.f : pr[k], p.plain ? 0.f : pr[p.k2 + k]);
            if (p.plain) { c.ry = c.ly; c.rx = c.lx; }
            int64_t o = ((int64_t)b * p.t * p.n + pos) * p.k2 + k;
            p.a[o] = a_s[i];
            if (p.idx) { p.idx[o * 4 + 0] = c.ly; p.idx[o * 4 + 1] = c.ry; p.idx[o * 4 + 2] = c.lx; p.idx[o * 4 + 3] = c.rx; }
        }
    }
    const bool cok = c0 + lane < p.c;
    for (int pos = w; pos < p.t * p.n; pos += WALK_THREADS / 64) {
        int tt = pos / p.n, nn = pos - tt * p.n;
        const float* pr = off_s + pos * 2 * p.k2;
        float zacc = 0.f;
        if (nn >= nb) {                                    // padding actor of a shorter clip: defined output (0), no walk
            if (cok) p.z[((int64_t)b * p.t * p.n + pos) * p.c + c0 + lane] = 0.f;
            continue;
        }
        for (int k = 0; k < p.k2; ++k) {
            float sk;
            if (p.plain) {                                 // the feature at the lattice point itself (pos_0 + pos_k: integers inside the padded grid)
                const int r = k / p.kw, s = k - r * p.kw;
                sk = tile[((p.pt + tt + p.ky0 + r * p.ratio) * p.wp + p.pl + nn + p.kx0 + s * p.ratio) * CH + lane];
            } else {
            Corner c = corners(p, wpc, tt, nn, k, pr[k], pr[p.k2 + k]);
            float wy_l = coef(c.py, c.ly), wy_r = coef(c.py, c.ry), wx_l = coef(c.px, c.lx), wx_r = coef(c.px, c.rx);
            float v_lt = tile[(c.ly * p.wp + c.lx) * CH + lane], v_rb = tile[(c.ry * p.wp + c.rx) * CH + lane];
            float v_lb = tile[(c.ry * p.wp + c.lx) * CH + lane], v_rt = tile[(c.ly * p.wp + c.rx) * CH + lane];
            // same association as the reference: lt*coe_lt + rb*coe_rb + lb*coe_lb + rt*coe_rt  (:255-258)
            sk = v_lt * (wy_l * wx_l) + v_rb * (wy_r * wx_r);
            sk = sk + v_lb * (wy_r * wx_l);
            sk = sk + v_rt * (wy_l * wx_r);
            }
            if (p.mad && cok) p.mad[(((int64_t)b * p.t * p.n + pos) * p.k2 + k) * p.c + c0 + lane] = sk;
            zacc += sk * a_s[pos * p.k2 + k];
        }
        if (cok) p.z[((int64_t)b * p.t * p.n + pos) * p.c + c0 + lane] = zacc;
    }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// Backward: one workgroup of WALK_BWD_WAVES waves per (clip, 64-channel chunk, group of WALK_BWD_WAVES positions) -- one position per
// wave, so the k2 dependent steps of a position are the whole critical path and a 4-clip batch still fills the chip (the first version ran
// all T*N positions of a clip chunk in one workgroup: 64 workgroups, 139 us).  Feature gradients of the groups meet in dx through native
// fp32 atomics (dx is zeroed by the call); offset / relation gradients are per-(chunk) partial sums, plain stores.
// GLOBAL_DX: the second LDS tile (dP) does not fit next to P (large T x N grids with wide / dilated kernels: 5x5 with ratio >= 2, 7x7,
// ratio 4 at T = 10): the feature gradient then goes straight to dx with global fp32 atomics (cells outside the clip's grid are dropped
// at the source).  Slower per position, but any grid whose forward tile fits can also be trained.
constexpr int WALK_BWD_WAVES = 4;
template <bool GLOBAL_DX>
__global__ __launch_bounds__(WALK_BWD_WAVES * 64) void din_walk_bwd_kernel(WalkK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cells = p.hp * p.wp;
    float* tile = smem;                       // P
    float* dtile = tile + cells * CH;         // dP (LDS mode only)
    float* off_s = dtile + (GLOBAL_DX ? 0 : cells * CH);        // t*n*2*k2 offsets
    float* a_s = off_s + p.t * p.n * 2 * p.k2; // t*n*k2 saved relation weights
    const int nchunks = (p.c + CH - 1) / CH;
    const int ngroups = p.ngroups;
    const int grp = blockIdx.x % ngroups, bc = blockIdx.x / ngroups;
    const int b = bc / nchunks, chunk = bc % nchunks, c0 = chunk * CH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nb = clip_n(p, b), wpc = nb + 2 * p.pl;
    stage_tile(p, p.x, b, c0, tile);
    stage_offsets(p, b, off_s);
    for (int i = threadIdx.x; i < p.t * p.n * p.k2; i += (int)blockDim.x) a_s[i] = p.a[(int64_t)b * p.t * p.n * p.k2 + i];
    if (!GLOBAL_DX)
        for (int i = threadIdx.x; i < cells * CH; i += (int)blockDim.x) dtile[i] = 0.f;
    __syncthreads();
    const bool cok = c0 + lane < p.c;
    // feature-gradient scatter of one corner: LDS tile, or dx itself when the tile does not fit
    auto scatter = [&](int cy, int cx, float v) {
        if (!GLOBAL_DX) { atomicAdd(&dtile[(cy * p.wp + cx) * CH + lane], v); return; }
        const int tt2 = cy - p.pt, nn2 = cx - p.pl;
        if (cok && tt2 >= 0 && tt2 < p.t && nn2 >= 0 && nn2 < nb && v != 0.f)
            atomicAdd(&p.dx[((int64_t)(b * p.t + tt2) * p.n + nn2) * p.c + c0 + lane], v);
    };
    // per-chunk partial sums, plain stores (each (chunk, position, k) is written exactly once): scratch[chunk][ d_off [b,t,n,2k2] | d_a [b,t,n,k2] ];
    // din_walk_bwd_finish_kernel adds the chunks in a fixed order
    float* d_off = p.scratch + (int64_t)chunk * p.b * p.t * p.n * 3 * p.k2;
    float* d_a = d_off + (int64_t)p.b * p.t * p.n * 2 * p.k2;
    for (int pos = grp * WALK_BWD_WAVES + w; pos < p.t * p.n; pos += ngroups * WALK_BWD_WAVES)
    if (pos % p.n >= nb) {                                // padding actor: no gradient (the finish kernel sums every chunk's slot)
        const int64_t gpos = (int64_t)b * p.t * p.n + pos;
        for (int j = lane; j < 3 * p.k2; j += 64) {
            if (j < 2 * p.k2) d_off[gpos * 2 * p.k2 + j] = 0.f; else d_a[gpos * p.k2 + (j - 2 * p.k2)] = 0.f;
        }
    } else {
        int tt = pos / p.n, nn = pos - tt * p.n;
        const int64_t gpos = (int64_t)b * p.t * p.n + pos;
        const float* pr = off_s + pos * 2 * p.k2;
        const float g = cok ? p.gz[gpos * p.c + c0 + lane] : 0.f;
        for (int k = 0; k < p.k2; ++k) {
            const float ak = a_s[pos * p.k2 + k];
            if (p.plain) {
                const int r = k / p.kw, s2 = k - r * p.kw;
                const int cy = p.pt + tt + p.ky0 + r * p.ratio, cx = p.pl + nn + p.kx0 + s2 * p.ratio;
                const float sk = tile[(cy * p.wp + cx) * CH + lane];
                scatter(cy, cx, ak * g);
                const float d_s = wave_sum(g * sk);
                if (lane == 0) { d_off[gpos * 2 * p.k2 + k] = 0.f; d_off[gpos * 2 * p.k2 + p.k2 + k] = 0.f; d_a[gpos * p.k2 + k] = d_s; }
                continue;
            }
            Corner c = corners(p, wpc, tt, nn, k, pr[k], pr[p.k2 + k]);
            float wy_l = coef(c.py, c.ly), wy_r = coef(c.py, c.ry), wx_l = coef(c.px, c.lx), wx_r = coef(c.px, c.rx);
            const int i_lt = (c.ly * p.wp + c.lx) * CH + lane, i_rb = (c.ry * p.wp + c.rx) * CH + lane;
            const int i_lb = (c.ry * p.wp + c.lx) * CH + lane, i_rt = (c.ly * p.wp + c.rx) * CH + lane;
            float v_lt = tile[i_lt], v_rb = tile[i_rb], v_lb = tile[i_lb], v_rt = tile[i_rt];
            float sk = v_lt * (wy_l * wx_l) + v_rb * (wy_r * wx_r) + v_lb * (wy_r * wx_l) + v_rt * (wy_l * wx_r);
            // feature gradient: scatter-add a_k * w_corner * gz into the padded tile (LDS atomics; waves may collide)
            const float ag = ak * g;
            scatter(c.ly, c.lx, ag * (wy_l * wx_l));
            scatter(c.ry, c.rx, ag * (wy_r * wx_r));
            scatter(c.ry, c.lx, ag * (wy_r * wx_l));
            scatter(c.ly, c.rx, ag * (wy_l * wx_r));
            // per-corner <gz, P_corner> and <gz, S_k> over this channel chunk
            float d_s = wave_sum(g * sk);
            float d_lt = wave_sum(g * v_lt), d_rb = wave_sum(g * v_rb), d_lb = wave_sum(g * v_lb), d_rt = wave_sum(g * v_rt);
            if (lane == 0) {
                // d/d py of (1-|py-cy|) = -sign(py-cy); clamp passes gradient on the closed interval (Q8, Q9)
                const float my = (c.py0 >= 0.f && c.py0 <= (float)(p.phy >= 0 ? p.phy : (p.ihy >= 0 ? p.ihy : p.hp - 1))) ? 1.f : 0.f;
                const float mx = (c.px0 >= 0.f && c.px0 <= (float)(p.phx >= 0 ? p.phx : (p.ihx >= 0 ? p.ihx : wpc - 1))) ? 1.f : 0.f;
                const float sy_l = -sgn(c.py - (float)c.ly), sy_r = -sgn(c.py - (float)c.ry);
                const float sx_l = -sgn(c.px - (float)c.lx), sx_r = -sgn(c.px - (float)c.rx);
                float doy = d_lt * sy_l * wx_l + d_rb * sy_r * wx_r + d_lb * sy_r * wx_l + d_rt * sy_l * wx_r;
                float dox = d_lt * wy_l * sx_l + d_rb * wy_r * sx_r + d_lb * wy_r * sx_l + d_rt * wy_l * sx_r;
                d_off[gpos * 2 * p.k2 + k] = my * ak * doy;
                d_off[gpos * 2 * p.k2 + p.k2 + k] = mx * ak * dox;
                d_a[gpos * p.k2 + k] = d_s;
            }
        }
    }
    if (GLOBAL_DX) return;
    __syncthreads();
    // un-pad and add this group's share: dx[b,t,n,c] += dP[pt+t][pl+n][c]
    for (int q = w; q < p.t * p.n; q += WALK_BWD_WAVES) {
        int tt = q / p.n, nn = q - tt * p.n;
        if (nn >= nb) continue;                            // cells beyond the clip's grid are zero padding: their gradient is dropped
        const float v = dtile[((p.pt + tt) * p.wp + p.pl + nn) * CH + lane];
        if (cok && v != 0.f) atomicAdd(&p.dx[((int64_t)b * p.t * p.n + q) * p.c + c0 + lane], v);
    }
}

// dpred[.., 0:2k2] = d offset ; dpred[.., 2k2:3k2] = a_k (dA_k - sum_j a_j dA_j)
__global__ void din_walk_bwd_finish_kernel(const float* __restrict__ scratch, const float* __restrict__ a, float* __restrict__ dpred,
                                           int64_t positions, int k2, int cp, int scale_factor, int nchunks) {
    // one wave per position: lane j < 3*k2 adds the chunks of its value (fixed order), the softmax backward needs the dot over k
    const int64_t pos = blockIdx.x;
    const int j = threadIdx.x;
    const int64_t cstride = positions * 3 * k2;
    float v = 0.f;
    if (j < 2 * k2) {
        for (int c = 0; c < nchunks; ++c) v += scratch[c * cstride + pos * 2 * k2 + j];
        dpred[pos * cp + j] = v;
    } else if (j < 3 * k2) {
        for (int c = 0; c < nchunks; ++c) v += scratch[c * cstride + positions * 2 * k2 + pos * k2 + (j - 2 * k2)];
    }
    if (scale_factor) {
        const bool is_a = j >= 2 * k2 && j < 3 * k2;
        const float ak = is_a ? a[pos * k2 + (j - 2 * k2)] : 0.f;
        float dot = is_a ? ak * v : 0.f;
        __shared__ float part[4];
        dot = wave_sum(dot);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = dot;
        __syncthreads();
        float tot = 0.f;
        for (int q = 0; q < (int)blockDim.x / 64; ++q) tot += part[q];
        if (is_a) dpred[pos * cp + j] = ak * (v - tot);
    }
}

int fill(WalkK& p, int cp, int b, int t, int n, int c, int kh, int kw, int ratio, int scale_factor, int plain, const int32_t* clamp,
         const int32_t* n_per_clip) {
    DIN_REQUIRE(b > 0 && t > 0 && n > 0 && c > 0 && kh > 0 && kw > 0 && ratio > 0, "din_walk: bad shape");
    DIN_REQUIRE(kh * kw <= MAXK2, "din_walk: ST kernel larger than %d taps unsupported", MAXK2);
    DIN_REQUIRE(cp >= (scale_factor ? 3 : 2) * kh * kw, "din_walk: pred pixel stride too small");
    p.cp = cp; p.b = b; p.t = t; p.n = n; p.c = c; p.kh = kh; p.kw = kw; p.ratio = ratio; p.scale_factor = scale_factor;
    p.pt = (kh - 1) / 2 * ratio; p.pl = (kw - 1) / 2 * ratio;
    p.hp = t + 2 * p.pt; p.wp = n + 2 * p.pl; p.k2 = kh * kw;
    // lattice start = floor(-((k-1)*ratio) / 2)   (dynamic_infer_module.py:388-389)
    auto fl2 = [](int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };
    p.ky0 = fl2(-((kh - 1) * ratio)); p.kx0 = fl2(-((kw - 1) * ratio));
    p.plain = plain ? 1 : 0;
    p.ihy = p.ihx = p.phy = p.phx = -1;
    p.n_per_clip = n_per_clip;
    if (clamp) {
        DIN_REQUIRE(!n_per_clip, "din_walk: clamp overrides and n_per_clip are mutually exclusive");
        // index maxima can never leave the LDS tile (the reference would index outside its padded map there: undefined)
        p.ihy = clamp[0] < p.hp - 1 ? clamp[0] : p.hp - 1; p.ihx = clamp[1] < p.wp - 1 ? clamp[1] : p.wp - 1;
        p.phy = clamp[2]; p.phx = clamp[3];
        DIN_REQUIRE(p.ihy >= 0 && p.ihx >= 0 && p.phy >= 0 && p.phx >= 0, "din_walk: negative clamp maximum");
    }
    return DIN_OK;
}

}  // namespace

extern "C" {

int din_walk_fwd(const float* x, const float* pred, int cp, int b, int t, int n, int c, int kh, int kw, int ratio,
                 int scale_factor, int plain, const int32_t* clamp, const int32_t* n_per_clip, float* z, float* a, int32_t* idx, float* mad,
                 void* stream) {
    DIN_REQUIRE(x && pred && z && a, "din_walk_fwd: null pointer");
    WalkK p{};
    if (int e = fill(p, cp, b, t, n, c, kh, kw, ratio, scale_factor, plain, clamp, n_per_clip)) return e;
    p.x = x; p.pred = pred; p.z = z; p.a = a; p.idx = idx; p.mad = mad;
    size_t lds = ((size_t)p.hp * p.wp * CH + (size_t)t * n * 3 * p.k2) * sizeof(float);
    DIN_REQUIRE(lds <= 160 * 1024, "din_walk_fwd: T x N grid too large for one LDS tile (%zu bytes)", lds);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(din_walk_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        DIN_FAIL(DIN_E_LAUNCH, "din_walk_fwd: cannot raise the dynamic LDS limit to %zu bytes", lds);
    int nchunks = (c + CH - 1) / CH;
    hipLaunchKernelGGL(din_walk_fwd_kernel, dim3(b * nchunks), dim3(WALK_THREADS), lds, as_stream(stream), p);
    DIN_CHECK_LAUNCH("din_walk_fwd");
    return DIN_OK;
}

int din_walk_bwd(const float* x, const float* pred, int cp, const float* a, const float* gz, int b, int t, int n, int c,
                 int kh, int kw, int ratio, int scale_factor, int plain, const int32_t* clamp, const int32_t* n_per_clip, float* dx,
                 float* dpred, float* scratch, void* stream) {
    DIN_REQUIRE(x && pred && a && gz && dx && dpred && scratch, "din_walk_bwd: null pointer");
    WalkK p{};
    if (int e = fill(p, cp, b, t, n, c, kh, kw, ratio, scale_factor, plain, clamp, n_per_clip)) return e;
    p.x = x; p.pred = pred; p.a = const_cast<float*>(a); p.gz = gz; p.dx = dx; p.scratch = scratch;
    hipStream_t st = as_stream(stream);
    int64_t positions = (int64_t)b * t * n;
    const size_t tile_b = (size_t)p.hp * p.wp * CH * sizeof(float), tab_b = (size_t)t * n * 3 * p.k2 * sizeof(float);
    // feature gradient scattered straight into dx (fp32 L2 atomics): measured 320 -> 118 us on 32 clips x 36 positions against the LDS dP tile,
    // whose ds_add_f32 traffic alone cost 230 us (a knock-out of the LDS atomics: 95 us).  DIN_WALK_BWD_GLOBAL=0 keeps the LDS tile when it fits.
    const char* gx = DIN_OPT("DIN_WALK_BWD_GLOBAL");
    const bool global_dx = 2 * tile_b + tab_b > 160 * 1024 || !(gx && atoi(gx) == 0);
    const size_t lds = (global_dx ? 1 : 2) * tile_b + tab_b;
    DIN_REQUIRE(lds <= 160 * 1024, "din_walk_bwd: T x N grid too large for one LDS tile (%zu bytes)", lds);
    auto raise = [&](const void* fn) -> int {
        if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            DIN_FAIL(DIN_E_LAUNCH, "din_walk_bwd: cannot raise the dynamic LDS limit to %zu bytes", lds);
        return DIN_OK;
    };
    if (int e = raise(global_dx ? reinterpret_cast<const void*>(din_walk_bwd_kernel<true>) : reinterpret_cast<const void*>(din_walk_bwd_kernel<false>))) return e;
    if (hipMemsetAsync(dx, 0, sizeof(float) * positions * c, st) != hipSuccess) DIN_FAIL(DIN_E_LAUNCH, "din_walk_bwd: memset");
    int nchunks = (c + CH - 1) / CH;
    // every workgroup stages the clip's whole padded tile: as few position groups as still fill the chip (~4 workgroups per CU); one wave per
    // position (the round-1 launch) cost 298 us on 32 clips x 36 positions, almost all of it re-staging.  DIN_WALK_BWD_GROUPS overrides.
    const int full = (t * n + WALK_BWD_WAVES - 1) / WALK_BWD_WAVES;
    int ngroups = (1024 + b * nchunks - 1) / (b * nchunks);
    { const char* gv = DIN_OPT("DIN_WALK_BWD_GROUPS"); if (gv && atoi(gv) > 0) ngroups = atoi(gv); }
    if (ngroups > full) ngroups = full;
    p.ngroups = ngroups;
    if (global_dx) hipLaunchKernelGGL(din_walk_bwd_kernel<true>, dim3(b * nchunks * ngroups), dim3(WALK_BWD_WAVES * 64), lds, st, p);
    else hipLaunchKernelGGL(din_walk_bwd_kernel<false>, dim3(b * nchunks * ngroups), dim3(WALK_BWD_WAVES * 64), lds, st, p);
    DIN_CHECK_LAUNCH("din_walk_bwd");
    const int fin_threads = (3 * p.k2 + 63) / 64 * 64;
    hipLaunchKernelGGL(din_walk_bwd_finish_kernel, dim3((unsigned)positions), dim3(fin_threads), 0, st, scratch, a, dpred,
                       positions, p.k2, cp, scale_factor, nchunks);
    DIN_CHECK_LAUNCH("din_walk_bwd_finish");
    return DIN_OK;
}

}  // extern "C"
