// Standalone harness of csrc/conv_line.hip (the loader / consumer tap-line kernel): checks it against a naive fp32 kernel on random
// operands and times it in steady state.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o line_probe line_probe.hip
#define DIN_LINE_KNOCK 1
#include "conv_line32.hip"
#include "conv_line64.hip"
#include "conv_line_bar.hip"
#include <vector>
#include <functional>
#include <random>
#include <cstring>
#include <cmath>
void din_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
using din_line::LineK;

__global__ void ref_kernel(LineK p, float* out) {        // one thread per (position, channel)
    const long long total = (long long)p.Q * p.Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i / p.Cout), co = (int)(i - (long long)q * p.Cout);
        const int b = q % p.L;
        const bf16_t* x = reinterpret_cast<const bf16_t*>(p.in);
        const bf16_t* w = reinterpret_cast<const bf16_t*>(p.w) + (long long)co * p.wld * 8;
        float s = 0.f;
        for (int t = 0; t < p.taps; ++t) {
            const int sh = p.shift0 + t * p.dshift;
            if (b + sh < 0 || b + sh >= p.L) continue;
            const bf16_t* xp = x + (long long)din_line::walk_pixel(p, q + sh) * p.ldi + p.cioff;
            const bf16_t* wp = w + t * p.cpt * 8;
            for (int c = 0; c < p.cpt * 8; ++c) s += bf16_to_f32(xp[c]) * bf16_to_f32(wp[c]);
        }
        const int pix = din_line::walk_pixel(p, q);
        if (p.flags & DIN_CONV_BIAS) s += p.bias[co];
        if (p.flags & DIN_CONV_RELU) s = fmaxf(s, 0.f);
        if (p.flags & DIN_CONV_MASK) { if (!(bf16_to_f32(reinterpret_cast<const bf16_t*>(p.mask)[(long long)pix * p.ldm + p.moff + co]) > 0.f)) s = 0.f; }
        out[(long long)pix * p.Cout + co] = s;
    }
}
__global__ void cmp_kernel(const bf16_t* got, int ldo, int cooff, const float* ref, int Q, int C, float* maxerr, int* nbad) {
    const long long total = (long long)Q * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i / C), c = (int)(i - (long long)q * C);
        const float g = bf16_to_f32(got[(long long)q * ldo + cooff + c]), r = ref[i];
        const float e = fabsf(g - r), tol = 0.02f + 0.01f * fabsf(r);
        if (!(e <= tol)) atomicAdd(nbad, 1);
        atomicMax(reinterpret_cast<int*>(maxerr), __float_as_int(e));
    }
}
static bf16_t h_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

struct Case { const char* name; int NB, H, W, Cin, Cout; bool along_x; int taps, shift0, dshift, flags, bn; };

template <typename K> static void fill(K& k6, const LineK& k) {
    k6.in = k.in; k6.w = k.w; k6.out = k.out; k6.bias = k.bias; k6.mask = k.mask; k6.err = k.err; k6.prof = nullptr;
    k6.L = k.L; k6.OUTER = k.OUTER; k6.HW = k.HW; k6.strideA = k.strideA; k6.strideB = k.strideB; k6.Q = k.Q;
    k6.ldi = k.ldi; k6.cioff = k.cioff; k6.ldo = k.ldo; k6.cooff = k.cooff; k6.ldm = k.ldm; k6.moff = k.moff;
    k6.Cout = k.Cout; k6.cpt = k.cpt; k6.ncb = (k.cpt + 7) / 8; k6.taps = k.taps; k6.shift0 = k.shift0; k6.dshift = k.dshift; k6.wld = k.wld;
    k6.flags = k.flags; k6.in_bytes = k.in_bytes; k6.w_bytes = k.w_bytes;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 3000;
    const Case cases[] = {
        {"6e 7x1 192->192 fwd", 96, 43, 78, 192, 192, false, 7, -3, 1, DIN_CONV_BIAS | DIN_CONV_RELU, 192},
        {"6e 1x7 192->192 fwd", 96, 43, 78, 192, 192, true, 7, -3, 1, DIN_CONV_BIAS | DIN_CONV_RELU, 192},
        {"6e 7x1 192->192 dgrad+mask", 96, 43, 78, 192, 192, false, 7, 3, -1, DIN_CONV_MASK, 192},
        {"6c 1x7 160->192 fwd", 96, 43, 78, 160, 192, true, 7, -3, 1, DIN_CONV_BIAS | DIN_CONV_RELU, 192},
        {"6c 7x1 160->160 fwd", 96, 43, 78, 160, 160, false, 7, -3, 1, DIN_CONV_BIAS | DIN_CONV_RELU, 160},
        {"6c 1x7 160->160 dgrad+mask", 96, 43, 78, 160, 160, true, 7, 3, -1, DIN_CONV_MASK, 160},
        {"6b 7x1 128->128 fwd", 96, 43, 78, 128, 128, false, 7, -3, 1, DIN_CONV_BIAS | DIN_CONV_RELU, 128},
        {"6b 1x7 128->128 dgrad+mask", 96, 43, 78, 128, 128, true, 7, 3, -1, DIN_CONV_MASK, 128},
        {"small 7x1 192->192 (3 img)", 3, 43, 78, 192, 192, false, 7, -3, 1, DIN_CONV_BIAS | DIN_CONV_RELU, 192},
        {"ZERO pixels: 6e 7x1 192->192 fwd", 96, 43, 78, 192, 192, false, 7, -3, 1, DIN_CONV_BIAS | DIN_CONV_RELU, 192},
        {"HALF-ZERO pixels (post-ReLU-like): 6e 7x1 fwd", 96, 43, 78, 192, 192, false, 7, -3, 1, DIN_CONV_BIAS | DIN_CONV_RELU, 192},
    };
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, reps %d\n", prop.name, ncu, reps);
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    int* d_err; hipMalloc(&d_err, 4);
    for (const Case& c : cases) {
        const int Q = c.NB * c.H * c.W, cpt = c.Cin / 8, nk = (c.taps * cpt + 7) / 8, wld = nk * 8;
        const size_t xin = (size_t)Q * c.Cin, wel = (size_t)256 * wld * 8, oel = (size_t)Q * c.Cout;
        std::vector<bf16_t> hx(xin), hw(wel, 0), hm(oel);
        std::vector<float> hb(c.Cout);
        for (auto& v : hx) v = h_bf16(nd(rng));
        if (c.name[0] == 'Z') for (auto& v : hx) v = 0;
        if (c.name[0] == 'H') for (auto& v : hx) if ((int16_t)v < 0) v = 0;          // ReLU of N(0,1): half zeros
        for (int r = 0; r < c.Cout; ++r) for (int k = 0; k < c.taps * cpt * 8; ++k) hw[(size_t)r * wld * 8 + k] = h_bf16(0.05f * nd(rng));
        for (auto& v : hm) v = h_bf16(nd(rng));
        for (auto& v : hb) v = 0.1f * nd(rng);
        bf16_t *dx, *dw, *dm, *dout; float *db, *dref, *dmax; int* dbad;
        hipMalloc(&dx, xin * 2); hipMalloc(&dw, wel * 2); hipMalloc(&dm, oel * 2); hipMalloc(&dout, oel * 2); hipMalloc(&db, c.Cout * 4);
        hipMalloc(&dref, oel * 4); hipMalloc(&dmax, 4); hipMalloc(&dbad, 4);
        hipMemcpy(dx, hx.data(), xin * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), wel * 2, hipMemcpyHostToDevice);
        hipMemcpy(dm, hm.data(), oel * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), c.Cout * 4, hipMemcpyHostToDevice);
        LineK k{};
        k.in = dx; k.w = dw; k.out = dout; k.bias = db; k.mask = dm; k.err = d_err;
        k.L = c.along_x ? c.W : c.H; k.OUTER = c.along_x ? c.H : c.W; k.HW = c.H * c.W;
        k.strideA = c.along_x ? c.W : 1; k.strideB = c.along_x ? 1 : c.W; k.Q = Q;
        k.ldi = c.Cin; k.cioff = 0; k.ldo = c.Cout; k.cooff = 0; k.ldm = c.Cout; k.moff = 0;
        k.Cout = c.Cout; k.cpt = cpt; k.ncb = (cpt + 7) / 8; k.taps = c.taps; k.shift0 = c.shift0; k.dshift = c.dshift; k.wld = wld;
        k.flags = c.flags; k.in_bytes = (long long)xin * 2; k.w_bytes = (long long)wel * 2;
        k.n_co_tiles = 1; k.ntiles = (Q + 255) / 256;
        hipLaunchKernelGGL(ref_kernel, dim3(4096), dim3(256), 0, 0, k, dref);
        din_line64::LineK k6{}; fill(k6, k);
        din_lineb::LineK kb{}; fill(kb, k);
        struct Runner { const char* name; std::function<int()> go; };
        const Runner runners[] = {
            {"gen 1 (flags, 64-channel stages)", [&] { return c.bn == 160 ? -1 : din_line64::launch_line(k6, c.bn, ncu, 0, 0); }},
            {"gen 3 (barrier) variant 0", [&] { return din_lineb::launch_lineb(kb, c.bn, ncu, 0, 0); }},
            {"gen 3 (barrier) variant 1", [&] { return din_lineb::launch_lineb(kb, c.bn, ncu, 0, 1); }},
            {"gen 3 + stagger s_sleep 1", [&] { return din_lineb::launch_lineb(kb, c.bn, ncu, 0, 2); }},
            {"gen 3 + stagger s_sleep 2", [&] { return din_lineb::launch_lineb(kb, c.bn, ncu, 0, 3); }},
            {"gen 3 + stagger s_sleep 3", [&] { return din_lineb::launch_lineb(kb, c.bn, ncu, 0, 4); }},
        };
        for (const Runner& r : runners) {
            hipMemset(dout, 0xff, oel * 2); hipMemset(dmax, 0, 4); hipMemset(dbad, 0, 4); hipMemset(d_err, 0, 4);
            if (r.go()) continue;
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("%-30s %-34s FAILED: %s\n", c.name, r.name, hipGetErrorString(e)); return 1; }
            hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, 0, dout, c.Cout, 0, dref, Q, c.Cout, dmax, dbad);
            hipDeviceSynchronize();
            float maxerr; int nbad; hipMemcpy(&maxerr, dmax, 4, hipMemcpyDeviceToHost); hipMemcpy(&nbad, dbad, 4, hipMemcpyDeviceToHost);
            const int n = c.NB > 3 ? reps : reps / 4 + 1;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 20; ++i) r.go();
            hipEventRecord(e0);
            for (int i = 0; i < n; ++i) r.go();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / n, tf = 2.0 * Q * c.Cout * c.Cin * c.taps / us / 1e6;
            printf("%-30s %-34s %8.1f us %7.1f TF   max |err| %.4f  bad %d / %zu\n", c.name, r.name, us, tf, maxerr, nbad, oel);
            if (&r == &runners[1] && (c.bn == 192 || c.bn == 128) && c.NB > 3) {           // cycle accounting of the barrier kernel
                const int grid = k.ntiles < ncu ? k.ntiles : ncu;
                uint32_t* dprof; hipMalloc(&dprof, (size_t)grid * 10 * 8 * 4); hipMemset(dprof, 0, (size_t)grid * 10 * 8 * 4);
                din_lineb::LineK kp = kb; kp.prof = dprof;
                for (int i = 0; i < 3; ++i) din_lineb::launch_lineb(kp, c.bn, ncu, 0, 0);
                hipDeviceSynchronize();
                std::vector<uint32_t> hp((size_t)grid * 10 * 8);
                hipMemcpy(hp.data(), dprof, hp.size() * 4, hipMemcpyDeviceToHost);
                double cs[5] = {0, 0, 0, 0, 0}, ls[5] = {0, 0, 0, 0, 0};
                for (int g = 0; g < grid; ++g) for (int w = 0; w < 10; ++w) for (int f = 0; f < 5; ++f) (w < 8 ? cs : ls)[f] += hp[((size_t)g * 10 + w) * 8 + f];
                for (int f = 0; f < 5; ++f) { cs[f] /= grid * 8.0; ls[f] /= grid * 2.0; }
                printf("    consumer wave (mean): %.0f cycles total, %.0f at barriers (%.0f of it at each tile's first), %.0f epilogue\n", cs[0], cs[1], cs[3], cs[2]);
                printf("    loader wave   (mean): %.0f cycles total, %.0f waiting for landings, %.0f at barriers, %.0f issuing, %.0f stages\n", ls[0], ls[1], ls[2], ls[3], ls[4]);
                hipFree(dprof);
            }
            fflush(stdout);
        }
        hipFree(dx); hipFree(dw); hipFree(dm); hipFree(dout); hipFree(db); hipFree(dref); hipFree(dmax); hipFree(dbad);
    }
    return 0;
}
