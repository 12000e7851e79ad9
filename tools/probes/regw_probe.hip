// harness of conv1x1_regw.hip: checks against a naive fp32 kernel, times in steady state.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o regw_probe regw_probe.hip
#include "conv1x1_regw.hip"
#include <vector>
#include <random>
#include <cstring>
#include <cmath>
void din_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
using din_regw::RegwK;
__global__ void ref_kernel(RegwK p, int K, float* out) {
    const long long total = (long long)p.M * p.Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / p.Cout), co = (int)(i - (long long)m * p.Cout);
        const bf16_t* x = reinterpret_cast<const bf16_t*>(p.in) + (long long)m * p.ldi + p.cioff;
        const bf16_t* w = reinterpret_cast<const bf16_t*>(p.w) + (long long)co * p.wld * 8;
        float s = (p.flags & DIN_CONV_BIAS) ? p.bias[co] : 0.f;
        for (int c = 0; c < K; ++c) s += bf16_to_f32(x[c]) * bf16_to_f32(w[c]);
        if (p.flags & DIN_CONV_RELU) s = fmaxf(s, 0.f);
        out[i] = s;
    }
}
__global__ void cmp_kernel(const bf16_t* got, int ldo, int cooff, const float* ref, int M, int C, float* maxerr, int* nbad) {
    const long long total = (long long)M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i / C), c = (int)(i - (long long)q * C);
        const float gv = bf16_to_f32(got[(long long)q * ldo + cooff + c]), r = ref[i];
        const float e = fabsf(gv - r), tol = 0.02f + 0.01f * fabsf(r);
        if (!(e <= tol)) atomicAdd(nbad, 1);
        atomicMax(reinterpret_cast<int*>(maxerr), __float_as_int(e));
    }
}
static bf16_t h_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000, dwin = argc > 2 ? atoi(argv[2]) : 4;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    std::mt19937 rng(99); std::normal_distribution<float> nd(0.f, 1.f);
    for (int cas = 0; cas < 6; ++cas) {
        const int NB = cas == 2 ? 3 : 96, M = NB * 43 * 78, K = 768, Cout = cas == 1 ? 160 : cas == 3 ? 576 : cas == 4 ? 768 : cas == 5 ? 384 : 192, wld = K / 8;
        const int ncls = (Cout + 191) / 192;
        const size_t xin = (size_t)M * K, wel = (size_t)(ncls * 192 + 64) * K, oel = (size_t)M * Cout;
        std::vector<bf16_t> hx(xin), hw(wel, 0); std::vector<float> hb(Cout);
        for (auto& v : hx) { float f = nd(rng); v = h_bf16(f > 0.f ? f : 0.f); }                 // post-ReLU-like input: half zeros
        for (int r = 0; r < Cout; ++r) for (int k = 0; k < K; ++k) hw[(size_t)r * K + k] = h_bf16(0.05f * nd(rng));
        for (auto& v : hb) v = 0.1f * nd(rng);
        bf16_t *dx, *dw, *dout; float *db, *dref, *dmax; int* dbad;
        hipMalloc(&dx, xin * 2); hipMalloc(&dw, wel * 2); hipMalloc(&dout, oel * 2); hipMalloc(&db, Cout * 4); hipMalloc(&dref, oel * 4); hipMalloc(&dmax, 4); hipMalloc(&dbad, 4);
        hipMemcpy(dx, hx.data(), xin * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), wel * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice);
        hipMemset(dout, 0xff, oel * 2); hipMemset(dmax, 0, 4); hipMemset(dbad, 0, 4);
        RegwK k{};
        k.in = dx; k.w = dw; k.out = dout; k.bias = db; k.M = M; k.ldi = K; k.cioff = 0; k.ldo = Cout; k.cooff = 0; k.Cout = Cout; k.wld = wld;
        k.flags = DIN_CONV_BIAS | DIN_CONV_RELU; k.co_base = 0; k.ncls = ncls; k.in_bytes = (long long)xin * 2; k.w_bytes = (long long)wel * 2;
        din_regw::launch_regw(k, ncu, 0, dwin);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(e)); return 1; }
        hipLaunchKernelGGL(ref_kernel, dim3(4096), dim3(256), 0, 0, k, K, dref);
        hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, 0, dout, Cout, 0, dref, M, Cout, dmax, dbad);
        hipDeviceSynchronize();
        float maxerr; int nbad; hipMemcpy(&maxerr, dmax, 4, hipMemcpyDeviceToHost); hipMemcpy(&nbad, dbad, 4, hipMemcpyDeviceToHost);
        const int n = NB > 3 ? reps : reps / 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 20; ++i) din_regw::launch_regw(k, ncu, 0, dwin);
        hipEventRecord(e0);
        for (int i = 0; i < n; ++i) din_regw::launch_regw(k, ncu, 0, dwin);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / n;
        printf("1x1 %d -> %d, %d pixels: %8.1f us  %7.1f TF  %5.2f TB/s (input + output bytes)   max |err| %.4f  bad %d / %zu\n", K, Cout, M, us,
               2.0 * M * K * Cout / us / 1e6, ((double)xin * 2 + (double)oel * 2) / us / 1e6, maxerr, nbad, oel);
        if (cas == 0) {
            static const char* kn[] = {"", "no MFMAs", "no transfers after the prologue", "no fragment reads", "no barriers", "a quarter of the epilogue"};
            for (int kx = 1; kx <= 5; ++kx) {
                for (int i = 0; i < 20; ++i) din_regw::launch_regw(k, ncu, 0, 4, kx);
                hipEventRecord(e0);
                for (int i = 0; i < n; ++i) din_regw::launch_regw(k, ncu, 0, 4, kx);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                printf("    knock-out (results wrong): %-34s %8.1f us\n", kn[kx], ms * 1e3 / n);
            }
            const int grid = ncu; uint32_t* dp; hipMalloc(&dp, grid * 4 * 8 * 4); hipMemset(dp, 0, grid * 4 * 8 * 4);
            RegwK kp = k; kp.prof = dp;
            for (int i = 0; i < 3; ++i) din_regw::launch_regw(kp, ncu, 0, dwin);
            hipDeviceSynchronize();
            std::vector<uint32_t> hp(grid * 4 * 8); hipMemcpy(hp.data(), dp, hp.size() * 4, hipMemcpyDeviceToHost);
            double a[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < grid * 4; ++i) for (int f = 0; f < 5; ++f) a[f] += hp[i * 8 + f];
            for (int f = 0; f < 5; ++f) a[f] /= grid * 4;
            printf("    wave (mean): %.0f cycles total: %.0f loading its filters, %.0f waiting for landings, %.0f at barriers, %.0f epilogue\n", a[0], a[1], a[2], a[3], a[4]);
            hipFree(dp);
        }
        fflush(stdout);
        hipFree(dx); hipFree(dw); hipFree(dout); hipFree(db); hipFree(dref); hipFree(dmax); hipFree(dbad);
    }
    return 0;
}
