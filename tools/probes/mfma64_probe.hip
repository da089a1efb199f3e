// Issue-rate probe for FP64 on gfx950: vector FMA, the two FP64 MFMA shapes, and whether a SIMD runs the two side by side.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma64_probe.hip -o gpurun_out/mfma64_probe ; run on the GPU box.
// Prints TFLOP/s per mode (FMA = 2 flops per lane; mfma 16x16x4 = 2048, 4x4x4 (4 blocks) = 512 flops per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

enum { kFma = 1, kM16 = 2, kM4 = 4 };

// role(wave) picks what a wave issues; mixed modes give different waves (or one wave) different instruction kinds
template <int MODE_A, int MODE_B, bool SPLIT>
__global__ __launch_bounds__(512) void k_probe(double *out, int iters, double a0, double b0)
{
    const int wave = threadIdx.x >> 6;
    // SPLIT: waves 0-3 run MODE_A, waves 4-7 MODE_B (one of each per SIMD); else every wave runs MODE_A | MODE_B interleaved
    const int mode = SPLIT ? (wave < 4 ? MODE_A : MODE_B) : (MODE_A | MODE_B);
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 + threadIdx.x * 1e-9 + i;
    d4 c16[4];
    for (int i = 0; i < 4; ++i) c16[i] = d4{0., 0., 0., 0.};
    double c4[8];
    for (int i = 0; i < 8; ++i) c4[i] = 0.;
    const double a = a0 * 0.5 + threadIdx.x * 1e-12, b = b0;
    for (int it = 0; it < iters; ++it) {
        if (mode & kFma) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
        }
        if (mode & kM16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[i], c16[i], 0, 0, 0);
        }
        if (mode & kM4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[i], c4[i], 0, 0, 0);
        }
    }
    double s = 0.;
    for (int i = 0; i < 8; ++i) s += x[i] + c4[i];
    for (int i = 0; i < 4; ++i) s += c16[i][0] + c16[i][1] + c16[i][2] + c16[i][3];
    if (s == 1.2345e300) out[threadIdx.x] = s;
}

template <int A, int B, bool SPLIT>
static void run(const char *name, double *out, int wg_per_cu)
{
    const int iters = 20000, nblk = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<A, B, SPLIT>), dim3(nblk), dim3(512), 0, 0, out, 100, 1.0, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_probe<A, B, SPLIT>), dim3(nblk), dim3(512), 0, 0, out, iters, 1.0, 1e-3);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // flops issued
    auto flops_of = [&](int mode, double nwaves) {
        double f = 0.;
        if (mode & kFma) f += 8 * 128.;
        if (mode & kM16) f += 4 * 2048.;
        if (mode & kM4) f += 8 * 512.;
        return f * nwaves * iters;
    };
    const double nw = (double)nblk * 8;
    double fa, fb;
    if (SPLIT) { fa = flops_of(A, nw / 2); fb = flops_of(B, nw / 2); }
    else { fa = flops_of(A, nw); fb = flops_of(B, nw); }
    printf("%-34s wg/cu %d  %8.3f ms   A %7.2f TF  B %7.2f TF  sum %7.2f TF\n", name, wg_per_cu, ms, fa / ms * 1e-9, fb / ms * 1e-9,
           (fa + fb) / ms * 1e-9);
}

int main()
{
    double *out;
    hipMalloc(&out, 4096 * sizeof(double));
    for (int w = 1; w <= 2; ++w) {
        run<kFma, 0, false>("fma only", out, w);
        run<kM16, 0, false>("mfma 16x16x4 only", out, w);
        run<kM4, 0, false>("mfma 4x4x4 only", out, w);
        run<kFma, kM16, true>("split waves: fma | mfma16", out, w);
        run<kFma, kM4, true>("split waves: fma | mfma4", out, w);
        run<kFma, kM16, false>("one wave: fma + mfma16", out, w);
        run<kFma, kM4, false>("one wave: fma + mfma4", out, w);
        run<kFma, kFma, true>("split waves: fma | fma", out, w);
    }
    return 0;
}
