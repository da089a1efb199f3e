// What does a grid-wide seam cost on this chip: a kernel boundary inside a replayed hipGraph, or a grid barrier inside one persistent launch?
//
// The question behind the review's "one cooperative kernel per inner iteration of the nside-128 CG level": an operator application there
// is a chain of ~7 dependent kernels of 7-27 us on ~256 workgroups, each reading what ALL workgroups of the previous one wrote (alm -> prep ->
// phase -> map -> phase -> partial sums -> alm -> mat-vec), 27 applications per top-level iteration.  This probe runs that shape with
// synthetic phases -- every workgroup reads a strip written by OTHER workgroups in the previous phase, does `work` dependent FMA rounds, writes
// its strip -- (a) as one kernel per phase, the whole chain captured into a hipGraph and replayed (what qcinv.multigrid does), (b) as ONE
// launch of 256 resident workgroups with a grid barrier at every seam (XCD-hierarchical counter barrier with agent-scope release / acquire,
// MI355X_MICROARCH.md "barrier-xcd"; flat counter as a second form).  Results of (a) and (b) are compared word for word (a stale read shows).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/seam_probe tools/probes/seam_probe.hip && tools/probes/seam_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kWG = 256, kThreads = 256, kStrip = 1024;  // doubles per workgroup per phase (8 KB: 2 MB per phase over the grid)

struct Bar {                      // every word on a 128-byte line of its own
    unsigned xcc_cnt[8][32];      // arrivals per XCD (monotonic)
    unsigned top[32];             // XCDs arrived (monotonic)
    unsigned gen[8][32];          // generation released per XCD
    unsigned flat[32];            // flat form: one counter
    unsigned census[8][32];       // workgroups resident per XCD (counted in the first phase)
    unsigned timeout[32];
};

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned ld_rlx(unsigned *p) { return __hip_atomic_load((gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(unsigned *p, unsigned v) { __hip_atomic_store((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_rlx(unsigned *p, unsigned v) { return __hip_atomic_fetch_add((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7; }  // HW_REG_XCC_ID, bits 3:0

__device__ __forceinline__ bool spin_until(unsigned *p, unsigned want, unsigned *tmo)
{
    for (unsigned n = 0; ld_rlx(p) < want; ++n) {
        __builtin_amdgcn_s_sleep(1);
        if (n > (1u << 22)) { st_rlx(tmo, 1u); return false; }
    }
    return true;
}

// every storing wave has drained its stores (s_waitcnt vmcnt(0)) and the workgroup has met at a barrier before lane 0 gets here
__device__ __forceinline__ void grid_barrier_flat(Bar *b, unsigned epoch)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        add_rlx(b->flat, 1u);
        spin_until(b->flat, epoch * gridDim.x, b->timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// XCD-hierarchical: the last arriver of an XCD writes back that XCD's L2 once and arrives at the top counter; the last XCD releases all
__device__ __forceinline__ void grid_barrier_xcd(Bar *b, unsigned epoch, int xcc, unsigned n_in_xcc, unsigned nxcc)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = add_rlx(b->xcc_cnt[xcc], 1u);
        if (old + 1 == epoch * n_in_xcc) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // buffer_wbl2 sc1: the dirty lines of this XCD's L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = add_rlx(b->top, 1u);
            if (t + 1 == epoch * nxcc)
                for (int x = 0; x < 8; ++x) st_rlx(b->gen[x], epoch);
        }
        spin_until(b->gen[xcc], epoch, b->timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// one phase of one (virtual) workgroup: read the strip another workgroup wrote in the previous phase, `work` rounds of a dependent FMA chain, write
__device__ __forceinline__ void phase_body(int wg, int nwg, int phase, int work, const double *__restrict__ in, double *__restrict__ out)
{
    const int src = (wg * 37 + 11 * phase + 5) % nwg;
    for (int i = threadIdx.x; i < kStrip; i += kThreads) {
        double v = in[(size_t)src * kStrip + i];
        double w = in[(size_t)((src + 101) % nwg) * kStrip + (kStrip - 1 - i)];
        double a = v * 0.5 + w * 0.25 + 1e-3 * phase;
        for (int k = 0; k < work; ++k) a = fma(a, 0.999, 1e-4 * w);
        out[(size_t)wg * kStrip + i] = a;
    }
}

__global__ __launch_bounds__(kThreads) void k_phase(int phase, int work, const double *__restrict__ in, double *__restrict__ out)
{
    phase_body(blockIdx.x, gridDim.x, phase, work, in, out);
}

template <int FORM>  // 0: XCD-hierarchical barrier, 1: flat counter
__global__ __launch_bounds__(kThreads) void k_persistent(int nphase, int work, double *__restrict__ bufA, double *__restrict__ bufB, Bar *b)
{
    __shared__ unsigned s_n, s_nx;
    const int xcc = xcc_id();
    unsigned epoch = 0;
    if (FORM == 0) {  // census through one flat barrier: how many workgroups of this grid live on my XCD, how many XCDs hold any
        if (threadIdx.x == 0) add_rlx(b->census[xcc], 1u);
        grid_barrier_flat(b, ++epoch);
        if (threadIdx.x == 0) {
            unsigned nx = 0;
            for (int x = 0; x < 8; ++x) nx += ld_rlx(b->census[x]) > 0;
            s_n = ld_rlx(b->census[xcc]); s_nx = nx;
        }
        __syncthreads();
    }
    const unsigned n_in = s_n, nx = s_nx;
    unsigned e2 = 0;
    for (int p = 0; p < nphase; ++p) {
        const double *in = (p & 1) ? bufB : bufA;
        double *out = (p & 1) ? bufA : bufB;
        phase_body(blockIdx.x, gridDim.x, p, work, in, out);
        if (p + 1 < nphase) {
            if (FORM == 0) grid_barrier_xcd(b, ++e2, xcc, n_in, nx);
            else grid_barrier_flat(b, ++epoch);
        }
    }
}

int main(int argc, char **argv)
{
    const int nphase = argc > 1 ? atoi(argv[1]) : 7 * 27;  // seams of the nside-128 level of one temperature iteration
    const char *order = argc > 2 ? argv[2] : "012";        // the order in which the three forms are timed at each phase length (0 graph, 1 xcd barrier, 2 flat barrier)
    const int reps = 20;
    CHK(hipSetDevice(0));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    double *A, *B, *A2, *B2;
    Bar *bar;
    const size_t nb = (size_t)kWG * kStrip * sizeof(double);
    CHK(hipMalloc(&A, nb)); CHK(hipMalloc(&B, nb)); CHK(hipMalloc(&A2, nb)); CHK(hipMalloc(&B2, nb)); CHK(hipMalloc(&bar, sizeof(Bar)));
    std::vector<double> h0((size_t)kWG * kStrip), fin_h[3] = {h0, h0, h0};
    for (size_t i = 0; i < h0.size(); ++i) h0[i] = 1.0 + 1e-3 * (double)(i % 977);
    printf("seam probe: %d workgroups x %d threads, %d phases (%d seams), 8 KB strip per workgroup and phase, every read crosses workgroups; forms timed in the order %s\n", kWG,
           kThreads, nphase, nphase - 1, order);
    printf("%8s %14s %14s %14s %14s | %12s %12s %12s | %s\n", "work", "graph us/phase", "xcd-bar", "flat-bar", "eager launches", "seam(graph)", "seam(xcd)", "seam(flat)", "check");
    double base[3] = {0, 0, 0};
    for (int work : {0, 200, 800, 2000, 4000}) {
        // (a) one kernel per phase, captured and replayed
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int p = 0; p < nphase; ++p)
            hipLaunchKernelGGL(k_phase, dim3(kWG), dim3(kThreads), 0, st, p, work, (p & 1) ? B : A, (p & 1) ? A : B);
        CHK(hipStreamEndCapture(st, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double t[4];
        for (int fi = 0; fi < 4; ++fi) {
            const int form = fi < 3 ? order[fi] - '0' : 3;  // 3: the same kernels launched eagerly on the stream (no graph), timed last
            double best = 1e30;
            for (int r = 0; r < reps + 2; ++r) {
                double *a = (form == 0 || form == 3) ? A : A2, *bb = (form == 0 || form == 3) ? B : B2;
                CHK(hipMemcpyAsync(a, h0.data(), nb, hipMemcpyHostToDevice, st));
                CHK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
                CHK(hipStreamSynchronize(st));
                auto t0 = std::chrono::steady_clock::now();
                if (form == 0) CHK(hipGraphLaunch(ge, st));
                else if (form == 3) { for (int p = 0; p < nphase; ++p) hipLaunchKernelGGL(k_phase, dim3(kWG), dim3(kThreads), 0, st, p, work, (p & 1) ? B : A, (p & 1) ? A : B); }
                else if (form == 1) hipLaunchKernelGGL(k_persistent<0>, dim3(kWG), dim3(kThreads), 0, st, nphase, work, a, bb, bar);
                else hipLaunchKernelGGL(k_persistent<1>, dim3(kWG), dim3(kThreads), 0, st, nphase, work, a, bb, bar);
                CHK(hipStreamSynchronize(st));
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (r >= 2 && us < best) best = us;
            }
            t[form] = best / nphase;
            if (form < 3) CHK(hipMemcpy(fin_h[form].data(), (nphase & 1) ? (form == 0 ? B : B2) : (form == 0 ? A : A2), nb, hipMemcpyDeviceToHost));
        }
        // word-for-word check of the three forms' final buffers
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += (fin_h[0][i] != fin_h[1][i]) + (fin_h[0][i] != fin_h[2][i]);
        unsigned tmo = 0;
        CHK(hipMemcpy(&tmo, bar->timeout, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (work == 0) { base[0] = t[0]; base[1] = t[1]; base[2] = t[2]; }
        printf("%8d %14.2f %14.2f %14.2f %14.2f | %12.2f %12.2f %12.2f | %s%s\n", work, t[0], t[1], t[2], t[3], base[0], base[1], base[2],
               bad ? "MISMATCH (a barrier form vs graph)" : "both barrier forms == graph", tmo ? " TIMEOUT" : "");
        CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    }
    printf("(us/phase = wall time of the whole chain / phases, best of %d; the work = 0 row prices the bare seam; the difference between a\n"
           " column's rows is the phase body, the same code in all three forms)\n", reps);
    return 0;
}
