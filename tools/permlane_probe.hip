#include <hip/hip_runtime.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap32(double &a, double &b) {
    // a: vdst, b: vsrc ; after: lanes<32: (a own, b = partner's a) ; lanes>=32: (a = partner's b, b own)
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    v2u r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    v2u r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(r1.x, r0.x); b = __hiloint2double(r1.y, r0.y);
}
__device__ __forceinline__ void swap16(double &a, double &b) {
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    v2u r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    v2u r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(r1.x, r0.x); b = __hiloint2double(r1.y, r0.y);
}
__device__ __forceinline__ double dpp_xor8(double v) {  // row_ror:8 == lane ^ 8 within a row of 16
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_quad(double v, int) { return v; }
__global__ void k(double* out) {
    double a = threadIdx.x, b = 100 + threadIdx.x;
    swap32(a, b);
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
    double c = threadIdx.x, d = 100 + threadIdx.x;
    swap16(c, d);
    out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
    out[256 + threadIdx.x] = dpp_xor8((double)threadIdx.x);
    int q = threadIdx.x;
    out[320 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, q, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2] = xor 1
    out[384 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, q, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1] = xor 2
    out[448 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, q, 0x124, 0xf, 0xf, false); // row_ror:4
}
int main() {
    double* d; hipMalloc(&d, 512 * 8); k<<<1, 64>>>(d); double h[512]; hipMemcpy(h, d, 512 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"swap32 a", "swap32 b", "swap16 a", "swap16 b", "xor8", "xor1", "xor2", "ror4"};
    for (int r = 0; r < 8; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %g", h[r * 64 + i]); printf("\n"); }
    return 0;
}
