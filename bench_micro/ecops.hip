// Per-operation cost microbenchmark: chains of fp_mul / fp_sqr / xyzz_madd / xyzz_dbl / xyzz_add per
// lane, all CUs, at 1/2/4 waves per SIMD.  Reports ns per op per lane-wave and the implied fraction
// of the measured v_mad_u64_u32 peak (91 MADs per mul, 55 per sqr).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../bulletproofs_amd/csrc/ec.h"
using namespace bp;
#define N_IT 500
template <int KIND>
__global__ __launch_bounds__(256) void k_ops(uint32_t* out, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    fp a = fp_zero(), b = fp_zero();
    for (int i = 0; i < 9; i++) { a.v[i] = (seed * 2654435761u + tid * 40503u + i * 977u) & BP_M29; b.v[i] = (seed + tid * 7919u + i * 31u) & BP_M29; }
    a.v[8] &= 0xFFFFFF; b.v[8] &= 0xFFFFFF;
    // a curve-ish point is not needed for timing: formulas are branch-free except the (never taken) P==0 case
    affine P; P.x = a; P.y = b;
    xyzz acc; acc.X = b; acc.Y = a; acc.ZZ = fp_add(a, b); acc.ZZZ = fp_sub(a, b);
    xyzz other = acc; other.X = fp_add(acc.X, a);
    fp r = a;
    for (int it = 0; it < N_IT; it++) {
        if (KIND == 0) r = fp_mul(r, b);
        else if (KIND == 1) r = fp_sqr(r);
        else if (KIND == 2) xyzz_madd(acc, P);
        else if (KIND == 3) acc = xyzz_dbl(acc);
        else if (KIND == 4) xyzz_add(acc, other);
        else if (KIND == 5) r = fp_sub(fp_add(r, b), a);
    }
    uint32_t x = 0;
    for (int i = 0; i < 9; i++) x ^= r.v[i] ^ acc.X.v[i] ^ acc.Y.v[i] ^ acc.ZZ.v[i] ^ acc.ZZZ.v[i];
    out[tid] = x;
}
typedef void (*kfn)(uint32_t*, uint32_t);
int main() {
    const char* names[] = {"fp_mul", "fp_sqr", "xyzz_madd", "xyzz_dbl", "xyzz_add", "fp_add+fp_sub"};
    const double mads[] = {91, 55, 8 * 91 + 2 * 55, 6 * 91 + 3 * 55, 12 * 91 + 2 * 55, 0};
    kfn fns[] = {k_ops<0>, k_ops<1>, k_ops<2>, k_ops<3>, k_ops<4>, k_ops<5>};
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 6; kind++) for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(fns[kind], dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fns[kind], dim3(blocks), dim3(256), 0, 0, d, 2u);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static uint32_t hbuf[256 * 4 * 256];
        hipMemcpy(hbuf, d, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost);
        uint64_t cks = 0;
        for (int i = 0; i < blocks * 256; i++) cks = cks * 1000003u + hbuf[i];
        double ops = (double)blocks * 256 * N_IT;
        double ns_per_op_simd = ms * 1e6 / ((double)wps * N_IT);  // per wave-op per SIMD
        printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ns_per_wave_op_per_simd\": %.1f, \"Gops_per_s\": %.2f, \"TMAD_per_s\": %.2f, \"frac_of_31.2T\": %.3f, \"checksum\": \"%016llx\"}\n",
               names[kind], wps, ns_per_op_simd, ops / (ms * 1e-3) / 1e9, ops * mads[kind] / (ms * 1e-3) / 1e12, ops * mads[kind] / (ms * 1e-3) / 31.2e12, (unsigned long long)cks);
    }
    return 0;
}
