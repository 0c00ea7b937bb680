// Does v_mad_u64_u32's rate depend on WHICH registers it reads?  (VGPR bank = register number mod 4.)
// Eight independent accumulator chains v[0:1] .. v[14:15]; the two 32-bit sources are pinned per variant:
//   0  same two registers for every chain (what peak_mad32 measures)
//   1  sources in the accumulator's own banks (acc v[2k:2k+1] -> banks (2k)%4, (2k+1)%4; src0, src1 chosen in those banks)
//   2  sources in the two banks the accumulator does not use
//   3  both sources in ONE bank, different from the accumulator's
//   4  two sources per chain in distinct registers, banks as they come (compiler-like)
// One JSON line per (variant, waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 20000
#define MADS(acc0, s0, s1) "v_mad_u64_u32 v[" #acc0 ":" #acc0 "+1], vcc, v" #s0 ", v" #s1 ", v[" #acc0 ":" #acc0 "+1]\n"
template <int V>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = seed + tid * 2654435761u;
    // load 32 VGPRs with data through explicit registers
    asm volatile(
        "v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %0\n v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n"
        "v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n"
        "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
        "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
        :: "v"(x) : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
    for (int it = 0; it < ITERS; it++) {
        if (V == 0)
            asm volatile(MADS(0, 16, 17) MADS(2, 16, 17) MADS(4, 16, 17) MADS(6, 16, 17) MADS(8, 16, 17) MADS(10, 16, 17) MADS(12, 16, 17) MADS(14, 16, 17) ::: "vcc","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15");
        else if (V == 1)  // acc banks (0,1) or (2,3); sources in the same two banks
            asm volatile(MADS(0, 16, 17) MADS(2, 18, 19) MADS(4, 20, 21) MADS(6, 22, 23) MADS(8, 24, 25) MADS(10, 26, 27) MADS(12, 28, 29) MADS(14, 30, 31) ::: "vcc","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15");
        else if (V == 2)  // sources in the other two banks
            asm volatile(MADS(0, 18, 19) MADS(2, 16, 17) MADS(4, 22, 23) MADS(6, 20, 21) MADS(8, 26, 27) MADS(10, 24, 25) MADS(12, 30, 31) MADS(14, 28, 29) ::: "vcc","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15");
        else if (V == 3)  // both sources in one bank (bank 2 for acc in banks 0,1; bank 0 for acc in banks 2,3)
            asm volatile(MADS(0, 18, 22) MADS(2, 16, 20) MADS(4, 26, 30) MADS(6, 24, 28) MADS(8, 18, 26) MADS(10, 16, 24) MADS(12, 22, 30) MADS(14, 20, 28) ::: "vcc","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15");
        else  // mixed
            asm volatile(MADS(0, 17, 20) MADS(2, 19, 21) MADS(4, 16, 27) MADS(6, 25, 30) MADS(8, 18, 23) MADS(10, 29, 24) MADS(12, 31, 22) MADS(14, 26, 28) ::: "vcc","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15");
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v0, v2\n v_xor_b32 %0, %0, v4\n v_xor_b32 %0, %0, v6\n v_xor_b32 %0, %0, v8\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v12\n v_xor_b32 %0, %0, v14\n v_xor_b32 %0, %0, v1\n" : "=v"(r));
    out[tid] = r;
}
typedef void (*kfn)(uint32_t*, uint32_t);
int main() {
    kfn fns[] = {k<0>, k<1>, k<2>, k<3>, k<4>};
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int v = 0; v < 5; v++)
        for (int wps = 1; wps <= 8; wps *= 2) {
            int blocks = 256 * wps;
            hipLaunchKernelGGL(fns[v], dim3(blocks), dim3(256), 0, 0, d, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(fns[v], dim3(blocks), dim3(256), 0, 0, d, 2u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double lane_ops = (double)blocks * 256 * ITERS * 8;
            printf("{\"variant\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"T_lane_mad_per_s\": %.2f, \"ns_per_wave_mad_per_simd\": %.3f}\n", v, wps, ms,
                   lane_ops / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)wps * ITERS * 8));
        }
    return 0;
}
