// Peak-issue microbenchmark for the VALU integer roofline (SURVEY.md §8d "Peak denominator").
// Measures sustained wave-instruction throughput of the instructions the field arithmetic is made
// of, at 1/2/4/8 waves per SIMD, on all 256 CUs.  Output: one JSON line per (instruction, occupancy)
// with G-lane-ops/s (= 1e9 per-lane results per second) and cycles per wave-instruction per SIMD
// assuming the measured in-kernel clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ITERS 65536
#define CHAINS 8

template <int KIND>
__global__ __launch_bounds__(256) void k_peak(uint32_t* out, uint64_t* clocks, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = seed + tid * 2654435761u, b = (seed ^ 0x9e3779b9u) + tid;
    uint64_t acc[CHAINS];
    double facc[CHAINS];
    float sacc[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) {
        acc[k] = ((uint64_t)(a + k) << 32) | (b + 3 * k);
        facc[k] = 1.0 + 1e-9 * (double)(tid + k);
        sacc[k] = 1.0f + 1e-6f * (float)(tid + k);
    }
    double fa = 1.0000001, fb = 1e-12;
    float sa = 1.00001f, sb = 1e-7f;
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < CHAINS; k++) {
            if (KIND == 0) {  // v_mad_u64_u32
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 1) {  // v_mul_lo_u32
                uint32_t lo = (uint32_t)acc[k];
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                acc[k] = lo;
            } else if (KIND == 2) {  // v_mul_hi_u32
                uint32_t lo = (uint32_t)acc[k];
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                acc[k] = lo;
            } else if (KIND == 3) {  // v_add_u32
                uint32_t lo = (uint32_t)acc[k];
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                acc[k] = lo;
            } else if (KIND == 4) {  // v_lshl_add_u64 (64-bit add)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) % CHAINS]));
            } else if (KIND == 5) {  // v_mad_u32_u24
                uint32_t lo = (uint32_t)acc[k];
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));
                acc[k] = lo;
            } else if (KIND == 6) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(facc[k]) : "v"(fa), "v"(fb));
            } else if (KIND == 7) {  // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(sacc[k]) : "v"(sa), "v"(sb));
            } else if (KIND == 8) {  // v_addc_co_u32 chain (carry add)
                uint32_t lo = (uint32_t)acc[k];
                asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo) : "v"(a) : "vcc");
                acc[k] = lo;
            } else if (KIND == 9) {  // v_mov_b32
                uint32_t lo = (uint32_t)acc[k];
                asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(a + k));
                acc[k] = lo;
            } else if (KIND == 10) {  // v_mul_u32_u24 + v_mul_hi_u32_u24 pair
                uint32_t lo = (uint32_t)acc[k];
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a));
                acc[k] = lo;
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t x = 0;
    double fx = 0;
    float sx = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++) {
        x ^= acc[k];
        fx += facc[k];
        sx += sacc[k];
    }
    out[tid] = (uint32_t)x ^ (uint32_t)(x >> 32) ^ (uint32_t)fx ^ (uint32_t)sx;
    if (threadIdx.x == 0) {
        clocks[2 * blockIdx.x] = t1 - t0;
        clocks[2 * blockIdx.x + 1] = r1 - r0;
    }
}

typedef void (*kfn)(uint32_t*, uint64_t*, uint32_t);

int main() {
    const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_u32", "v_lshl_add_u64", "v_mad_u32_u24",
                           "v_fma_f64",     "v_fma_f32",    "v_addc_co_u32", "v_mov_b32", "v_mul_hi_u32_u24"};
    kfn fns[] = {k_peak<0>, k_peak<1>, k_peak<2>, k_peak<3>, k_peak<4>, k_peak<5>, k_peak<6>, k_peak<7>, k_peak<8>, k_peak<9>, k_peak<10>};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    fprintf(stderr, "device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* d_out;
    uint64_t* d_clk;
    int maxblocks = cus * 8;
    hipMalloc(&d_out, (size_t)maxblocks * 256 * 4);
    hipMalloc(&d_clk, (size_t)maxblocks * 16);
    uint64_t* h_clk = (uint64_t*)malloc((size_t)maxblocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int kind = 0; kind < 11; kind++) {
        for (int wps = 1; wps <= 8; wps *= 2) {
            int blocks = cus * wps;  // 256 threads = 4 waves = 1 wave per SIMD per block
            hipLaunchKernelGGL(fns[kind], dim3(blocks), dim3(256), 0, 0, d_out, d_clk, 12345u);  // warm
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            const int reps = 3;
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(fns[kind], dim3(blocks), dim3(256), 0, 0, d_out, d_clk, 12345u + r);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h_clk, d_clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
            double avgclk = 0, avgreal = 0;
            for (int i = 0; i < blocks; i++) {
                avgclk += (double)h_clk[2 * i];
                avgreal += (double)h_clk[2 * i + 1];
            }
            avgclk /= blocks;
            avgreal /= blocks;
            double ghz = avgclk / (avgreal * 10.0);  // s_memrealtime ticks at 100 MHz
            double lane_ops = (double)blocks * 256 * (double)ITERS * CHAINS * reps;
            double gops = lane_ops / (ms * 1e-3) / 1e9;
            // wave-instructions per SIMD: wps waves * ITERS*CHAINS each; s_memtime ticks at 100 MHz on gfx9 (constant clock)
            double wave_instr_per_simd = (double)wps * ITERS * CHAINS;
            double sec_per_launch = ms * 1e-3 / reps;
            printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"G_lane_ops_per_s\": %.1f, \"ns_per_wave_instr_per_simd\": %.3f, "
                   "\"memtime_ticks\": %.0f, \"ms_per_launch\": %.4f, \"in_kernel_ghz\": %.3f, \"cycles_per_wave_instr_per_simd\": %.3f}\n",
                   names[kind], wps, gops, sec_per_launch * 1e9 / wave_instr_per_simd, avgclk, ms / reps, ghz, avgclk / wave_instr_per_simd);
        }
    }
    return 0;
}
