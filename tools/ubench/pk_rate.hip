// Micro-benchmark: issue rate of scalar fp32 FMA vs packed fp32 ops vs packed int16 ops on gfx950.
// build: hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f, d = 0.5f;
    const f2 pc = {c, c}, pd = {d, d};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (MODE == 0) {           // 8 independent scalar FMAs
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if constexpr (MODE == 1) {    // 8 independent packed FMAs (16 FMAs)
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
            } else if constexpr (MODE == 2) {    // packed mul / add
                asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
                             "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
            } else if constexpr (MODE == 3) {    // packed int16 add / max
                asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_max_i16 %1, %1, %9\n v_pk_add_u16 %2, %2, %8\n v_pk_max_i16 %3, %3, %9\n"
                             "v_pk_add_u16 %4, %4, %8\n v_pk_max_i16 %5, %5, %9\n v_pk_add_u16 %6, %6, %8\n v_pk_max_i16 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if constexpr (MODE == 4) {    // scalar mul / add
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else {                             // v_and / v_perm (integer full-rate reference)
                asm volatile("v_and_b32 %0, %0, %8\n v_perm_b32 %1, %1, %9, %8\n v_and_b32 %2, %2, %8\n v_perm_b32 %3, %3, %9, %8\n"
                             "v_and_b32 %4, %4, %8\n v_perm_b32 %5, %5, %9, %8\n v_and_b32 %6, %6, %8\n v_perm_b32 %7, %7, %9, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                          p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int MODE>
void run(const char *name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;
    float *out;
    hipMalloc(&out, size_t(blocks) * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD: waves_per_simd * iters * 64
    const double inst = double(waves_per_simd) * iters * 64;
    printf("%-28s waves/SIMD %d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_pk_mul/add_f32", w);
        run<4>("v_mul/add_f32", w);
        run<3>("v_pk_add_u16/max_i16", w);
        run<5>("v_and/v_perm", w);
    }
    return 0;
}
