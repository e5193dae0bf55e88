// Micro-benchmark: the data-mover ceiling of the front end's two geometries on the same buffer pair.
//   with prefixes : every symbol period (2552 cf32 = 20 416 B) read whole, 3 072 B written per symbol
//   without       : the last 2048 cf32 (16 384 B) of every period read, the 4 032 B in front skipped, same writes
// One wave per run of 25 symbols, 12 waves per CU (a 51 KB LDS request, as the kernel has), streaming (nt) accesses.
// Several output buffers are tried (the HBM domain of a plain allocation is luck); min and max are printed.
// build: hipcc -O3 --offload-arch=gfx950 gap_read.hip -o gap_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v4 __attribute__((ext_vector_type(4)));

template <bool SKIP_CP>
__global__ __launch_bounds__(256) void mover(const char *in, char *out, size_t n_syms, int syms_per_wave) {
    const int lane = threadIdx.x & 63;
    const size_t wave = size_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    v4 acc = {0, 0, 0, 0};
    for (int c = 0; c < syms_per_wave; c++) {
        const size_t sym = wave * syms_per_wave + c;
        if (sym >= n_syms) break;
        const char *p = in + sym * 20416 + (SKIP_CP ? 4032 : 0);
        constexpr int LOADS = SKIP_CP ? 16 : 20;
        v4 v[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; i++) {
            // the whole-period read is 19.94 KB: the last load of the 20 is partial, as in the kernel
            const size_t off = size_t(i) * 1024 + lane * 16;
            if (SKIP_CP || off < 20416) v[i] = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p + off));
            else v[i] = v4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < LOADS; i++) acc += v[i];
        v4 *o = reinterpret_cast<v4 *>(out + sym * 3072) + lane;
#pragma unroll
        for (int i = 0; i < 3; i++) __builtin_nontemporal_store(acc, o + 64 * i);
    }
}

template <bool SKIP_CP>
float run(const char *in, char *out, size_t n_syms, int spw) {
    const size_t waves = (n_syms + spw - 1) / spw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mover<SKIP_CP>), dim3(unsigned((waves + 3) / 4)), dim3(256), 51 * 1024, 0, in, out, n_syms, spw);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0) best = std::min(best, ms);
    }
    return best;
}

int main() {
    const size_t n_frames = 16384, n_syms = n_frames * 76;
    const size_t in_bytes = n_syms * 20416 + 4096, out_bytes = n_syms * 3072;
    char *in;
    if (hipMalloc(&in, in_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0, in_bytes);
    std::vector<char *> outs;
    for (int k = 0; k < 12; k++) {
        char *o;
        if (hipMalloc(&o, out_bytes) != hipSuccess) break;
        outs.push_back(o);
    }
    for (int spw : {25, 76}) {
        std::vector<float> a, b;
        for (char *o : outs) { a.push_back(run<false>(in, o, n_syms, spw)); b.push_back(run<true>(in, o, n_syms, spw)); }
        const double by_a = double(n_syms) * (20416 + 3072), by_b = double(n_syms) * (16384 + 3072);
        const float amin = *std::min_element(a.begin(), a.end()), amax = *std::max_element(a.begin(), a.end());
        const float bmin = *std::min_element(b.begin(), b.end()), bmax = *std::max_element(b.begin(), b.end());
        printf("%2d symbols per wave, %zu output buffers tried\n", spw, outs.size());
        printf("  whole periods read   : %.3f .. %.3f ms = %.0f .. %.0f GB/s\n", amin, amax, by_a / amin / 1e6, by_a / amax / 1e6);
        printf("  prefixes skipped     : %.3f .. %.3f ms = %.0f .. %.0f GB/s  (time ratio best/best %.3f; bytes ratio %.3f)\n", bmin, bmax,
               by_b / bmin / 1e6, by_b / bmax / 1e6, bmin / amin, by_b / by_a);
    }
    return 0;
}
