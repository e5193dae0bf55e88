// Micro-benchmark: cache-policy bits (sc0 / nt / sc1) on the loads and stores of a data mover with the OFDM kernel's
// shape: 12 waves per CU, each wave streams 25 "symbols" of 20 KB (twenty 16-byte loads per lane) and writes 3 KB per
// symbol.  Every (load policy, store policy) pair runs on the same buffers in one process, interleaved.
// build: hipcc -O3 --offload-arch=gfx950 cache_policy.hip -o cache_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// aux: bit 0 = sc0, bit 1 = nt, bit 4 = sc1
template <int LD, int ST>
__global__ __launch_bounds__(256) void mover(const char *in, char *out, int n_chunks, int chunks_per_wave) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    v4u acc = {0u, 0u, 0u, 0u};
    for (int c = 0; c < chunks_per_wave; c++) {
        const int chunk = wave * chunks_per_wave + c;
        if (chunk >= n_chunks) break;
        // a descriptor per chunk (wave-uniform base in SGPRs), 32-bit lane offset
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in) + size_t(chunk) * 20480, 0, 20480, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out + size_t(chunk) * 3072, 0, 3072, 0x00020000);
        v4u v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16, i * 1024, LD);
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i];
#pragma unroll
        for (int i = 0; i < 3; i++) __builtin_amdgcn_raw_buffer_store_b128(acc, rout, lane * 16, i * 1024, ST);
    }
}

struct Variant { const char *name; void (*launch)(const char *, char *, int, int, unsigned); };
template <int LD, int ST>
void launch(const char *in, char *out, int n_chunks, int cpw, unsigned grid) {
    hipLaunchKernelGGL((mover<LD, ST>), dim3(grid), dim3(256), 51 * 1024, 0, in, out, n_chunks, cpw);
}

int main() {
    const size_t n_chunks = size_t(16384) * 76;              // the bench launch: 16 384 frames x 76 symbols
    const size_t in_bytes = n_chunks * 20480, out_bytes = n_chunks * 3072;
    char *in, *out;
    if (hipMalloc(&in, in_bytes) != hipSuccess || hipMalloc(&out, out_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0, in_bytes);
    const int cpw = 25;
    const unsigned grid = unsigned(((n_chunks + cpw - 1) / cpw + 3) / 4);
#define V(LD, ST) Variant{"ld " #LD " st " #ST, launch<LD, ST>}
    std::vector<Variant> vs = {V(0, 0), V(2, 0), V(0, 2), V(2, 2), V(1, 2), V(16, 2), V(17, 2), V(18, 2), V(3, 2), V(19, 2),
                               V(2, 1), V(2, 16), V(2, 17), V(2, 18), V(2, 3), V(2, 19), V(18, 18), V(19, 19)};
    std::vector<std::vector<float>> ms(vs.size());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 5; round++)
        for (size_t k = 0; k < vs.size(); k++) {
            hipEventRecord(e0);
            for (int r = 0; r < 3; r++) vs[k].launch(in, out, int(n_chunks), cpw, grid);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (round > 0) ms[k].push_back(t / 3);
        }
    printf("# aux bits: 1 = sc0, 2 = nt, 16 = sc1.  %.2f GB read + %.2f GB written per launch\n", in_bytes / 1e9, out_bytes / 1e9);
    for (size_t k = 0; k < vs.size(); k++) {
        float mn = 1e9f, sum = 0;
        for (float t : ms[k]) { mn = t < mn ? t : mn; sum += t; }
        printf("%-16s mean %.3f ms  min %.3f ms  %.0f GB/s\n", vs[k].name, sum / ms[k].size(), mn, (in_bytes + out_bytes) / mn / 1e6);
    }
    return 0;
}
