// Micro-benchmark: map of the read/write interference domains of MI355X's HBM.  One 256 GB arena; a small mover
// (2 GB read + 0.3 GB written, the OFDM kernel's access shape, noise data, non-temporal) is timed with its input window
// at arena offset i x 8 GB and its output window at j x 8 GB.  Slow cells = input and output in the same domain.
// build: hipcc -O3 --offload-arch=gfx950 domain_map.hip -o domain_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mover(const char *in, char *out, int n_chunks, int cpw) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    v4u acc = {1u, 2u, 3u, 4u};
    for (int c = 0; c < cpw; c++) {
        const int chunk = wave * cpw + c;
        if (chunk >= n_chunks) break;
        const v4u *p = reinterpret_cast<const v4u *>(in + size_t(chunk) * 20416) + lane;
        v4u v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = __builtin_nontemporal_load(p + 64 * i);
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i];
        v4u *o = reinterpret_cast<v4u *>(out + size_t(chunk) * 3072) + lane;
#pragma unroll
        for (int i = 0; i < 3; i++) __builtin_nontemporal_store(acc, o + 64 * i);
    }
}

__global__ void fill_noise(unsigned *p, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        unsigned x = unsigned(i) * 2654435761u + unsigned(i >> 32) * 40503u + 12345u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (x & 0x807fffffu) | 0x3f000000u;
    }
}

int main(int argc, char **argv) {
    const int step_gb = argc > 1 ? atoi(argv[1]) : 8;
    const size_t arena_bytes = size_t(256) << 30;
    char *arena;
    if (hipMalloc(&arena, arena_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_noise, dim3(8192), dim3(256), 0, 0, reinterpret_cast<unsigned *>(arena), arena_bytes / 4);
    hipDeviceSynchronize();
    const int n_chunks = 100000, cpw = 8;                      // 2.04 GB in, 0.31 GB out; 12 500 waves = 4 rounds of 3072
    const unsigned grid = unsigned(((n_chunks + cpw - 1) / cpw + 3) / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = int((arena_bytes >> 30) / step_gb) - 1;
    printf("arena %p, 256 GB; rows = input window at i x %d GB, columns = output window at j x %d GB + 4 GB; us per launch\n      ", (void *)arena, step_gb, step_gb);
    for (int j = 0; j < N; j++) printf(" %4d", j * step_gb + 4);
    printf("\n");
    for (int i = 0; i < N; i++) {
        printf("%4d  ", i * step_gb);
        for (int j = 0; j < N; j++) {
            const char *in = arena + (size_t(i) * step_gb << 30);
            char *out = arena + ((size_t(j) * step_gb + 4) << 30);
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(mover, dim3(grid), dim3(256), 51 * 1024, 0, in, out, n_chunks, cpw);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (rep > 0 && t < best) best = t;
            }
            printf(" %4.0f", best * 1e3f);
        }
        printf("\n");
    }
    return 0;
}
