// Micro-benchmark: do two READ streams interfere when they come from the same HBM domain?  One 256 GB arena; half of
// the waves stream a 1 GB window at i x 16 GB, the other half one at j x 16 GB (non-temporal 16-byte loads, the OFDM
// kernel's shape, noise data).  Compare with domain_map.hip (read + write).
// build: hipcc -O3 --offload-arch=gfx950 domain_read.hip -o domain_read
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void reader(const char *a, const char *b, char *sink, int n_chunks, int cpw) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const char *in = (wave & 1) ? b : a;
    v4u acc = {1u, 2u, 3u, 4u};
    for (int c = 0; c < cpw; c++) {
        const int chunk = (wave >> 1) * cpw + c;
        if (chunk >= n_chunks) break;
        const v4u *p = reinterpret_cast<const v4u *>(in + size_t(chunk) * 20416) + lane;
        v4u v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = __builtin_nontemporal_load(p + 64 * i);
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i];
    }
    if (acc.x == 0x12345u && acc.y == 7u) reinterpret_cast<v4u *>(sink)[wave] = acc;
}

__global__ void fill_noise(unsigned *p, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        unsigned x = unsigned(i) * 2654435761u + unsigned(i >> 32) * 40503u + 12345u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (x & 0x807fffffu) | 0x3f000000u;
    }
}

int main() {
    const int step_gb = 16;
    const size_t arena_bytes = size_t(256) << 30;
    char *arena, *sink;
    if (hipMalloc(&arena, arena_bytes) != hipSuccess || hipMalloc(&sink, 1 << 24) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_noise, dim3(8192), dim3(256), 0, 0, reinterpret_cast<unsigned *>(arena), arena_bytes / 4);
    hipDeviceSynchronize();
    const int n_chunks = 50000, cpw = 8;                       // 1.02 GB per window, 12 500 waves in all
    const unsigned grid = unsigned(((2 * n_chunks + cpw - 1) / cpw + 3) / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = int((arena_bytes >> 30) / step_gb) - 1;
    printf("rows = window A at i x %d GB, columns = window B at j x %d GB + 8 GB; us per launch (2.04 GB read)\n      ", step_gb, step_gb);
    for (int j = 0; j < N; j++) printf(" %4d", j * step_gb + 8);
    printf("\n");
    for (int i = 0; i < N; i++) {
        printf("%4d  ", i * step_gb);
        for (int j = 0; j < N; j++) {
            const char *a = arena + (size_t(i) * step_gb << 30), *b = arena + ((size_t(j) * step_gb + 8) << 30);
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(reader, dim3(grid), dim3(256), 51 * 1024, 0, a, b, sink, n_chunks, cpw);
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
