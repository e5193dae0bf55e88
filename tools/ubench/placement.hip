// Micro-benchmark: why does the same HBM-bound launch take 5.4 ms on one buffer and 6.0 ms on another?
// A data mover with the OFDM kernel's shape (12 waves per CU, 25 chunks of 20 KB per wave, 3 KB written per chunk,
// non-temporal accesses) is timed on several hipMalloc'ed input / output buffers, read-only, write-only and both, and
// at different offsets inside one allocation.
// build: hipcc -O3 --offload-arch=gfx950 placement.hip -o placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool RD, bool WR>
__global__ __launch_bounds__(256) void mover(const char *in, char *out, int n_chunks, int cpw, size_t in_stride, size_t out_stride) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    v4u acc = {1u, 2u, 3u, 4u};
    for (int c = 0; c < cpw; c++) {
        const int chunk = wave * cpw + c;
        if (chunk >= n_chunks) break;
        if (RD) {
            const v4u *p = reinterpret_cast<const v4u *>(in + size_t(chunk) * in_stride) + lane;
            v4u v[20];
#pragma unroll
            for (int i = 0; i < 20; i++) v[i] = __builtin_nontemporal_load(p + 64 * i);
#pragma unroll
            for (int i = 0; i < 20; i++) acc += v[i];
        }
        if (WR) {
            v4u *o = reinterpret_cast<v4u *>(out + size_t(chunk) * out_stride) + lane;
#pragma unroll
            for (int i = 0; i < 3; i++) __builtin_nontemporal_store(acc, o + 64 * i);
        } else if (acc.x == 0x12345u && acc.y == 77u) {
            reinterpret_cast<v4u *>(out)[wave] = acc;
        }
    }
}

// sample values matter: the same launch moves zeros ~6 % faster than noise-like floats
__global__ void fill_noise(unsigned *p, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        unsigned x = unsigned(i) * 2654435761u + unsigned(i >> 32) * 40503u + 12345u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (x & 0x807fffffu) | 0x3f000000u;                 // +-[0.5, 1): every mantissa bit random
    }
}

static hipEvent_t e0, e1;
template <bool RD, bool WR>
float run(const char *in, char *out, int n_chunks, size_t in_stride = 20480, size_t out_stride = 3072) {
    const int cpw = 25;
    const unsigned grid = unsigned(((n_chunks + cpw - 1) / cpw + 3) / 4);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mover<RD, WR>), dim3(grid), dim3(256), 51 * 1024, 0, in, out, n_chunks, cpw, in_stride, out_stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep > 0 && t < best) best = t;
    }
    return best;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 5;
    const bool noise = !(argc > 2 && argv[2][0] == 'z');       // "z": zero-filled input instead of noise
    printf("%s input\n", noise ? "noise" : "zero");
    const int n_chunks = 16384 * 76;
    const size_t in_bytes = size_t(n_chunks) * 20480 + (size_t(64) << 20), out_bytes = size_t(n_chunks) * 3072 + (size_t(64) << 20);
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<char *> in(K), out(K);
    for (int k = 0; k < K; k++) {
        if (hipMalloc(&in[k], in_bytes) != hipSuccess || hipMalloc(&out[k], out_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        if (noise) hipLaunchKernelGGL(fill_noise, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned *>(in[k]), in_bytes / 4); else hipMemset(in[k], 0, in_bytes);
    }
    printf("in  "); for (int k = 0; k < K; k++) printf(" %p", (void *)in[k]); printf("\nout "); for (int k = 0; k < K; k++) printf(" %p", (void *)out[k]); printf("\n");
    printf("read-only, per input buffer (ms):  "); for (int k = 0; k < K; k++) printf(" %.3f", run<true, false>(in[k], out[0], n_chunks)); printf("\n");
    printf("write-only, per output buffer (ms):"); for (int k = 0; k < K; k++) printf(" %.3f", run<false, true>(in[0], out[k], n_chunks)); printf("\n");
    printf("read + write: rows = input buffer, columns = output buffer\n");
    for (int i = 0; i < K; i++) { printf("   in%d ", i); for (int j = 0; j < K; j++) printf(" %.3f", run<true, true>(in[i], out[j], n_chunks)); printf("\n"); }
    printf("read + write on (in0, out0), input shifted by k x 4 KB:  "); for (int k = 0; k < 8; k++) printf(" %.3f", run<true, true>(in[0] + size_t(k) * 4096, out[0], n_chunks)); printf("\n");
    printf("read + write on (in0, out0), input shifted by k x 2 MB:  "); for (int k = 0; k < 8; k++) printf(" %.3f", run<true, true>(in[0] + (size_t(k) << 21), out[0], n_chunks)); printf("\n");
    printf("read + write on (in0, out0), output shifted by k x 4 KB: "); for (int k = 0; k < 8; k++) printf(" %.3f", run<true, true>(in[0], out[0] + size_t(k) * 4096, n_chunks)); printf("\n");
    printf("read + write on (in0, out0), output shifted by k x 2 MB: "); for (int k = 0; k < 8; k++) printf(" %.3f", run<true, true>(in[0], out[0] + (size_t(k) << 21), n_chunks)); printf("\n");
    printf("read + write on (in0, out0), chunk stride 20480 + k x 256 B: "); for (int k = 0; k < 8; k++) printf(" %.3f", run<true, true>(in[0], out[0], int(size_t(n_chunks) * 20480 / (20480 + 256 * k)), 20480 + 256 * k, 3072)); printf("\n");
    // one arena: input at its start, output at a varying distance behind it
    for (int k = 0; k < K; k++) { hipFree(in[k]); hipFree(out[k]); }
    char *arena;
    const size_t arena_bytes = size_t(72) << 30;
    if (hipMalloc(&arena, arena_bytes) != hipSuccess) { printf("arena alloc failed\n"); return 1; }
    if (noise) hipLaunchKernelGGL(fill_noise, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned *>(arena), arena_bytes / 4); else hipMemset(arena, 0, arena_bytes);
    printf("arena %p: input at +0, output at +26 GB + x\n", (void *)arena);
    printf("  x = k x 256 MB:"); for (int k = 0; k < 16; k++) printf(" %.3f", run<true, true>(arena, arena + (size_t(26) << 30) + (size_t(k) << 28), n_chunks)); printf("\n");
    printf("  x = k x 4 GB:  "); for (int k = 0; k < 10; k++) printf(" %.3f", run<true, true>(arena, arena + (size_t(26) << 30) + (size_t(k) << 32), n_chunks)); printf("\n");
    printf("  input at +k x 4 GB, output at +66 GB:"); for (int k = 0; k < 10; k++) printf(" %.3f", run<true, true>(arena + (size_t(k) << 32), arena + (size_t(66) << 30), n_chunks)); printf("\n");
    printf("  x = k x 16 MB: "); for (int k = 0; k < 16; k++) printf(" %.3f", run<true, true>(arena, arena + (size_t(26) << 30) + (size_t(k) << 24), n_chunks)); printf("\n");
    return 0;
}
