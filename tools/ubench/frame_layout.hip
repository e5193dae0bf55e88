// Micro-benchmark: a data mover with the fused OFDM kernel's exact geometry -- frames of 76 symbols of 2552 samples
// (20 416 B) behind a 2656-sample null symbol, three runs per frame (25/25/25 data symbols + each run's reference
// symbol), one wave per run, per symbol 20 loads of 1 KB (prefix + body, as the kernel) and one 3 KB store into the
// frame's soft-bit row -- timed for different frame strides of the input and of the output.
// build: hipcc -O3 --offload-arch=gfx950 frame_layout.hip -o frame_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int SYM = 2552 * 8, NULLB = 2656 * 8;

template <int SLEEP>
__global__ __launch_bounds__(256) void mover(const char *in, char *out, int n_items, size_t in_stride, size_t out_stride) {
    const int lane = threadIdx.x & 63;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= n_items) return;
    const int frame = item / 3, part = item - 3 * frame;
    const int l_first = 25 * part, l_last = 25 * (part + 1);       // symbols l_first..l_last of 0..75 (first = reference)
    const char *fin = in + size_t(frame) * in_stride + NULLB;
    char *fout = out + size_t(frame) * out_stride;
    v4u acc = {1u, 2u, 3u, 4u};
    for (int l = l_first; l <= l_last; l++) {
        const v4u *p = reinterpret_cast<const v4u *>(fin + size_t(l) * SYM) + lane;
        v4u v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) if (i < 19 || lane < 60) v[i] = __builtin_nontemporal_load(p + 64 * i); else v[i] = acc;
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i];
        // stand-in for the arithmetic of a symbol: the wave has nothing in flight for SLEEP x 1024 cycles
#pragma unroll
        for (int k = 0; k < SLEEP; k++) __builtin_amdgcn_s_sleep(16);
        if (l > l_first) {
            v4u *o = reinterpret_cast<v4u *>(fout + size_t(l - 1) * 3072) + lane;
#pragma unroll
            for (int i = 0; i < 3; i++) __builtin_nontemporal_store(acc, o + 64 * i);
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
template <int SLEEP = 0>
float run(const char *in, char *out, int n_frames, size_t in_stride, size_t out_stride) {
    const int n_items = 3 * n_frames;
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mover<SLEEP>, dim3(unsigned((n_items + 3) / 4)), dim3(256), 51 * 1024, 0, in, out, n_items, in_stride, out_stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep > 0 && t < best) best = t;
    }
    return best;
}

int main(int argc, char **argv) {
    const bool noise = !(argc > 1 && argv[1][0] == 'z');       // "z": zero-filled input instead of noise
    const int n = 16384;
    const size_t FR = size_t(196608) * 8, SO = 230400;
    const size_t in_bytes = size_t(n) * (FR + (1 << 20)), out_bytes = size_t(n) * (SO + (1 << 18));
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int K = 3;
    char *in[K], *out[K];
    for (int k = 0; k < K; k++) {
        if (hipMalloc(&in[k], in_bytes) != hipSuccess || hipMalloc(&out[k], out_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        if (noise) hipLaunchKernelGGL(fill_noise, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned *>(in[k]), in_bytes / 4); else hipMemset(in[k], 0, in_bytes);
    }
    const size_t ipads[] = {0, 256, 1024, 4096, 4096 + 256, 16384, 20416, 65536 + 256, 262144, 524288 + 4096 + 256};
    const size_t opads[] = {0, 256, 1024, 2048, 3072, 4096 + 256, 31744 /* -> 262144 */, 65536 + 256};
    // a wave that computes between its loads has nothing in flight meanwhile: the same mover with s_sleep between a
    // symbol's loads and the next symbol's (1024 cycles per unit), on every buffer pair
    for (int k = 0; k < K; k++)
        printf("pair %d, idle 0 / 2 / 4 / 6 / 8 / 12 / 16 x 1024 cycles per symbol: %.3f %.3f %.3f %.3f %.3f %.3f %.3f ms\n", k,
               run<0>(in[k], out[k], n, FR, SO), run<2>(in[k], out[k], n, FR, SO), run<4>(in[k], out[k], n, FR, SO),
               run<6>(in[k], out[k], n, FR, SO), run<8>(in[k], out[k], n, FR, SO), run<12>(in[k], out[k], n, FR, SO),
               run<16>(in[k], out[k], n, FR, SO));
    if (argc > 2) return 0;
    for (int k = 0; k < K; k++) {
        printf("%s input, buffer pair %d: rows = input frame stride 1572864 + pad, columns = output frame stride 230400 + pad; ms per 16384 frames\n      ", noise ? "noise" : "zero", k);
        for (size_t op : opads) printf(" %7zu", op);
        printf("\n");
        for (size_t ip : ipads) {
            printf("%7zu", ip);
            for (size_t op : opads) printf(" %7.3f", run(in[k], out[k], n, FR + ip, SO + op));
            printf("\n");
        }
    }
    return 0;
}
