// Micro-benchmark: HBM read ceiling for the OFDM kernel's access pattern (each wave streams 20 KB symbols with
// 16-byte loads) with and without a 13 % write stream.  build: hipcc -O3 --offload-arch=gfx950 read_bw.hip -o read_bw
#include <hip/hip_runtime.h>
#include <cstdio>

// variant: writes batched every BATCH chunks (BATCH*3 KB contiguous), optionally non-temporal
template <int BATCH, bool NT>
__global__ __launch_bounds__(256) void rdw(const float4 *in, float4 *out, size_t n_chunks, int chunks_per_wave) {
    const int lane = threadIdx.x & 63;
    const size_t wave = size_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int c = 0; c < chunks_per_wave; c++) {
        const size_t chunk = wave * chunks_per_wave + c;
        if (chunk >= n_chunks) break;
        const float4 *p = in + chunk * (20 * 64) + lane;
        float4 v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = p[64 * i];
#pragma unroll
        for (int i = 0; i < 20; i++) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        if ((c % BATCH) == BATCH - 1) {
            float4 *o = out + (chunk - (BATCH - 1)) * 192 + lane;
#pragma unroll
            for (int i = 0; i < 3 * BATCH; i++) {
                if (NT) {
                    typedef float v4 __attribute__((ext_vector_type(4)));
                    v4 t = {acc.x, acc.y, acc.z, acc.w};
                    __builtin_nontemporal_store(t, reinterpret_cast<v4 *>(o + 64 * i));
                } else o[64 * i] = acc;
            }
        }
    }
}

// variant: WK KB written per 20 KB read (contiguous per wave)
template <int WK>
__global__ __launch_bounds__(256) void rdmix(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n_chunks, int chunks_per_wave) {
    const int lane = threadIdx.x & 63;
    const size_t wave = size_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int c = 0; c < chunks_per_wave; c++) {
        const size_t chunk = wave * chunks_per_wave + c;
        if (chunk >= n_chunks) break;
        const float4 *p = in + chunk * (20 * 64) + lane;
        float4 v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = p[64 * i];
#pragma unroll
        for (int i = 0; i < 20; i++) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        float4 *o = out + chunk * (WK * 64) + lane;
#pragma unroll
        for (int i = 0; i < WK; i++) o[64 * i] = acc;
    }
}
template <int WK>
void runmix(const float4 *in, float4 *out, size_t bytes) {
    const size_t n_chunks = bytes / (20 * 1024);
    const int cpw = 24;
    const size_t waves = (n_chunks + cpw - 1) / cpw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rdmix<WK>), dim3(unsigned((waves + 3) / 4)), dim3(256), 0, 0, in, out, n_chunks, cpw);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rdb = double(n_chunks) * 20 * 1024, wrb = double(n_chunks) * WK * 1024;
    printf("read 20 KB + write %2d KB per chunk: %.3f ms  read %.0f  write %.0f  total %.0f GB/s\n", WK, ms, rdb / ms / 1e6,
           wrb / ms / 1e6, (rdb + wrb) / ms / 1e6);
}

template <int BATCH, bool NT>
void runw(const char *name, const float4 *in, float4 *out, size_t bytes, int chunks_per_wave) {
    const size_t n_chunks = bytes / (20 * 1024);
    const size_t waves = (n_chunks + chunks_per_wave - 1) / chunks_per_wave;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rdw<BATCH, NT>), dim3(unsigned((waves + 3) / 4)), dim3(256), 0, 0, in, out, n_chunks, chunks_per_wave);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rdb = double(n_chunks) * 20 * 1024, wrb = double(n_chunks) * 3072;
    printf("%-34s batch %2d nt %d: %.3f ms  total %.0f GB/s\n", name, BATCH, int(NT), ms, (rdb + wrb) / ms / 1e6);
}

template <int LOADS, bool WRITE>
__global__ __launch_bounds__(256) void rd(const float4 *in, float4 *out, size_t n_chunks, int chunks_per_wave) {
    const int lane = threadIdx.x & 63;
    const size_t wave = size_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int c = 0; c < chunks_per_wave; c++) {
        const size_t chunk = wave * chunks_per_wave + c;
        if (chunk >= n_chunks) break;
        const float4 *p = in + chunk * (LOADS * 64) + lane;
        float4 v[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; i++) v[i] = p[64 * i];
#pragma unroll
        for (int i = 0; i < LOADS; i++) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        if (WRITE) {                       // 3 x 16 B per lane per chunk = 3 KB of 20 KB
            float4 *o = out + chunk * 192 + lane;
            o[0] = acc; o[64] = acc; o[128] = acc;
        }
    }
    if (!WRITE && acc.x == 12345.f) out[wave] = acc;
}

template <int LOADS, bool WRITE>
void run(const char *name, const float4 *in, float4 *out, size_t bytes, int chunks_per_wave, int wg_lds) {
    const size_t n_chunks = bytes / (LOADS * 1024);
    const size_t waves = (n_chunks + chunks_per_wave - 1) / chunks_per_wave;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd<LOADS, WRITE>), dim3(unsigned((waves + 3) / 4)), dim3(256), wg_lds, 0, in, out, n_chunks, chunks_per_wave);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rdb = double(n_chunks) * LOADS * 1024, wrb = WRITE ? double(n_chunks) * 3072 : 0;
    printf("%-34s chunks/wave %3d lds %6d: %.3f ms  read %.0f GB/s  write %.0f GB/s  total %.0f GB/s\n", name, chunks_per_wave, wg_lds, ms,
           rdb / ms / 1e6, wrb / ms / 1e6, (rdb + wrb) / ms / 1e6);
}

int main() {
    const size_t bytes = size_t(24) << 30;
    float4 *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0, bytes);
    for (int cpw : {1, 8, 25, 75}) {
        run<20, false>("read-only, 20 KB chunks", in, out, bytes, cpw, 0);
        run<20, true>("read + 15% write, 20 KB chunks", in, out, bytes, cpw, 0);
    }
    // same occupancy as the OFDM kernel: 3 workgroups (12 waves) per CU via a 51 KB LDS request
    for (int cpw : {8, 25}) {
        run<20, false>("read-only, 12 waves/CU", in, out, bytes, cpw, 51 * 1024);
        run<20, true>("read + 15% write, 12 waves/CU", in, out, bytes, cpw, 51 * 1024);
    }
    run<16, false>("read-only, 16 KB chunks", in, out, bytes, 25, 0);
    runmix<1>(in, out, bytes); runmix<3>(in, out, bytes); runmix<6>(in, out, bytes); runmix<10>(in, out, bytes);
    runmix<16>(in, out, bytes); runmix<20>(in, out, bytes);
    runw<1, false>("read + write", in, out, bytes, 24);
    runw<1, true>("read + write", in, out, bytes, 24);
    runw<2, false>("read + write", in, out, bytes, 24);
    runw<4, false>("read + write", in, out, bytes, 24);
    runw<4, true>("read + write", in, out, bytes, 24);
    runw<8, false>("read + write", in, out, bytes, 24);
    runw<8, true>("read + write", in, out, bytes, 24);
    return 0;
}
