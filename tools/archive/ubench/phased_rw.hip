// Micro-benchmark: does the 13 % write stream of the OFDM kernel cost less when the whole chip writes in bursts?
// A mover of the kernel's geometry (every wave streams 20 KB "symbols" with 16-byte streaming loads and owes 3 KB of output
// per symbol, 12 waves per CU) holds its output back and stores it only while the chip-wide 100 MHz clock (s_memrealtime) is
// inside a window of W ticks every P ticks, or when NB symbols are pending (mode 1), or waits for the window (mode 2).
// build: hipcc -O3 --offload-arch=gfx950 phased_rw.hip -o phased_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4 __attribute__((ext_vector_type(4)));

__device__ inline v4 ld_nt(const v4 *p) { return __builtin_nontemporal_load(p); }
__device__ inline void st_nt(v4 *p, v4 v) { __builtin_nontemporal_store(v, p); }

template <int NB>
__global__ __launch_bounds__(256) void gated(const v4 *in, v4 *out, size_t n_chunks, int cpw, unsigned P, unsigned W, int mode,
                                              unsigned long long *stat) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const size_t wave = size_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    v4 acc = {0, 0, 0, 0};
    int pending = 0;
    unsigned forced = 0, in_window = 0, waited = 0;
    for (int c = 0; c < cpw; c++) {
        const size_t chunk = wave * cpw + c;
        if (chunk >= n_chunks) break;
        const v4 *p = in + chunk * (20 * 64) + lane;
        v4 v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = ld_nt(p + 64 * i);
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i];
        pending++;
        bool flush = mode == 0;
        if (!flush) {
            const bool full = pending == NB || c == cpw - 1 || chunk + 1 >= n_chunks;
            bool open = unsigned(wall_clock64() % P) < W;
            if (full && !open && mode == 2) {
                for (int spin = 0; spin < 4096 && !open; spin++) {
                    __builtin_amdgcn_s_sleep(8);
                    open = unsigned(wall_clock64() % P) < W;
                    waited++;
                }
            }
            flush = full || open;
            if (flush) { if (open) in_window++; else forced++; }
        }
        if (flush) {
            for (int k = pending - 1; k >= 0; k--) {
                v4 *o = out + (chunk - k) * 192 + lane;
                st_nt(o, acc); st_nt(o + 64, acc); st_nt(o + 128, acc);
            }
            pending = 0;
        }
    }
    if (lane == 0 && stat && mode) {
        atomicAdd(stat + 0, (unsigned long long)in_window);
        atomicAdd(stat + 1, (unsigned long long)forced);
        atomicAdd(stat + 2, (unsigned long long)waited);
    }
}

// lock step: every wave reads only outside the window and writes only inside it
__global__ __launch_bounds__(256) void lockstep(const v4 *in, v4 *out, size_t n_chunks, int cpw, unsigned P, unsigned W, int nb) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const size_t wave = size_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    v4 acc = {0, 0, 0, 0};
    int pending = 0;
    for (int c = 0; c < cpw; c++) {
        const size_t chunk = wave * cpw + c;
        if (chunk >= n_chunks) break;
        for (int spin = 0; spin < 4096 && unsigned(wall_clock64() % P) < W; spin++) __builtin_amdgcn_s_sleep(4);   // no reads in the write window
        const v4 *p = in + chunk * (20 * 64) + lane;
        v4 v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = ld_nt(p + 64 * i);
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i];
        pending++;
        if (pending == nb || c == cpw - 1 || chunk + 1 >= n_chunks) {
            for (int spin = 0; spin < 4096 && unsigned(wall_clock64() % P) >= W; spin++) __builtin_amdgcn_s_sleep(4);
            for (int k = pending - 1; k >= 0; k--) {
                v4 *o = out + (chunk - k) * 192 + lane;
                st_nt(o, acc); st_nt(o + 64, acc); st_nt(o + 128, acc);
            }
            pending = 0;
        }
    }
}

// lock step with the front end's arithmetic emulated: `work` rounds of 80 independent FMAs on the loaded registers per symbol
// (27 rounds ~ 4 300 VALU cycles per symbol-wave, what the fused kernel spends), so that a sleeping wave costs arithmetic too
__global__ __launch_bounds__(256) void lockstep_work(const v4 *in, v4 *out, size_t n_chunks, int cpw, unsigned P, unsigned W, int nb, int work) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const size_t wave = size_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    v4 acc = {0, 0, 0, 0};
    int pending = 0;
    for (int c = 0; c < cpw; c++) {
        const size_t chunk = wave * cpw + c;
        if (chunk >= n_chunks) break;
        if (W) for (int spin = 0; spin < 4096 && unsigned(wall_clock64() % P) < W; spin++) __builtin_amdgcn_s_sleep(4);
        const v4 *p = in + chunk * (20 * 64) + lane;
        v4 v[20];
#pragma unroll
        for (int i = 0; i < 20; i++) v[i] = ld_nt(p + 64 * i);
        for (int r = 0; r < work; r++) {
#pragma unroll
            for (int i = 0; i < 20; i++) v[i] = v[i] * 1.0001f + 0.5f;
        }
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i];
        pending++;
        if (pending == nb || c == cpw - 1 || chunk + 1 >= n_chunks) {
            if (W) for (int spin = 0; spin < 4096 && unsigned(wall_clock64() % P) >= W; spin++) __builtin_amdgcn_s_sleep(4);
            for (int k = pending - 1; k >= 0; k--) {
                v4 *o = out + (chunk - k) * 192 + lane;
                st_nt(o, acc); st_nt(o + 64, acc); st_nt(o + 128, acc);
            }
            pending = 0;
        }
    }
}

static float time_it(void (*launch)(void *), void *arg) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0); launch(arg); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

struct Args { const v4 *in; v4 *out; size_t n_chunks; int cpw; unsigned P, W; int mode, nb; unsigned long long *stat; unsigned grid; int lds; };
template <int NB> static void launch_gated(void *a_) {
    Args &a = *static_cast<Args *>(a_);
    hipLaunchKernelGGL((gated<NB>), dim3(a.grid), dim3(256), a.lds, 0, a.in, a.out, a.n_chunks, a.cpw, a.P, a.W, a.mode, a.stat);
}
static int g_work = 27;
static void launch_lock_work(void *a_) {
    Args &a = *static_cast<Args *>(a_);
    hipLaunchKernelGGL(lockstep_work, dim3(a.grid), dim3(256), a.lds, 0, a.in, a.out, a.n_chunks, a.cpw, a.P, a.W, a.nb, g_work);
}
static void launch_lock(void *a_) {
    Args &a = *static_cast<Args *>(a_);
    hipLaunchKernelGGL(lockstep, dim3(a.grid), dim3(256), a.lds, 0, a.in, a.out, a.n_chunks, a.cpw, a.P, a.W, a.nb);
}

int main(int argc, char **argv) {
    const size_t bytes = size_t(24) << 30;
    v4 *in, *out; unsigned long long *stat;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes / 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&stat, 64);
    hipMemset(in, 0, bytes);
    Args a{in, out, bytes / (20 * 1024), 25, 0, 0, 0, 1, stat, 0, 51 * 1024};
    a.grid = unsigned(((a.n_chunks + a.cpw - 1) / a.cpw + 3) / 4);
    const double gb = double(a.n_chunks) * (20 * 1024 + 3072) / 1e9;
    a.mode = 0;
    float base = time_it(launch_gated<1>, &a);
    printf("%.2f GB per launch (reads 20 KB + writes 3 KB per symbol, streaming), 12 waves per CU, 25 symbols per wave\n", gb);
    printf("plain (each symbol's 3 KB stored at once)                       %.3f ms  %.0f GB/s\n", base, gb / base * 1e3);
    auto row = [&](const char *what, int nb, unsigned P_us, double share, float ms, bool stats) {
        unsigned long long h[3] = {0, 0, 0};
        if (stats) hipMemcpy(h, stat, 24, hipMemcpyDeviceToHost);
        printf("%-22s hold <= %d  period %4u us  window %2.0f %%   %.3f ms  %.0f GB/s  (%+.1f %%)", what, nb, P_us, share * 100, ms, gb / ms * 1e3,
               (ms / base - 1) * 100);
        if (stats) printf("   flushes in window %llu forced %llu sleeps %llu", h[0], h[1], h[2]);
        printf("\n");
    };
    const bool full = argc > 1;
    if (full) for (int mode : {1, 2})
        for (unsigned P_us : {4u, 10u, 20u, 40u, 80u, 160u})
            for (double share : {0.15, 0.3}) {
                a.mode = mode; a.P = P_us * 100; a.W = unsigned(a.P * share);
                const char *what = mode == 1 ? "gated, forced if full" : "gated, waits if full";
                hipMemset(stat, 0, 24); float t2 = time_it(launch_gated<2>, &a); hipMemset(stat, 0, 24); launch_gated<2>(&a); hipDeviceSynchronize(); row(what, 2, P_us, share, t2, true);
                hipMemset(stat, 0, 24); float t4 = time_it(launch_gated<4>, &a); hipMemset(stat, 0, 24); launch_gated<4>(&a); hipDeviceSynchronize(); row(what, 4, P_us, share, t4, true);
                hipMemset(stat, 0, 24); float t8 = time_it(launch_gated<8>, &a); hipMemset(stat, 0, 24); launch_gated<8>(&a); hipDeviceSynchronize(); row(what, 8, P_us, share, t8, true);
            }
    if (full) for (int nb : {1, 2, 4, 8})
        for (unsigned P_us : {10u, 20u, 40u, 80u, 160u})
            for (double share : {0.15, 0.25}) {
                a.nb = nb; a.P = P_us * 100; a.W = unsigned(a.P * share);
                row("lock step", nb, P_us, share, time_it(launch_lock, &a), false);
            }
    for (int work : {14, 27, 40}) {
        g_work = work;
        a.nb = 1; a.P = 1000; a.W = 0;
        const float wbase = time_it(launch_lock_work, &a);
        printf("with %d rounds of 80 FMAs per symbol, no gating: %.3f ms  %.0f GB/s\n", work, wbase, gb / wbase * 1e3);
        for (int nb : {2, 3, 4, 8})
            for (unsigned P_us : {12u, 16u, 20u, 24u, 32u, 40u, 60u})
                for (double share : {0.15, 0.25}) {
                    a.nb = nb; a.P = P_us * 100; a.W = unsigned(a.P * share);
                    const float ms = time_it(launch_lock_work, &a);
                    printf("  lock step + work %d  hold <= %d  period %4u us  window %2.0f %%   %.3f ms  (%+.1f %%)\n", work, nb, P_us, share * 100, ms, (ms / wbase - 1) * 100);
                }
    }
    a.mode = 0;
    float again = time_it(launch_gated<1>, &a);
    printf("plain again                                                      %.3f ms\n", again);
    return 0;
}
