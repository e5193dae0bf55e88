// Micro-benchmark: calibration of SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE on gfx950 and the LDS cost of every
// access pattern of ofdm_wave_kernel, each pattern in a kernel of its own so that a rocprofv3 --pmc pass gives one
// row per pattern.  Each kernel also times itself with s_memtime: shader cycles per wave-instruction per CU at
// 12 waves per CU (3 workgroups x 4 waves, the fused kernel's occupancy).
// build: hipcc -O3 --offload-arch=gfx950 -I../../sdrplusplus-dab-radio-plugin_amd/csrc lds_conflict.hip -o lds_conflict
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#include "dab_tables.hpp"

using namespace dab;

enum Pat { RD64_FREE, RD64_2WAY, RD64_4WAY, WR64_FREE, WR64_2WAY, WR64_4WAY, EX1, EX2, TW1_NATURAL, TW2_NATURAL, TW1_TABLE, TW2_TABLE,
           SCATTER_B8, SCATTER_B16, NIDX_RD32, STG_RD128, NPAT };
static const char *NAMES[NPAT] = {
    "calib ds_read_b64 conflict-free (lane*8)", "calib ds_read_b64 2-way (lane*16)", "calib ds_read_b64 4-way (lane*32)",
    "calib ds_write_b64 conflict-free (lane*8)", "calib ds_write_b64 2-way (lane*16)", "calib ds_write_b64 4-way (lane*32)",
    "ofdm exchange 1 (32 wr64 + 32 rd64)", "ofdm exchange 2 (32 wr64 + 32 rd64)", "ofdm step-1 twiddles, natural table (15 rd64)",
    "ofdm step-2 twiddles, natural table (32 rd64)", "ofdm step-1 twiddles, [k1][n2] table (15 rd64)",
    "ofdm step-2 twiddles, product table layout (32 rd64)", "ofdm soft-bit scatter (48 wr8)", "soft-bit scatter as (re,im) pairs (24 wr16)",
    "ofdm nidx reads (12 rd32)", "ofdm staging reads (3 rd128)"};
static const int OPS[NPAT] = {32, 32, 32, 32, 32, 32, 64, 64, 15, 32, 15, 32, 48, 24, 12, 3};

struct Lds {
    float2 tw[2048];
    uint32_t nidx[12 * 64];
    float2 t1[15 * 16];
    float2 ex[4][1024];
};

template <int PAT>
__global__ __launch_bounds__(256, 3) void k(const float2 *twiddle, const uint32_t *nidx, unsigned long long *ticks, float *sink, int iters) {
    __shared__ Lds sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 256) sm.tw[i] = twiddle[i];
    for (int i = tid; i < 12 * 64; i += 256) sm.nidx[i] = nidx[i];
    if (tid < 240) sm.t1[tid] = twiddle[(8 * (tid & 15) * ((tid >> 4) + 1)) & 2047];
    for (int i = tid; i < 4096; i += 256) (&sm.ex[0][0])[i] = make_float2(float(i), 1.f);
    __syncthreads();
    float2 *ex = sm.ex[wave];
    const float2 *tw = sm.tw;
    int n2 = lane >> 2, p = lane & 3, k1v = lane & 15, kk = lane >> 4;
    float2 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = make_float2(float(lane + i), float(i));
    float2 acc = make_float2(0.f, 0.f);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        asm volatile("" : "+v"(n2), "+v"(p), "+v"(k1v), "+v"(kk));
        if constexpr (PAT == RD64_FREE || PAT == RD64_2WAY || PAT == RD64_4WAY) {
            const int st = PAT == RD64_FREE ? 1 : PAT == RD64_2WAY ? 2 : 4;
#pragma unroll
            for (int r = 0; r < 32; r++) { const float2 v = ex[(lane * st + r * 8) & 1023]; acc.x += v.x; acc.y += v.y; }
        } else if constexpr (PAT == WR64_FREE || PAT == WR64_2WAY || PAT == WR64_4WAY) {
            const int st = PAT == WR64_FREE ? 1 : PAT == WR64_2WAY ? 2 : 4;
#pragma unroll
            for (int r = 0; r < 32; r++) ex[(lane * st + r * 8) & 1023] = x[r & 15];
        } else if constexpr (PAT == EX1) {
            const int w1_base = p * 256, w1_r = n2 ^ (p << 2);
            const int r1b = p * 256 + n2 * 16, r1x = (p << 2) ^ ((n2 >> 1) & 3);
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
#pragma unroll
                for (int k1 = 0; k1 < 16; k1++) ex[w1_base + k1 * 16 + (w1_r ^ ((k1 >> 1) & 3))] = x[k1];
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = ex[r1b + (m ^ r1x)];
            }
        } else if constexpr (PAT == EX2) {
            const int w2_base = p * 256 + (n2 ^ (p << 2)), r2_base = kk * 16;
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) {
                    ex[w2_base + k2 * 16] = x[k2];
                    ex[w2_base + k2 * 16 + 128] = x[8 + k2];
                }
#pragma unroll
                for (int cc = 0; cc < 2; cc++)
#pragma unroll
                    for (int n3 = 0; n3 < 8; n3++) x[8 * cc + n3] = ex[n3 * 128 + cc * 64 + r2_base + (k1v ^ ((n3 >> 1) << 2))];
            }
        } else if constexpr (PAT == TW1_NATURAL) {
#pragma unroll
            for (int k1 = 1; k1 < 16; k1++) { const float2 t = tw[(8 * n2) * k1]; acc.x += t.x; acc.y += t.y; }
        } else if constexpr (PAT == TW2_NATURAL) {
            const int i0 = 2 * p * n2, st0 = 32 * p, i1 = i0 + n2, st1 = st0 + 16;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                const float2 a = tw[i0 + st0 * k2], b = tw[i1 + st1 * k2];
                acc.x += a.x + b.x; acc.y += a.y + b.y;
            }
        } else if constexpr (PAT == TW1_TABLE) {
#pragma unroll
            for (int k1 = 1; k1 < 16; k1++) { const float2 t = sm.t1[(k1 - 1) * 16 + n2]; acc.x += t.x; acc.y += t.y; }
        } else if constexpr (PAT == TW2_TABLE) {
            // the product's layout (ofdm_kernels.hip, WaveLds): even samples in blocks of 49 entries, odd ones in blocks of 64
            const int hi = n2 >> 3, kq = (n2 >> 2) & 1, klo = n2 & 3;
            const int b0 = p ? hi * 24 + kq * 12 + (p - 1) * 4 + klo : 48;
            const int b1 = 784 + hi * 32 + kq * 16 + p * 4 + klo;
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                const float2 a = tw[b0 + k2 * 49], b = tw[b1 + k2 * 64];
                acc.x += a.x + b.x; acc.y += a.y + b.y;
            }
        } else if constexpr (PAT == SCATTER_B8) {
            uint8_t *stg = reinterpret_cast<uint8_t *>(ex);
#pragma unroll
            for (int j = 0; j < 24; j++) {
                const uint32_t nd = sm.nidx[(j >> 1) * 64 + lane];
                const uint32_t ni = (j & 1) ? (nd >> 16) : (nd & 0xFFFFu);
                stg[ni] = uint8_t(j + it);
                stg[1536 + ni] = uint8_t(j - it);
            }
        } else if constexpr (PAT == SCATTER_B16) {
            uint16_t *stg = reinterpret_cast<uint16_t *>(ex);
#pragma unroll
            for (int j = 0; j < 24; j++) {
                const uint32_t nd = sm.nidx[(j >> 1) * 64 + lane];
                const uint32_t ni = (j & 1) ? (nd >> 16) : (nd & 0xFFFFu);
                stg[ni] = uint16_t(j + it);
            }
        } else if constexpr (PAT == NIDX_RD32) {
#pragma unroll
            for (int j = 0; j < 12; j++) acc.x += __uint_as_float(sm.nidx[j * 64 + lane]);
        } else {
            const uint4 *sv = reinterpret_cast<const uint4 *>(ex) + lane;
            const uint4 a = sv[0], b = sv[64], c = sv[128];
            acc.x += __uint_as_float(a.x ^ b.y ^ c.z);
            acc.y += __uint_as_float(a.w ^ b.x ^ c.y);
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 16; i++) { acc.x += x[i].x; acc.y += x[i].y; }
    if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
    sink[blockIdx.x * 256 + tid] = acc.x + acc.y + ex[lane].x;
}

static float2 *d_tw;
static uint32_t *d_nidx;
static unsigned long long *d_ticks;
static float *d_sink;

template <int PAT>
void run() {
    const int blocks = 256 * 3, iters = 2000;
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, d_tw, d_nidx, d_ticks, d_sink, 20);
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, d_tw, d_nidx, d_ticks, d_sink, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(size_t(blocks) * 4);
    (void)hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += double(v);
    s /= double(h.size());
    // 12 waves per CU share the LDS: cycles per wave-instruction per CU = wave cycles / (ops * iters * 12)
    printf("%-52s %7.1f cycles per wave per pass, %5.2f cycles per DS instruction per CU\n", NAMES[PAT], s / iters,
           s / iters / OPS[PAT] / 12.0);
}

int main() {
    std::vector<float2> tw(2048), twp(2048);
    for (int m = 0; m < 2048; m++) tw[m] = make_float2(float(cos(-2.0 * M_PI * m / 2048.0)), float(sin(-2.0 * M_PI * m / 2048.0)));
    const std::vector<int32_t> mapper = make_mapper();
    std::vector<int> n_of_bin(2048, -1);
    for (int n = 0; n < NB_CARRIERS; n++) n_of_bin[carrier_bin(mapper[n])] = n;
    std::vector<uint16_t> nvj(24 * 64);
    for (int j = 0; j < 24; j++)
        for (int v = 0; v < 64; v++) {
            int bin = v + 64 * (j < 12 ? j : j + 8);
            if (j == 0 && v == 0) bin = 768;
            nvj[((j >> 1) * 64 + v) * 2 + (j & 1)] = uint16_t(n_of_bin[bin]);
        }
    (void)hipMalloc(&d_tw, 2048 * 8);
    (void)hipMalloc(&d_nidx, 24 * 64 * 2);
    (void)hipMalloc(&d_ticks, 768 * 4 * 8);
    (void)hipMalloc(&d_sink, 768 * 256 * 4);
    (void)hipMemcpy(d_tw, tw.data(), 2048 * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_nidx, nvj.data(), 24 * 64 * 2, hipMemcpyHostToDevice);
    run<RD64_FREE>(); run<RD64_2WAY>(); run<RD64_4WAY>(); run<WR64_FREE>(); run<WR64_2WAY>(); run<WR64_4WAY>();
    run<EX1>(); run<EX2>(); run<TW1_NATURAL>(); run<TW2_NATURAL>(); run<TW1_TABLE>(); run<TW2_TABLE>();
    run<SCATTER_B8>(); run<SCATTER_B16>(); run<NIDX_RD32>(); run<STG_RD128>();
    return 0;
}
