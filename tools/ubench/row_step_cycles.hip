// Micro-benchmark for the decoder that was NOT built: four codewords per wavefront, one 16-lane row
// each, four trellis states per lane -- the add-compare-select of one step for all four codewords, written as the kernel
// would write it (rotating state layout: in four of six phases the butterfly partner is a lane of the same row, reached
// by DPP; in the other two it is another register of the same lane), with the branch metrics and the survivor bits,
// against the step of the wave-per-codeword decoder that exists (csrc/viterbi_kernels.hip, rot_step: one state per lane).
// Measured: shader cycles per wave-step with 1, 2 and 4 waves per SIMD (one workgroup of 4k waves per CU), i.e. the
// rates a launch of 256 .. 4096 wavefronts sees.  What it shows is in profiles/r05_row_decoder_rejected.md.
// build: hipcc -O3 --offload-arch=gfx950 row_step_cycles.hip -o row_step_cycles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }

__device__ __forceinline__ void acs(int &m, int other, int c, int ct, unsigned &dec) {
    const int x = m + c, xt = m + ct, y = other - c;
    m = max(x, y);
    asm("v_cmp_gt_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(dec) : "v"(y), "v"(xt) : "vcc");
}

// one state per lane (the existing kernel's step, DPP phases only)
__global__ void k_lane_state(const int *soft, int *out, int steps) {
    const int lane = threadIdx.x & 63;
    int m = lane == 0 ? 0 : -4096;
    unsigned dec = 0;
    const int tab = 0x01FF01FF ^ (lane * 0x01010101), thr = -(lane & 1);
    int w = soft[lane & 15];
    for (int t = 0; t < steps; t += 4) {
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {
            const int c = __builtin_amdgcn_sdot4(tab, w, 0, false), ct = c + thr;
            int other;
            if (ph == 0) other = dpp<0xB1>(m); else if (ph == 1) other = dpp<0x4E>(m); else if (ph == 2) other = dpp<0x141>(m); else other = dpp<0x128>(m);
            acs(m, other, c, ct, dec);
            w = __builtin_amdgcn_alignbit(w, w, 8) + int(dec & 1);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = m + int(dec);
}

// four states per lane, one codeword per 16-lane row: four DPP phases, two in-lane phases
__global__ void k_row_states(const int *soft, int *out, int steps) {
    const int lane = threadIdx.x & 63;
    int m[4];
    unsigned dec[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) m[k] = (lane & 15) + k == 0 ? 0 : -4096;
    const int tab = 0x01FF01FF ^ (lane * 0x01010101);
    const int thr0 = -(lane & 1), thr1 = -((lane >> 1) & 1);
    int w = soft[lane >> 4];                                   // each row its own codeword's soft word
    for (int t = 0; t < steps; t += 6) {
#pragma unroll
        for (int ph = 0; ph < 6; ph++) {
            // branch metrics of the lane's four states: one dot product, the other three by the slot bits' sign patterns
            const int a = __builtin_amdgcn_sdot4(tab, w, 0, false);
            const int b = int8_t(w >> 8) * 2, d = int8_t(w >> 16) * 2;
            const int c[4] = {a, a - b, a - d, a - b - d};
            int other[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (ph == 0) other[k] = dpp<0xB1>(m[k]);
                else if (ph == 1) other[k] = dpp<0x4E>(m[k]);
                else if (ph == 2) other[k] = dpp<0x141>(m[k]);
                else if (ph == 3) other[k] = dpp<0x128>(m[k]);
                else if (ph == 4) other[k] = m[k ^ 1];
                else other[k] = m[k ^ 2];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) acs(m[k], other[k], c[k], c[k] + ((k & 1) ? thr1 : thr0), dec[k]);
            w = __builtin_amdgcn_alignbit(w, w, 8) + int(dec[0] & 1);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = m[0] + m[1] + m[2] + m[3] + int(dec[0] ^ dec[1] ^ dec[2] ^ dec[3]);
}

int main() {
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) n_cu = prop.multiProcessorCount;
    int *soft, *out;
    (void)hipMalloc(&soft, 64 * sizeof(int));
    (void)hipMalloc(&out, size_t(n_cu) * 1024 * sizeof(int));
    std::vector<int> h(64);
    for (int i = 0; i < 64; i++) h[i] = 0x11223344 * (i + 1);
    (void)hipMemcpy(soft, h.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int steps = 1536 * 8;
    const double mhz = double(prop.clockRate) / 1e3;
    std::printf("%d CUs, clock %.0f MHz (nominal; cycles below use it), %d trellis steps per wave\n", n_cu, mhz, steps);
    std::printf("%-42s %14s %14s %14s\n", "cycles per wave-step at waves/SIMD =", "1", "2", "4");
    for (int which = 0; which < 2; which++) {
        double cyc[3];
        for (int ki = 0; ki < 3; ki++) {
            const int k = 1 << ki;
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                (void)hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_lane_state, dim3(n_cu), dim3(256 * k), 0, 0, soft, out, steps);
                else hipLaunchKernelGGL(k_row_states, dim3(n_cu), dim3(256 * k), 0, 0, soft, out, steps);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep) best = ms < best ? ms : best;
            }
            cyc[ki] = double(best) * 1e-3 * mhz * 1e6 / steps / k;          // per wave-step, per SIMD
        }
        std::printf("%-42s %14.1f %14.1f %14.1f\n", which == 0 ? "one state per lane (1 codeword per wave)" : "four states per lane (4 codewords per wave)",
                    cyc[0], cyc[1], cyc[2]);
        if (which == 1)
            std::printf("%-42s %14.1f %14.1f %14.1f\n", "   ... per codeword-step (/4)", cyc[0] / 4, cyc[1] / 4, cyc[2] / 4);
    }
    return 0;
}
