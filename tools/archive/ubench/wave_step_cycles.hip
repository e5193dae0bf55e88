// Micro-benchmark: what one trellis step of the wave-per-codeword Viterbi (csrc/viterbi_kernels.hip, rot_step) costs a
// wavefront that runs ALONE on its SIMD -- the situation of the plugin's one frame per call -- measured inside the kernel
// with s_memtime (shader-clock ticks).  (a) issue cost of each instruction kind of the step, eight independent copies in a
// row; (b) the step itself as the compiler emits it (dependent chain metric -> exchange/subtract -> max, compare ->
// s_xor -> two v_writelane, the branch metric's v_mov + v_dot4c beside it), for the single-DPP phases and for the
// v_permlane32_swap phase.  build: hipcc -O3 --offload-arch=gfx950 wave_step_cycles.hip -o wave_step_cycles
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned long long u64;

#define REP8(X) X X X X X X X X

enum { ADD, SUB_DPP, MAX, CMP_SGPR, CMP_XOR_WRITELANE, WRITELANE, MOV_DOT4C, PERMLANE32, LDS_BCAST, STEP_DPP, STEP_PERMLANE,
       CMP_WRITELANE, CMP_ADDC, CMP_FAR_XOR_WRITELANE, CMP_XOR_FAR_WRITELANE, CMP_XOR, XOR_WRITELANE, STEP_DPP_LATE, STEP_DPP_LATE_RAW, STEP_DPP_LATE_X32, STEP_DPP_ADDC, STEP_DPP_NONE, NMODES };
static const char *NAMES[NMODES] = {
    "v_add_u32 (8 independent)", "v_sub_u32_dpp quad_perm (8 independent)", "v_max_i32 (8 independent)",
    "v_cmp_gt_i32_e64 -> SGPR pair (8 independent)", "v_cmp -> s_xor_b64 -> 2 x v_writelane (per group)", "v_writelane_b32 from a ready SGPR",
    "v_mov_b32 0 + v_dot4c_i32_i8 (per pair)", "2 x v_mov + v_permlane32_swap + v_cndmask (per group)", "ds_read2_b32 broadcast + wait (per read)",
    "whole step, single-DPP phase", "whole step, v_permlane32_swap phase",
    "v_cmp -> s_nop 1 -> 2 x v_writelane (no s_xor)", "v_cmp -> vcc -> v_addc_co (bit into the lane's own word)",
    "v_cmp, 6 x v_add, s_xor, 2 x v_writelane", "v_cmp, s_xor, 6 x v_add, 2 x v_writelane", "v_cmp -> s_xor_b64 only", "s_xor_b64 -> 2 x v_writelane only",
    "whole step, single-DPP, hand-over one step late", "whole step, late, raw ballot (s_nop 1 + 2 x v_writelane, no s_xor)",
    "whole step, late, 2 x s_xor_b32 literal + 2 x v_writelane", "whole step, v_cmp -> vcc -> v_addc_co instead of the hand-over",
    "whole step without any hand-over (v_cmp only)"};

template <int MODE>
__global__ __launch_bounds__(64) void k(u64 *ticks, int iters) {
    __shared__ int lds[256];
    lds[threadIdx.x] = threadIdx.x * 7;
    __syncthreads();
    int m0 = threadIdx.x, m1 = m0 + 1, m2 = m0 + 2, m3 = m0 + 3, m4 = m0 + 4, m5 = m0 + 5, m6 = m0 + 6, m7 = m0 + 7;
    int c = 5 - int(threadIdx.x & 3), kk = int(threadIdx.x & 1), tab = 0x01ff01ff, w = 0x11223344, vm = 0;
    int x = 0, xt = 0, y = 0, cacc = 0, t0r = 0, t1r = 0;
    u64 lr = 0;
    const u64 mask = 0xAAAAAAAAAAAAAAAAull;
    const bool upper = threadIdx.x >= 32;
    const u64 t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == ADD) {
            asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7) : "v"(c));
        } else if constexpr (MODE == SUB_DPP) {
            asm volatile(REP8("v_sub_u32_dpp %0, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_u32_dpp %2, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_u32_dpp %4, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_u32_dpp %6, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_u32_dpp %1, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_u32_dpp %3, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_u32_dpp %5, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_u32_dpp %7, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7) : "v"(c));
        } else if constexpr (MODE == MAX) {
            asm volatile(REP8("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n"
                              "v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8\n")
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7) : "v"(c));
        } else if constexpr (MODE == CMP_SGPR) {
            asm volatile(REP8("v_cmp_gt_i32_e64 s[20:21], %0, %1\n v_cmp_gt_i32_e64 s[22:23], %2, %3\n v_cmp_gt_i32_e64 s[24:25], %4, %5\n"
                              "v_cmp_gt_i32_e64 s[26:27], %6, %7\n v_cmp_gt_i32_e64 s[28:29], %1, %0\n v_cmp_gt_i32_e64 s[30:31], %3, %2\n"
                              "v_cmp_gt_i32_e64 s[36:37], %5, %4\n v_cmp_gt_i32_e64 s[34:35], %7, %6\n")
                         :: "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s36", "s37", "s34", "s35");
        } else if constexpr (MODE == CMP_XOR_WRITELANE) {
            asm volatile(REP8("v_cmp_gt_i32_e64 s[20:21], %1, %2\n s_xor_b64 s[20:21], s[20:21], %3\n v_writelane_b32 %0, s20, 3\n v_writelane_b32 %0, s21, 35\n")
                         : "+v"(vm) : "v"(m0), "v"(m1), "s"(mask) : "s20", "s21", "scc");
        } else if constexpr (MODE == WRITELANE) {
            asm volatile(REP8("v_writelane_b32 %0, %1, 1\n v_writelane_b32 %0, %1, 2\n v_writelane_b32 %0, %1, 3\n v_writelane_b32 %0, %1, 4\n"
                              "v_writelane_b32 %0, %1, 5\n v_writelane_b32 %0, %1, 6\n v_writelane_b32 %0, %1, 7\n v_writelane_b32 %0, %1, 8\n")
                         : "+v"(vm) : "s"(iters));
        } else if constexpr (MODE == MOV_DOT4C) {
            asm volatile(REP8("v_mov_b32 %0, 0\n v_dot4c_i32_i8 %0, %2, %3\n v_mov_b32 %1, 0\n v_dot4c_i32_i8 %1, %2, %3\n"
                              "v_mov_b32 %0, 0\n v_dot4c_i32_i8 %0, %2, %3\n v_mov_b32 %1, 0\n v_dot4c_i32_i8 %1, %2, %3\n")
                         : "+v"(cacc), "+v"(x) : "v"(tab), "v"(w));
        } else if constexpr (MODE == PERMLANE32) {
            asm volatile(REP8("v_mov_b32 %1, %0\n v_mov_b32 %2, %0\n s_nop 1\n v_permlane32_swap_b32 %1, %2\n v_cndmask_b32 %0, %1, %2, %3\n")
                         : "+v"(m0), "+v"(t0r), "+v"(t1r) : "s"(mask));
        } else if constexpr (MODE == LDS_BCAST) {
            asm volatile(REP8("ds_read2_b32 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)\n") : "=v"(lr) : "v"(16));
        } else if constexpr (MODE == STEP_DPP) {
            // eight steps: x = M + c; xt = x - k; y = dpp(M) - c; M = max(x, y); cmp y > xt; xor; two writelanes; beside it the
            // next branch metric (v_mov 0 + v_dot4c)
            asm volatile(REP8("v_add_u32 %1, %0, %5\n v_sub_u32 %2, %1, %6\n v_mov_b32 %4, 0\n"
                              "v_sub_u32_dpp %3, %0, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_dot4c_i32_i8 %4, %7, %8\n v_max_i32 %0, %1, %3\n v_cmp_gt_i32_e64 s[20:21], %3, %2\n"
                              "s_xor_b64 s[20:21], s[20:21], %10\n v_writelane_b32 %9, s20, 3\n v_writelane_b32 %9, s21, 35\n")
                         : "+v"(m0), "+v"(x), "+v"(xt), "+v"(y), "+v"(cacc) : "v"(c), "v"(kk), "v"(tab), "v"(w), "v"(vm), "s"(mask) : "s20", "s21", "scc");
        } else if constexpr (MODE == CMP_WRITELANE) {
            asm volatile(REP8("v_cmp_gt_i32_e64 s[20:21], %1, %2\n s_nop 1\n v_writelane_b32 %0, s20, 3\n v_writelane_b32 %0, s21, 35\n")
                         : "+v"(vm) : "v"(m0), "v"(m1) : "s20", "s21");
        } else if constexpr (MODE == CMP_ADDC) {
            asm volatile(REP8("v_cmp_gt_i32_e32 vcc, %1, %2\n v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n")
                         : "+v"(vm) : "v"(m0), "v"(m1) : "vcc");
        } else if constexpr (MODE == CMP_FAR_XOR_WRITELANE) {
            asm volatile(REP8("v_cmp_gt_i32_e64 s[20:21], %1, %2\n v_add_u32 %4, %4, %2\n v_add_u32 %5, %5, %2\n v_add_u32 %6, %6, %2\n"
                              "v_add_u32 %7, %7, %2\n v_add_u32 %8, %8, %2\n v_add_u32 %9, %9, %2\n"
                              "s_xor_b64 s[20:21], s[20:21], %3\n v_writelane_b32 %0, s20, 3\n v_writelane_b32 %0, s21, 35\n")
                         : "+v"(vm) : "v"(m0), "v"(m1), "s"(mask), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7) : "s20", "s21", "scc");
        } else if constexpr (MODE == CMP_XOR_FAR_WRITELANE) {
            asm volatile(REP8("v_cmp_gt_i32_e64 s[20:21], %1, %2\n s_xor_b64 s[20:21], s[20:21], %3\n v_add_u32 %4, %4, %2\n v_add_u32 %5, %5, %2\n"
                              "v_add_u32 %6, %6, %2\n v_add_u32 %7, %7, %2\n v_add_u32 %8, %8, %2\n v_add_u32 %9, %9, %2\n"
                              "v_writelane_b32 %0, s20, 3\n v_writelane_b32 %0, s21, 35\n")
                         : "+v"(vm) : "v"(m0), "v"(m1), "s"(mask), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7) : "s20", "s21", "scc");
        } else if constexpr (MODE == CMP_XOR) {
            asm volatile(REP8("v_cmp_gt_i32_e64 s[20:21], %0, %1\n s_xor_b64 s[20:21], s[20:21], %2\n")
                         :: "v"(m0), "v"(m1), "s"(mask) : "s20", "s21", "scc");
        } else if constexpr (MODE == XOR_WRITELANE) {
            asm volatile(REP8("s_xor_b64 s[20:21], s[20:21], %1\n v_writelane_b32 %0, s20, 3\n v_writelane_b32 %0, s21, 35\n")
                         : "+v"(vm) : "s"(mask) : "s20", "s21", "scc");
        } else if constexpr (MODE == STEP_DPP_LATE) {
            // as STEP_DPP, but the s_xor and the writelanes of a step sit behind the NEXT step's add / sub / dpp / dot
            asm volatile(REP8("v_add_u32 %1, %0, %5\n v_sub_u32 %2, %1, %6\n v_mov_b32 %4, 0\n"
                              "v_sub_u32_dpp %3, %0, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_dot4c_i32_i8 %4, %7, %8\n"
                              "s_xor_b64 s[22:23], s[20:21], %10\n v_max_i32 %0, %1, %3\n v_writelane_b32 %9, s22, 3\n v_writelane_b32 %9, s23, 35\n"
                              "v_cmp_gt_i32_e64 s[20:21], %3, %2\n")
                         : "+v"(m0), "+v"(x), "+v"(xt), "+v"(y), "+v"(cacc) : "v"(c), "v"(kk), "v"(tab), "v"(w), "v"(vm), "s"(mask)
                         : "s20", "s21", "s22", "s23", "scc");
        } else if constexpr (MODE == STEP_DPP_LATE_RAW) {
            asm volatile(REP8("v_add_u32 %1, %0, %5\n v_sub_u32 %2, %1, %6\n v_mov_b32 %4, 0\n"
                              "v_sub_u32_dpp %3, %0, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_dot4c_i32_i8 %4, %7, %8\n"
                              "s_nop 1\n v_writelane_b32 %9, s20, 3\n v_writelane_b32 %9, s21, 35\n v_max_i32 %0, %1, %3\n"
                              "v_cmp_gt_i32_e64 s[20:21], %3, %2\n")
                         : "+v"(m0), "+v"(x), "+v"(xt), "+v"(y), "+v"(cacc) : "v"(c), "v"(kk), "v"(tab), "v"(w), "v"(vm), "s"(mask)
                         : "s20", "s21", "s22", "s23", "scc");
        } else if constexpr (MODE == STEP_DPP_LATE_X32) {
            asm volatile(REP8("v_add_u32 %1, %0, %5\n v_sub_u32 %2, %1, %6\n v_mov_b32 %4, 0\n"
                              "v_sub_u32_dpp %3, %0, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_dot4c_i32_i8 %4, %7, %8\n"
                              "s_xor_b32 s22, s20, 0xcccccccc\n s_xor_b32 s23, s21, 0xcccccccc\n v_writelane_b32 %9, s22, 3\n v_writelane_b32 %9, s23, 35\n"
                              "v_max_i32 %0, %1, %3\n v_cmp_gt_i32_e64 s[20:21], %3, %2\n")
                         : "+v"(m0), "+v"(x), "+v"(xt), "+v"(y), "+v"(cacc) : "v"(c), "v"(kk), "v"(tab), "v"(w), "v"(vm), "s"(mask)
                         : "s20", "s21", "s22", "s23", "scc");
        } else if constexpr (MODE == STEP_DPP_ADDC) {
            asm volatile(REP8("v_add_u32 %1, %0, %6\n v_sub_u32 %2, %1, %7\n v_mov_b32 %4, 0\n"
                              "v_sub_u32_dpp %3, %0, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_dot4c_i32_i8 %4, %8, %9\n v_max_i32 %0, %1, %3\n v_cmp_gt_i32_e32 vcc, %3, %2\n"
                              "v_addc_co_u32_e32 %5, vcc, %5, %5, vcc\n")
                         : "+v"(m0), "+v"(x), "+v"(xt), "+v"(y), "+v"(cacc), "+v"(vm) : "v"(c), "v"(kk), "v"(tab), "v"(w) : "vcc");
        } else if constexpr (MODE == STEP_DPP_NONE) {
            asm volatile(REP8("v_add_u32 %1, %0, %5\n v_sub_u32 %2, %1, %6\n v_mov_b32 %4, 0\n"
                              "v_sub_u32_dpp %3, %0, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_dot4c_i32_i8 %4, %7, %8\n v_max_i32 %0, %1, %3\n v_cmp_gt_i32_e64 s[20:21], %3, %2\n")
                         : "+v"(m0), "+v"(x), "+v"(xt), "+v"(y), "+v"(cacc) : "v"(c), "v"(kk), "v"(tab), "v"(w) : "s20", "s21");
        } else if constexpr (MODE == STEP_PERMLANE) {
            asm volatile(REP8("v_add_u32 %1, %0, %5\n v_mov_b32 %11, %0\n v_mov_b32 %12, %0\n v_sub_u32 %2, %1, %6\n v_mov_b32 %4, 0\n"
                              "v_permlane32_swap_b32 %11, %12\n v_dot4c_i32_i8 %4, %7, %8\n v_cndmask_b32 %3, %11, %12, %10\n v_sub_u32 %3, %3, %5\n"
                              "v_max_i32 %0, %1, %3\n v_cmp_gt_i32_e64 s[20:21], %3, %2\n"
                              "s_xor_b64 s[20:21], s[20:21], %10\n v_writelane_b32 %9, s20, 3\n v_writelane_b32 %9, s21, 35\n")
                         : "+v"(m0), "+v"(x), "+v"(xt), "+v"(y), "+v"(cacc) : "v"(c), "v"(kk), "v"(tab), "v"(w), "v"(vm), "s"(mask), "v"(t0r), "v"(t1r)
                         : "s20", "s21", "scc");
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7 + x + xt + y + cacc + vm + t0r + t1r + int(lr) == 0x7fffffff && upper) ticks[63] = 1;   // keep everything alive
}

template <int MODE>
static void run(u64 *d, int per_iter) {
    const int iters = 2000;
    u64 h = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-60s %6.2f shader cycles each\n", NAMES[MODE], double(h) / (double(iters) * per_iter));
}

int main() {
    u64 *d;
    (void)hipMalloc(&d, 64 * sizeof(u64));
    printf("one wavefront alone on its SIMD (s_memtime ticks)\n");
    run<ADD>(d, 64); run<SUB_DPP>(d, 64); run<MAX>(d, 64); run<CMP_SGPR>(d, 64); run<CMP_XOR_WRITELANE>(d, 8); run<WRITELANE>(d, 64);
    run<MOV_DOT4C>(d, 32); run<PERMLANE32>(d, 8); run<LDS_BCAST>(d, 8); run<STEP_DPP>(d, 8); run<STEP_PERMLANE>(d, 8);
    run<CMP_WRITELANE>(d, 8); run<CMP_ADDC>(d, 8); run<CMP_FAR_XOR_WRITELANE>(d, 8); run<CMP_XOR_FAR_WRITELANE>(d, 8); run<CMP_XOR>(d, 8);
    run<XOR_WRITELANE>(d, 8); run<STEP_DPP_LATE>(d, 8); run<STEP_DPP_LATE_RAW>(d, 8); run<STEP_DPP_LATE_X32>(d, 8);
    run<STEP_DPP_ADDC>(d, 8); run<STEP_DPP_NONE>(d, 8);
    (void)hipFree(d);
    return 0;
}
