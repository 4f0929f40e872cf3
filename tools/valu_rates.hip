// valu_rates.hip -- issue-rate microbenchmark for the instruction classes the solver kernels are made
// of (gfx950).  Answers the question DESIGN.md section 5 rests on: how many cycles of a SIMD does one
// wave64 instruction of each class take, with one and with two wavefronts per SIMD?
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates tools/valu_rates.hip && ./valu_rates out.json
//
// Method.  One workgroup of 4 * W wavefronts per CU (W wavefronts per SIMD; the dispatcher places the
// wavefronts of a workgroup round-robin on the four SIMDs), every wavefront runs ITERS iterations of
// a block of 128 instructions of ONE class written in assembly on fixed registers:
//   "thr": eight independent register sets (throughput: nothing waits for a result)
//   "dep": every instruction reads the previous one's result (latency of a dependent chain)
// and reads the shader clock (s_memtime) before and after.  cycles per instruction of a wavefront =
// (t1 - t0) / (ITERS * 128); a SIMD with W wavefronts issues W instructions in that time.  The
// constant-rate counter (s_memrealtime, 100 MHz) is read as well, which gives the shader clock the
// run actually had.  The whole chip is loaded (one workgroup per CU) so that the clock is the one a
// full kernel sees.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(e)                                                                        \
    do {                                                                                \
        hipError_t e_ = (e);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                     \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

struct Sample {
    unsigned long long cycles, realtime;
};

#define CLOBBERS                                                                                           \
    "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",      \
        "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37",  \
        "v38", "v39", "v40", "v41", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48",  \
        "s49", "s50", "s51", "s52", "s53", "s54", "s55", "memory"

// registers: v[10:25] eight 64-bit accumulators, v[30:33] two 64-bit operands (1.0000001, 0.9999999),
// v34 / v35 32-bit operands, v[36:37] an LDS address pair
#define SETUP                                     \
    "v_mov_b32 v30, 0x00000001\n"                 \
    "v_mov_b32 v31, 0x3ff00000\n"                 \
    "v_mov_b32 v32, 0xffffffff\n"                 \
    "v_mov_b32 v33, 0x3fefffff\n"                 \
    "v_mov_b32 v34, 0x12345\n"                    \
    "v_mov_b32 v35, 0x54321\n"                    \
    "v_mov_b32 v36, 0\n"                          \
    "s_mov_b32 s40, 0x00000001\n"                 \
    "s_mov_b32 s41, 0x3ff00000\n"                 \
    "s_mov_b32 s44, 0x33333333\n"                 \
    "s_mov_b32 s45, 0x33333333\n"                 \
    "s_mov_b32 vcc_lo, 0x55555555\n"              \
    "s_mov_b32 vcc_hi, 0x55555555\n"              \
    "v_mov_b32 v10, v30\n v_mov_b32 v11, v31\n"   \
    "v_mov_b32 v12, v30\n v_mov_b32 v13, v31\n"   \
    "v_mov_b32 v14, v30\n v_mov_b32 v15, v31\n"   \
    "v_mov_b32 v16, v30\n v_mov_b32 v17, v31\n"   \
    "v_mov_b32 v18, v30\n v_mov_b32 v19, v31\n"   \
    "v_mov_b32 v20, v30\n v_mov_b32 v21, v31\n"   \
    "v_mov_b32 v22, v30\n v_mov_b32 v23, v31\n"   \
    "v_mov_b32 v24, v30\n v_mov_b32 v25, v31\n"

// a block = .rept 16 of eight instructions
#define BLOCK8(i0, i1, i2, i3, i4, i5, i6, i7) \
    ".rept 16\n" i0 "\n" i1 "\n" i2 "\n" i3 "\n" i4 "\n" i5 "\n" i6 "\n" i7 "\n.endr\n"

#define BENCH_KERNEL(name, body)                                                                    \
    __global__ __launch_bounds__(512) void name(Sample* out, int iters) {                           \
        __shared__ double lds[1024];                                                                 \
        lds[threadIdx.x] = 1.0;                                                                      \
        lds[threadIdx.x + 512] = 1.0;                                                                \
        __syncthreads();                                                                             \
        asm volatile(SETUP ::: CLOBBERS);                                                            \
        unsigned long long r0, r1, t0, t1;                                                           \
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)); \
        for (int i = 0; i < iters; ++i) asm volatile(body ::: CLOBBERS);                             \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" \
                     : "=s"(t1), "=s"(r1));                                                          \
        if ((threadIdx.x & 63) == 0) {                                                               \
            Sample s;                                                                                \
            s.cycles = t1 - t0;                                                                      \
            s.realtime = r1 - r0;                                                                    \
            out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;                              \
        }                                                                                            \
    }

// ---- throughput forms (eight independent destinations) ----
BENCH_KERNEL(k_fma_f64_thr,
             BLOCK8("v_fma_f64 v[10:11], v[30:31], v[32:33], v[10:11]", "v_fma_f64 v[12:13], v[30:31], v[32:33], v[12:13]",
                    "v_fma_f64 v[14:15], v[30:31], v[32:33], v[14:15]", "v_fma_f64 v[16:17], v[30:31], v[32:33], v[16:17]",
                    "v_fma_f64 v[18:19], v[30:31], v[32:33], v[18:19]", "v_fma_f64 v[20:21], v[30:31], v[32:33], v[20:21]",
                    "v_fma_f64 v[22:23], v[30:31], v[32:33], v[22:23]", "v_fma_f64 v[24:25], v[30:31], v[32:33], v[24:25]"))
BENCH_KERNEL(k_fma_f64_sgpr_thr,
             BLOCK8("v_fma_f64 v[10:11], v[10:11], v[32:33], s[40:41]", "v_fma_f64 v[12:13], v[12:13], v[32:33], s[40:41]",
                    "v_fma_f64 v[14:15], v[14:15], v[32:33], s[40:41]", "v_fma_f64 v[16:17], v[16:17], v[32:33], s[40:41]",
                    "v_fma_f64 v[18:19], v[18:19], v[32:33], s[40:41]", "v_fma_f64 v[20:21], v[20:21], v[32:33], s[40:41]",
                    "v_fma_f64 v[22:23], v[22:23], v[32:33], s[40:41]", "v_fma_f64 v[24:25], v[24:25], v[32:33], s[40:41]"))
BENCH_KERNEL(k_mul_f64_thr,
             BLOCK8("v_mul_f64 v[10:11], v[10:11], v[32:33]", "v_mul_f64 v[12:13], v[12:13], v[30:31]",
                    "v_mul_f64 v[14:15], v[14:15], v[32:33]", "v_mul_f64 v[16:17], v[16:17], v[30:31]",
                    "v_mul_f64 v[18:19], v[18:19], v[32:33]", "v_mul_f64 v[20:21], v[20:21], v[30:31]",
                    "v_mul_f64 v[22:23], v[22:23], v[32:33]", "v_mul_f64 v[24:25], v[24:25], v[30:31]"))
BENCH_KERNEL(k_add_f64_thr,
             BLOCK8("v_add_f64 v[10:11], v[10:11], v[32:33]", "v_add_f64 v[12:13], v[12:13], -v[32:33]",
                    "v_add_f64 v[14:15], v[14:15], v[32:33]", "v_add_f64 v[16:17], v[16:17], -v[32:33]",
                    "v_add_f64 v[18:19], v[18:19], v[32:33]", "v_add_f64 v[20:21], v[20:21], -v[32:33]",
                    "v_add_f64 v[22:23], v[22:23], v[32:33]", "v_add_f64 v[24:25], v[24:25], -v[32:33]"))
BENCH_KERNEL(k_mov_b32_thr,
             BLOCK8("v_mov_b32 v10, v34", "v_mov_b32 v11, v35", "v_mov_b32 v12, v34", "v_mov_b32 v13, v35",
                    "v_mov_b32 v14, v34", "v_mov_b32 v15, v35", "v_mov_b32 v16, v34", "v_mov_b32 v17, v35"))
BENCH_KERNEL(k_mov_b64_thr,
             BLOCK8("v_mov_b64 v[10:11], v[30:31]", "v_mov_b64 v[12:13], v[32:33]", "v_mov_b64 v[14:15], v[30:31]",
                    "v_mov_b64 v[16:17], v[32:33]", "v_mov_b64 v[18:19], v[30:31]", "v_mov_b64 v[20:21], v[32:33]",
                    "v_mov_b64 v[22:23], v[30:31]", "v_mov_b64 v[24:25], v[32:33]"))
BENCH_KERNEL(k_cndmask_b32_thr,
             BLOCK8("v_cndmask_b32 v10, v34, v35, vcc", "v_cndmask_b32 v11, v35, v34, vcc", "v_cndmask_b32 v12, v34, v35, vcc",
                    "v_cndmask_b32 v13, v35, v34, vcc", "v_cndmask_b32 v14, v34, v35, vcc", "v_cndmask_b32 v15, v35, v34, vcc",
                    "v_cndmask_b32 v16, v34, v35, vcc", "v_cndmask_b32 v17, v35, v34, vcc"))
BENCH_KERNEL(k_add_u32_thr,
             BLOCK8("v_add_u32 v10, v10, v34", "v_add_u32 v11, v11, v35", "v_add_u32 v12, v12, v34", "v_add_u32 v13, v13, v35",
                    "v_add_u32 v14, v14, v34", "v_add_u32 v15, v15, v35", "v_add_u32 v16, v16, v34", "v_add_u32 v17, v17, v35"))
BENCH_KERNEL(k_xor_b32_thr,
             BLOCK8("v_xor_b32 v10, v10, v34", "v_xor_b32 v11, v11, v35", "v_xor_b32 v12, v12, v34", "v_xor_b32 v13, v13, v35",
                    "v_xor_b32 v14, v14, v34", "v_xor_b32 v15, v15, v35", "v_xor_b32 v16, v16, v34", "v_xor_b32 v17, v17, v35"))
BENCH_KERNEL(k_fma_f32_thr,
             BLOCK8("v_fma_f32 v10, v34, v35, v10", "v_fma_f32 v11, v34, v35, v11", "v_fma_f32 v12, v34, v35, v12",
                    "v_fma_f32 v13, v34, v35, v13", "v_fma_f32 v14, v34, v35, v14", "v_fma_f32 v15, v34, v35, v15",
                    "v_fma_f32 v16, v34, v35, v16", "v_fma_f32 v17, v34, v35, v17"))
BENCH_KERNEL(k_readlane_b32_thr,
             BLOCK8("v_readlane_b32 s42, v34, 3", "v_readlane_b32 s43, v35, 5", "v_readlane_b32 s44, v34, 7", "v_readlane_b32 s45, v35, 9",
                    "v_readlane_b32 s46, v34, 11", "v_readlane_b32 s47, v35, 13", "v_readlane_b32 s48, v34, 15", "v_readlane_b32 s49, v35, 17"))
BENCH_KERNEL(k_writelane_b32_thr,
             BLOCK8("v_writelane_b32 v10, s40, 3", "v_writelane_b32 v11, s41, 5", "v_writelane_b32 v12, s40, 7", "v_writelane_b32 v13, s41, 9",
                    "v_writelane_b32 v14, s40, 11", "v_writelane_b32 v15, s41, 13", "v_writelane_b32 v16, s40, 15", "v_writelane_b32 v17, s41, 17"))
BENCH_KERNEL(k_mad_u64_u32_thr,
             BLOCK8("v_mad_u64_u32 v[10:11], s[42:43], v34, v35, v[10:11]", "v_mad_u64_u32 v[12:13], s[44:45], v34, v35, v[12:13]",
                    "v_mad_u64_u32 v[14:15], s[46:47], v34, v35, v[14:15]", "v_mad_u64_u32 v[16:17], s[48:49], v34, v35, v[16:17]",
                    "v_mad_u64_u32 v[18:19], s[42:43], v34, v35, v[18:19]", "v_mad_u64_u32 v[20:21], s[44:45], v34, v35, v[20:21]",
                    "v_mad_u64_u32 v[22:23], s[46:47], v34, v35, v[22:23]", "v_mad_u64_u32 v[24:25], s[48:49], v34, v35, v[24:25]"))
BENCH_KERNEL(k_mul_lo_u32_thr,
             BLOCK8("v_mul_lo_u32 v10, v34, v35", "v_mul_lo_u32 v11, v35, v34", "v_mul_lo_u32 v12, v34, v35", "v_mul_lo_u32 v13, v35, v34",
                    "v_mul_lo_u32 v14, v34, v35", "v_mul_lo_u32 v15, v35, v34", "v_mul_lo_u32 v16, v34, v35", "v_mul_lo_u32 v17, v35, v34"))
BENCH_KERNEL(k_cvt_f64_u32_thr,
             BLOCK8("v_cvt_f64_u32 v[10:11], v34", "v_cvt_f64_u32 v[12:13], v35", "v_cvt_f64_u32 v[14:15], v34", "v_cvt_f64_u32 v[16:17], v35",
                    "v_cvt_f64_u32 v[18:19], v34", "v_cvt_f64_u32 v[20:21], v35", "v_cvt_f64_u32 v[22:23], v34", "v_cvt_f64_u32 v[24:25], v35"))
BENCH_KERNEL(k_rsq_f64_thr,
             BLOCK8("v_rsq_f64 v[10:11], v[30:31]", "v_rsq_f64 v[12:13], v[32:33]", "v_rsq_f64 v[14:15], v[30:31]", "v_rsq_f64 v[16:17], v[32:33]",
                    "v_rsq_f64 v[18:19], v[30:31]", "v_rsq_f64 v[20:21], v[32:33]", "v_rsq_f64 v[22:23], v[30:31]", "v_rsq_f64 v[24:25], v[32:33]"))
BENCH_KERNEL(k_rcp_f64_thr,
             BLOCK8("v_rcp_f64 v[10:11], v[30:31]", "v_rcp_f64 v[12:13], v[32:33]", "v_rcp_f64 v[14:15], v[30:31]", "v_rcp_f64 v[16:17], v[32:33]",
                    "v_rcp_f64 v[18:19], v[30:31]", "v_rcp_f64 v[20:21], v[32:33]", "v_rcp_f64 v[22:23], v[30:31]", "v_rcp_f64 v[24:25], v[32:33]"))
BENCH_KERNEL(k_rndne_f64_thr,
             BLOCK8("v_rndne_f64 v[10:11], v[30:31]", "v_rndne_f64 v[12:13], v[32:33]", "v_rndne_f64 v[14:15], v[30:31]", "v_rndne_f64 v[16:17], v[32:33]",
                    "v_rndne_f64 v[18:19], v[30:31]", "v_rndne_f64 v[20:21], v[32:33]", "v_rndne_f64 v[22:23], v[30:31]", "v_rndne_f64 v[24:25], v[32:33]"))
BENCH_KERNEL(k_cmp_f64_thr,
             BLOCK8("v_cmp_lt_f64 vcc, v[30:31], v[32:33]", "v_cmp_gt_f64 vcc, v[30:31], v[32:33]", "v_cmp_lt_f64 vcc, v[30:31], v[32:33]",
                    "v_cmp_gt_f64 vcc, v[30:31], v[32:33]", "v_cmp_lt_f64 vcc, v[30:31], v[32:33]", "v_cmp_gt_f64 vcc, v[30:31], v[32:33]",
                    "v_cmp_lt_f64 vcc, v[30:31], v[32:33]", "v_cmp_gt_f64 vcc, v[30:31], v[32:33]"))
BENCH_KERNEL(k_max_f64_thr,
             BLOCK8("v_max_f64 v[10:11], v[10:11], v[32:33]", "v_max_f64 v[12:13], v[12:13], v[30:31]", "v_max_f64 v[14:15], v[14:15], v[32:33]",
                    "v_max_f64 v[16:17], v[16:17], v[30:31]", "v_max_f64 v[18:19], v[18:19], v[32:33]", "v_max_f64 v[20:21], v[20:21], v[30:31]",
                    "v_max_f64 v[22:23], v[22:23], v[32:33]", "v_max_f64 v[24:25], v[24:25], v[30:31]"))
BENCH_KERNEL(k_ds_read_b64_thr,
             BLOCK8("ds_read_b64 v[10:11], v36", "ds_read_b64 v[12:13], v36 offset:8", "ds_read_b64 v[14:15], v36 offset:16",
                    "ds_read_b64 v[16:17], v36 offset:24", "ds_read_b64 v[18:19], v36 offset:32", "ds_read_b64 v[20:21], v36 offset:40",
                    "ds_read_b64 v[22:23], v36 offset:48", "ds_read_b64 v[24:25], v36 offset:56\n s_waitcnt lgkmcnt(4)"))
BENCH_KERNEL(k_ds_bpermute_b32_thr,
             BLOCK8("ds_bpermute_b32 v10, v36, v34", "ds_bpermute_b32 v11, v36, v35", "ds_bpermute_b32 v12, v36, v34", "ds_bpermute_b32 v13, v36, v35",
                    "ds_bpermute_b32 v14, v36, v34", "ds_bpermute_b32 v15, v36, v35", "ds_bpermute_b32 v16, v36, v34",
                    "ds_bpermute_b32 v17, v36, v35\n s_waitcnt lgkmcnt(4)"))
BENCH_KERNEL(k_mov_dpp_thr,
             BLOCK8("v_mov_b32_dpp v10, v34 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v11, v35 row_shr:1 row_mask:0xf bank_mask:0xf",
                    "v_mov_b32_dpp v12, v34 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v13, v35 row_shr:1 row_mask:0xf bank_mask:0xf",
                    "v_mov_b32_dpp v14, v34 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v15, v35 row_shr:1 row_mask:0xf bank_mask:0xf",
                    "v_mov_b32_dpp v16, v34 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v17, v35 row_shr:1 row_mask:0xf bank_mask:0xf"))
// the mix the solver kernels actually issue: FP64 arithmetic with 32-bit moves / selects in between
BENCH_KERNEL(k_mix_fma_mov_thr,
             BLOCK8("v_fma_f64 v[10:11], v[30:31], v[32:33], v[10:11]", "v_mov_b32 v26, v34", "v_fma_f64 v[12:13], v[30:31], v[32:33], v[12:13]",
                    "v_mov_b32 v27, v35", "v_fma_f64 v[14:15], v[30:31], v[32:33], v[14:15]", "v_cndmask_b32 v28, v34, v35, vcc",
                    "v_fma_f64 v[16:17], v[30:31], v[32:33], v[16:17]", "v_cndmask_b32 v29, v35, v34, vcc"))
// scalar instructions between vector ones: do they take vector issue slots of a lone wavefront?
BENCH_KERNEL(k_mix_fma_salu_thr,
             BLOCK8("v_fma_f64 v[10:11], v[30:31], v[32:33], v[10:11]", "s_mov_b32 s50, 0x1234", "v_fma_f64 v[12:13], v[30:31], v[32:33], v[12:13]",
                    "s_mov_b32 s51, 0x4321", "v_fma_f64 v[14:15], v[30:31], v[32:33], v[14:15]", "s_add_u32 s52, s50, s51",
                    "v_fma_f64 v[16:17], v[30:31], v[32:33], v[16:17]", "s_xor_b32 s53, s50, s51"))

// ---- v_cndmask_b32 in the contexts the kernels have it in (the plain back-to-back form above measured 16+
// cycles: which form is slow?) ----
BENCH_KERNEL(k_cndmask_e64_sgpr_thr,
             BLOCK8("v_cndmask_b32_e64 v10, v34, v35, s[44:45]", "v_cndmask_b32_e64 v11, v35, v34, s[44:45]",
                    "v_cndmask_b32_e64 v12, v34, v35, s[44:45]", "v_cndmask_b32_e64 v13, v35, v34, s[44:45]",
                    "v_cndmask_b32_e64 v14, v34, v35, s[44:45]", "v_cndmask_b32_e64 v15, v35, v34, s[44:45]",
                    "v_cndmask_b32_e64 v16, v34, v35, s[44:45]", "v_cndmask_b32_e64 v17, v35, v34, s[44:45]"))
BENCH_KERNEL(k_cndmask_pair_thr, // a 64-bit select: two v_cndmask_b32 on the halves, then an FP64 use
             BLOCK8("v_cndmask_b32 v10, v30, v32, vcc", "v_cndmask_b32 v11, v31, v33, vcc",
                    "v_fma_f64 v[18:19], v[10:11], v[32:33], v[18:19]", "v_cndmask_b32 v12, v30, v32, vcc",
                    "v_cndmask_b32 v13, v31, v33, vcc", "v_fma_f64 v[20:21], v[12:13], v[32:33], v[20:21]",
                    "v_fma_f64 v[22:23], v[30:31], v[32:33], v[22:23]", "v_fma_f64 v[24:25], v[30:31], v[32:33], v[24:25]"))
BENCH_KERNEL(k_cmp_cndmask_thr, // compare -> select, the way a ternary compiles
             BLOCK8("v_cmp_lt_f64 vcc, v[30:31], v[32:33]", "v_cndmask_b32 v10, v30, v32, vcc", "v_cndmask_b32 v11, v31, v33, vcc",
                    "v_fma_f64 v[18:19], v[30:31], v[32:33], v[18:19]", "v_cmp_gt_f64 vcc, v[30:31], v[32:33]",
                    "v_cndmask_b32 v12, v30, v32, vcc", "v_cndmask_b32 v13, v31, v33, vcc",
                    "v_fma_f64 v[20:21], v[30:31], v[32:33], v[20:21]"))
BENCH_KERNEL(k_cndmask_mov_alt_thr,
             BLOCK8("v_cndmask_b32 v10, v34, v35, vcc", "v_mov_b32 v18, v34", "v_cndmask_b32 v11, v35, v34, vcc", "v_mov_b32 v19, v35",
                    "v_cndmask_b32 v12, v34, v35, vcc", "v_mov_b32 v20, v34", "v_cndmask_b32 v13, v35, v34, vcc", "v_mov_b32 v21, v35"))
BENCH_KERNEL(k_cndmask_dep,
             BLOCK8("v_cndmask_b32 v10, v10, v35, vcc", "v_cndmask_b32 v10, v10, v34, vcc", "v_cndmask_b32 v10, v10, v35, vcc",
                    "v_cndmask_b32 v10, v10, v34, vcc", "v_cndmask_b32 v10, v10, v35, vcc", "v_cndmask_b32 v10, v10, v34, vcc",
                    "v_cndmask_b32 v10, v10, v35, vcc", "v_cndmask_b32 v10, v10, v34, vcc"))
BENCH_KERNEL(k_cndmask_x2_fma_thr, // two selects per FP64 instruction
             BLOCK8("v_cndmask_b32 v10, v34, v35, vcc", "v_cndmask_b32 v11, v35, v34, vcc", "v_fma_f64 v[18:19], v[30:31], v[32:33], v[18:19]",
                    "v_cndmask_b32 v12, v34, v35, vcc", "v_cndmask_b32 v13, v35, v34, vcc", "v_fma_f64 v[20:21], v[30:31], v[32:33], v[20:21]",
                    "v_cndmask_b32 v14, v34, v35, vcc", "v_cndmask_b32 v15, v35, v34, vcc"))
BENCH_KERNEL(k_salu_only_thr,
             BLOCK8("s_mov_b32 s50, 0x1234", "s_mov_b32 s51, 0x4321", "s_add_u32 s52, s50, s51", "s_xor_b32 s53, s50, s51",
                    "s_mov_b32 s54, 0x1234", "s_mov_b32 s55, 0x4321", "s_and_b32 s52, s50, s51", "s_or_b32 s53, s50, s51"))
BENCH_KERNEL(k_fma_f64_x3_salu_thr, // three FP64 instructions per scalar one
             BLOCK8("v_fma_f64 v[10:11], v[30:31], v[32:33], v[10:11]", "v_fma_f64 v[12:13], v[30:31], v[32:33], v[12:13]",
                    "v_fma_f64 v[14:15], v[30:31], v[32:33], v[14:15]", "s_mov_b32 s50, 0x1234",
                    "v_fma_f64 v[16:17], v[30:31], v[32:33], v[16:17]", "v_fma_f64 v[18:19], v[30:31], v[32:33], v[18:19]",
                    "v_fma_f64 v[20:21], v[30:31], v[32:33], v[20:21]", "s_mov_b32 s51, 0x4321"))
// ---- dependent chains (latency) ----
BENCH_KERNEL(k_fma_f64_dep,
             BLOCK8("v_fma_f64 v[10:11], v[10:11], v[32:33], v[30:31]", "v_fma_f64 v[10:11], v[10:11], v[32:33], v[30:31]",
                    "v_fma_f64 v[10:11], v[10:11], v[32:33], v[30:31]", "v_fma_f64 v[10:11], v[10:11], v[32:33], v[30:31]",
                    "v_fma_f64 v[10:11], v[10:11], v[32:33], v[30:31]", "v_fma_f64 v[10:11], v[10:11], v[32:33], v[30:31]",
                    "v_fma_f64 v[10:11], v[10:11], v[32:33], v[30:31]", "v_fma_f64 v[10:11], v[10:11], v[32:33], v[30:31]"))
BENCH_KERNEL(k_mul_f64_dep,
             BLOCK8("v_mul_f64 v[10:11], v[10:11], v[32:33]", "v_mul_f64 v[10:11], v[10:11], v[30:31]", "v_mul_f64 v[10:11], v[10:11], v[32:33]",
                    "v_mul_f64 v[10:11], v[10:11], v[30:31]", "v_mul_f64 v[10:11], v[10:11], v[32:33]", "v_mul_f64 v[10:11], v[10:11], v[30:31]",
                    "v_mul_f64 v[10:11], v[10:11], v[32:33]", "v_mul_f64 v[10:11], v[10:11], v[30:31]"))
BENCH_KERNEL(k_mov_b32_dep,
             BLOCK8("v_mov_b32 v10, v11", "v_mov_b32 v11, v10", "v_mov_b32 v10, v11", "v_mov_b32 v11, v10", "v_mov_b32 v10, v11",
                    "v_mov_b32 v11, v10", "v_mov_b32 v10, v11", "v_mov_b32 v11, v10"))
BENCH_KERNEL(k_rsq_f64_dep,
             BLOCK8("v_rsq_f64 v[10:11], v[10:11]", "v_rsq_f64 v[10:11], v[10:11]", "v_rsq_f64 v[10:11], v[10:11]", "v_rsq_f64 v[10:11], v[10:11]",
                    "v_rsq_f64 v[10:11], v[10:11]", "v_rsq_f64 v[10:11], v[10:11]", "v_rsq_f64 v[10:11], v[10:11]", "v_rsq_f64 v[10:11], v[10:11]"))
BENCH_KERNEL(k_ds_read_b64_dep,
             BLOCK8("ds_read_b64 v[10:11], v36\n s_waitcnt lgkmcnt(0)", "ds_read_b64 v[10:11], v36\n s_waitcnt lgkmcnt(0)",
                    "ds_read_b64 v[10:11], v36\n s_waitcnt lgkmcnt(0)", "ds_read_b64 v[10:11], v36\n s_waitcnt lgkmcnt(0)",
                    "ds_read_b64 v[10:11], v36\n s_waitcnt lgkmcnt(0)", "ds_read_b64 v[10:11], v36\n s_waitcnt lgkmcnt(0)",
                    "ds_read_b64 v[10:11], v36\n s_waitcnt lgkmcnt(0)", "ds_read_b64 v[10:11], v36\n s_waitcnt lgkmcnt(0)"))
BENCH_KERNEL(k_readlane_fma_dep, // VALU writes an SGPR, the next VALU reads it (the SGPR-spill pattern)
             BLOCK8("v_readlane_b32 s42, v30, 3\n v_readlane_b32 s43, v31, 3", "v_fma_f64 v[10:11], v[10:11], v[32:33], s[42:43]",
                    "v_readlane_b32 s44, v30, 5\n v_readlane_b32 s45, v31, 5", "v_fma_f64 v[12:13], v[12:13], v[32:33], s[44:45]",
                    "v_readlane_b32 s42, v30, 7\n v_readlane_b32 s43, v31, 7", "v_fma_f64 v[14:15], v[14:15], v[32:33], s[42:43]",
                    "v_readlane_b32 s44, v30, 9\n v_readlane_b32 s45, v31, 9", "v_fma_f64 v[16:17], v[16:17], v[32:33], s[44:45]"))

struct Entry {
    const char* name;
    void (*fn)(Sample*, int);
    int instr_per_block; // instructions of the measured class per loop iteration
    const char* note;
};

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : "valu_rates.json";
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int n_cu = prop.multiProcessorCount;
    const int iters = 2000;
    const Entry entries[] = {
        {"v_fma_f64", k_fma_f64_thr, 128, "thr"},
        {"v_fma_f64 (scalar addend)", k_fma_f64_sgpr_thr, 128, "thr"},
        {"v_mul_f64", k_mul_f64_thr, 128, "thr"},
        {"v_add_f64", k_add_f64_thr, 128, "thr"},
        {"v_max_f64", k_max_f64_thr, 128, "thr"},
        {"v_cmp_lt_f64", k_cmp_f64_thr, 128, "thr"},
        {"v_rndne_f64", k_rndne_f64_thr, 128, "thr"},
        {"v_rsq_f64", k_rsq_f64_thr, 128, "thr"},
        {"v_rcp_f64", k_rcp_f64_thr, 128, "thr"},
        {"v_cvt_f64_u32", k_cvt_f64_u32_thr, 128, "thr"},
        {"v_mov_b32", k_mov_b32_thr, 128, "thr"},
        {"v_mov_b64", k_mov_b64_thr, 128, "thr"},
        {"v_cndmask_b32", k_cndmask_b32_thr, 128, "thr"},
        {"v_add_u32", k_add_u32_thr, 128, "thr"},
        {"v_xor_b32", k_xor_b32_thr, 128, "thr"},
        {"v_fma_f32", k_fma_f32_thr, 128, "thr"},
        {"v_mov_b32_dpp", k_mov_dpp_thr, 128, "thr"},
        {"v_readlane_b32", k_readlane_b32_thr, 128, "thr"},
        {"v_writelane_b32", k_writelane_b32_thr, 128, "thr"},
        {"v_mad_u64_u32", k_mad_u64_u32_thr, 128, "thr"},
        {"v_mul_lo_u32", k_mul_lo_u32_thr, 128, "thr"},
        {"ds_read_b64", k_ds_read_b64_thr, 128, "thr"},
        {"ds_bpermute_b32", k_ds_bpermute_b32_thr, 128, "thr"},
        {"mix: v_fma_f64 + v_mov_b32/v_cndmask_b32 alternating", k_mix_fma_mov_thr, 128, "thr"},
        {"mix: v_fma_f64 + SALU alternating (vector instructions counted)", k_mix_fma_salu_thr, 64, "thr"},
        {"v_cndmask_b32_e64 (SGPR-pair mask)", k_cndmask_e64_sgpr_thr, 128, "thr"},
        {"mix: 64-bit select (2 v_cndmask_b32) feeding v_fma_f64, 4 cndmask + 4 fma per 8", k_cndmask_pair_thr, 128, "thr"},
        {"mix: v_cmp_f64 -> 2 v_cndmask_b32, + v_fma_f64 (2 cmp + 4 cndmask + 2 fma per 8)", k_cmp_cndmask_thr, 128, "thr"},
        {"mix: v_cndmask_b32 / v_mov_b32 alternating", k_cndmask_mov_alt_thr, 128, "thr"},
        {"mix: 2 v_cndmask_b32 per v_fma_f64 (6 + 2 per 8... see source)", k_cndmask_x2_fma_thr, 128, "thr"},
        {"SALU only (s_mov / s_add / s_xor ...)", k_salu_only_thr, 128, "thr"},
        {"mix: 3 v_fma_f64 per SALU (all 8 counted)", k_fma_f64_x3_salu_thr, 128, "thr"},
        {"v_cndmask_b32", k_cndmask_dep, 128, "dep"},
        {"v_fma_f64", k_fma_f64_dep, 128, "dep"},
        {"v_mul_f64", k_mul_f64_dep, 128, "dep"},
        {"v_mov_b32", k_mov_b32_dep, 128, "dep"},
        {"v_rsq_f64", k_rsq_f64_dep, 128, "dep"},
        {"ds_read_b64 (+ wait)", k_ds_read_b64_dep, 128, "dep"},
        {"v_readlane_b32 x2 -> v_fma_f64 reading the SGPR pair (3 instructions per step)", k_readlane_fma_dep, 192, "dep"},
    };
    Sample* d_out = nullptr;
    const int max_waves = n_cu * 8;
    CHECK(hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(Sample) * max_waves));
    std::vector<Sample> h(max_waves);
    std::string js = "{\n \"device\": \"" + std::string(prop.gcnArchName) + "\", \"compute_units\": " + std::to_string(n_cu) +
                     ",\n \"method\": \"tools/valu_rates.hip: one workgroup of 4 W wavefronts per CU (W per SIMD), " +
                     std::to_string(iters) + " iterations of a 128-instruction assembly block per wavefront, s_memtime around the loop; "
                     "cycles_per_instr = median over wavefronts of (t1 - t0) / instructions; a SIMD issues W instructions of W wavefronts in that time\",\n"
                     " \"rows\": [\n";
    bool first = true;
    for (const Entry& e : entries) {
        for (int W = 1; W <= 2; ++W) {
            const int block = 64 * 4 * W;
            // warm-up + measured run
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(e.fn, dim3(n_cu), dim3(block), 0, 0, d_out, iters);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
            }
            const int waves = n_cu * 4 * W;
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Sample) * waves, hipMemcpyDeviceToHost));
            std::vector<double> cpi(waves), mhz(waves);
            for (int i = 0; i < waves; ++i) {
                cpi[i] = (double)h[i].cycles / ((double)iters * e.instr_per_block);
                mhz[i] = h[i].realtime ? (double)h[i].cycles / ((double)h[i].realtime / 100.0) : 0.0;
            }
            std::sort(cpi.begin(), cpi.end());
            std::sort(mhz.begin(), mhz.end());
            const double med = cpi[waves / 2], lo = cpi[0], hi = cpi[waves - 1];
            char buf[640];
            snprintf(buf, sizeof buf,
                     "%s  {\"instr\": \"%s\", \"form\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_wave\": %.4f, "
                     "\"min\": %.4f, \"max\": %.4f, \"simd_cycles_per_instr\": %.4f, \"shader_clock_mhz\": %.1f}",
                     first ? "" : ",\n", e.name, e.note, W, med, lo, hi, med / W, mhz[waves / 2]);
            js += buf;
            first = false;
            printf("%-78s %-3s W=%d  %.3f cycles/instr/wave  (%.3f SIMD cycles per instr)  clock %.0f MHz\n", e.name, e.note, W,
                   med, med / W, mhz[waves / 2]);
        }
    }
    js += "\n ]\n}\n";
    FILE* f = fopen(out_path, "w");
    if (f) {
        fputs(js.c_str(), f);
        fclose(f);
    }
    CHECK(hipFree(d_out));
    return 0;
}
