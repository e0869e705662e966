// valu_issue_bench.hip -- what does one wave64 instruction COST a gfx950 SIMD, per instruction class?
//
// VERDICT r1 asked for the measured VALU ceiling of the render kernel instead of an assumed "4 cycles per wave
// instruction".  This micro-benchmark issues long runs of independent instructions of ONE class from W waves per SIMD
// (W = 1, 2, 5, 6: the render kernel ran at 5 in rounds 1-2, runs at 6 since round 3) on every SIMD of the chip and reports
//     cycles per wave-instruction per SIMD = wave's elapsed shader cycles (s_memtime) / (instructions x W),
// the wall-clock rate, and the shader clock (s_memtime ticks / wall time).  The render kernel's instruction mix
// (rocprofv3 SQ_INSTS_VALU_* counters) priced with these costs is the "weighted issue floor" in bench.py's
// roofline.valu block (scripts/make_pmc_json.py).
//
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/bin/valu_issue_bench scripts/valu_issue_bench.hip
// run:   scripts/bin/valu_issue_bench > profiles/r03/valu_issue_costs.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int UNROLL = 32;   // independent instructions per loop body (eight destinations, four rounds)
constexpr int ITERS = 8192;  // loop trips: 262144 instructions per wave

// eight independent destinations so that no instruction waits for the previous one's result; the eight instructions
// of a round sit in ONE asm statement (between separate statements that clobber vcc the compiler puts an s_nop)
#define OUT8(c) [d0] c(d0), [d1] c(d1), [d2] c(d2), [d3] c(d3), [d4] c(d4), [d5] c(d5), [d6] c(d6), [d7] c(d7)
#define IN16 [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5), [a6] "v"(a6), [a7] "v"(a7), \
             [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [b4] "v"(b4), [b5] "v"(b5), [b6] "v"(b6), [b7] "v"(b7)
#define CV(x) "+v"(x)
#define CS(x) "+s"(x)
#define ROUND8(S, C) asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) : OUT8(C) : IN16 : "vcc", "scc", "s20", "s21", "s22", "s23");
#define REP32(S) ROUND8(S, CV) ROUND8(S, CV) ROUND8(S, CV) ROUND8(S, CV)
#define REP32S(S) ROUND8(S, CS) ROUND8(S, CS) ROUND8(S, CS) ROUND8(S, CS)

#define DECL_F32(p, v) float p##0 = v, p##1 = v + 1, p##2 = v + 2, p##3 = v + 3, p##4 = v + 4, p##5 = v + 5, p##6 = v + 6, p##7 = v + 7;
#define DECL_F64(p, v) double p##0 = v, p##1 = v + 1, p##2 = v + 2, p##3 = v + 3, p##4 = v + 4, p##5 = v + 5, p##6 = v + 6, p##7 = v + 7;
#define DECL_U32(p, v) unsigned p##0 = v, p##1 = v + 1, p##2 = v + 2, p##3 = v + 3, p##4 = v + 4, p##5 = v + 5, p##6 = v + 6, p##7 = v + 7;
#define SINK8(p) (p##0 + p##1 + p##2 + p##3 + p##4 + p##5 + p##6 + p##7)

// one instruction of round position k
#define D(k) "%[d" #k "]"
#define A(k) "%[a" #k "]"
#define B(k) "%[b" #k "]"
#define I_ADD_F32(k) "v_add_f32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_MUL_F32(k) "v_mul_f32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_FMA_F32(k) "v_fma_f32 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_MIN_F32(k) "v_min_f32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_SQRT_F32(k) "v_sqrt_f32 " D(k) ", " A(k) "\n"
#define I_CMP_F32(k) "v_cmp_lt_f32 vcc, " A(k) ", " B(k) "\n"
#define I_CNDMASK(k) "v_cndmask_b32 " D(k) ", " A(k) ", " B(k) ", vcc\n"
#define I_MOV(k) "v_mov_b32 " D(k) ", " A(k) "\n"
#define I_AND(k) "v_and_b32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_ADD_U32(k) "v_add_u32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_LSHL_ADD(k) "v_lshl_add_u32 " D(k) ", " A(k) ", 2, " B(k) "\n"
#define I_MUL_LO(k) "v_mul_lo_u32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_ADD_F64(k) "v_add_f64 " D(k) ", " A(k) ", " B(k) "\n"
#define I_MUL_F64(k) "v_mul_f64 " D(k) ", " A(k) ", " B(k) "\n"
#define I_FMA_F64(k) "v_fma_f64 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_MIN_F64(k) "v_min_f64 " D(k) ", " A(k) ", " B(k) "\n"
#define I_MAX_F64(k) "v_max_f64 " D(k) ", " A(k) ", " B(k) "\n"
#define I_CMP_F64(k) "v_cmp_lt_f64 vcc, " A(k) ", " B(k) "\n"
#define I_RCP_F64(k) "v_rcp_f64 " D(k) ", " A(k) "\n"
#define I_RSQ_F64(k) "v_rsq_f64 " D(k) ", " A(k) "\n"
#define I_RNDNE_F64(k) "v_rndne_f64 " D(k) ", " A(k) "\n"
#define I_CVT_F64_F32(k) "v_cvt_f64_f32 " D(k) ", " A(k) "\n"
#define I_CVT_F32_F64(k) "v_cvt_f32_f64 " D(k) ", " A(k) "\n"
#define I_CVT_I32_F32(k) "v_cvt_i32_f32 " D(k) ", " A(k) "\n"
#define I_READLANE(k) "v_readlane_b32 " D(k) ", " A(k) ", 5\n"
#define I_WRITELANE(k) "v_writelane_b32 " D(k) ", 17, 7\n"
#define I_DPP_MOV(k) "v_mov_b32_dpp " D(k) ", " A(k) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_S_ADD(k) "s_add_u32 " D(k) ", " D(k) ", 3\n"
#define I_DS_READ(k) "ds_read_b32 " D(k) ", " A(k) "\n"
#define I_SUB_F32(k) "v_sub_f32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_MAX_F32(k) "v_max_f32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_FMAC_F32(k) "v_fmac_f32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_PK_ADD_F32(k) "v_pk_add_f32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_PK_MUL_F32(k) "v_pk_mul_f32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_PK_FMA_F32(k) "v_pk_fma_f32 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_ADD_F32_SGPR(k) "v_add_f32 " D(k) ", s20, " B(k) "\n"
#define I_CNDMASK_E64(k) "v_cndmask_b32 " D(k) ", " A(k) ", " B(k) ", s[20:21]\n"
#define I_CMP_F32_E64(k) "v_cmp_lt_f32 s[22:23], " A(k) ", " B(k) "\n"
#define I_CMP_F64_E64(k) "v_cmp_lt_f64 s[22:23], " A(k) ", " B(k) "\n"
#define I_CMP_U32(k) "v_cmp_lt_u32 vcc, " A(k) ", " B(k) "\n"
#define I_CMPX_F32(k) "v_cmpx_ge_f32 " A(k) ", " A(k) "\n"
#define I_CMP_CND(k) "v_cmp_lt_f32 vcc, " A(k) ", " B(k) "\n v_cndmask_b32 " D(k) ", " A(k) ", " B(k) ", vcc\n"
#define I_CNDMASK_E64_VCC(k) "v_cndmask_b32_e64 " D(k) ", " A(k) ", " B(k) ", vcc\n"
#define I_CMP_CND_SGPR(k) "v_cmp_lt_f32 s[22:23], " A(k) ", " B(k) "\n v_cndmask_b32 " D(k) ", " A(k) ", " B(k) ", s[22:23]\n"
#define I_CMP_CND_E64_VCC(k) "v_cmp_lt_f32 vcc, " A(k) ", " B(k) "\n v_cndmask_b32_e64 " D(k) ", " A(k) ", " B(k) ", vcc\n"
#define I_ADD_CO(k) "v_add_co_u32 " D(k) ", vcc, " A(k) ", " B(k) "\n"
#define I_ADDC_CO(k) "v_addc_co_u32 " D(k) ", vcc, " A(k) ", " B(k) ", vcc\n"
#define I_BFI(k) "v_bfi_b32 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_OR(k) "v_or_b32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_XOR(k) "v_xor_b32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_LSHLREV(k) "v_lshlrev_b32 " D(k) ", 3, " A(k) "\n"
#define I_BFE(k) "v_bfe_u32 " D(k) ", " A(k) ", 3, 5\n"
#define I_AND_OR(k) "v_and_or_b32 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_MIN_U32(k) "v_min_u32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_MED3_F32(k) "v_med3_f32 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_MAD_U32_U24(k) "v_mad_u32_u24 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_CVT_F32_U32(k) "v_cvt_f32_u32 " D(k) ", " A(k) "\n"
#define I_SQRT_F64(k) "v_sqrt_f64 " D(k) ", " A(k) "\n"
#define I_LDEXP_F64(k) "v_ldexp_f64 " D(k) ", " A(k) ", 3\n"
#define I_DIV_SCALE_F64(k) "v_div_scale_f64 " D(k) ", vcc, " A(k) ", " B(k) ", " A(k) "\n"
#define I_DIV_FMAS_F64(k) "v_div_fmas_f64 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_DIV_FIXUP_F64(k) "v_div_fixup_f64 " D(k) ", " A(k) ", " B(k) ", " D(k) "\n"
#define I_RCP_F32(k) "v_rcp_f32 " D(k) ", " A(k) "\n"
#define I_READFIRSTLANE(k) "v_readfirstlane_b32 " D(k) ", " A(k) "\n"
#define I_MBCNT(k) "v_mbcnt_lo_u32_b32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_S_AND_B64(k) "s_and_b64 s[22:23], s[20:21], vcc\n"
#define I_S_CSELECT(k) "s_cselect_b32 " D(k) ", " D(k) ", 7\n"
#define I_BPERMUTE(k) "ds_bpermute_b32 " D(k) ", " A(k) ", " B(k) "\n"
#define I_ABS_SUB_F64(k) "v_add_f64 " D(k) ", |" A(k) "|, -" B(k) "\n"

#define COMMA ,
struct Result {
    unsigned long long cycles;  // this wave's s_memtime ticks for the whole run
    unsigned int hw_id, xcc_id;  // HW_REG_HW_ID (wave / SIMD / CU / SH / SE) and HW_REG_XCC_ID: which SIMD the wave ran on
};
extern __shared__ unsigned char dyn_lds[];  // sized by the launch so that exactly W workgroups fit a CU's 160 KB

#define KERNEL(name, DECLS, BODY, SINK)                                                                  \
    __global__ __launch_bounds__(256) void name(Result *out, float seed, int do_store) {                  \
        DECLS                                                                                             \
        if (do_store == 54321) dyn_lds[threadIdx.x] = 1;                                                  \
        asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 vcc, exec" ::: "s20", "s21", "vcc");          \
        __syncthreads();                                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int it = 0; it < ITERS; ++it) {                                                              \
            BODY                                                                                          \
        }                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0) {                                                                    \
            Result &r_ = out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6];                               \
            r_.cycles = t1 - t0;                                                                          \
            r_.hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11));                                         \
            r_.xcc_id = __builtin_amdgcn_s_getreg(20 | (3 << 11));                                        \
        }                                                                                                 \
        if (do_store == 12345) reinterpret_cast<volatile float *>(out)[threadIdx.x] = static_cast<float>(SINK); \
    }

KERNEL(k_add_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_ADD_F32), SINK8(d))
KERNEL(k_mul_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_MUL_F32), SINK8(d))
KERNEL(k_fma_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_FMA_F32), SINK8(d))
KERNEL(k_sub_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_SUB_F32), SINK8(d))
KERNEL(k_max_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_MAX_F32), SINK8(d))
KERNEL(k_fmac_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_FMAC_F32), SINK8(d))
KERNEL(k_pk_add_f32, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_PK_ADD_F32), SINK8(d))
KERNEL(k_pk_mul_f32, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_PK_MUL_F32), SINK8(d))
KERNEL(k_pk_fma_f32, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_PK_FMA_F32), SINK8(d))
KERNEL(k_add_f32_sgpr, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_ADD_F32_SGPR), SINK8(d))
KERNEL(k_cndmask_e64, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CNDMASK_E64), SINK8(d))
KERNEL(k_cmp_f32_e64, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CMP_F32_E64), SINK8(d))
KERNEL(k_cmp_f64_e64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_CMP_F64_E64), SINK8(d))
KERNEL(k_cmp_u32, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_CMP_U32), SINK8(d))
KERNEL(k_cmpx_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CMPX_F32), SINK8(d))
KERNEL(k_cmp_cnd, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CMP_CND), SINK8(d))
KERNEL(k_cndmask_e64_vcc, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CNDMASK_E64_VCC), SINK8(d))
KERNEL(k_cmp_cnd_sgpr, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CMP_CND_SGPR), SINK8(d))
KERNEL(k_cmp_cnd_e64_vcc, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CMP_CND_E64_VCC), SINK8(d))
KERNEL(k_add_co, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_ADD_CO), SINK8(d))
KERNEL(k_addc_co, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_ADDC_CO), SINK8(d))
KERNEL(k_bfi, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_BFI), SINK8(d))
KERNEL(k_or, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_OR), SINK8(d))
KERNEL(k_xor, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_XOR), SINK8(d))
KERNEL(k_lshlrev, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_LSHLREV), SINK8(d))
KERNEL(k_bfe, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_BFE), SINK8(d))
KERNEL(k_and_or, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_AND_OR), SINK8(d))
KERNEL(k_min_u32, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_MIN_U32), SINK8(d))
KERNEL(k_med3_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_MED3_F32), SINK8(d))
KERNEL(k_mad_u32_u24, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_MAD_U32_U24), SINK8(d))
KERNEL(k_cvt_f32_u32, DECL_F32(d, seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_CVT_F32_U32), SINK8(d))
KERNEL(k_sqrt_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_SQRT_F64), SINK8(d))
KERNEL(k_ldexp_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_LDEXP_F64), SINK8(d))
KERNEL(k_div_scale_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_DIV_SCALE_F64), SINK8(d))
KERNEL(k_div_fmas_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_DIV_FMAS_F64), SINK8(d))
KERNEL(k_div_fixup_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_DIV_FIXUP_F64), SINK8(d))
KERNEL(k_abs_sub_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_ABS_SUB_F64), SINK8(d))
KERNEL(k_rcp_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_RCP_F32), SINK8(d))
KERNEL(k_readfirstlane, unsigned d0 = 0 COMMA d1 = 0 COMMA d2 = 0 COMMA d3 = 0 COMMA d4 = 0 COMMA d5 = 0 COMMA d6 = 0 COMMA d7 = 0; DECL_U32(a, (unsigned)seed * 2 + threadIdx.x) DECL_U32(b, 3u),
       REP32S(I_READFIRSTLANE), SINK8(d))
KERNEL(k_mbcnt, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_MBCNT), SINK8(d))
KERNEL(k_s_and_b64, unsigned d0 = 0 COMMA d1 = 0 COMMA d2 = 0 COMMA d3 = 0 COMMA d4 = 0 COMMA d5 = 0 COMMA d6 = 0 COMMA d7 = 0; DECL_U32(a, 2u) DECL_U32(b, 3u),
       REP32S(I_S_AND_B64), SINK8(d))
KERNEL(k_s_cselect, unsigned d0 = 0 COMMA d1 = 0 COMMA d2 = 0 COMMA d3 = 0 COMMA d4 = 0 COMMA d5 = 0 COMMA d6 = 0 COMMA d7 = 0; DECL_U32(a, 2u) DECL_U32(b, 3u),
       REP32S(I_S_CSELECT), SINK8(d))
KERNEL(k_min_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_MIN_F32), SINK8(d))
KERNEL(k_sqrt_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_SQRT_F32), SINK8(d))
KERNEL(k_cmp_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CMP_F32), SINK8(d))
KERNEL(k_cndmask, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CNDMASK), SINK8(d))
KERNEL(k_mov, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_MOV), SINK8(d))
KERNEL(k_dpp_mov, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_DPP_MOV), SINK8(d))
KERNEL(k_and_b32, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_AND), SINK8(d))
KERNEL(k_add_u32, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_ADD_U32), SINK8(d))
KERNEL(k_lshl_add_u32, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_LSHL_ADD), SINK8(d))
KERNEL(k_mul_lo_u32, DECL_U32(d, (unsigned)seed) DECL_U32(a, (unsigned)seed * 2) DECL_U32(b, (unsigned)seed * 3), REP32(I_MUL_LO), SINK8(d))
KERNEL(k_add_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_ADD_F64), SINK8(d))
KERNEL(k_mul_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_MUL_F64), SINK8(d))
KERNEL(k_fma_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_FMA_F64), SINK8(d))
KERNEL(k_min_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_MIN_F64), SINK8(d))
KERNEL(k_max_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_MAX_F64), SINK8(d))
KERNEL(k_cmp_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_CMP_F64), SINK8(d))
KERNEL(k_rcp_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_RCP_F64), SINK8(d))
KERNEL(k_rsq_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_RSQ_F64), SINK8(d))
KERNEL(k_rndne_f64, DECL_F64(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_RNDNE_F64), SINK8(d))
KERNEL(k_cvt_f64_f32, DECL_F64(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CVT_F64_F32), SINK8(d))
KERNEL(k_cvt_f32_f64, DECL_F32(d, seed) DECL_F64(a, seed * 2) DECL_F64(b, seed * 3), REP32(I_CVT_F32_F64), SINK8(d))
KERNEL(k_cvt_i32_f32, DECL_U32(d, (unsigned)seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_CVT_I32_F32), SINK8(d))
KERNEL(k_readlane, unsigned d0 = 0 COMMA d1 = 0 COMMA d2 = 0 COMMA d3 = 0 COMMA d4 = 0 COMMA d5 = 0 COMMA d6 = 0 COMMA d7 = 0; DECL_U32(a, (unsigned)seed * 2 + threadIdx.x) DECL_U32(b, 3u),
       REP32S(I_READLANE), SINK8(d))
KERNEL(k_writelane, DECL_U32(d, (unsigned)seed) DECL_U32(a, 2u) DECL_U32(b, 3u), REP32(I_WRITELANE), SINK8(d))
KERNEL(k_s_add_u32, unsigned d0 = 0 COMMA d1 = 0 COMMA d2 = 0 COMMA d3 = 0 COMMA d4 = 0 COMMA d5 = 0 COMMA d6 = 0 COMMA d7 = 0; DECL_U32(a, 2u) DECL_U32(b, 3u),
       REP32S(I_S_ADD), SINK8(d))

// LDS reads: conflict-free dword per lane, addresses in the first 1 KB
__global__ __launch_bounds__(256) void k_ds_read_b32(Result *out, float seed, int do_store) {
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = i * (unsigned)seed;
    __syncthreads();
    unsigned a0 = (threadIdx.x & 63) * 4, a1 = a0 + 256, a2 = a0 + 512, a3 = a0 + 768, a4 = a0, a5 = a1, a6 = a2, a7 = a3;
    unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
    unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
    (void)b0; (void)b1; (void)b2; (void)b3; (void)b4; (void)b5; (void)b6; (void)b7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        REP32(I_DS_READ)
        asm volatile("s_waitcnt lgkmcnt(8)");
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        Result &r_ = out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6];
        r_.cycles = t1 - t0;
        r_.hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        r_.xcc_id = __builtin_amdgcn_s_getreg(20 | (3 << 11));
    }
    if (do_store == 54321) dyn_lds[threadIdx.x] = 1;
    if (do_store == 12345) reinterpret_cast<volatile unsigned *>(out)[threadIdx.x] = SINK8(d);
}

typedef void (*Kern)(Result *, float, int);
struct Entry {
    const char *name;
    const char *pmc_class;  // the SQ_INSTS_VALU_* bucket this instruction is counted in ("OTHER": no arithmetic bucket)
    Kern k;
};

int main(int argc, char **argv) {
    const Entry entries[] = {
        {"v_add_f32", "ADD_F32", k_add_f32}, {"v_mul_f32", "MUL_F32", k_mul_f32}, {"v_fma_f32", "FMA_F32", k_fma_f32},
        {"v_min_f32", "OTHER", k_min_f32}, {"v_sqrt_f32", "TRANS_F32", k_sqrt_f32}, {"v_cmp_lt_f32", "OTHER", k_cmp_f32},
        {"v_cndmask_b32", "OTHER", k_cndmask}, {"v_mov_b32", "OTHER", k_mov}, {"v_mov_b32_dpp", "OTHER", k_dpp_mov},
        {"v_and_b32", "INT32", k_and_b32}, {"v_add_u32", "INT32", k_add_u32}, {"v_lshl_add_u32", "INT32", k_lshl_add_u32},
        {"v_mul_lo_u32", "INT32", k_mul_lo_u32},
        {"v_add_f64", "ADD_F64", k_add_f64}, {"v_mul_f64", "MUL_F64", k_mul_f64}, {"v_fma_f64", "FMA_F64", k_fma_f64},
        {"v_min_f64", "OTHER", k_min_f64}, {"v_max_f64", "OTHER", k_max_f64}, {"v_cmp_lt_f64", "OTHER", k_cmp_f64},
        {"v_rcp_f64", "TRANS_F64", k_rcp_f64}, {"v_rsq_f64", "TRANS_F64", k_rsq_f64}, {"v_rndne_f64", "OTHER", k_rndne_f64},
        {"v_cvt_f64_f32", "CVT", k_cvt_f64_f32}, {"v_cvt_f32_f64", "CVT", k_cvt_f32_f64}, {"v_cvt_i32_f32", "CVT", k_cvt_i32_f32},
        {"v_readlane_b32", "OTHER", k_readlane}, {"v_writelane_b32", "OTHER", k_writelane},
        {"s_add_u32", "SALU", k_s_add_u32}, {"ds_read_b32", "LDS", k_ds_read_b32},
        {"v_sub_f32", "ADD_F32", k_sub_f32}, {"v_max_f32", "OTHER", k_max_f32}, {"v_fmac_f32", "FMA_F32", k_fmac_f32},
        {"v_pk_add_f32", "ADD_F32", k_pk_add_f32}, {"v_pk_mul_f32", "MUL_F32", k_pk_mul_f32}, {"v_pk_fma_f32", "FMA_F32", k_pk_fma_f32},
        {"v_add_f32 (SGPR source)", "ADD_F32", k_add_f32_sgpr}, {"v_cndmask_b32 (SGPR-pair mask, VOP3)", "OTHER", k_cndmask_e64},
        {"v_cmp_lt_f32 -> SGPR pair", "OTHER", k_cmp_f32_e64}, {"v_cmp_lt_f64 -> SGPR pair", "OTHER", k_cmp_f64_e64},
        {"v_cmp_lt_u32", "OTHER", k_cmp_u32}, {"v_cmpx_ge_f32", "OTHER", k_cmpx_f32},
        {"v_cmp_lt_f32 + v_cndmask_b32 (pair, cost per instruction)", "OTHER", k_cmp_cnd},
        {"v_cndmask_b32_e64 (VOP3 encoding, vcc mask)", "OTHER", k_cndmask_e64_vcc},
        {"v_cmp_lt_f32 -> s[22:23] + v_cndmask_b32 s[22:23] (pair, cost per instruction)", "OTHER", k_cmp_cnd_sgpr},
        {"v_cmp_lt_f32 -> vcc + v_cndmask_b32_e64 vcc (pair, cost per instruction)", "OTHER", k_cmp_cnd_e64_vcc},
        {"v_add_co_u32 (writes vcc)", "INT32", k_add_co}, {"v_addc_co_u32 (reads and writes vcc)", "INT32", k_addc_co},
        {"v_bfi_b32", "INT32", k_bfi}, {"v_or_b32", "INT32", k_or}, {"v_xor_b32", "INT32", k_xor}, {"v_lshlrev_b32", "INT32", k_lshlrev},
        {"v_bfe_u32", "INT32", k_bfe}, {"v_and_or_b32", "INT32", k_and_or}, {"v_min_u32", "INT32", k_min_u32},
        {"v_med3_f32", "OTHER", k_med3_f32}, {"v_mad_u32_u24", "INT32", k_mad_u32_u24}, {"v_cvt_f32_u32", "CVT", k_cvt_f32_u32},
        {"v_sqrt_f64", "TRANS_F64", k_sqrt_f64}, {"v_ldexp_f64", "OTHER", k_ldexp_f64}, {"v_div_scale_f64", "OTHER", k_div_scale_f64},
        {"v_div_fmas_f64", "OTHER", k_div_fmas_f64}, {"v_div_fixup_f64", "OTHER", k_div_fixup_f64},
        {"v_add_f64 (|a|, -b modifiers)", "ADD_F64", k_abs_sub_f64}, {"v_rcp_f32", "TRANS_F32", k_rcp_f32},
        {"v_readfirstlane_b32", "OTHER", k_readfirstlane}, {"v_mbcnt_lo_u32_b32", "OTHER", k_mbcnt},
        {"s_and_b64", "SALU", k_s_and_b64}, {"s_cselect_b32", "SALU", k_s_cselect},
    };
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const int max_w = 6;
    (void)argc; (void)argv;
    Result *d_out;
    CHECK(hipMalloc(&d_out, sizeof(Result) * cus * max_w * 4 + 4096));
    std::vector<Result> h(static_cast<size_t>(cus) * max_w * 4);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double insts = static_cast<double>(UNROLL) * ITERS;
    printf("{\n \"device\": \"%s\", \"compute_units\": %d, \"instructions_per_wave\": %.0f,\n", prop.gcnArchName, cus, insts);
    printf(" \"_comment\": \"cycles = shader cycles (s_memtime) one wave needs per instruction, divided by the waves per SIMD: the "
           "cost of one wave64 instruction to the SIMD's issue; wall_* from HIP events over the whole launch\",\n \"classes\": {\n");
    bool first = true;
    for (const Entry &en : entries) {
        printf("%s  \"%s\": {\"pmc_class\": \"%s\"", first ? "" : ",\n", en.name, en.pmc_class);
        first = false;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(en.k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        for (int w : {1, 2, 5, 6}) {
            const int blocks = cus * w;  // one 4-wave workgroup per CU and per wave-per-SIMD
            // dynamic LDS so that exactly w workgroups fit a CU's 160 KB: the dispatcher cannot pile 8 on one CU
            // (LDS is handed out in granules of 1 280 bytes: 31 744 -> 32 000 x 5 = 160 000; 26 624 -> 26 880 x 6 = 161 280 <= 163 840)
            const size_t lds = w == 1 ? 96 * 1024 : (w == 2 ? 64 * 1024 : (w == 5 ? 32 * 1024 - 1024 : 26 * 1024));
            hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), lds, 0, d_out, 1.5f, 0);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), lds, 0, d_out, 1.5f, 0);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Result) * blocks * 4, hipMemcpyDeviceToHost));
            double sum = 0, mx = 0;
            std::vector<int> per_simd(8 * 4096, 0);  // waves per (xcc, se, sh, cu, simd)
            int max_per_simd = 0;
            for (int i = 0; i < blocks * 4; ++i) {
                sum += static_cast<double>(h[i].cycles);
                if (static_cast<double>(h[i].cycles) > mx) mx = static_cast<double>(h[i].cycles);
                const unsigned id = h[i].hw_id;  // [5:4] simd, [11:8] cu, [12] sh, [15:13] se
                const unsigned key = ((h[i].xcc_id & 7) << 12) | (((id >> 13) & 7) << 9) | (((id >> 12) & 1) << 8) | (((id >> 8) & 15) << 4) | ((id >> 4) & 3);
                if (++per_simd[key] > max_per_simd) max_per_simd = per_simd[key];
            }
            const double mean = sum / (blocks * 4);
            // if the dispatcher spread the workgroups evenly, a SIMD ran w waves: per-instruction cost to the SIMD
            printf(", \"w%d\": {\"cycles\": %.3f, \"cycles_slowest_wave\": %.3f, \"wall_us\": %.1f, \"clock_mhz\": %.0f, \"max_waves_on_a_simd\": %d}", w,
                   mean / insts / w, mx / insts / w, ms * 1e3, mx / (ms * 1e-3) / 1e6, max_per_simd);
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n }\n}\n");
    return 0;
}
