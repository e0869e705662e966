// valu_issue_bench.hip -- what does one wave64 instruction COST a gfx950 SIMD, per instruction class?
//
// VERDICT r1 asked for the measured VALU ceiling of the render kernel instead of an assumed "4 cycles per wave
// instruction".  This micro-benchmark issues long runs of independent instructions of ONE class from W waves per SIMD
// (W = 1, 2, 5: the render kernel runs at 5) on every SIMD of the chip and reports
//     cycles per wave-instruction per SIMD = wave's elapsed shader cycles (s_memtime) / (instructions x W),
// the wall-clock rate, and the shader clock (s_memtime ticks / wall time).  The render kernel's instruction mix
// (rocprofv3 SQ_INSTS_VALU_* counters) priced with these costs is the "weighted issue floor" in bench.py's
// roofline.valu block (scripts/make_pmc_json.py).
//
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/bin/valu_issue_bench scripts/valu_issue_bench.hip
// run:   scripts/bin/valu_issue_bench > profiles/r02/valu_issue_costs.json
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
#define ROUND8(S, C) asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) : OUT8(C) : IN16 : "vcc", "scc");
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

#define COMMA ,
struct Result {
    unsigned long long cycles;  // this wave's s_memtime ticks for the whole run
};

#define KERNEL(name, DECLS, BODY, SINK)                                                                  \
    __global__ __launch_bounds__(256) void name(Result *out, float seed, int do_store) {                  \
        DECLS                                                                                             \
        __syncthreads();                                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int it = 0; it < ITERS; ++it) {                                                              \
            BODY                                                                                          \
        }                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6].cycles = t1 - t0;  \
        if (do_store == 12345) reinterpret_cast<volatile float *>(out)[threadIdx.x] = static_cast<float>(SINK); \
    }

KERNEL(k_add_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_ADD_F32), SINK8(d))
KERNEL(k_mul_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_MUL_F32), SINK8(d))
KERNEL(k_fma_f32, DECL_F32(d, seed) DECL_F32(a, seed * 2) DECL_F32(b, seed * 3), REP32(I_FMA_F32), SINK8(d))
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
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6].cycles = t1 - t0;
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
    };
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const int max_w = 5;
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
        for (int w : {1, 2, 5}) {
            const int blocks = cus * w;  // one 4-wave workgroup per CU and per wave-per-SIMD
            hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.5f, 0);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.5f, 0);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Result) * blocks * 4, hipMemcpyDeviceToHost));
            double sum = 0, mx = 0;
            for (int i = 0; i < blocks * 4; ++i) {
                sum += static_cast<double>(h[i].cycles);
                if (static_cast<double>(h[i].cycles) > mx) mx = static_cast<double>(h[i].cycles);
            }
            const double mean = sum / (blocks * 4);
            // if the dispatcher spread the workgroups evenly, a SIMD ran w waves: per-instruction cost to the SIMD
            printf(", \"w%d\": {\"cycles\": %.3f, \"cycles_slowest_wave\": %.3f, \"wall_us\": %.1f, \"clock_mhz\": %.0f}", w,
                   mean / insts / w, mx / insts / w, ms * 1e3, mx / (ms * 1e-3) / 1e6);
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n }\n}\n");
    return 0;
}
