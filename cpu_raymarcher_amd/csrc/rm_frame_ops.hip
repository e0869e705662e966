// rm_frame_ops.hip -- rank 0's side of the row-tile shard (gfx950): ONE kernel that puts the gathered stripes of
// every rank at their place in the row-major frame and combines the ranks' partial diagnostics.
//
// Replaces the fan-in of the reference's main thread: `depthBuffer.set(tile, yStart * width)` per worker result
// (src/main.ts:461-468) and, for the diagnostics (src/main.ts:528-548: a sum, a max and a min, which combine
// exactly from per-rank partial results), the single pass over the gathered counters.  HBM-bound byte movement:
// 16-byte loads and stores, one workgroup per (stripe, column chunk); no reshaping into anything else.
#include <hip/hip_runtime.h>

#include "rm_kernels.h"

namespace {

// stripe_src[s] = (rank << 16) | local stripe index of frame stripe s within that rank's packed rows
__global__ __launch_bounds__(256) void assemble_kernel(const unsigned char *__restrict__ gathered, long long rank_stride,
                                                       long long section_offset, int row_bytes, int height, int stripe_rows,
                                                       const int *__restrict__ stripe_src, int n_stripes,
                                                       unsigned char *__restrict__ frame, long long acc_offset, int world,
                                                       RmDiagDevice *acc, int chunks_per_stripe) {
    const int s = blockIdx.x / chunks_per_stripe, chunk = blockIdx.x - s * chunks_per_stripe;
    if (s < n_stripes) {
        const int src = stripe_src[s];
        const int rank = src >> 16, local = src & 0xFFFF;
        const int y0 = s * stripe_rows;
        const int rows = min(stripe_rows, height - y0);
        const long long bytes = static_cast<long long>(rows) * row_bytes;  // a stripe is contiguous on both sides
        const unsigned char *from = gathered + rank * rank_stride + section_offset + static_cast<long long>(local) * stripe_rows * row_bytes;
        unsigned char *to = frame + static_cast<long long>(y0) * row_bytes;
        if (((reinterpret_cast<uintptr_t>(from) | reinterpret_cast<uintptr_t>(to) | static_cast<uintptr_t>(bytes)) & 15) == 0) {
            const long long n16 = bytes >> 4;
            const uint4 *f4 = reinterpret_cast<const uint4 *>(from);
            uint4 *t4 = reinterpret_cast<uint4 *>(to);
            const long long step = static_cast<long long>(chunks_per_stripe) * 256;
            long long i = static_cast<long long>(chunk) * 256 + threadIdx.x;
            for (; i + 3 * step < n16; i += 4 * step) {  // four 16-B loads in flight per lane
                const uint4 a = f4[i], b = f4[i + step], c = f4[i + 2 * step], d = f4[i + 3 * step];
                t4[i] = a;
                t4[i + step] = b;
                t4[i + 2 * step] = c;
                t4[i + 3 * step] = d;
            }
            for (; i < n16; i += step) t4[i] = f4[i];
        } else {
            for (long long i = static_cast<long long>(chunk) * 256 + threadIdx.x; i < bytes; i += static_cast<long long>(chunks_per_stripe) * 256)
                to[i] = from[i];
        }
    }
    // combined diagnostics: sums add, max of max, min of min (a rank without rows holds the neutral elements)
    if (blockIdx.x == 0 && threadIdx.x == 0 && acc && acc_offset >= 0) {
        unsigned long long ts = 0, ti = 0;
        unsigned int mx = 0, mn = 0xFFFFFFFFu;
        for (int r = 0; r < world; ++r) {
            const RmDiagDevice *p = reinterpret_cast<const RmDiagDevice *>(gathered + r * rank_stride + acc_offset);
            ts += p->total_sdf;
            ti += p->total_iters;
            mx = p->max_sdf > mx ? p->max_sdf : mx;
            mn = p->min_sdf < mn ? p->min_sdf : mn;
        }
        acc->total_sdf = ts;
        acc->total_iters = ti;
        acc->max_sdf = mx;
        acc->min_sdf = mn;
        acc->pad = 0;
    }
}

}  // namespace

hipError_t rm_launch_assemble(const unsigned char *gathered, int64_t rank_stride, int64_t section_offset, int32_t row_bytes,
                              int32_t height, int32_t stripe_rows, const int32_t *stripe_src, int32_t n_stripes,
                              unsigned char *frame, int64_t acc_offset, int32_t world, RmDiagDevice *acc, hipStream_t stream) {
    if (n_stripes <= 0 && !(acc && acc_offset >= 0)) return hipSuccess;
    // ~1024 workgroups (4 per CU) whatever the stripe height
    int chunks = n_stripes > 0 ? (1024 + n_stripes - 1) / n_stripes : 1;
    const long long stripe_bytes = static_cast<long long>(stripe_rows) * row_bytes;
    const long long max_chunks = (stripe_bytes / 16 + 255) / 256;
    if (chunks > max_chunks) chunks = static_cast<int>(max_chunks > 0 ? max_chunks : 1);
    const unsigned blocks = static_cast<unsigned>((n_stripes > 0 ? n_stripes : 1) * chunks);
    hipLaunchKernelGGL(assemble_kernel, dim3(blocks), dim3(256), 0, stream, gathered, static_cast<long long>(rank_stride),
                       static_cast<long long>(section_offset), row_bytes, height, stripe_rows, stripe_src, n_stripes, frame,
                       static_cast<long long>(acc_offset), world, acc, chunks);
    return hipGetLastError();
}
