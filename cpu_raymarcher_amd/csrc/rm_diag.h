// rm_diag.h -- the diagnostics of the reference's frame loop (src/main.ts:528-548: sum / max / min of the SDF-evaluation
// counters, sum of the iteration counters) produced by the render kernels themselves.
//
// Round 2 ran two more launches per frame (reduce_init_kernel + reduce_kernel) that re-read the 4 B/pixel the render
// kernel had just written: 12.6 us per 4K frame, ~7 % of a 1/8-row shard.  The counters are in registers when a pixel is
// stored, so the render kernels accumulate them there:
//   * per wave in LDS (v2: LDS atomics at every pixel store; v1: once, at the wave's only store),
//   * per launch in a block of RM_DIAG_SLOTS accumulator slots in device memory, 128 B apart (atomics on one cache line
//     serialise: 12 ns each, measured on reduce_kernel), one flush per wave,
//   * the LAST wave of the launch (a count of flushed waves per slot, then a count of completed slots) combines the
//     slots, writes the 32-byte result (RmDiagDevice, the layout rm_reduce_counters_enqueue writes) and leaves the block
//     -- and the launch's tile-queue heads -- zeroed for their next user: no initialising launch, no memset.
// Ordering: a wave's four data atomics RETURN their old values and the wave waits for them before it counts itself as
// flushed, so they are performed (at device scope: every XCD sees them) before the count can reach its target; there is
// no fence, whose release half would write back every dirty line of the XCD's L2 -- the pixel stores.  The last wave
// reads the slots with device-scope atomic loads.  min is kept as max(0xFFFFFFFF - sdfEval): all accumulators start at 0.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "rm_kernels.h"

#define RM_DIAG_SLOTS 256

struct RmDiagSlot {
    unsigned long long sdf, iters;
    unsigned int mx, mn_inv, done, pad0;
    unsigned int pad[24];
};
struct RmDiagBlock {
    RmDiagSlot slot[RM_DIAG_SLOTS];
    unsigned int slots_done;
    unsigned int pad[31];
};

namespace rmd {

// The launch's last wave (all 64 lanes): every slot is complete.  Lanes read slots lane, lane + 64, ..., zero them, and a
// wave reduction gives the result.  Out of line: it runs once per launch, and inlined into the render kernels its
// unrolled loads raised their register need (40 spilled VGPRs in the headline instantiation).
__device__ __forceinline__ void diag_finalise_body(RmDiagBlock *blk, RmDiagDevice *out, unsigned int *tile_counters, unsigned int nslots) {
    const unsigned int lane = __lane_id();
    unsigned long long ts = 0, ti = 0;
    unsigned int mxa = 0, mia = 0;
    for (unsigned int k = lane; k < nslots; k += 64) {
        RmDiagSlot *q = &blk->slot[k];
        const unsigned long long a = __hip_atomic_load(&q->sdf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(&q->iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int c = __hip_atomic_load(&q->mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int d = __hip_atomic_load(&q->mn_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ts += a;
        ti += b;
        mxa = c > mxa ? c : mxa;
        mia = d > mia ? d : mia;
        __hip_atomic_store(&q->sdf, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&q->iters, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&q->mx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&q->mn_inv, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&q->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int off = 32; off > 0; off >>= 1) {
        ts += __shfl_down(ts, off);
        ti += __shfl_down(ti, off);
        const unsigned int omx = __shfl_down(mxa, off), omi = __shfl_down(mia, off);
        mxa = omx > mxa ? omx : mxa;
        mia = omi > mia ? omi : mia;
    }
    if (lane == 0 && out) {
        out->total_sdf = ts;
        out->total_iters = ti;
        out->max_sdf = mxa;
        out->min_sdf = 0xFFFFFFFFu - mia;  // no pixel at all: UINT_MAX, as rm_launch_reduce_init leaves it
        out->pad = 0;
    }
    if (lane == 0) __hip_atomic_store(&blk->slots_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tile_counters) __hip_atomic_store(&tile_counters[lane * 64u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the v2 kernels' 64 queue heads, 256 bytes apart (RM_QSTRIDE)
}

__device__ __attribute__((noinline)) static void diag_finalise(RmDiagBlock *blk, RmDiagDevice *out, unsigned int *tile_counters,
                                                               unsigned int nslots) {
    diag_finalise_body(blk, out, tile_counters, nslots);
}

// FINAL_INLINE: the caller is itself an out-of-line function (the one-ray-per-lane kernels' epilogue): a nested call
// would need a stack frame.
// Called by EVERY lane of EVERY wave of the launch, all 64 lanes active, exactly once; (sdf, iters, mx, mn_inv) are the
// wave's totals (wave-uniform), `unit` numbers the launch's waves 0 .. units - 1.
template <bool FINAL_INLINE = false>
__device__ __forceinline__ void diag_flush_wave(RmDiagBlock *blk, RmDiagDevice *out, unsigned int *tile_counters,
                                                unsigned long long sdf, unsigned long long iters, unsigned int mx,
                                                unsigned int mn_inv, unsigned int unit, unsigned int units, int lane) {
    const unsigned int s = unit % RM_DIAG_SLOTS;
    const unsigned int expected = units / RM_DIAG_SLOTS + (s < units % RM_DIAG_SLOTS ? 1u : 0u);
    const unsigned int nslots = units < RM_DIAG_SLOTS ? units : static_cast<unsigned int>(RM_DIAG_SLOTS);
    RmDiagSlot *sl = &blk->slot[s];
    unsigned int last = 0;
    if (lane == 0) {
        if (mn_inv != 0u) {  // the wave stored at least one pixel (mn_inv >= 0xFFFF0000 then)
            const unsigned long long r0 = atomicAdd(&sl->sdf, sdf), r1 = atomicAdd(&sl->iters, iters);
            const unsigned int r2 = atomicMax(&sl->mx, mx), r3 = atomicMax(&sl->mn_inv, mn_inv);
            asm volatile("" ::"v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");  // the four have been performed
        }
        if (atomicAdd(&sl->done, 1u) + 1u == expected) {  // this slot is complete
            if (atomicAdd(&blk->slots_done, 1u) + 1u == nslots) last = 1;
        }
    }
    if (__builtin_amdgcn_readfirstlane(static_cast<int>(last))) {
        if (FINAL_INLINE) diag_finalise_body(blk, out, tile_counters, nslots);
        else diag_finalise(blk, out, tile_counters, nslots);
    }
}

}  // namespace rmd
