// rm_rtc.h -- run-time specialisation of the one-ray-per-lane kernels for ONE scene's expression forest.
//
// The reference evaluates an SDF operator tree by virtual calls (Primitive.sdf, primitive.ts:33-39, and the overrides in
// primitive_operations/*.ts).  The interpreter of rm_program.h walks the same tree as an instruction stream: every
// instruction costs a scalar-cache round trip, a branch tree and LDS traffic whatever it computes.  A GPU's answer to a
// scene that changes rarely and is evaluated 10^8 times per frame is to compile it: the forest becomes straight-line HIP
// (the same formula functions of rm_program.h, called in the interpreter's order, every matrix and parameter a literal)
// and hiprtc builds render_body<ACCEL, OTHER, 4> of rm_kernels.hip around it for gfx950.  The sources compiled at run time
// are the ones this library was built from (embedded at build time), with the library's own flags (-ffp-contract=off).
#pragma once
#include <string>
#include <vector>

#include "rm_scene_host.h"
#include "rm_types.h"

namespace rmrtc {

struct Kernel {
    void *module = nullptr;       // hipModule_t
    void *render = nullptr;       // hipFunction_t of rm_rtc_render(RmRenderParams)
    void *distance = nullptr;     // hipFunction_t of rm_rtc_distance(RmRenderParams, const float *, int64_t, double *, uint32_t *)
    std::string name;             // "rm_rtc_render<ACCEL, OTHER>" + length tag, as rm_last_kernel reports it
    double compile_seconds = 0;
};

// Straight-line source of a scene's programs: defines rmd::rm_rtc_object_sdf(int obj, const Vec3f &p, double time).
// Empty when the forest is too large to be worth a compile (more than kMaxObjects objects or kMaxInstructions instructions).
constexpr int kMaxObjects = 32, kMaxInstructions = 512;
// `prune`: smooth unions / subtractions over spheres, boxes and tori skip operands that provably cannot matter (rm_rtc.cpp).
// bvh / bvh_prims: the scene's BVH (may be empty); up to eight leaves are emitted as code (rm_rtc_bvh_distance / _next_interval).
// require_bvh: no source at all when a BVH is given and cannot be emitted (plain primitive lists: their data-driven kernels are
// the better choice once the tree has to be walked in memory anyway).
std::string scene_source(const std::vector<RmInstr> &prog, const std::vector<int32_t> &obj_ranges, const std::vector<rmh::ProgTreeNode> &tree,
                         const std::vector<int32_t> &roots, bool prune, const std::vector<RmBvhNode> &bvh, const std::vector<int32_t> &bvh_prims,
                         bool require_bvh);

// hiprtc is loaded on first use (dlopen): false + reason when this machine has none.
bool available(std::string *why);

// Compiles rm_kernels.hip for (accel, other, length) around scene_src.  With load_module the code object is loaded into the
// current device's context and the kernel handles are looked up; without (a machine with no GPU: tests of the build) only the
// compile is done.  *log receives hiprtc's log (resource-usage remarks included when want_remarks).
bool compile(const std::string &scene_src, int accel, bool other, bool length_sqrt, bool load_module, bool want_remarks, Kernel &out,
             std::string &log);
void release(Kernel &k);

// The v2 wave loop (rm_render_v2.hip) compiled for ONE launch configuration: rm_v2_fix() assigns every configuration parameter
// (rm_v2_fields.h) its value as a literal; the kernel applies it to the parameter block wherever it (re)loads it.
std::string v2_fixed_source(const RmRenderParams &p, bool with_counts);  // with_counts: the scene's counts too (rm_v2_fields.h says when not)
uint64_t v2_fixed_hash(const RmRenderParams &p, int variant_bits);  // cheap identity of (configuration, instantiation) for the per-launch lookup
// accel / lds / ur / rel: the instantiation render_kernel_v2<ACCEL, LDS, UR, REL> the launcher chose.  A kernel that spills or
// uses scratch is refused (log says so).
bool compile_v2(const std::string &fixed_src, int accel, bool lds, bool ur, bool rel, bool length_sqrt, bool load_module, Kernel &out, std::string &log);
constexpr int kV2Bit = 1 << 8;  // in the `accel` argument of compile_cached / compile_async: ACCEL | LDS << 2 | UR << 3 | REL << 4 | kV2Bit

// compile(..., load_module = true) through a process-wide cache keyed by (device, source text, accel, other, length): a host
// that switches back to a scene it has rendered before (the reference's preset menu) gets the loaded module back instead of
// another 1.5 - 8 s compile.  Cached modules stay loaded for the life of the process (at most kCacheEntries; beyond that
// kernels are compiled per use and released with their scene: *cached tells which).
constexpr int kCacheEntries = 256;
bool compile_cached(int device, const std::string &scene_src, int accel, bool other, bool length_sqrt, Kernel &out, std::string &log, bool *cached);
// The same without waiting: 1 = the kernel is in the cache (out is set), 0 = a background thread is compiling it (started by
// this call or an earlier one; the caller renders with its ahead-of-time kernels meanwhile -- same bytes), -1 = that compile
// failed or the cache is full (log says which).
int compile_async(int device, const std::string &scene_src, int accel, bool other, bool length_sqrt, Kernel &out, std::string &log);

}  // namespace rmrtc
