// rm_rtc.cpp -- see rm_rtc.h.  Host only.
#include "rm_rtc.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>

// The device sources this library was built from, as the assembler found them at build time (csrc/Makefile runs in this
// directory and lists them as prerequisites of this object).
#define RM_EMBED(sym, file)                                                                                   \
    __asm__(".pushsection .rodata\n .hidden " #sym "\n .global " #sym "\n" #sym ":\n .incbin \"" file "\"\n .byte 0\n .popsection\n"); \
    extern "C" const char sym[];
RM_EMBED(rm_src_kernels_hip, "rm_kernels.hip")
RM_EMBED(rm_src_device_h, "rm_device.h")
RM_EMBED(rm_src_jsmath_h, "rm_jsmath.h")
RM_EMBED(rm_src_types_h, "rm_types.h")
RM_EMBED(rm_src_program_h, "rm_program.h")
RM_EMBED(rm_src_bvh_list_h, "rm_bvh_list.h")
RM_EMBED(rm_src_kernels_h, "rm_kernels.h")
RM_EMBED(rm_src_diag_h, "rm_diag.h")

namespace rmrtc {

namespace {

// ---- source generation -----------------------------------------------------------------------------------------------
// Literals are hexadecimal floating point: every binary32 / binary64 value prints and parses exactly.
std::string lit_d(double v) {
    char b[64];
    std::snprintf(b, sizeof b, "%a", v);
    return std::string("(") + b + ")";
}
std::string lit_f(float v) {
    char b[64];
    std::snprintf(b, sizeof b, "%af", static_cast<double>(v));
    return std::string("(") + b + ")";
}
std::string mat(const char *name, const float *m) {
    std::string s = std::string("const float ") + name + "[16] = {";
    for (int k = 0; k < 16; ++k) s += lit_f(m[k]) + (k < 15 ? ", " : "");
    return s + "};";
}
std::string slot(int s, char c) { return "s" + std::to_string(s) + c; }

// One object: the interpreter's loop (rm_program.h, program_sdf) unrolled over this object's instructions -- same
// formula functions, same order, slot numbers and stack depths resolved here.
bool emit_object(const RmInstr *ins, int count, int index, std::string &out) {
    int max_slot = 0, max_val = 0, sp = 0;
    std::string body;
    for (int pc = 0; pc < count; ++pc) {
        const RmInstr &I = ins[pc];
        const int op = I.op;
        body += "    {  // " + std::to_string(pc) + ": op " + std::to_string(op) + "\n";
        if (op >= 20) {
            if (op == 20) {
                if (sp < 1) return false;
                body += "        v" + std::to_string(sp - 1) + " = v" + std::to_string(sp - 1) + " - " + lit_d(I.p[0]) + ";\n";
            } else {
                if (sp < 2) return false;
                const std::string d1 = "v" + std::to_string(sp - 2), d2 = "v" + std::to_string(sp - 1);
                body += "        " + d1 + " = " + (op == 21 ? "post_smooth_union(" : "post_smooth_subtraction(") + d1 + ", " + d2 + ", " + lit_d(I.p[0]) +
                        " * 4.0);\n";
                sp -= 1;
            }
            body += "    }\n";
            continue;
        }
        if (I.src < 0 || I.src >= RM_PROG_MAX_SLOTS || I.dst < 0 || I.dst >= RM_PROG_MAX_SLOTS) return false;
        max_slot = std::max(max_slot, I.src);
        const int fT = (I.flags & 1) | (((I.flags >> 2) & 1) << 2), fI = ((I.flags >> 1) & 1) | (((I.flags >> 3) & 1) << 2);
        body += "        " + mat("T", I.T) + "\n        float lx, ly, lz;\n";
        body += "        transform_mat4(T, " + std::to_string(fT) + ", " + slot(I.src, 'x') + ", " + slot(I.src, 'y') + ", " + slot(I.src, 'z') +
                ", lx, ly, lz);\n";  // primitive.ts:34-35
        if (op < 10) {
            const std::string v = "v" + std::to_string(sp);
            if (op == 1) body += "        " + v + " = leaf_box(lx, ly, lz, " + lit_d(I.p[0]) + ", " + lit_d(I.p[1]) + ", " + lit_d(I.p[2]) + ");\n";
            else if (op == 2) body += "        " + v + " = leaf_torus(lx, ly, lz, " + lit_d(I.p[0]) + ", " + lit_d(I.p[1]) + ");\n";
            else if (op == 3) {
                body += "        const double prm[6] = {";
                for (int k = 0; k < 6; ++k) body += lit_d(I.p[k]) + (k < 5 ? ", " : "");
                body += "};\n        " + v + " = mandelbulb_sdf(prm, lx, ly, lz, time);\n";
            } else if (op == 0) body += "        " + v + " = leaf_sphere(lx, ly, lz, " + lit_d(I.p[0]) + ");\n";
            else return false;
            sp += 1;
            max_val = std::max(max_val, sp);
            if (sp > RM_PROG_MAX_VALS) return false;
            body += "    }\n";
            continue;
        }
        max_slot = std::max(max_slot, I.dst);
        body += "        float wx, wy, wz;\n";
        if (op == 15) {
            body += "        pre_animated_translate(lx, ly, lz, " + lit_d(I.p[0]) + ", " + lit_d(I.p[1]) + ", " + lit_d(I.p[2]) + ", " + lit_d(I.p[3]) + ", " +
                    lit_d(I.p[4]) + ", time, wx, wy, wz);\n";
        } else {
            body += "        " + mat("Ti", I.Tinv) + "\n";
            body += "        transform_mat4(Ti, " + std::to_string(fI) + ", lx, ly, lz, wx, wy, wz);\n";
            if (op == 13) body += "        pre_twist(" + lit_d(I.p[0]) + ", wx, wy, wz);\n";
            else if (op == 14) body += "        pre_repetition(" + lit_d(I.p[0]) + ", " + lit_d(I.p[1]) + ", " + lit_d(I.p[2]) + ", wx, wy, wz);\n";
            else if (op != 10 && op != 11 && op != 12) return false;
        }
        body += "        " + slot(I.dst, 'x') + " = wx;\n        " + slot(I.dst, 'y') + " = wy;\n        " + slot(I.dst, 'z') + " = wz;\n    }\n";
    }
    if (sp < 1) return false;
    out += "__device__ __forceinline__ double rm_rtc_obj_" + std::to_string(index) + "(float s0x, float s0y, float s0z, double time) {\n";
    for (int s = 1; s <= max_slot; ++s) out += "    float " + slot(s, 'x') + " = 0.f, " + slot(s, 'y') + " = 0.f, " + slot(s, 'z') + " = 0.f;\n";
    for (int v = 0; v < std::max(max_val, 1); ++v) out += "    double v" + std::to_string(v) + " = 0.0;\n";
    out += body + "    return v0;\n}\n";
    return true;
}

// ---- hiprtc, loaded on first use ----------------------------------------------------------------------------------------
struct Rtc {
    void *lib = nullptr;
    std::string why;
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
};
Rtc &rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.why = std::string("libhiprtc.so not found: ") + (dlerror() ? dlerror() : "");
            return;
        }
#define RM_SYM(field, sym)                                                    \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym));        \
    if (!r.field) {                                                           \
        r.why = "libhiprtc.so lacks " #sym;                                   \
        r.lib = nullptr;                                                      \
        return;                                                               \
    }
        RM_SYM(create, hiprtcCreateProgram)
        RM_SYM(compile, hiprtcCompileProgram)
        RM_SYM(log_size, hiprtcGetProgramLogSize)
        RM_SYM(log, hiprtcGetProgramLog)
        RM_SYM(code_size, hiprtcGetCodeSize)
        RM_SYM(code, hiprtcGetCode)
        RM_SYM(destroy, hiprtcDestroyProgram)
#undef RM_SYM
    });
    return r;
}

}  // namespace

std::string scene_source(const std::vector<RmInstr> &prog, const std::vector<int32_t> &obj_ranges) {
    const int n_obj = static_cast<int>(obj_ranges.size() / 2);
    if (n_obj < 1 || n_obj > kMaxObjects || prog.size() > static_cast<size_t>(kMaxInstructions)) return std::string();
    std::string out = "// generated by rm_rtc.cpp: the scene's expression programs, one function per object\nnamespace rmd {\n";
    for (int r = 0; r < n_obj; ++r) {
        const int first = obj_ranges[2 * r], count = obj_ranges[2 * r + 1];
        if (first < 0 || count < 1 || static_cast<size_t>(first) + static_cast<size_t>(count) > prog.size()) return std::string();
        if (!emit_object(prog.data() + first, count, r, out)) return std::string();
    }
    out += "__device__ __forceinline__ double rm_rtc_object_sdf(int obj, const Vec3f &p, double time) {\n    switch (obj) {\n";
    for (int r = 0; r < n_obj; ++r)
        out += "        case " + std::to_string(r) + ": return rm_rtc_obj_" + std::to_string(r) + "(p.x, p.y, p.z, time);\n";
    out += "        default: return 0.0;\n    }\n}\n}  // namespace rmd\n";
    return out;
}

bool available(std::string *why) {
    Rtc &r = rtc();
    if (!r.lib && why) *why = r.why;
    return r.lib != nullptr;
}

bool compile(const std::string &scene_src, int accel, bool other, bool length_sqrt, bool load_module, bool want_remarks, Kernel &out,
             std::string &log) {
    Rtc &r = rtc();
    if (!r.lib) {
        log = r.why;
        return false;
    }
    if (scene_src.empty()) {
        log = "no specialised source for this scene";
        return false;
    }
    const std::string main_src = "#define RM_RTC 1\n#define RM_RTC_ACCEL " + std::to_string(accel == 1 || accel == 2 ? accel : 0) + "\n#define RM_RTC_OTHER " +
                                 (other ? "1" : "0") + "\n" + (length_sqrt ? "#define RM_LENGTH_SQRT 1\n" : "") + "#include \"rm_kernels.hip\"\n";
    const char *names[] = {"rm_kernels.hip", "rm_device.h", "rm_jsmath.h", "rm_types.h", "rm_program.h", "rm_bvh_list.h", "rm_kernels.h", "rm_diag.h",
                           "rm_rtc_scene.inc"};
    const char *texts[] = {rm_src_kernels_hip, rm_src_device_h, rm_src_jsmath_h, rm_src_types_h, rm_src_program_h, rm_src_bvh_list_h, rm_src_kernels_h,
                           rm_src_diag_h, scene_src.c_str()};
    hiprtcProgram prog = nullptr;
    if (r.create(&prog, main_src.c_str(), "rm_rtc_main.hip", 9, texts, names) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return false;
    }
    // the flags of csrc/Makefile
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-amdgpu-inline-max-bb=100000"};
    if (want_remarks) opts.push_back("-Rpass-analysis=kernel-resource-usage");
    const auto t0 = std::chrono::steady_clock::now();
    const hiprtcResult res = r.compile(prog, static_cast<int>(opts.size()), opts.data());
    out.compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t ls = 0;
    r.log_size(prog, &ls);
    log.assign(ls, '\0');
    if (ls) r.log(prog, &log[0]);
    while (!log.empty() && log.back() == '\0') log.pop_back();
    if (res != HIPRTC_SUCCESS) {
        r.destroy(&prog);
        if (log.empty()) log = "hiprtcCompileProgram failed";
        return false;
    }
    size_t cs = 0;
    r.code_size(prog, &cs);
    std::vector<char> code(cs);
    r.code(prog, code.data());
    r.destroy(&prog);
    out.name = std::string("rm_rtc_render<") + std::to_string(accel) + ", " + (other ? "true" : "false") + ">" + (length_sqrt ? " [length=sqrt]" : "");
    if (!load_module) return true;
    hipModule_t mod = nullptr;
    hipFunction_t fr = nullptr, fd = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
        log += "\nhipModuleLoadData failed";
        (void)hipGetLastError();
        return false;
    }
    if (hipModuleGetFunction(&fr, mod, "rm_rtc_render") != hipSuccess || hipModuleGetFunction(&fd, mod, "rm_rtc_distance") != hipSuccess) {
        log += "\nthe compiled module lacks rm_rtc_render / rm_rtc_distance";
        (void)hipGetLastError();
        (void)hipModuleUnload(mod);
        return false;
    }
    out.module = mod;
    out.render = fr;
    out.distance = fd;
    return true;
}

void release(Kernel &k) {
    if (k.module) (void)hipModuleUnload(static_cast<hipModule_t>(k.module));
    k = Kernel();
}

}  // namespace rmrtc
