// rm_rtc.cpp -- see rm_rtc.h.  Host only.
#include "rm_rtc.h"

#include "rm_v2_fields.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

// The device sources this library was built from, as the assembler found them at build time (csrc/Makefile runs in this
// directory and lists them as prerequisites of this object).
#define RM_EMBED(sym, file)                                                                                   \
    __asm__(".pushsection .rodata\n .hidden " #sym "\n .global " #sym "\n" #sym ":\n .incbin \"" file "\"\n .byte 0\n .popsection\n"); \
    extern "C" const char sym[];
RM_EMBED(rm_src_kernels_hip, "rm_kernels.hip")
RM_EMBED(rm_src_render_v2_hip, "rm_render_v2.hip")
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

// One object.  Every node of the tree gets a value variable v<n>; operators that move the point (a PRE half in the instruction
// stream) a point of their own q<n>{x,y,z}.  The arithmetic of a value is the interpreter's (rm_program.h, program_sdf): the
// same formula functions on the same operands; only the order in which independent subtrees are visited is this file's.
//
// Exact pruning of smooth unions / subtractions.  smoothUnion(d1, d2) = min(d1, d2) - h^2 / (4k) with h = max(k - |d1 - d2|, 0)
// (smoothUnion.ts:31-34; k4 = 4 * smoothness below) IS d1 once d2 - d1 >= k4: h is exactly +0, the quotient +0, and
// x - (+0) = x.  Likewise smoothSubtraction = max(d1, -d2) + h^2 / (4k), h = max(k4 - |d1 + d2|, 0), is d1 + 0.0 once
// d1 + d2 >= k4 and -d2 + 0.0 once d1 + d2 <= -k4.  Inside a subtree of spheres, boxes and tori under affine transforms,
// Round, SmoothUnion and SmoothSubtraction ("bounded"), every node gets a binary32 interval [lo, hi] that contains its
// value -- leaves from their own formula evaluated in binary32 (twenty instructions against a hundred and fifty binary64
// ones) with an error margin, operators from their operands' -- and an operand whose interval proves it cannot matter is
// not evaluated at all.  The Chicken (nine unions of ten boxes, k = 1e-4) evaluates one or two boxes per call instead of ten.
struct Gen {
    const RmInstr *prog;
    const rmh::ProgTreeNode *tree;
    int n_tree;
    std::string code;
    std::vector<int> value_nodes;
    bool ok = true, prune = true;

    static bool affine(const float *m) { return m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f; }
    static bool rigid(const float *m) {
        if (!affine(m)) return false;
        for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b) {
                const double dot = double(m[4 * a]) * m[4 * b] + double(m[4 * a + 1]) * m[4 * b + 1] + double(m[4 * a + 2]) * m[4 * b + 2];
                if (!(std::fabs(dot - (a == b ? 1.0 : 0.0)) <= 1e-6)) return false;
            }
        for (int k = 0; k < 16; ++k)
            if (!std::isfinite(m[k])) return false;
        return true;
    }
    const RmInstr &main_of(int n) const { return prog[tree[n].main]; }
    int kind(int n) const {  // 0 leaf, 1 round, 2 smooth union, 3 smooth subtraction, 4 point operator without a POST half
        if (tree[n].main < 0) return 4;
        const int op = main_of(n).op;
        return op < 10 ? 0 : (op == 20 ? 1 : (op == 21 ? 2 : 3));
    }
    bool bounded(int n) const {
        const rmh::ProgTreeNode &t = tree[n];
        switch (kind(n)) {
            case 0: {
                const RmInstr &I = main_of(n);
                if (I.op > 2 || !affine(I.T)) return false;
                for (int k = 0; k < 16; ++k)
                    if (!std::isfinite(I.T[k]) || std::fabs(I.T[k]) > 1e6f) return false;
                for (int k = 0; k < 3; ++k)
                    if (!std::isfinite(I.p[k]) || std::fabs(I.p[k]) > 1e6) return false;
                return true;
            }
            case 1: return std::isfinite(main_of(n).p[0]) && (t.pre < 0 || (affine(prog[t.pre].T) && affine(prog[t.pre].Tinv))) && bounded(t.a);
            case 2:
            case 3: {
                const double k4 = main_of(n).p[0] * 4.0;
                return t.pre < 0 && k4 > 0.0 && std::isfinite(k4) && bounded(t.a) && bounded(t.b);
            }
            default: return false;
        }
    }
    // Worth the interval pass: three or more leaves, or a blend narrow enough (k4 <= 0.01) that one operand nearly always
    // decides alone.  (Two leaves under a wide blend -- the "Smooth Union" presets, k4 = 0.8 -- lost 1.5 % to the estimates.)
    int leaves(int n) const { return kind(n) == 0 ? 1 : leaves(tree[n].a) + (tree[n].b >= 0 ? leaves(tree[n].b) : 0); }
    double narrowest(int n) const {
        if (kind(n) == 0) return INFINITY;
        double w = std::min(narrowest(tree[n].a), tree[n].b >= 0 ? narrowest(tree[n].b) : static_cast<double>(INFINITY));
        if (kind(n) == 2 || kind(n) == 3) w = std::min(w, main_of(n).p[0] * 4.0);
        return w;
    }
    bool worth_pruning(int n) const { return has_binary(n) && (leaves(n) >= 3 || narrowest(n) <= 0.01); }
    bool has_binary(int n) const {
        const int k = kind(n);
        if (k == 0) return false;
        if (k == 2 || k == 3) return true;
        return has_binary(tree[n].a);
    }
    static std::string P(const std::string &base, char c) { return base + c; }
    std::string v(int n) const { return "v" + std::to_string(n); }
    void line(int ind, const std::string &t) { code += std::string(static_cast<size_t>(4 * ind), ' ') + t + "\n"; }

    // local = transformMat4(point, T) of instruction I into lx, ly, lz (primitive.ts:34-35)
    void emit_local(int ind, const RmInstr &I, const std::string &pt) {
        const int fT = (I.flags & 1) | (((I.flags >> 2) & 1) << 2);
        line(ind, mat("T", I.T));
        line(ind, "float lx, ly, lz;");
        line(ind, "transform_mat4(T, " + std::to_string(fT) + ", " + P(pt, 'x') + ", " + P(pt, 'y') + ", " + P(pt, 'z') + ", lx, ly, lz);");
    }
    // `pinned`: the leaf sits under an `if` of the pruning code.  Its arithmetic is pure, and the optimiser hoisted the first
    // sixty instructions of every guarded leaf above its guard (measured: more VALU instructions than without pruning); a
    // volatile (empty) asm statement on the point cannot be speculated, and everything computed from it stays below the branch.
    void emit_leaf(int ind, int n, const std::string &pt_in, bool pinned = false) {
        const RmInstr &I = main_of(n);
        line(ind, "{  // node " + std::to_string(n) + ": leaf " + std::to_string(I.op));
        std::string pt = pt_in;
        if (pinned) {
            line(ind + 1, "float gx = " + P(pt_in, 'x') + ", gy = " + P(pt_in, 'y') + ", gz = " + P(pt_in, 'z') + ";");
            line(ind + 1, "asm volatile(\"\" : \"+v\"(gx), \"+v\"(gy), \"+v\"(gz));");
            pt = "g";
        }
        emit_local(ind + 1, I, pt);
        if (I.op == 1) line(ind + 1, v(n) + " = leaf_box(lx, ly, lz, " + lit_d(I.p[0]) + ", " + lit_d(I.p[1]) + ", " + lit_d(I.p[2]) + ");");
        else if (I.op == 2) line(ind + 1, v(n) + " = leaf_torus(lx, ly, lz, " + lit_d(I.p[0]) + ", " + lit_d(I.p[1]) + ");");
        else if (I.op == 3) {
            std::string prm = "const double prm[6] = {";
            for (int k = 0; k < 6; ++k) prm += lit_d(I.p[k]) + (k < 5 ? ", " : "");
            line(ind + 1, prm + "};");
            line(ind + 1, v(n) + " = mandelbulb_sdf(prm, lx, ly, lz, time);");
        } else if (I.op == 0) line(ind + 1, v(n) + " = leaf_sphere(lx, ly, lz, " + lit_d(I.p[0]) + ");");
        else ok = false;
        line(ind, "}");
    }
    // the PRE half of node n: the point its operands see, into q<n>{x,y,z} (declared by the caller)
    void emit_pre(int ind, int n, const std::string &pt) {
        const RmInstr &I = prog[tree[n].pre];
        const std::string q = "q" + std::to_string(n);
        line(ind, "{  // node " + std::to_string(n) + ": operator " + std::to_string(I.op) + ", the point its operands see");
        emit_local(ind + 1, I, pt);
        line(ind + 1, "float wx, wy, wz;");
        if (I.op == 15) {
            line(ind + 1, "pre_animated_translate(lx, ly, lz, " + lit_d(I.p[0]) + ", " + lit_d(I.p[1]) + ", " + lit_d(I.p[2]) + ", " + lit_d(I.p[3]) + ", " +
                              lit_d(I.p[4]) + ", time, wx, wy, wz);");
        } else {
            const int fI = ((I.flags >> 1) & 1) | (((I.flags >> 3) & 1) << 2);
            line(ind + 1, mat("Ti", I.Tinv));
            line(ind + 1, "transform_mat4(Ti, " + std::to_string(fI) + ", lx, ly, lz, wx, wy, wz);");  // "convert local position back to world space"
            if (I.op == 13) line(ind + 1, "pre_twist(" + lit_d(I.p[0]) + ", wx, wy, wz);");
            else if (I.op == 14) line(ind + 1, "pre_repetition(" + lit_d(I.p[0]) + ", " + lit_d(I.p[1]) + ", " + lit_d(I.p[2]) + ", wx, wy, wz);");
            else if (I.op < 10 || I.op > 12) ok = false;
        }
        line(ind + 1, P(q, 'x') + " = wx;");
        line(ind + 1, P(q, 'y') + " = wy;");
        line(ind + 1, P(q, 'z') + " = wz;");
        line(ind, "}");
    }
    std::string post_expr(int n) const {
        const rmh::ProgTreeNode &t = tree[n];
        const RmInstr &I = main_of(n);
        if (I.op == 20) return v(t.a) + " - " + lit_d(I.p[0]);  // round.ts:24
        return std::string(I.op == 21 ? "post_smooth_union(" : "post_smooth_subtraction(") + v(t.a) + ", " + v(t.b) + ", " + lit_d(I.p[0]) + " * 4.0)";
    }

    // ---- any node, no pruning ----
    void emit_node(int ind, int n, const std::string &pt) {
        if (!ok || n < 0 || n >= n_tree) {
            ok = false;
            return;
        }
        value_nodes.push_back(n);
        const rmh::ProgTreeNode &t = tree[n];
        if (prune && kind(n) != 0 && bounded(n) && worth_pruning(n)) return emit_bounded(ind, n, pt);
        if (kind(n) == 0) return emit_leaf(ind, n, pt);
        std::string child_pt = pt;
        if (t.pre >= 0) {
            child_pt = "q" + std::to_string(n);
            line(ind, "float " + P(child_pt, 'x') + ", " + P(child_pt, 'y') + ", " + P(child_pt, 'z') + ";");
            emit_pre(ind, n, pt);
        }
        emit_node(ind, t.a, child_pt);
        if (kind(n) == 2 || kind(n) == 3) emit_node(ind, t.b, child_pt);
        if (kind(n) == 4) line(ind, v(n) + " = " + v(t.a) + ";");
        else line(ind, v(n) + " = " + post_expr(n) + ";");
    }

    // ---- a bounded subtree: points, intervals, then only the operands that can matter ----
    void bounded_points(int ind, int n, const std::string &pt, std::vector<std::string> &point_of) {
        const rmh::ProgTreeNode &t = tree[n];
        point_of[static_cast<size_t>(n)] = pt;
        if (kind(n) == 0) return;
        std::string child_pt = pt;
        if (t.pre >= 0) {
            child_pt = "q" + std::to_string(n);
            line(ind, "float " + P(child_pt, 'x') + ", " + P(child_pt, 'y') + ", " + P(child_pt, 'z') + ";");
            emit_pre(ind, n, pt);
        }
        bounded_points(ind, t.a, child_pt, point_of);
        if (t.b >= 0) bounded_points(ind, t.b, child_pt, point_of);
    }
    void bounded_intervals(int ind, int n, const std::vector<std::string> &point_of) {
        const rmh::ProgTreeNode &t = tree[n];
        const std::string lo = "lo" + std::to_string(n), hi = "hi" + std::to_string(n);
        if (kind(n) == 0) {
            // the leaf's own formula in binary32 on binary32 local coordinates, +- a margin that covers both roundings: the
            // formulas are 1-Lipschitz in the local point, whose binary32 form is within 4 ulp of the magnitudes summed
            // (|m_ij| |p_j|, |t_i|) of the exact path's; the margin is ~40 times that.
            const RmInstr &I = main_of(n);
            const std::string &pt = point_of[static_cast<size_t>(n)];
            double mmax = 1.0, tsum = 0.0, psum = 0.0;
            for (int k = 0; k < 12; ++k) mmax = std::max(mmax, std::fabs(double(I.T[k])));
            for (int k = 12; k < 15; ++k) tsum += std::fabs(double(I.T[k]));
            for (int k = 0; k < 3; ++k) psum += std::fabs(I.p[k]);
            const float A = std::nextafter(static_cast<float>(1e-5 * 3.0 * mmax), INFINITY), B = std::nextafter(static_cast<float>(1e-5 * (tsum + psum + 1.0)), INFINITY);
            line(ind, "float " + lo + ", " + hi + ";");
            line(ind, "{  // node " + std::to_string(n) + ": binary32 estimate of leaf " + std::to_string(I.op));
            const std::string x = P(pt, 'x'), y = P(pt, 'y'), z = P(pt, 'z');
            if (I.flags & 4) {
                line(ind + 1, "const float ex = " + x + " + " + lit_f(I.T[12]) + ", ey = " + y + " + " + lit_f(I.T[13]) + ", ez = " + z + " + " + lit_f(I.T[14]) + ";");
            } else {
                for (int r = 0; r < 3; ++r)
                    line(ind + 1, std::string("const float e") + "xyz"[r] + " = " + lit_f(I.T[r]) + " * " + x + " + " + lit_f(I.T[4 + r]) + " * " + y + " + " + lit_f(I.T[8 + r]) +
                                      " * " + z + " + " + lit_f(I.T[12 + r]) + ";");
            }
            if (I.op == 0) {
                line(ind + 1, "const float est = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) - " + lit_f(static_cast<float>(I.p[0])) + ";");
            } else if (I.op == 1) {
                line(ind + 1, "const float b0 = __builtin_fabsf(ex) - " + lit_f(static_cast<float>(I.p[0])) + ", b1 = __builtin_fabsf(ey) - " + lit_f(static_cast<float>(I.p[1])) +
                                  ", b2 = __builtin_fabsf(ez) - " + lit_f(static_cast<float>(I.p[2])) + ";");
                line(ind + 1, "const float o0 = __builtin_fmaxf(b0, 0.f), o1 = __builtin_fmaxf(b1, 0.f), o2 = __builtin_fmaxf(b2, 0.f);");
                line(ind + 1, "const float est = __builtin_amdgcn_sqrtf(o0 * o0 + o1 * o1 + o2 * o2) + __builtin_fminf(__builtin_fmaxf(b0, __builtin_fmaxf(b1, b2)), 0.f);");
            } else {
                line(ind + 1, "const float qx = __builtin_amdgcn_sqrtf(ex * ex + ez * ez) - " + lit_f(static_cast<float>(I.p[0])) + ";");
                line(ind + 1, "const float est = __builtin_amdgcn_sqrtf(qx * qx + ey * ey) - " + lit_f(static_cast<float>(I.p[1])) + ";");
            }
            line(ind + 1, "const float err = (__builtin_fabsf(" + x + ") + __builtin_fabsf(" + y + ") + __builtin_fabsf(" + z + ")) * " + lit_f(A) + " + " + lit_f(B) + ";");
            line(ind + 1, lo + " = est - err;");
            line(ind + 1, hi + " = est + err;");
            line(ind, "}");
            return;
        }
        bounded_intervals(ind, t.a, point_of);
        if (t.b >= 0) bounded_intervals(ind, t.b, point_of);
        const std::string la = "lo" + std::to_string(t.a), ha = "hi" + std::to_string(t.a);
        const RmInstr &I = main_of(n);
        if (kind(n) == 1) {
            const float r = static_cast<float>(I.p[0]);
            const float ra = std::nextafter(static_cast<float>(std::fabs(I.p[0]) + 1.0), INFINITY);
            line(ind, "const float " + lo + " = (" + la + " - " + lit_f(r) + ") - (__builtin_fabsf(" + la + ") + " + lit_f(ra) + ") * 1e-6f;");
            line(ind, "const float " + hi + " = (" + ha + " - " + lit_f(r) + ") + (__builtin_fabsf(" + ha + ") + " + lit_f(ra) + ") * 1e-6f;");
            return;
        }
        const std::string lb = "lo" + std::to_string(t.b), hb = "hi" + std::to_string(t.b);
        const double k4 = I.p[0] * 4.0;
        const float q = std::nextafter(static_cast<float>(k4 * 0.25 * (1.0 + 1e-5) + 1e-7), INFINITY);  // h^2 / (4 k) <= k4 / 4
        if (kind(n) == 2) {
            line(ind, "const float " + lo + " = __builtin_fminf(" + la + ", " + lb + ") - " + lit_f(q) + ", " + hi + " = __builtin_fminf(" + ha + ", " + hb + ");");
        } else {
            line(ind, "const float " + lo + " = __builtin_fmaxf(" + la + ", -" + hb + "), " + hi + " = __builtin_fmaxf(" + ha + ", -" + lb + ") + " + lit_f(q) + ";");
        }
    }
    void bounded_eval(int ind, int n, const std::vector<std::string> &point_of) {
        const rmh::ProgTreeNode &t = tree[n];
        value_nodes.push_back(n);
        if (kind(n) == 0) {
            return emit_leaf(ind, n, point_of[static_cast<size_t>(n)], true);
        }
        if (kind(n) == 1) {
            bounded_eval(ind, t.a, point_of);
            line(ind, v(n) + " = " + post_expr(n) + ";");
            return;
        }
        const RmInstr &I = main_of(n);
        const double k4 = I.p[0] * 4.0;
        const float K = std::nextafter(static_cast<float>(k4 * (1.0 + 1e-5) + 1e-6), INFINITY);
        const std::string id = std::to_string(n), la = "lo" + std::to_string(t.a), ha = "hi" + std::to_string(t.a), lb = "lo" + std::to_string(t.b),
                          hb = "hi" + std::to_string(t.b);
        // (the margin of the binary32 comparison itself: 1e-6 of the magnitudes compared)
        if (kind(n) == 2) {
            line(ind, "const bool na" + id + " = !(" + la + " - " + hb + " > " + lit_f(K) + " + 1e-6f * (__builtin_fabsf(" + la + ") + __builtin_fabsf(" + hb + ")));");
            line(ind, "const bool nb" + id + " = !(" + lb + " - " + ha + " > " + lit_f(K) + " + 1e-6f * (__builtin_fabsf(" + lb + ") + __builtin_fabsf(" + ha + ")));");
        } else {
            line(ind, "const bool na" + id + " = !(" + ha + " + " + hb + " < -(" + lit_f(K) + " + 1e-6f * (__builtin_fabsf(" + ha + ") + __builtin_fabsf(" + hb + "))));");
            line(ind, "const bool nb" + id + " = !(" + la + " + " + lb + " > " + lit_f(K) + " + 1e-6f * (__builtin_fabsf(" + la + ") + __builtin_fabsf(" + lb + ")));");
        }
        line(ind, "if (na" + id + ") {");
        bounded_eval(ind + 1, t.a, point_of);
        line(ind, "}");
        line(ind, "if (nb" + id + ") {");
        bounded_eval(ind + 1, t.b, point_of);
        line(ind, "}");
        line(ind, "if (na" + id + " && nb" + id + ") {");
        line(ind + 1, "asm volatile(\"\" : \"+v\"(" + v(t.a) + "));  // (pinned below the branch, as the leaves are)");
        line(ind + 1, v(n) + " = " + post_expr(n) + ";");
        line(ind, "} else {");
        if (kind(n) == 2) line(ind + 1, v(n) + " = (na" + id + " ? " + v(t.a) + " : " + v(t.b) + ") - 0.0;");
        else line(ind + 1, v(n) + " = (na" + id + " ? " + v(t.a) + " : -" + v(t.b) + ") + 0.0;");
        line(ind, "}");
    }
    void emit_bounded(int ind, int n, const std::string &pt) {
        value_nodes.pop_back();  // (bounded_eval lists the node itself)
        std::vector<std::string> point_of(static_cast<size_t>(n_tree));
        line(ind, "// bounded subtree at node " + std::to_string(n) + ": points, intervals, then only the operands that can matter");
        bounded_points(ind, n, pt, point_of);
        bounded_intervals(ind, n, point_of);
        bounded_eval(ind, n, point_of);
    }
};

bool emit_object(const std::vector<RmInstr> &prog, const std::vector<rmh::ProgTreeNode> &tree, int root, int index, bool prune, std::string &out) {
    Gen g{prog.data(), tree.data(), static_cast<int>(tree.size())};
    for (const rmh::ProgTreeNode &t : tree)
        if (t.pre >= static_cast<int>(prog.size()) || t.main >= static_cast<int>(prog.size()) || t.a >= static_cast<int>(tree.size()) || t.b >= static_cast<int>(tree.size()))
            return false;
    g.prune = prune;
    g.emit_node(1, root, "p");
    if (!g.ok) return false;
    out += "__device__ __forceinline__ double rm_rtc_obj_" + std::to_string(index) + "(float px, float py, float pz, double time) {\n";
    for (int n : g.value_nodes) out += "    double v" + std::to_string(n) + " = 0.0;\n";
    out += g.code + "    return v" + std::to_string(root) + ";\n}\n";
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


// The scene's BVH as code.  The one-ray-per-lane kernels walk the node array in global memory: two dependent loads per march
// step (node, object id) plus the walk of bvh_next_interval, at five waves per SIMD -- a ONE-node tree cost a 4K frame 0.2 ms
// (a single sphere: 0.53 ms through the tree, 0.32 ms without acceleration although that evaluates more).  Expression scenes
// have a handful of objects, so their leaves become literals: BVH.getPrimitivesAt (bvh.ts:95-121) is "every leaf whose box
// contains p" (boxes nest bit for bit -- checked here --, so a leaf's ancestors contain what it contains, and the binary32
// slab test is monotone in the box, so a ray that hits a leaf hits its ancestors); the objects of a leaf are called by name.
bool emit_bvh(const std::vector<RmBvhNode> &bvh, const std::vector<int32_t> &prims, int n_obj, std::string &out) {
    const int n = static_cast<int>(bvh.size());
    if (n < 1 || n > 64) return false;
    std::vector<int> parent(static_cast<size_t>(n), -1), leaves;
    int listed = 0;
    for (int i = 0; i < n; ++i) {
        if (bvh[i].skip <= i || bvh[i].skip > n) return false;
        if (bvh[i].leaf < 0) {
            for (int c = i + 1; c < bvh[i].skip; c = bvh[c].skip) {
                if (c >= n || bvh[c].skip <= c) return false;
                parent[c] = i;
            }
            continue;
        }
        const int first = bvh[i].leaf >> 8, cnt = bvh[i].leaf & 0xFF;
        if (cnt == 0) continue;
        if (first < 0 || static_cast<size_t>(first) + static_cast<size_t>(cnt) > prims.size()) return false;
        for (int j = 0; j < cnt; ++j)
            if (prims[first + j] < 0 || prims[first + j] >= n_obj) return false;
        leaves.push_back(i);
        listed += cnt;
    }
    if (leaves.empty() || leaves.size() > 8 || listed > 32) return false;
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(bvh[i].lo[k]) || !std::isfinite(bvh[i].hi[k])) return false;
        if (parent[i] >= 0)
            for (int k = 0; k < 3; ++k)
                if (!(bvh[i].lo[k] >= bvh[parent[i]].lo[k] && bvh[i].hi[k] <= bvh[parent[i]].hi[k])) return false;
    }
    auto box = [&](int i) {
        const RmBvhNode &nd = bvh[i];
        return "const float lo[3] = {" + lit_f(nd.lo[0]) + ", " + lit_f(nd.lo[1]) + ", " + lit_f(nd.lo[2]) + "}, hi[3] = {" + lit_f(nd.hi[0]) + ", " + lit_f(nd.hi[1]) +
               ", " + lit_f(nd.hi[2]) + "};";
    };
    out += "#define RM_RTC_BVH_LEAVES " + std::to_string(leaves.size()) + "\n";
    // bvh_next_interval of rm_kernels.hip over the leaves, in node order (the order of the reference's traversal)
    out += "__device__ __forceinline__ bool rm_rtc_bvh_next_interval(const Ray &r, const RayInv &ri, double keyT, int keyOrd, Interval &out) {\n    bool have = false;\n";
    for (int i : leaves) {
        out += "    {  // node " + std::to_string(i) + "\n        " + box(i) + "\n        double tE, tX;\n";
        out += "        if (slab_inv(lo, hi, r, ri, tE, tX) && !(tX < 0.0) && !(tE > RM_MAX_DIST)) {  // bvh.ts:145,151\n";
        out += "            const double cE = tE > 0.0 ? tE : 0.0, cX = tX < RM_MAX_DIST ? tX : RM_MAX_DIST;\n";
        out += "            const bool after = cE > keyT || (cE == keyT && " + std::to_string(i) + " > keyOrd);\n";
        out += "            const bool better = !have || cE < out.tEnter;\n";
        out += "            if (after && better) {\n                out.tEnter = cE;\n                out.tExit = cX;\n                out.ord = " + std::to_string(i) +
               ";\n                have = true;\n            }\n        }\n    }\n";
    }
    out += "    return have;\n}\n";
    // bvh_distance: scene.ts:167-181 over the same leaves, the all-object fallback of scene.ts:173 behind it
    out += "__device__ __forceinline__ double rm_rtc_bvh_distance(const Vec3f &p, double time, uint32_t &count) {\n    double closest = RM_MAX_DIST;\n    uint32_t found = 0;\n";
    for (int i : leaves) {
        const int first = bvh[i].leaf >> 8, cnt = bvh[i].leaf & 0xFF;
        out += "    {  // node " + std::to_string(i) + "\n        " + box(i) + "\n        if (box_contains(lo, hi, p)) {\n";
        for (int j = 0; j < cnt; ++j)
            out += "            closest = js_min_nan(rm_rtc_obj_" + std::to_string(prims[first + j]) + "(p.x, p.y, p.z, time), closest);\n";
        out += "            found += " + std::to_string(cnt) + "u;\n        }\n    }\n";
    }
    out += "    if (found == 0) {\n";
    for (int r = 0; r < n_obj; ++r) out += "        closest = js_min_nan(rm_rtc_obj_" + std::to_string(r) + "(p.x, p.y, p.z, time), closest);\n";
    out += "        found = " + std::to_string(n_obj) + "u;\n    }\n    count += found;\n    return closest;\n}\n";
    return true;
}

}  // namespace

std::string scene_source(const std::vector<RmInstr> &prog, const std::vector<int32_t> &obj_ranges, const std::vector<rmh::ProgTreeNode> &tree,
                         const std::vector<int32_t> &roots, bool prune, const std::vector<RmBvhNode> &bvh, const std::vector<int32_t> &bvh_prims, bool require_bvh) {
    const int n_obj = static_cast<int>(obj_ranges.size() / 2);
    if (n_obj < 1 || n_obj > kMaxObjects || prog.size() > static_cast<size_t>(kMaxInstructions) || roots.size() != static_cast<size_t>(n_obj)) return std::string();
    std::string out = "// generated by rm_rtc.cpp: the scene's expression trees, one function per object\nnamespace rmd {\n";
    for (int r = 0; r < n_obj; ++r)
        if (roots[r] < 0 || roots[r] >= static_cast<int>(tree.size()) || !emit_object(prog, tree, roots[r], r, prune, out)) return std::string();
    out += "__device__ __forceinline__ double rm_rtc_object_sdf(int obj, const Vec3f &p, double time) {\n    switch (obj) {\n";
    for (int r = 0; r < n_obj; ++r)
        out += "        case " + std::to_string(r) + ": return rm_rtc_obj_" + std::to_string(r) + "(p.x, p.y, p.z, time);\n";
    out += "        default: return 0.0;\n    }\n}\n";
    std::string tree_code;
    if (emit_bvh(bvh, bvh_prims, n_obj, tree_code)) out += tree_code;
    else if (require_bvh && !bvh.empty()) return std::string();
    out += "}  // namespace rmd\n";
    return out;
}

bool available(std::string *why) {
    Rtc &r = rtc();
    if (!r.lib && why) *why = r.why;
    return r.lib != nullptr;
}

namespace {

// one hiprtc compile of one embedded source file around one generated header
struct Unit {
    std::string defines;      // #define lines ahead of the #include of `main_file`
    const char *main_file;    // "rm_kernels.hip" | "rm_render_v2.hip"
    const char *gen_name;     // the generated header's name
    const std::string *gen;   // ... and text
    const char *render_fn;    // kernel to look up (out.render)
    const char *distance_fn;  // second kernel (out.distance) or null
    std::string display;      // what rm_last_kernel reports
    bool no_spills;           // refuse a kernel that spills VGPRs or uses scratch (the v2 wave loop: the ahead-of-time build holds that line, and
                              // hipcc 7.2 can place a spill ahead of an EXEC restore: profiles/r03/spill_exec_hazard.txt)
};

// the value behind "<key>: N" of the resource-usage remarks of function `fn` (-1: not found)
long remark_of(const std::string &log, const char *fn, const char *key) {
    size_t at = log.find(std::string("Function Name: ") + fn);
    if (at == std::string::npos) return -1;
    const size_t next = log.find("Function Name: ", at + 15);
    at = log.find(std::string(key) + ": ", at);
    if (at == std::string::npos || (next != std::string::npos && at > next)) return -1;
    return std::strtol(log.c_str() + at + std::strlen(key) + 2, nullptr, 10);
}

bool compile_unit(const Unit &u, bool load_module, bool want_remarks, Kernel &out, std::string &log) {
    Rtc &r = rtc();
    if (!r.lib) {
        log = r.why;
        return false;
    }
    // RM_RTC_DEFINES (environment; experiments only): names to #define ahead of the sources, separated by blanks
    std::string extra;
    if (const char *env = std::getenv("RM_RTC_DEFINES")) {
        std::string word;
        for (const char *c = env;; ++c) {
            if (*c && *c != ' ') word += *c;
            else {
                if (!word.empty()) extra += "#define " + word + " 1\n";
                word.clear();
                if (!*c) break;
            }
        }
    }
    const std::string main_src = extra + u.defines + "#include \"" + u.main_file + "\"\n";
    const char *names[] = {"rm_kernels.hip", "rm_render_v2.hip", "rm_device.h", "rm_jsmath.h", "rm_types.h", "rm_program.h", "rm_bvh_list.h", "rm_kernels.h",
                           "rm_diag.h", u.gen_name};
    const char *texts[] = {rm_src_kernels_hip, rm_src_render_v2_hip, rm_src_device_h, rm_src_jsmath_h, rm_src_types_h, rm_src_program_h, rm_src_bvh_list_h,
                           rm_src_kernels_h, rm_src_diag_h, u.gen->c_str()};
    hiprtcProgram prog = nullptr;
    if (r.create(&prog, main_src.c_str(), "rm_rtc_main.hip", 10, texts, names) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return false;
    }
    // the flags of csrc/Makefile
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-amdgpu-inline-max-bb=100000"};
    if (want_remarks || u.no_spills) opts.push_back("-Rpass-analysis=kernel-resource-usage");
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
    if (u.no_spills) {
        const long spills = remark_of(log, u.render_fn, "VGPRs Spill"), scratch = remark_of(log, u.render_fn, "ScratchSize [bytes/lane]");
        if (spills != 0 || scratch != 0) {
            r.destroy(&prog);
            log = "refused: " + std::string(u.render_fn) + " spills " + std::to_string(spills) + " VGPRs, " + std::to_string(scratch) + " bytes of scratch per lane\n" + log;
            return false;
        }
    }
    size_t cs = 0;
    r.code_size(prog, &cs);
    std::vector<char> code(cs);
    r.code(prog, code.data());
    r.destroy(&prog);
    out.name = u.display;
    if (!load_module) return true;
    hipModule_t mod = nullptr;
    hipFunction_t fr = nullptr, fd = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
        log += "\nhipModuleLoadData failed";
        (void)hipGetLastError();
        return false;
    }
    if (hipModuleGetFunction(&fr, mod, u.render_fn) != hipSuccess || (u.distance_fn && hipModuleGetFunction(&fd, mod, u.distance_fn) != hipSuccess)) {
        log += "\nthe compiled module lacks its kernels";
        (void)hipGetLastError();
        (void)hipModuleUnload(mod);
        return false;
    }
    out.module = mod;
    out.render = fr;
    out.distance = fd;
    return true;
}

template <typename T>
std::string lit_of(T v) { return std::to_string(v); }
template <>
std::string lit_of<uint32_t>(uint32_t v) { return std::to_string(v) + "u"; }
template <>
std::string lit_of<float>(float v) { return lit_f(v); }
template <>
std::string lit_of<double>(double v) { return lit_d(v); }

}  // namespace

bool compile(const std::string &scene_src, int accel, bool other, bool length_sqrt, bool load_module, bool want_remarks, Kernel &out,
             std::string &log) {
    if (scene_src.empty()) {
        log = "no specialised source for this scene";
        return false;
    }
    Unit u;
    u.defines = "#define RM_RTC 1\n#define RM_RTC_ACCEL " + std::to_string(accel == 1 || accel == 2 ? accel : 0) + "\n#define RM_RTC_OTHER " + (other ? "1" : "0") + "\n" +
                (length_sqrt ? "#define RM_LENGTH_SQRT 1\n" : "");
    u.main_file = "rm_kernels.hip";
    u.gen_name = "rm_rtc_scene.inc";
    u.gen = &scene_src;
    u.render_fn = "rm_rtc_render";
    u.distance_fn = "rm_rtc_distance";
    u.display = std::string("rm_rtc_render<") + std::to_string(accel) + ", " + (other ? "true" : "false") + ">" + (length_sqrt ? " [length=sqrt]" : "");
    u.no_spills = false;
    return compile_unit(u, load_module, want_remarks, out, log);
}

// ---- the v2 wave loop with a launch configuration's parameters as literals (rm_v2_fields.h) -------------------------------------
std::string v2_fixed_source(const RmRenderParams &p, bool with_counts) {
    std::string s = "// generated by rm_rtc.cpp: the configuration parameters of one launch of the v2 wave loop (rm_v2_fields.h)\n"
                    "__device__ __forceinline__ void rm_v2_fix(RmRenderParams &C) {\n";
#define RM_X(f) s += "    C." #f " = " + lit_of(p.f) + ";\n";
    RM_V2_FIXED_SCALARS(RM_X)
    if (with_counts) {
        RM_V2_FIXED_COUNTS(RM_X)
    }
#undef RM_X
#define RM_X(f, n) \
    for (int k = 0; k < n; ++k) s += "    C." #f "[" + std::to_string(k) + "] = " + lit_of(p.f[k]) + ";\n";
    RM_V2_FIXED_ARRAYS(RM_X)
#undef RM_X
    return s + "}\n";
}

uint64_t v2_fixed_hash(const RmRenderParams &p, int variant_bits) {
    uint64_t h = 1469598103934665603ull ^ static_cast<uint64_t>(variant_bits);
    auto mix = [&](const void *d, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(d);
        for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
#define RM_X(f) mix(&p.f, sizeof p.f);
    RM_V2_FIXED_SCALARS(RM_X)
    RM_V2_FIXED_COUNTS(RM_X)
#undef RM_X
#define RM_X(f, n) mix(p.f, sizeof p.f);
    RM_V2_FIXED_ARRAYS(RM_X)
#undef RM_X
    return h;
}

bool compile_v2(const std::string &fixed_src, int accel, bool lds, bool ur, bool rel, bool length_sqrt, bool load_module, Kernel &out, std::string &log) {
    Unit u;
    u.defines = "#define RM_RTC_V2 1\n#define RM_RTC_V2_ACCEL " + std::to_string(accel) + "\n#define RM_RTC_V2_LDS " + (lds ? "1" : "0") + "\n#define RM_RTC_V2_UR " +
                (ur ? "1" : "0") + "\n#define RM_RTC_V2_REL " + (rel ? "1" : "0") + "\n" + (length_sqrt ? "#define RM_LENGTH_SQRT 1\n" : "");
    u.main_file = "rm_render_v2.hip";
    u.gen_name = "rm_v2_fixed.inc";
    u.gen = &fixed_src;
    u.render_fn = "rm_rtc_render_v2";
    u.distance_fn = nullptr;
    u.display = "rm_rtc_render_v2";
    u.no_spills = true;
    return compile_unit(u, load_module, false, out, log);
}

namespace {
// (never destroyed: a background compile may outlive static destruction at process exit)
struct Cache {
    std::mutex mu;
    std::map<std::string, Kernel> done;
    std::map<std::string, std::string> failed;  // key -> log
    std::map<std::string, bool> running;
};
Cache &cache() {
    static Cache *c = new Cache();
    return *c;
}
// `accel` carries the instantiation: scene kernels (other = marcher family) the acceleration structure; the v2 wave loop
// (kV2Bit set) ACCEL | LDS << 2 | UR << 3 | REL << 4
std::string cache_key(int device, const std::string &scene_src, int accel, bool other, bool length_sqrt) {
    const char *env = std::getenv("RM_RTC_DEFINES");
    return std::to_string(device) + "|" + std::to_string(accel) + (other ? "|o" : "|s") + (length_sqrt ? "|q|" : "|h|") + (env ? env : "") + "|" + scene_src;
}
bool compile_any(const std::string &src, int accel, bool other, bool length_sqrt, Kernel &out, std::string &log) {
    if (accel & kV2Bit) return compile_v2(src, accel & 3, (accel >> 2) & 1, (accel >> 3) & 1, (accel >> 4) & 1, length_sqrt, true, out, log);
    return compile(src, accel, other, length_sqrt, true, false, out, log);
}
}  // namespace

bool compile_cached(int device, const std::string &scene_src, int accel, bool other, bool length_sqrt, Kernel &out, std::string &log, bool *cached) {
    Cache &c = cache();
    const std::string key = cache_key(device, scene_src, accel, other, length_sqrt);
    std::lock_guard<std::mutex> lock(c.mu);  // (a compile in the background of the same key finishes first: it holds no lock while compiling,
                                             //  so this call may compile a second copy; the first one stored wins)
    auto it = c.done.find(key);
    if (it != c.done.end()) {
        out = it->second;
        out.compile_seconds = 0;
        if (cached) *cached = true;
        return true;
    }
    if (!compile_any(scene_src, accel, other, length_sqrt, out, log)) return false;
    const bool keep = c.done.size() < static_cast<size_t>(kCacheEntries);
    if (keep) c.done.emplace(key, out);
    if (cached) *cached = keep;
    return true;
}

int compile_async(int device, const std::string &scene_src, int accel, bool other, bool length_sqrt, Kernel &out, std::string &log) {
    Cache &c = cache();
    const std::string key = cache_key(device, scene_src, accel, other, length_sqrt);
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.done.find(key);
        if (it != c.done.end()) {
            out = it->second;
            out.compile_seconds = 0;
            return 1;
        }
        auto f = c.failed.find(key);
        if (f != c.failed.end()) {
            log = f->second;
            return -1;
        }
        if (c.running.count(key)) return 0;
        if (c.done.size() >= static_cast<size_t>(kCacheEntries)) {
            log = "the kernel cache is full";
            return -1;
        }
        c.running[key] = true;
    }
    std::thread([key, device, scene_src, accel, other, length_sqrt] {
        Kernel k;
        std::string text;
        const bool ok = hipSetDevice(device) == hipSuccess && compile_any(scene_src, accel, other, length_sqrt, k, text);
        Cache &cc = cache();
        std::lock_guard<std::mutex> lock(cc.mu);
        if (ok && !cc.done.count(key)) cc.done.emplace(key, k);
        else if (!ok) cc.failed[key] = text;
        cc.running.erase(key);
    }).detach();
    return 0;
}

void release(Kernel &k) {
    if (k.module) (void)hipModuleUnload(static_cast<hipModule_t>(k.module));
    k = Kernel();
}

}  // namespace rmrtc
