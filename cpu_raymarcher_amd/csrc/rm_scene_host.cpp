// rm_scene_host.cpp -- see rm_scene_host.h.
//
// Arithmetic contract: the reference computes in JS doubles and
// rounds to binary32 wherever it stores into a gl-matrix vector/matrix, so every helper
// here evaluates in double, one rounding per operation (build with -ffp-contract=off),
// and narrows to float exactly where the reference stores.
#include "rm_scene_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace rmh {

namespace {

inline float to_f32(double v) { return static_cast<float>(v); }

// ---- 4x4 helpers on Float32Array-like storage (column-major, gl-matrix 3.x) ----------

struct Mat4 {
    float m[16];
    static Mat4 identity() {
        Mat4 r;
        for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        return r;
    }
};

// mat4.rotateY(out, a, rad) for out != a
Mat4 rotate_y(const Mat4 &a, double rad) {
    const double s = std::sin(rad), c = std::cos(rad);
    Mat4 r = a;
    for (int k = 0; k < 4; ++k) {
        const double row0 = a.m[k], row2 = a.m[8 + k];
        r.m[k] = to_f32(row0 * c - row2 * s);
        r.m[8 + k] = to_f32(row0 * s + row2 * c);
    }
    return r;
}

// mat4.rotateX(out, a, rad) for out != a
Mat4 rotate_x(const Mat4 &a, double rad) {
    const double s = std::sin(rad), c = std::cos(rad);
    Mat4 r = a;
    for (int k = 0; k < 4; ++k) {
        const double row1 = a.m[4 + k], row2 = a.m[8 + k];
        r.m[4 + k] = to_f32(row1 * c + row2 * s);
        r.m[8 + k] = to_f32(row2 * c - row1 * s);
    }
    return r;
}

// mat4.translate(out, a, v) for out != a
Mat4 translate(const Mat4 &a, const float v[3]) {
    Mat4 r = a;
    const double x = v[0], y = v[1], z = v[2];
    for (int k = 0; k < 4; ++k)
        r.m[12 + k] = to_f32(double(a.m[k]) * x + double(a.m[4 + k]) * y + double(a.m[8 + k]) * z +
                             double(a.m[12 + k]));
    return r;
}

// JS Math.min / Math.max for finite-or-infinite, non-NaN operands with -0 < +0
inline double js_min2(double a, double b) {
    if (a == 0.0 && b == 0.0) return std::signbit(a) ? a : b;
    return a < b ? a : b;
}
inline double js_max2(double a, double b) {
    if (a == 0.0 && b == 0.0) return std::signbit(a) ? b : a;
    return a > b ? a : b;
}

struct Box {
    float lo[3], hi[3];
};

inline bool boxes_touch(const Box &a, const float *blo, const float *bhi) {  // boundingBox.ts:24-30
    for (int k = 0; k < 3; ++k)
        if (!(a.lo[k] <= bhi[k] && a.hi[k] >= blo[k])) return false;
    return true;
}

double box_gap(const Box &a, const float *blo, const float *bhi) {  // boundingBox.ts:33-47
    double g[3];
    for (int k = 0; k < 3; ++k) {
        g[k] = 0.0;
        if (a.hi[k] < blo[k]) g[k] = double(blo[k]) - double(a.hi[k]);
        else if (bhi[k] < a.lo[k]) g[k] = double(a.lo[k]) - double(bhi[k]);
    }
    return js_hypot3(g[0], g[1], g[2]);
}

// computeBounds (boundingBox.ts:158-169): left fold of merge; values are f32 already
Box bounds_of(const HostScene &s, const int32_t *ids, int n) {
    Box b;
    if (n == 0) {
        std::memset(&b, 0, sizeof b);
        return b;
    }
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = s.prim_lo[3 * ids[0] + k];
        b.hi[k] = s.prim_hi[3 * ids[0] + k];
    }
    for (int i = 1; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = to_f32(js_min2(b.lo[k], s.prim_lo[3 * ids[i] + k]));
            b.hi[k] = to_f32(js_max2(b.hi[k], s.prim_hi[3 * ids[i] + k]));
        }
    return b;
}

// ---- BVH (bvh.ts:29-92), emitted directly in right-first pre-order ---------------------

struct BvhBuilder {
    HostScene &s;
    std::string &err;
    bool ok = true;

    int emit(std::vector<int32_t> &ids, const Box &box, int depth) {
        const int me = static_cast<int>(s.bvh.size());
        RmBvhNode node;
        std::memcpy(node.lo, box.lo, sizeof node.lo);
        std::memcpy(node.hi, box.hi, sizeof node.hi);
        node.skip = 0;
        node.leaf = -1;
        s.bvh.push_back(node);
        s.bvh_depth = std::max(s.bvh_depth, depth);
        const int n = static_cast<int>(ids.size());

        auto make_leaf = [&]() {
            if (n > RM_BVH_LEAF_MAX) {
                ok = false;
                err = "BVH leaf with more than 255 primitives (depth limit 20 reached)";
            }
            const int first = static_cast<int>(s.bvh_prims.size());
            s.bvh_prims.insert(s.bvh_prims.end(), ids.begin(), ids.end());
            s.bvh[me].leaf = (first << 8) | (n & 0xFF);
            s.bvh[me].skip = static_cast<int>(s.bvh.size());
            s.bvh_leaves++;
            return me;
        };

        if (depth >= 20 || n <= 2) return make_leaf();  // bvh.ts:52

        // bvh.ts:58-63: longest axis of the f32 extent, strict > so ties keep x, then y
        float ext[3];
        for (int k = 0; k < 3; ++k) ext[k] = to_f32(double(box.hi[k]) - double(box.lo[k]));
        int axis = 0;
        if (ext[1] > ext[0]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;

        // bvh.ts:66-70: stable sort by world position on that axis (centre, f32)
        std::vector<int32_t> order(ids);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            const float ca = s.world_pos[3 * a + axis], cb = s.world_pos[3 * b + axis];
            return double(ca) - double(cb) < 0.0;
        });
        const int mid = n / 2;  // bvh.ts:73
        std::vector<int32_t> left(order.begin(), order.begin() + mid);
        std::vector<int32_t> right(order.begin() + mid, order.end());
        if (left.empty() || right.empty()) return make_leaf();  // bvh.ts:78-81

        const Box lb = bounds_of(s, left.data(), static_cast<int>(left.size()));
        const Box rb = bounds_of(s, right.data(), static_cast<int>(right.size()));
        emit(right, rb, depth + 1);  // popped first by bvh.ts:160-161
        emit(left, lb, depth + 1);
        s.bvh[me].skip = static_cast<int>(s.bvh.size());
        return me;
    }
};

// ---- point-query grid over the BVH leaves -------------------------------------------------
//
// Not a reference structure: an index over the leaves the reference's recursive descent
// (bvh.ts:101-121) would reach.  Leaf boxes are unions of the padded primitive boxes and
// every ancestor box is the exact f32 min/max of its leaves' boxes, so "p lies in leaf box L"
// already implies "p lies in every ancestor of L": the set the descent returns is exactly
// {leaves whose box contains p}.  The grid lists, per cell, every leaf whose box overlaps the
// cell grown by 1 % of a cell (the device computes the cell index in binary32, error ~1e-5
// cell), and the device re-tests each listed leaf box with the reference's inclusive f32
// compares, so the result set is identical.
void build_point_query_grid(HostScene &s) {
    const int leaves = s.bvh_leaves;
    if (s.bvh.empty() || leaves == 0) return;
    int g = static_cast<int>(std::ceil(std::cbrt(static_cast<double>(leaves)) * 3.0));
    g = std::max(2, std::min(g, 32));
    size_t cells = 1;
    for (int k = 0; k < 3; ++k) {
        const double ext = double(s.root_max[k]) - double(s.root_min[k]);
        s.pq_dim[k] = ext > 0 ? g : 1;
        s.pq_origin[k] = s.root_min[k];
        s.pq_inv[k] = ext > 0 ? static_cast<float>(s.pq_dim[k] / ext) : 0.0f;
        cells *= static_cast<size_t>(s.pq_dim[k]);
    }
    std::vector<std::vector<uint16_t>> per_cell(cells);
    bool ok = s.bvh.size() < 65536;
    for (size_t i = 0; ok && i < s.bvh.size(); ++i) {
        const RmBvhNode &nd = s.bvh[i];
        if (nd.leaf < 0 || (nd.leaf & 0xFF) == 0) continue;
        int c0[3], c1[3];
        for (int k = 0; k < 3; ++k) {
            // + 0.0101 in world units: the v2 kernel's fused normal evaluation asks, from the cell of the hit point, about
            // the three points 0.01 beside it (raymarcher.ts:126-132; rm_render_v2.hip RM_NRM_DELTA), so a cell also lists
            // every leaf that can contain a point within that distance of the cell
            const double a = (double(nd.lo[k]) - 0.0101 - double(s.pq_origin[k])) * double(s.pq_inv[k]) - 0.01;
            const double b = (double(nd.hi[k]) + 0.0101 - double(s.pq_origin[k])) * double(s.pq_inv[k]) + 0.01;
            c0[k] = std::max(0, std::min(s.pq_dim[k] - 1, static_cast<int>(std::floor(a))));
            c1[k] = std::max(0, std::min(s.pq_dim[k] - 1, static_cast<int>(std::floor(b))));
        }
        for (int z = c0[2]; z <= c1[2]; ++z)
            for (int y = c0[1]; y <= c1[1]; ++y)
                for (int x = c0[0]; x <= c1[0]; ++x)
                    per_cell[(static_cast<size_t>(z) * s.pq_dim[1] + y) * s.pq_dim[0] + x].push_back(
                        static_cast<uint16_t>(i));
    }
    if (!ok) {  // node ids do not fit 16 bits: no grid, the device walks the tree
        s.pq_dim[0] = s.pq_dim[1] = s.pq_dim[2] = 0;
        return;
    }
    s.pq_cells.assign(cells, 0);
    for (size_t c = 0; c < cells; ++c) {
        const size_t n = per_cell[c].size();
        if (n >= 255 || s.pq_list.size() + n >= (1u << 24)) {
            s.pq_cells[c] = 255;  // too crowded: walk the tree for points of this cell
            continue;
        }
        s.pq_cells[c] = static_cast<uint32_t>(s.pq_list.size() << 8) | static_cast<uint32_t>(n);
        s.pq_list.insert(s.pq_list.end(), per_cell[c].begin(), per_cell[c].end());
    }

    // Nearest-candidate lists.  For a cell C (grown by 1 % like above) and sphere j with centre
    // c_j, radius r_j:  lb_j = max(0, dist(C, c_j)) - r_j  <=  Sphere.sdf_j(p)  <=  maxdist(C, c_j) - r_j = ub_j
    // for every p in C.  With U = min(10, min_j ub_j), a sphere with lb_j > U can never attain
    // min(10, min_j sdf_j(p)) for p in C, so evaluating only {j : lb_j <= U + margin} gives the
    // same minimum (min is order independent).  margin absorbs the difference between these
    // real-number bounds and the reference's rounded arithmetic (~1e-15) by many orders.
    // The candidate grid is finer than the leaf grid (cell diagonal << sphere spacing keeps the lists at a
    // handful of spheres): up to 48 cells per axis, bounded by the host work cells * n.
    if (s.general || s.program) return;  // nearest-candidate lists need exact sphere distances (general scenes keep bounding spheres here)
    const size_t n = s.spheres.size();
    if (n == 0 || n > 2048) return;  // used for scenes of <= 512 spheres by default (rm_api.cpp), on request up to 2048
    int ng = 48;
    while (ng > 4 && static_cast<unsigned long long>(ng) * ng * ng * n > 40000000ull) ng -= 4;
    size_t ncells = 1;
    for (int k = 0; k < 3; ++k) {
        const double ext = double(s.root_max[k]) - double(s.root_min[k]);
        s.nn_dim[k] = ext > 0 ? ng : 1;
        s.nn_inv[k] = ext > 0 ? static_cast<float>(s.nn_dim[k] / ext) : 0.0f;
        ncells *= static_cast<size_t>(s.nn_dim[k]);
    }
    s.nn_cells.assign(ncells, 255);
    const double margin = 1e-6;
    std::vector<double> lb(n);
    for (int z = 0; z < s.nn_dim[2]; ++z)
        for (int y = 0; y < s.nn_dim[1]; ++y)
            for (int x = 0; x < s.nn_dim[0]; ++x) {
                const int ci[3] = {x, y, z};
                double lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    const double w = s.nn_inv[k] > 0 ? 1.0 / double(s.nn_inv[k]) : 0.0;
                    // 3 % slack: the device's binary32 cell index is off by <= 1e-4 cell, and border cells
                    // receive the clamped indices of points that round outside the grid range
                    lo[k] = double(s.pq_origin[k]) + (ci[k] - 0.03) * w;
                    hi[k] = double(s.pq_origin[k]) + (ci[k] + 1.03) * w;
                }
                double U = 10.0;
                for (size_t j = 0; j < n; ++j) {
                    const double c[3] = {s.spheres[j].cx, s.spheres[j].cy, s.spheres[j].cz};
                    double dmin2 = 0, dmax2 = 0;
                    for (int k = 0; k < 3; ++k) {
                        const double below = lo[k] - c[k], above = c[k] - hi[k];
                        const double dmin = below > 0 ? below : (above > 0 ? above : 0.0);
                        const double dmax = std::max(std::fabs(c[k] - lo[k]), std::fabs(c[k] - hi[k]));
                        dmin2 += dmin * dmin;
                        dmax2 += dmax * dmax;
                    }
                    lb[j] = std::sqrt(dmin2) - s.radii[j];
                    const double ub = std::sqrt(dmax2) - s.radii[j];
                    if (ub < U) U = ub;
                }
                std::vector<uint16_t> cand;
                for (size_t j = 0; j < n; ++j)
                    if (lb[j] <= U + margin) cand.push_back(static_cast<uint16_t>(j));
                const size_t c_idx = (static_cast<size_t>(z) * s.nn_dim[1] + y) * s.nn_dim[0] + x;
                if (cand.size() >= 255 || s.nn_list.size() + cand.size() >= (1u << 24)) continue;  // stays 255
                s.nn_cells[c_idx] = static_cast<uint32_t>(s.nn_list.size() << 8) | static_cast<uint32_t>(cand.size());
                s.nn_list.insert(s.nn_list.end(), cand.begin(), cand.end());
            }
}

// ---- Octree (octree.ts:36-191) ----------------------------------------------------------

struct OctBuilder {
    HostScene &s;

    void fill(int me, const std::vector<int32_t> &ids, const Box &box, int depth) {
        RmOctNode &node = s.oct[me];
        std::memcpy(node.lo, box.lo, sizeof node.lo);
        std::memcpy(node.hi, box.hi, sizeof node.hi);
        node.first_child = -1;
        node.prim_first = 0;
        node.prim_count = 0;
        node.is_empty = 1;
        node.min_distance = 0.0;
        node.center[0] = node.center[1] = node.center[2] = 0.0f;
        node.sub_first = -1;
        const int n = static_cast<int>(ids.size());
        if (depth >= 6 || n <= 4) {  // octree.ts:60
            s.oct[me].prim_first = static_cast<int>(s.oct_prims.size());
            s.oct[me].prim_count = n;
            s.oct_prims.insert(s.oct_prims.end(), ids.begin(), ids.end());
            return;
        }
        float c[3];  // boundingBox.ts:108-114, stored f32
        for (int k = 0; k < 3; ++k) c[k] = to_f32((double(box.lo[k]) + double(box.hi[k])) / 2);
        Box kid[8];
        for (int i = 0; i < 8; ++i)
            for (int k = 0; k < 3; ++k) {
                const bool upper = (i >> k) & 1;  // index = x + 2y + 4z (octree.ts:69-71)
                kid[i].lo[k] = upper ? c[k] : box.lo[k];
                kid[i].hi[k] = upper ? box.hi[k] : c[k];
            }
        std::vector<int32_t> share[8];
        for (int32_t id : ids)  // octree.ts:93-103
            for (int i = 0; i < 8; ++i)
                if (boxes_touch(kid[i], &s.prim_lo[3 * id], &s.prim_hi[3 * id])) share[i].push_back(id);
        const int first = static_cast<int>(s.oct.size());
        s.oct.resize(s.oct.size() + 8);
        s.oct[me].first_child = first;
        std::memcpy(s.oct[me].center, c, sizeof c);
        for (int i = 0; i < 8; ++i) {
            if (!share[i].empty()) fill(first + i, share[i], kid[i], depth + 1);
            else fill(first + i, std::vector<int32_t>(), kid[i], 7);  // empty leaf, octree.ts:110-114
        }
    }

    double nearest_prim_box(const RmOctNode &node) const {  // octree.ts:155-160
        Box b;
        std::memcpy(b.lo, node.lo, sizeof b.lo);
        std::memcpy(b.hi, node.hi, sizeof b.hi);
        double best = std::numeric_limits<double>::infinity();
        const int n = static_cast<int>(s.prim_lo.size() / 3);
        for (int i = 0; i < n; ++i) {
            const double d = box_gap(b, &s.prim_lo[3 * i], &s.prim_hi[3 * i]);
            if (d < best) best = d;
        }
        return best != std::numeric_limits<double>::infinity() ? js_max2(0.0, best) : 0.0;
    }

    bool mark(int me) {  // octree.ts:149-191
        RmOctNode &node = s.oct[me];
        if (node.first_child < 0) {
            const bool has = node.prim_count > 0;
            node.is_empty = has ? 0 : 1;
            node.min_distance = has ? 0.0 : nearest_prim_box(node);
            s.oct_leaves++;
            if (!has) s.oct_empty++;
            s.oct_max_leaf = std::max(s.oct_max_leaf, node.prim_count);
            return has;
        }
        bool any = false;
        const int first = node.first_child;
        for (int i = 0; i < 8; ++i)
            if (mark(first + i)) any = true;
        s.oct[me].is_empty = any ? 0 : 1;
        s.oct[me].min_distance = any ? 0.0 : nearest_prim_box(s.oct[me]);
        return any;
    }
};

}  // namespace

// gl-matrix vec3.length / vec3.distance as the scene builder uses them (box.ts:33, smoothUnion.ts:45): Math.hypot in
// 3.0 - 3.4.3, or Math.sqrt(x*x + y*y + z*z) when the ctx's option `length` is 1 (set per build by the API layer).
namespace { thread_local int g_length_sqrt = 0; }
void set_length_mode(int use_sqrt) { g_length_sqrt = use_sqrt ? 1 : 0; }
double vec3_length_host(double x, double y, double z) {
    if (g_length_sqrt) return std::sqrt(x * x + y * y + z * z);
    return js_hypot3(x, y, z);
}

double js_hypot3(double x, double y, double z) {
    double v[3] = {std::fabs(x), std::fabs(y), std::fabs(z)};
    double big = 0.0;
    bool nan = false;
    for (double a : v) {
        if (a != a) nan = true;
        else if (a > big) big = a;
    }
    if (big == std::numeric_limits<double>::infinity()) return big;
    if (nan) return std::numeric_limits<double>::quiet_NaN();
    if (big == 0.0) return 0.0;
    double sum = 0.0, comp = 0.0;
    for (double a : v) {
        const double q = a / big;
        const double term = q * q - comp;
        const double next = sum + term;
        comp = (next - sum) - term;
        sum = next;
    }
    return std::sqrt(sum) * big;
}

bool preset_spheres(int index, std::vector<float> &centers, std::vector<double> &radii) {
    centers.clear();
    radii.clear();
    index = std::max(0, std::min(index, kPresetCount - 1));  // scene.ts:39
    auto add = [&](double x, double y, double z, double r) {
        // createSphere -> getTransform: translation stored f32 (sceneManager.ts:31)
        centers.push_back(to_f32(x));
        centers.push_back(to_f32(y));
        centers.push_back(to_f32(z));
        radii.push_back(r);
    };
    switch (index) {
        case 0:  // "Sphere"
            add(0, 0, 0, 1.5);
            return true;
        case 1:  // "Random Spheres"
            add(0.8, -0.3, 0.2, 0.4);
            add(-0.5, 0.9, -0.1, 0.5);
            add(0.2, 0.1, 0.8, 0.3);
            add(-0.9, -0.4, -0.6, 0.6);
            add(0.4, -0.8, 0.5, 0.35);
            add(-0.2, 0.6, -0.9, 0.4);
            add(0.7, 0.3, -0.4, 0.25);
            return true;
        case 2:  // "Grid of Spheres"
            for (int y = -1; y <= 1; ++y)
                for (int x = -1; x <= 1; ++x) add(x, y, 0, 0.3);
            return true;
        case 3: {  // "Dense Sphere Grid"
            const int grid = 5;
            const double spacing = 0.6;
            const double offset = (grid - 1) * spacing / 2;
            for (int x = 0; x < grid; ++x)
                for (int y = 0; y < grid; ++y)
                    for (int z = 0; z < grid; ++z)
                        add(x * spacing - offset, y * spacing - offset, z * spacing - offset, 0.15);
            return true;
        }
        case 4:  // "Atom"
            add(0, 0, 0, 0.5);
            add(1.2, 0, 0, 0.3);
            add(-1.2, 0, 0, 0.3);
            add(0, 1.2, 0, 0.3);
            add(0, -1.2, 0, 0.3);
            add(0, 0, 1.2, 0.3);
            add(0, 0, -1.2, 0.3);
            return true;
        default:
            return false;  // box / torus / mandelbulb / operator presets: out of scope
    }
}

// BVH / Octree over the padded primitive boxes already in s.prim_lo / prim_hi
static bool build_accel(HostScene &s, int n, std::string &err) {
    std::vector<int32_t> all(n);
    std::iota(all.begin(), all.end(), 0);
    if (s.accel == 2) {
        const Box root = bounds_of(s, all.data(), n);  // bvh.ts:38-41 (ctor bounds arg ignored)
        BvhBuilder b{s, err};
        b.emit(all, root, 0);
        if (!b.ok) return false;
        std::memcpy(s.root_min, root.lo, sizeof root.lo);
        std::memcpy(s.root_max, root.hi, sizeof root.hi);
        build_point_query_grid(s);  // the leaf grid only knows leaf boxes: it serves every primitive representation
    } else if (s.accel == 1) {
        Box root;  // scene.ts:81-85
        for (int k = 0; k < 3; ++k) {
            root.lo[k] = -10.0f;
            root.hi[k] = 10.0f;
        }
        s.oct.resize(1);
        OctBuilder b{s};
        b.fill(0, all, root, 0);
        b.mark(0);
        {  // flat findNode table (see rm_scene_host.h); left empty if a leaf is not a box of whole cells
            std::vector<int32_t> lut(64 * 64 * 64, -1);
            bool ok = true;
            for (size_t i = 0; i < s.oct.size() && ok; ++i) {
                const RmOctNode &nd = s.oct[i];
                if (nd.first_child >= 0) continue;
                int k0[3], k1[3];
                for (int a = 0; a < 3; ++a) {
                    const double f0 = (double(nd.lo[a]) + 10.0) / 0.3125, f1 = (double(nd.hi[a]) + 10.0) / 0.3125;
                    k0[a] = static_cast<int>(f0);
                    k1[a] = static_cast<int>(f1);
                    if (double(k0[a]) != f0 || double(k1[a]) != f1 || k0[a] < 0 || k1[a] > 64 || k0[a] >= k1[a]) ok = false;
                }
                if (!ok) break;
                for (int z = k0[2]; z < k1[2]; ++z)
                    for (int y = k0[1]; y < k1[1]; ++y)
                        for (int x = k0[0]; x < k1[0]; ++x) lut[(size_t(z) * 64 + y) * 64 + x] = static_cast<int32_t>(i);
            }
            for (int32_t v : lut)
                if (v < 0) ok = false;
            if (ok) s.oct_lut = std::move(lut);
        }
        if (!s.general) {  // sub-cell candidate lists of crowded leaves (see rm_scene_host.h)
            const double margin = 1e-6;
            for (size_t i = 0; i < s.oct.size(); ++i) {
                RmOctNode &nd = s.oct[i];
                const int n = nd.prim_count;
                if (nd.first_child >= 0 || n <= 8 || n > 255) continue;
                if (s.oct_sub_list.size() + size_t(RM_OCT_SUB) * RM_OCT_SUB * RM_OCT_SUB * size_t(n) >= (1u << 24)) break;
                nd.sub_first = static_cast<int32_t>(s.oct_sub_hdr.size());
                // a leaf has no split point: `center` carries the sub-cells per unit length instead, so the kernel indexes a
                // sub-cell with three multiplications and no division (binary32 quotient, as the kernel used to form it)
                for (int k = 0; k < 3; ++k) nd.center[k] = static_cast<float>(RM_OCT_SUB) / (nd.hi[k] - nd.lo[k]);
                std::vector<double> lb(n);
                for (int cz = 0; cz < RM_OCT_SUB; ++cz)
                    for (int cy = 0; cy < RM_OCT_SUB; ++cy)
                        for (int cx = 0; cx < RM_OCT_SUB; ++cx) {
                            const int ci[3] = {cx, cy, cz};
                            double lo[3], hi[3];
                            for (int k = 0; k < 3; ++k) {  // 3 % slack: the device's binary32 cell index, clamped at the border
                                const double w = (double(nd.hi[k]) - double(nd.lo[k])) / double(RM_OCT_SUB);
                                lo[k] = double(nd.lo[k]) + (ci[k] - 0.03) * w;
                                hi[k] = double(nd.lo[k]) + (ci[k] + 1.03) * w;
                            }
                            double U = std::numeric_limits<double>::infinity();
                            for (int j = 0; j < n; ++j) {
                                const int id = s.oct_prims[nd.prim_first + j];
                                const double c[3] = {s.spheres[id].cx, s.spheres[id].cy, s.spheres[id].cz};
                                double dmin2 = 0, dmax2 = 0;
                                for (int k = 0; k < 3; ++k) {
                                    const double below = lo[k] - c[k], above = c[k] - hi[k];
                                    const double dmin = below > 0 ? below : (above > 0 ? above : 0.0);
                                    const double dmax = std::max(std::fabs(c[k] - lo[k]), std::fabs(c[k] - hi[k]));
                                    dmin2 += dmin * dmin;
                                    dmax2 += dmax * dmax;
                                }
                                lb[j] = std::sqrt(dmin2) - s.radii[id];
                                U = std::min(U, std::sqrt(dmax2) - s.radii[id]);
                            }
                            const size_t first = s.oct_sub_list.size();
                            uint32_t cnt = 0;
                            for (int j = 0; j < n; ++j)
                                if (lb[j] <= U + margin) {
                                    s.oct_sub_list.push_back(static_cast<uint8_t>(j));
                                    ++cnt;
                                }
                            s.oct_sub_hdr.push_back(static_cast<uint32_t>(first << 8) | cnt);
                        }
            }
        }
        if (!s.general) {  // leaf-ordered sphere records for the device's leaf loops
            s.oct_recs.resize(s.oct_prims.size());
            for (size_t k = 0; k < s.oct_prims.size(); ++k) {
                const int id = s.oct_prims[k];
                const RmSphere &sp = s.spheres[id];
                s.oct_recs[k] = RmSphereRec{sp.cx, sp.cy, sp.cz, sp.rf, s.radii[id], id, 0};
            }
        }
        std::memcpy(s.root_min, root.lo, sizeof root.lo);
        std::memcpy(s.root_max, root.hi, sizeof root.hi);
    }
    return true;
}

bool build_scene(HostScene &s, const float *centers, const double *radii, int n, int accel,
                 std::string &err) {
    if (n < 0 || (n > 0 && (!centers || !radii))) {
        err = "bad sphere list";
        return false;
    }
    for (int i = 0; i < n; ++i) {
        if (!std::isfinite(centers[3 * i]) || !std::isfinite(centers[3 * i + 1]) ||
            !std::isfinite(centers[3 * i + 2]) || !std::isfinite(radii[i])) {
            err = "non-finite sphere centre or radius";
            return false;
        }
    }
    s = HostScene();
    s.accel = (accel == 1 || accel == 2) ? accel : 0;
    s.spheres.resize(n);
    s.radii.assign(radii, radii + n);
    s.prim_lo.resize(3 * size_t(n));
    s.prim_hi.resize(3 * size_t(n));
    for (int i = 0; i < n; ++i) {
        s.spheres[i] = RmSphere{centers[3 * i], centers[3 * i + 1], centers[3 * i + 2], to_f32(radii[i])};
        // BoundingBox.fromPrimitive (boundingBox.ts:133-154): local->world is identity
        // + translation, so each column has Math.hypot(1,0,0) = 1 and maxScale = 1.
        const double scale = js_max2(js_max2(js_hypot3(1, 0, 0), js_hypot3(0, 1, 0)), js_hypot3(0, 0, 1));
        const double pad = radii[i] * scale * 1.5;
        for (int k = 0; k < 3; ++k) {
            s.prim_lo[3 * i + k] = to_f32(double(centers[3 * i + k]) - pad);
            s.prim_hi[3 * i + k] = to_f32(double(centers[3 * i + k]) + pad);
        }
    }
    s.world_pos.assign(centers, centers + 3 * size_t(n));
    if (!build_accel(s, n, err)) return false;
    // BVH: every sphere lives in exactly one leaf, so the device arrays can be permuted into leaf order; a leaf's
    // spheres are then spheres[first .. first+count) and the id indirection (one dependent LDS read per sphere)
    // disappears.  min() and the counters do not depend on the order.
    if (s.accel == 2 && s.bvh_prims.size() == size_t(n) && n > 0) {
        std::vector<RmSphere> sp(n);
        std::vector<double> rd(n);
        std::vector<int32_t> pos(n, -1);
        bool perm = true;
        for (int k = 0; k < n; ++k) {
            const int id = s.bvh_prims[k];
            if (id < 0 || id >= n || pos[id] >= 0) {
                perm = false;
                break;
            }
            pos[id] = k;
            sp[k] = s.spheres[id];
            rd[k] = s.radii[id];
        }
        if (perm) {
            s.spheres.swap(sp);
            s.radii.swap(rd);
            std::iota(s.bvh_prims.begin(), s.bvh_prims.end(), 0);
            for (uint16_t &e : s.nn_list) e = static_cast<uint16_t>(pos[e]);
            s.leaf_order = true;
        }
    }
    return true;
}

// ---- general primitives -----------------------------------------------------------------------

namespace {

// mat4.invert (gl-matrix 3.x cofactor form); false when the determinant is falsy
bool invert4(const float a[16], float out[16]) {
    const double a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    const double a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
    const double b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10;
    const double b03 = a01 * a12 - a02 * a11, b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
    const double b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30, b08 = a20 * a33 - a23 * a30;
    const double b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
    double det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
    if (!(det != 0.0)) return false;
    det = 1.0 / det;
    float o[16];
    o[0] = to_f32((a11 * b11 - a12 * b10 + a13 * b09) * det);
    o[1] = to_f32((a02 * b10 - a01 * b11 - a03 * b09) * det);
    o[2] = to_f32((a31 * b05 - a32 * b04 + a33 * b03) * det);
    o[3] = to_f32((a22 * b04 - a21 * b05 - a23 * b03) * det);
    o[4] = to_f32((a12 * b08 - a10 * b11 - a13 * b07) * det);
    o[5] = to_f32((a00 * b11 - a02 * b08 + a03 * b07) * det);
    o[6] = to_f32((a32 * b02 - a30 * b05 - a33 * b01) * det);
    o[7] = to_f32((a20 * b05 - a22 * b02 + a23 * b01) * det);
    o[8] = to_f32((a10 * b10 - a11 * b08 + a13 * b06) * det);
    o[9] = to_f32((a01 * b08 - a00 * b10 - a03 * b06) * det);
    o[10] = to_f32((a30 * b04 - a31 * b02 + a33 * b00) * det);
    o[11] = to_f32((a21 * b02 - a20 * b04 - a23 * b00) * det);
    o[12] = to_f32((a11 * b07 - a10 * b09 - a12 * b06) * det);
    o[13] = to_f32((a00 * b09 - a01 * b07 + a02 * b06) * det);
    o[14] = to_f32((a31 * b01 - a30 * b03 - a32 * b00) * det);
    o[15] = to_f32((a20 * b03 - a21 * b01 + a22 * b00) * det);
    std::memcpy(out, o, sizeof o);
    return true;
}

// mat4.rotateZ (same arithmetic in place or not)
Mat4 rotate_z(const Mat4 &a, double rad) {
    const double s = std::sin(rad), c = std::cos(rad);
    Mat4 r = a;
    for (int k = 0; k < 4; ++k) {
        const double row0 = a.m[k], row1 = a.m[4 + k];
        r.m[k] = to_f32(row0 * c + row1 * s);
        r.m[4 + k] = to_f32(row1 * c - row0 * s);
    }
    return r;
}

}  // namespace

void make_transform(double x, double y, double z, const float *rot, float out16[16]) {
    Mat4 model = Mat4::identity();
    model.m[12] = to_f32(x);  // fromTranslation / fromRotationTranslationScale(identity quat, v, 1)
    model.m[13] = to_f32(y);
    model.m[14] = to_f32(z);
    if (rot) model = rotate_z(rotate_y(rotate_x(model, rot[0]), rot[1]), rot[2]);  // sceneManager.ts:26-29
    Mat4 inv = Mat4::identity();
    invert4(model.m, inv.m);  // sceneManager.ts:34-35
    std::memcpy(out16, inv.m, sizeof inv.m);
}

bool preset_prims(int index, std::vector<PrimDesc> &out) {
    out.clear();
    index = std::max(0, std::min(index, kPresetCount - 1));
    auto add = [&](int type, double x, double y, double z, const float *rot, double p0, double p1, double p2) {
        PrimDesc d;
        d.type = type;
        make_transform(x, y, z, rot, d.m);
        d.params[0] = p0;
        d.params[1] = p1;
        d.params[2] = p2;
        out.push_back(d);
    };
    if (index <= 4) {
        std::vector<float> c;
        std::vector<double> r;
        preset_spheres(index, c, r);
        for (size_t i = 0; i < r.size(); ++i) add(0, c[3 * i], c[3 * i + 1], c[3 * i + 2], nullptr, r[i], 0, 0);
        return true;
    }
    switch (index) {
        case 5: {  // "Torus": createTorus(0,0,0, 1.3, vec3.fromValues(-Math.PI/2, 0, 0)), minor = radius / 4
            const float rot[3] = {to_f32(-3.141592653589793 / 2), 0.0f, 0.0f};
            add(2, 0, 0, 0, rot, 1.3, 1.3 / 4, 0);
            return true;
        }
        case 7:  // "Cube"
            add(1, 0, 0, 0, nullptr, 1, 1, 1);
            return true;
        case 8:  // "Sphere and Cube"
            add(0, -0.7, 0, 0, nullptr, 0.5, 0, 0);
            add(1, 1, 0, 0, nullptr, 0.5, 0.5, 0.5);
            return true;
        case 9:  // "Pyramid of Boxes"
            add(1, 0, 0.5, 0, nullptr, 0.9, 0.25, 0.9);
            add(1, 0, 0, 0, nullptr, 0.6, 0.25, 0.6);
            add(1, 0, -0.5, 0, nullptr, 0.3, 0.25, 0.3);
            return true;
        default:
            return false;  // Round / SmoothUnion / Twist / Repetition / Mandelbulb ...: SURVEY 8(f) N4
    }
}

bool build_scene_general(HostScene &s, const PrimDesc *prims, int n, int accel, std::string &err) {
    if (n < 0 || (n > 0 && !prims)) {
        err = "bad primitive list";
        return false;
    }
    s = HostScene();
    s.accel = (accel == 1 || accel == 2) ? accel : 0;
    s.general = true;
    s.prims.resize(n);
    s.prim_lo.resize(3 * size_t(n));
    s.prim_hi.resize(3 * size_t(n));
    s.world_pos.resize(3 * size_t(n));
    for (int i = 0; i < n; ++i) {
        const PrimDesc &d = prims[i];
        if (d.type < 0 || d.type > 2) {
            err = "unknown primitive type";
            return false;
        }
        for (int k = 0; k < 16; ++k)
            if (!std::isfinite(d.m[k])) {
                err = "non-finite transform";
                return false;
            }
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(d.params[k])) {
                err = "non-finite primitive parameter";
                return false;
            }
        RmPrim &q = s.prims[i];
        std::memcpy(q.m, d.m, sizeof q.m);
        q.type = d.type;
        if (q.m[3] == 0.0f && q.m[7] == 0.0f && q.m[11] == 0.0f && q.m[15] == 1.0f) {
            q.type |= 0x100;  // w == 1: no divide
            if (q.m[0] == 1.0f && q.m[5] == 1.0f && q.m[10] == 1.0f && q.m[1] == 0.0f && q.m[2] == 0.0f && q.m[4] == 0.0f &&
                q.m[6] == 0.0f && q.m[8] == 0.0f && q.m[9] == 0.0f)
                q.type |= 0x200;  // pure translation
        }
        q.half[0] = q.half[1] = q.half[2] = 0.0f;
        q.a = q.b = 0.0;
        double local_radius;
        if (d.type == 1) {  // box.ts:8-11,32-34
            for (int k = 0; k < 3; ++k) q.half[k] = to_f32(d.params[k]);
            local_radius = vec3_length_host(q.half[0], q.half[1], q.half[2]);
        } else if (d.type == 2) {  // torus.ts:27-29
            q.a = d.params[0];
            q.b = d.params[1];
            local_radius = q.a + q.b;
        } else {
            q.a = d.params[0];
            local_radius = q.a;
        }
        // Primitive.getWorldPosition (primitive.ts:20-30) and BoundingBox.fromPrimitive
        // (boundingBox.ts:133-154): both invert the world->local matrix
        Mat4 l2w = Mat4::identity();
        const bool ok = invert4(q.m, l2w.m);
        const float *m = ok ? l2w.m : q.m;
        const double scale = js_max2(js_max2(js_hypot3(m[0], m[1], m[2]), js_hypot3(m[4], m[5], m[6])),
                                     js_hypot3(m[8], m[9], m[10]));
        const double pad = local_radius * scale * 1.5;
        for (int k = 0; k < 3; ++k) {
            s.world_pos[3 * i + k] = l2w.m[12 + k];
            s.prim_lo[3 * i + k] = to_f32(double(l2w.m[12 + k]) - pad);
            s.prim_hi[3 * i + k] = to_f32(double(l2w.m[12 + k]) + pad);
        }
        // Bounding sphere for the all-primitive fallback's candidate filter (rm_kernels.hip, all_prims_distance): for a RIGID
        // world->local matrix (orthonormal 3x3, bottom row 0 0 0 1) Primitive.sdf is a true distance in world space and the
        // primitive lies inside the ball (centre c = the point mapped to the local origin, radius R = |halfSize| for a box,
        // major + minor for a torus, |r| for a sphere), hence  |p - c| - R <= sdf(p) <= |p - c| + R  for every p.  Anything
        // else (scale, shear, projective rows) switches the filter off for the whole scene.
        {
            bool rigid = ok && (q.type & 0x100);
            const float *w = q.m;
            for (int a = 0; rigid && a < 3; ++a)
                for (int b = a; b < 3; ++b) {
                    const double dot = double(w[4 * a]) * w[4 * b] + double(w[4 * a + 1]) * w[4 * b + 1] + double(w[4 * a + 2]) * w[4 * b + 2];
                    if (std::fabs(dot - (a == b ? 1.0 : 0.0)) > 1e-6) rigid = false;  // 1e-6 x |p - c| stays inside the filter's 4e-6 margin
                }
            if (!rigid) s.prim_filter_ok = false;
            const double cabs = std::fabs(double(l2w.m[12])) + std::fabs(double(l2w.m[13])) + std::fabs(double(l2w.m[14]));
            RmSphere bs;
            bs.cx = l2w.m[12];
            bs.cy = l2w.m[13];
            bs.cz = l2w.m[14];
            // slack: the inverse's rounding (1e-6 relative on c), the 1e-6 orthonormality tolerance over the radius
            // (the torus: |p| - |a| - |b| <= sdf <= |p| + |a| + |b| for radii of either sign -- triangle inequality in the (r, y)
            // plane; a + b, the reference's bounding-box radius kept in local_radius for prim_lo / prim_hi, is not a bound when a
            // radius is negative: ADVICE r2)
            const double bound_radius = (q.type & 0xFF) == 2 ? std::fabs(q.a) + std::fabs(q.b) : std::fabs(local_radius);
            bs.rf = std::nextafter(static_cast<float>(bound_radius * (1.0 + 3e-5) + 1e-5 * (1.0 + cabs)), INFINITY);
            s.spheres.push_back(bs);
        }
    }
    if (n == 0) s.prim_filter_ok = false;
    return build_accel(s, n, err);
}

// ---- expression forests ---------------------------------------------------------

void scale_transform(float m[16], double x, double y, double z) {
    for (int i = 0; i < 4; ++i) {
        m[i] = to_f32(double(m[i]) * x);
        m[4 + i] = to_f32(double(m[4 + i]) * y);
        m[8 + i] = to_f32(double(m[8 + i]) * z);
    }
}

namespace {

struct Forest {
    const NodeDesc *nodes;
    int n;
    std::vector<Mat4> T;  // Primitive.transform per node

    // Primitive.transform: wrappers pass their operand's to super(), unions pass mat4.create()
    void derive_transforms() {
        T.resize(n);
        for (int i = 0; i < n; ++i) {
            const NodeDesc &d = nodes[i];
            if (d.type < 10) std::memcpy(T[i].m, d.m, sizeof T[i].m);
            else if (d.type == 11 || d.type == 12) T[i] = Mat4::identity();
            else T[i] = T[d.a];
        }
    }
    // getWorldPosition: primitive.ts:20-30 and the overrides (smoothUnion.ts:52-60 midpoint, others: operand a)
    void world_pos(int i, float out[3]) const {
        const NodeDesc &d = nodes[i];
        if (d.type == 11) {
            float p1[3], p2[3];
            world_pos(d.a, p1);
            world_pos(d.b, p2);
            for (int k = 0; k < 3; ++k) out[k] = to_f32((double(p1[k]) + double(p2[k])) / 2);
            return;
        }
        if (d.type >= 10) return world_pos(d.a, out);
        Mat4 l2w = Mat4::identity();
        invert4(T[i].m, l2w.m);
        for (int k = 0; k < 3; ++k) out[k] = l2w.m[12 + k];
    }
    // getLocalBoundingRadius of every class
    double local_radius(int i) const {
        const NodeDesc &d = nodes[i];
        switch (d.type) {
            case 0: return d.params[0];
            case 1: return vec3_length_host(to_f32(d.params[0]), to_f32(d.params[1]), to_f32(d.params[2]));
            case 2: return d.params[0] + d.params[1];
            case 3: return 2.5;                                   // mandelbulb.ts:80-83
            case 10: return local_radius(d.a) + d.params[0];      // round.ts:27-30
            case 13: return local_radius(d.a);                    // twist.ts:38-41
            case 14: return std::numeric_limits<double>::infinity();  // repetition.ts:31-34
            case 15: return local_radius(d.a) + d.params[3];      // animatedTranslate.ts:51-54
            case 12: return local_radius(d.a);                    // smoothSubstraction.ts:36-39
            default: {                                            // smoothUnion.ts:37-49
                float p1[3], p2[3];
                world_pos(d.a, p1);
                world_pos(d.b, p2);
                const double dist = vec3_length_host(double(p2[0]) - double(p1[0]), double(p2[1]) - double(p1[1]),
                                                     double(p2[2]) - double(p1[2]));
                return js_max2(local_radius(d.a), local_radius(d.b)) + dist * 0.5;
            }
        }
    }
    // spheres, boxes and tori under Round / SmoothUnion / SmoothSubtraction only
    bool plain_subtree(int i) const {
        const NodeDesc &d = nodes[i];
        if (d.type < 10) return d.type == 0 || d.type == 1 || d.type == 2;
        if (d.type == 10) return plain_subtree(d.a);
        if (d.type == 11 || d.type == 12) return plain_subtree(d.a) && plain_subtree(d.b);
        return false;
    }
    // post-order instruction stream of the subtree rooted at i, reading its point from slot `slot`
    mutable int max_slot = 0, max_vals = 0;
    bool compile(int i, int slot, int &depth_vals, std::vector<RmInstr> &out, std::string &err, std::vector<ProgTreeNode> &tree, int &tree_node) const {
        max_slot = std::max(max_slot, slot);
        if (slot + 1 >= RM_PROG_MAX_SLOTS) {
            err = "expression tree deeper than RM_PROG_MAX_SLOTS";
            return false;
        }
        const NodeDesc &d = nodes[i];
        RmInstr ins;
        std::memset(&ins, 0, sizeof ins);
        std::memcpy(ins.T, T[i].m, sizeof ins.T);
        Mat4 inv = Mat4::identity();
        invert4(T[i].m, inv.m);  // stays identity when singular, like mat4.invert's untouched `out`
        std::memcpy(ins.Tinv, inv.m, sizeof ins.Tinv);
        auto affine = [](const float *m) { return m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f; };
        auto shift = [&](const float *m) {  // pure translation: identity upper block on top of the affine bottom row
            return affine(m) && m[0] == 1.0f && m[5] == 1.0f && m[10] == 1.0f && m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f &&
                   m[6] == 0.0f && m[8] == 0.0f && m[9] == 0.0f;
        };
        ins.flags = (affine(ins.T) ? 1 : 0) | (affine(ins.Tinv) ? 2 : 0) | (shift(ins.T) ? 4 : 0) | (shift(ins.Tinv) ? 8 : 0);
        for (int k = 0; k < 6; ++k) ins.p[k] = d.params[k];
        if (d.type == 1 || d.type == 14 || d.type == 15)  // vec3 members are Float32Arrays
            for (int k = 0; k < 3; ++k) ins.p[k] = to_f32(d.params[k]);
        ins.op = d.type;
        ins.src = slot;
        ins.dst = slot + 1;
        tree_node = static_cast<int>(tree.size());
        tree.push_back(ProgTreeNode());
        const int me = tree_node;
        if (d.type < 10) {
            tree[me].main = static_cast<int>(out.size());
            out.push_back(ins);
            max_vals = std::max(max_vals, depth_vals + 1);
            if (++depth_vals > RM_PROG_MAX_VALS) {
                err = "expression needs more than RM_PROG_MAX_VALS pending values";
                return false;
            }
            return true;
        }
        // A Round / SmoothUnion / SmoothSubtraction node whose transform is the identity hands its point on unchanged
        // (transformMat4 by the identity and back: x*1 + 0 + 0 + 0, exact but for the sign of a zero coordinate): its PRE half
        // is dropped and the operands read the node's own slot.  Only above spheres, boxes and tori, whose distances do not
        // depend on the sign of a zero (|x|, x*x, hypot); a Twist or Mandelbulb below keeps the reference's every step.
        auto identity = [](const float *m) {
            for (int k = 0; k < 16; ++k)
                if (m[k] != ((k % 5 == 0) ? 1.0f : 0.0f)) return false;
            return true;
        };
        const bool pass_through = d.type >= 10 && d.type <= 12 && identity(ins.T) && identity(ins.Tinv) && plain_subtree(i);
        const int child_slot = pass_through ? slot : slot + 1;
        if (!pass_through) {
            tree[me].pre = static_cast<int>(out.size());
            out.push_back(ins);  // PRE
        }
        int child = -1;
        if (!compile(d.a, child_slot, depth_vals, out, err, tree, child)) return false;
        tree[me].a = child;
        const bool binary = d.type == 11 || d.type == 12;
        if (binary && !compile(d.b, child_slot, depth_vals, out, err, tree, child)) return false;
        if (binary) tree[me].b = child;
        if (binary || d.type == 10) {
            ins.op = d.type + 10;  // POST
            tree[me].main = static_cast<int>(out.size());
            out.push_back(ins);
            if (binary) --depth_vals;
        }
        return true;
    }
};

}  // namespace

bool preset_nodes(int index, std::vector<NodeDesc> &nodes, std::vector<int> &roots) {
    nodes.clear();
    roots.clear();
    index = std::max(0, std::min(index, kPresetCount - 1));
    const double PI = 3.141592653589793;
    auto leaf = [&](int type, double x, double y, double z, const float *rot, double p0, double p1, double p2) {
        NodeDesc d;
        d.type = type;
        make_transform(x, y, z, rot, d.m);
        for (double &v : d.params) v = 0;
        d.params[0] = p0;
        d.params[1] = p1;
        d.params[2] = p2;
        nodes.push_back(d);
        return static_cast<int>(nodes.size()) - 1;
    };
    auto sphere = [&](double x, double y, double z, double r) { return leaf(0, x, y, z, nullptr, r, 0, 0); };
    auto box = [&](double x, double y, double z, double hx, double hy, double hz, const float *rot = nullptr) {
        return leaf(1, x, y, z, rot, hx, hy, hz);
    };
    auto torus = [&](double x, double y, double z, double radius, const float *rot) {
        return leaf(2, x, y, z, rot, radius, radius / 4, 0);  // sceneManager.ts:47-49
    };
    auto op = [&](int type, int a, int b, double p0) {
        NodeDesc d;
        d.type = type;
        d.a = a;
        d.b = b;
        std::memset(d.m, 0, sizeof d.m);
        for (double &v : d.params) v = 0;
        d.params[0] = p0;
        nodes.push_back(d);
        return static_cast<int>(nodes.size()) - 1;
    };
    switch (index) {
        case 6:  // "Rounded Box"
            roots.push_back(op(10, box(0, 0, 0, 0.4, 0.4, 0.4), -1, 0.3));
            return true;
        case 10:  // "Smooth Union"
            roots.push_back(op(11, sphere(0, 0, 0, 0.5), box(0, 0.5, 0, 1, 0.2, 1), 0.2));
            return true;
        case 11: {  // "Smooth Subtraction"
            const float rot[3] = {0.0f, to_f32(PI / 4), 0.0f};
            const int a = op(10, box(0, 0, 0, 1, 1, 1, rot), -1, 0.1);
            roots.push_back(op(12, a, sphere(0, 0, 0, 0.9), 0.2));
            return true;
        }
        case 12: {  // "Smooth Union [A]": AnimatedTranslate(sphere, (1,0,0), 3.0, 0.005)
            const int a = op(15, sphere(0, 0, 0, 1), -1, 0);
            const float dir[3] = {1.0f, 0.0f, 0.0f};
            float nd[3];
            double len = double(dir[0]) * dir[0] + double(dir[1]) * dir[1] + double(dir[2]) * dir[2];
            if (len > 0) len = 1 / std::sqrt(len);  // vec3.normalize (animatedTranslate.ts:22-23)
            for (int k = 0; k < 3; ++k) nd[k] = to_f32(double(dir[k]) * len);
            nodes[a].params[0] = nd[0];
            nodes[a].params[1] = nd[1];
            nodes[a].params[2] = nd[2];
            nodes[a].params[3] = 3.0;
            nodes[a].params[4] = 0.005;
            roots.push_back(op(11, a, sphere(0, 0, 0, 1), 0.2));
            return true;
        }
        case 13: {  // "Mandelbulb [A]": createMandelbulb(0,0,0, 8, 80, true, -0.0001)
            const int m = leaf(3, 0, 0, 0, nullptr, 8, 80, 1);
            nodes[m].params[3] = -0.0001;
            scale_transform(nodes[m].m, 0.5, 0.5, 0.5);
            roots.push_back(m);
            return true;
        }
        case 14: {  // "Twisted Torus"
            const float rot[3] = {to_f32(-PI / 2), 0.0f, 0.0f};
            roots.push_back(op(13, torus(0, 0, 0, 1.3, rot), -1, 3));
            return true;
        }
        case 15: {  // "Infinite Spheres"
            const int r = op(14, sphere(0, 0, 0, 0.3), -1, 1.5);
            nodes[r].params[1] = 1.5;
            nodes[r].params[2] = 1.5;
            roots.push_back(r);
            return true;
        }
        case 16:  // "Screw"
            roots.push_back(op(10, op(13, box(0, 0, 0, 0.4, 1.5, 0.4), -1, 4.0), -1, 0.1));
            return true;
        case 17: {  // "Chicken"
            static const double b[10][6] = {
                {0, 0, 0, 0.6, 0.6, 0.8},      {0, -0.2, 0, 0.8, 0.4, 0.6},   {0, -0.8, 0.8, 0.4, 0.6, 0.3},
                {0, -0.8, 1.2, 0.4, 0.2, 0.2}, {0, -0.4, 1.0, 0.2, 0.2, 0.2}, {0.3, 1, 0, 0.1, 0.6, 0.01},
                {-0.3, 1, 0, 0.1, 0.6, 0.01},  {0, 1.6, 0.2, 0.6, 0.01, 0.2}, {0.3, 1.6, 0.5, 0.1, 0.01, 0.1},
                {-0.3, 1.6, 0.5, 0.1, 0.01, 0.1}};
            int acc = box(b[0][0], b[0][1], b[0][2], b[0][3], b[0][4], b[0][5]);
            for (int i = 1; i < 10; ++i) acc = op(11, acc, box(b[i][0], b[i][1], b[i][2], b[i][3], b[i][4], b[i][5]), 0.0001);
            roots.push_back(acc);
            return true;
        }
        case 18: {  // "67"
            const float r5[3] = {0.0f, 0.0f, to_f32(PI / 5)}, r7[3] = {0.0f, 0.0f, to_f32(PI / 7)};
            const float r2[3] = {0.0f, 0.0f, to_f32(PI / 2)}, rt[3] = {to_f32(-PI / 2), 0.0f, 0.0f};
            const int six_a = op(10, box(-1.25, -0.8, 0, 0.05, 0.7, 0.05, r5), -1, 0.20);
            const int six_b = op(10, torus(-1.25, 0.5, 0, 0.8, rt), -1, 0.05);
            roots.push_back(op(11, six_a, six_b, 0.0001));
            const int sev_a = op(10, box(1.35, 0, 0, 0.05, 1.5, 0.05, r7), -1, 0.20);
            const int sev_b = op(10, box(1.25, -1.4, 0, 0.05, 0.8, 0.05, r2), -1, 0.20);
            roots.push_back(op(11, sev_a, sev_b, 0.0001));
            return true;
        }
        default:
            return false;
    }
}

bool build_scene_nodes(HostScene &s, const NodeDesc *nodes, int n_nodes, const int *roots, int n_roots, int accel,
                       std::string &err) {
    if (n_nodes < 0 || n_roots < 0 || (n_nodes > 0 && !nodes) || (n_roots > 0 && !roots)) {
        err = "bad node list";
        return false;
    }
    for (int i = 0; i < n_nodes; ++i) {
        const NodeDesc &d = nodes[i];
        const bool leaf = d.type >= 0 && d.type <= 3, unary = d.type == 10 || (d.type >= 13 && d.type <= 15);
        const bool binary = d.type == 11 || d.type == 12;
        if (!leaf && !unary && !binary) {
            err = "unknown node type";
            return false;
        }
        if (!leaf && (d.a < 0 || d.a >= i || (binary && (d.b < 0 || d.b >= i)))) {
            err = "operand index must name an earlier node";
            return false;
        }
        if (leaf)
            for (int k = 0; k < 16; ++k)
                if (!std::isfinite(d.m[k])) {
                    err = "non-finite transform";
                    return false;
                }
        for (int k = 0; k < 6; ++k)
            if (!std::isfinite(d.params[k])) {
                err = "non-finite node parameter";
                return false;
            }
    }
    for (int r = 0; r < n_roots; ++r)
        if (roots[r] < 0 || roots[r] >= n_nodes) {
            err = "root index out of range";
            return false;
        }
    s = HostScene();
    s.accel = (accel == 1 || accel == 2) ? accel : 0;
    s.general = true;  // not the compact sphere path
    s.program = true;
    Forest f{nodes, n_nodes, {}};
    f.derive_transforms();
    s.prim_lo.resize(3 * size_t(n_roots));
    s.prim_hi.resize(3 * size_t(n_roots));
    s.world_pos.resize(3 * size_t(n_roots));
    for (int r = 0; r < n_roots; ++r) {
        const int root = roots[r];
        const int first = static_cast<int>(s.prog.size());
        int vals = 0;
        int tree_root = -1;
        if (!f.compile(root, 0, vals, s.prog, err, s.prog_tree, tree_root)) return false;
        s.prog_roots.push_back(tree_root);
        s.obj_ranges.push_back(first);
        s.obj_ranges.push_back(static_cast<int>(s.prog.size()) - first);
        // BoundingBox.fromPrimitive (boundingBox.ts:133-154) with the overridden getters
        float wp[3];
        f.world_pos(root, wp);
        const double local_radius = f.local_radius(root);
        Mat4 l2w = Mat4::identity();
        const bool ok = invert4(f.T[root].m, l2w.m);
        const float *m = ok ? l2w.m : f.T[root].m;
        const double scale = js_max2(js_max2(js_hypot3(m[0], m[1], m[2]), js_hypot3(m[4], m[5], m[6])),
                                     js_hypot3(m[8], m[9], m[10]));
        const double pad = local_radius * scale * 1.5;
        for (int k = 0; k < 3; ++k) {
            s.world_pos[3 * r + k] = wp[k];
            s.prim_lo[3 * r + k] = to_f32(double(wp[k]) - pad);
            s.prim_hi[3 * r + k] = to_f32(double(wp[k]) + pad);
        }
    }
    s.prog_slots = f.max_slot + 1;
    s.prog_vals = std::max(1, f.max_vals);
    if ((size_t(s.prog_slots) * 12 + size_t(s.prog_vals) * 8) * 256 > 64 * 1024) {
        err = "expression forest exceeds the interpreter's LDS budget (RM_PROG_MAX_SLOTS / RM_PROG_MAX_VALS)";
        return false;
    }
    return build_accel(s, n_roots, err);
}

bool leaf_objects(const HostScene &s, std::vector<RmInstr> &prog, std::vector<int32_t> &obj_ranges, std::vector<ProgTreeNode> &tree,
                  std::vector<int32_t> &roots) {
    prog.clear();
    obj_ranges.clear();
    tree.clear();
    roots.clear();
    if (s.program) return false;
    const size_t n = s.general ? s.prims.size() : s.spheres.size();
    if (n == 0 || (!s.general && s.radii.size() != n)) return false;
    for (size_t i = 0; i < n; ++i) {
        RmInstr ins;
        std::memset(&ins, 0, sizeof ins);
        if (s.general) {
            const RmPrim &q = s.prims[i];
            std::memcpy(ins.T, q.m, sizeof ins.T);
            ins.op = q.type & 0xFF;
            if (ins.op == 1) for (int k = 0; k < 3; ++k) ins.p[k] = q.half[k];  // Float32Array members (box.ts:8-11)
            else ins.p[0] = q.a, ins.p[1] = q.b;
        } else {
            make_transform(s.spheres[i].cx, s.spheres[i].cy, s.spheres[i].cz, nullptr, ins.T);  // sceneManager.ts:21-37 without a rotation
            ins.op = 0;
            ins.p[0] = s.radii[i];
        }
        if (ins.op < 0 || ins.op > 2) return false;
        Mat4 inv = Mat4::identity();
        invert4(ins.T, inv.m);
        std::memcpy(ins.Tinv, inv.m, sizeof ins.Tinv);
        for (int k = 0; k < 16; ++k)
            if (!std::isfinite(ins.T[k]) || !std::isfinite(ins.Tinv[k])) return false;
        for (int k = 0; k < 6; ++k)
            if (!std::isfinite(ins.p[k])) return false;
        auto affine = [](const float *m) { return m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f; };
        auto shift = [&](const float *m) {
            return affine(m) && m[0] == 1.0f && m[5] == 1.0f && m[10] == 1.0f && m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f && m[6] == 0.0f &&
                   m[8] == 0.0f && m[9] == 0.0f;
        };
        ins.flags = (affine(ins.T) ? 1 : 0) | (affine(ins.Tinv) ? 2 : 0) | (shift(ins.T) ? 4 : 0) | (shift(ins.Tinv) ? 8 : 0);
        ins.src = 0;
        ins.dst = 1;
        ProgTreeNode t;
        t.main = static_cast<int>(prog.size());
        obj_ranges.push_back(static_cast<int32_t>(prog.size()));
        obj_ranges.push_back(1);
        roots.push_back(static_cast<int32_t>(tree.size()));
        tree.push_back(t);
        prog.push_back(ins);
    }
    return true;
}

void camera_from_angles(double pitch, double yaw, float rot9[9], float origin3[3]) {
    const double half_pi = 3.141592653589793 / 2;
    const double p = js_min2(js_max2(pitch, -half_pi), half_pi);  // camera.ts:59
    const Mat4 orbit = rotate_x(rotate_y(Mat4::identity(), yaw), p);  // camera.ts:83-84
    const float back[3] = {0.0f, 0.0f, 3.0f};                       // camera.ts:13,86
    const Mat4 cam = translate(orbit, back);
    const int src[9] = {0, 1, 2, 4, 5, 6, 8, 9, 10};  // mat3.fromMat4
    for (int i = 0; i < 9; ++i) rot9[i] = cam.m[src[i]];
    origin3[0] = cam.m[12];
    origin3[1] = cam.m[13];
    origin3[2] = cam.m[14];
}

void phong_light_dir(float out[3]) {
    const float v[3] = {1.0f, -1.0f, 1.5f};
    double len = double(v[0]) * v[0] + double(v[1]) * v[1] + double(v[2]) * v[2];
    if (len > 0) len = 1 / std::sqrt(len);
    for (int k = 0; k < 3; ++k) out[k] = to_f32(double(v[k]) * len);
}

}  // namespace rmh
