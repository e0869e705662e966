// rm_scene_host.h -- host-side scene construction for the MI355X render path: sphere
// presets, camera matrices, BVH / Octree build and flattening into the device layout of
// rm_types.h.  Runs once per scene change (the reference rebuilds per tile per frame,
// raymarchWorker.ts:37-38).  Pure C++, no HIP.
#pragma once
#include <string>
#include <vector>

#include "rm_types.h"

namespace rmh {

constexpr int kPresetCount = 19;  // sceneManager.ts:102-357

// one primitive as the ABI hands it over (include/rm_raymarch.h: rm_prim)
struct PrimDesc {
    int type = 0;        // 0 sphere, 1 box, 2 torus
    float m[16];         // world -> local, gl-matrix layout
    double params[3];    // sphere: r; box: halfSize; torus: major, minor
};

// one node of an SDF expression forest as the ABI hands it over (include/rm_raymarch.h: rm_node).
// Operands must precede the node that uses them (a, b < own index).
struct NodeDesc {
    int type = 0;        // 0 sphere, 1 box, 2 torus, 3 mandelbulb, 10 round, 11 smooth union, 12 smooth subtraction,
                         // 13 twist, 14 repetition, 15 animated translate
    int a = -1, b = -1;  // operand node indices
    float m[16];         // leaves: world -> local.  Operators derive theirs (wrappers: the operand's; unions: identity)
    double params[6];    // see RmInstr::p
};

// The expression tree behind an object's instruction stream (HostScene::prog), for the run-time specialiser (rm_rtc.h):
// pre / main index the instructions of the node (-1: none -- an operator whose identity transform lets its operands read its
// own point has no PRE half; Twist / Repetition / AnimatedTranslate have no POST half), a / b the operand nodes.
struct ProgTreeNode {
    int pre = -1, main = -1, a = -1, b = -1;
};

struct HostScene {
    int accel = 0;
    int preset = 0;
    bool general = false;         // RmPrim records instead of RmSphere
    bool program = false;         // expression programs (RmInstr) instead of either
    int prog_slots = 1, prog_vals = 1;  // what the deepest program needs (device LDS sizing)
    std::vector<RmInstr> prog;
    std::vector<int32_t> obj_ranges;  // (first, count) per scene object
    std::vector<ProgTreeNode> prog_tree;  // nodes of every object's tree (instruction indices into prog)
    std::vector<int32_t> prog_roots;      // root node per scene object
    std::vector<RmPrim> prims;
    std::vector<float> world_pos; // Primitive.getWorldPosition() per primitive (BVH sort key)
    bool leaf_order = false;      // BVH sphere scenes: spheres / radii are stored in leaf order, bvh_prims is 0..n-1
    std::vector<RmSphere> spheres;
    std::vector<double> radii;
    std::vector<float> prim_lo, prim_hi;  // padded AABBs, 3 floats per primitive
    std::vector<RmBvhNode> bvh;
    std::vector<int32_t> bvh_prims;
    int bvh_leaves = 0, bvh_depth = 0;
    // Uniform grid over the BVH root box that accelerates BVH.getPrimitivesAt (bvh.ts:95-121):
    // cell -> leaves whose box may contain a point of that cell (conservative), so the device
    // tests only those leaf boxes.  pq_cells[c] = (offset << 8) | count, count 255 = "walk the tree".
    int pq_dim[3] = {0, 0, 0};
    float pq_origin[3] = {0, 0, 0}, pq_inv[3] = {0, 0, 0};
    std::vector<uint32_t> pq_cells;
    std::vector<uint16_t> pq_list;
    // Same cells: spheres that can attain min_j Sphere.sdf for SOME point of the cell (the
    // all-primitive fallback of scene.ts:173 then only evaluates these).  count 255 = no list.
    int nn_dim[3] = {0, 0, 0};
    float nn_inv[3] = {0, 0, 0};
    std::vector<uint32_t> nn_cells;
    std::vector<uint16_t> nn_list;
    std::vector<RmOctNode> oct;
    std::vector<int32_t> oct_prims;
    std::vector<RmSphereRec> oct_recs;  // oct_prims expanded to sphere records (sphere scenes)
    // Octree.findNode as one lookup: the tree is always the +-10 cube split at binary32 midpoints down to
    // depth 6 (scene.ts:81-85, octree.ts:60), so every leaf is a box of whole 0.3125-cells of a 64^3 grid
    // and every cell boundary -10 + 0.3125 k is exact in binary32.  oct_lut[(z*64 + y)*64 + x] = leaf node.
    std::vector<int32_t> oct_lut;
    // Crowded octree leaves (> 8 spheres): a grid of RM_OCT_SUB^3 sub-cells, each listing the positions (within the leaf's
    // list) of the spheres that can attain the leaf's minimum for SOME point of the sub-cell -- the same bound as the
    // BVH's nearest-candidate grid, but taken over the leaf's own list, which is what Scene.getDistance evaluates.
    std::vector<uint32_t> oct_sub_hdr;
    std::vector<uint8_t> oct_sub_list;
    bool prim_filter_ok = true;  // general scenes: every primitive's transform is rigid, `spheres` holds their bounding spheres
    int oct_leaves = 0, oct_empty = 0, oct_max_leaf = 0;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
};

// V8 Math.hypot for three arguments (used by boundingBox.ts:46,142-144)
double js_hypot3(double x, double y, double z);
// vec3.length / vec3.distance of the scene builder (box.ts:33, smoothUnion.ts:45): Math.hypot, or sqrt(x*x+y*y+z*z)
// after set_length_mode(1) (thread-local; the API layer sets it from the ctx before every build)
void set_length_mode(int use_sqrt);
double vec3_length_host(double x, double y, double z);

// sceneManager.ts:102-170: sphere-only presets 0..4; false for the others
bool preset_spheres(int index, std::vector<float> &centers, std::vector<double> &radii);

// Builds spheres + acceleration structure.  Returns false and sets err on bad input.
bool build_scene(HostScene &s, const float *centers, const double *radii, int n, int accel,
                 std::string &err);

// SceneManager.getTransform (sceneManager.ts:21-37): world->local matrix of a primitive placed
// at (x, y, z); rot == nullptr is the branch without a rotation argument
void make_transform(double x, double y, double z, const float *rot, float out16[16]);

// presets 0-4 (spheres) and 5, 7, 8, 9 (torus / boxes) as primitive descriptions; false for
// presets that need SDF operators or the Mandelbulb
bool preset_prims(int index, std::vector<PrimDesc> &out);

// general scenes (any mix of spheres, boxes, tori, rotated or not)
bool build_scene_general(HostScene &s, const PrimDesc *prims, int n, int accel, std::string &err);

// presets 6 and 10-18 (SDF operators, Mandelbulb: sceneManager.ts:178-356) as expression forests
bool preset_nodes(int index, std::vector<NodeDesc> &nodes, std::vector<int> &roots);

// scenes whose objects are expression trees (any preset can be expressed this way)
bool build_scene_nodes(HostScene &s, const NodeDesc *nodes, int n_nodes, const int *roots, int n_roots, int accel,
                       std::string &err);

// A scene of plain primitives (spheres; spheres / boxes / tori with transforms) as one single-leaf expression object per
// primitive, in the scene's primitive order -- what the run-time specialiser (rm_rtc.h) needs to emit such a scene as code.
// False when the scene is an expression forest already, is empty, or holds a non-finite number.
bool leaf_objects(const HostScene &s, std::vector<RmInstr> &prog, std::vector<int32_t> &obj_ranges, std::vector<ProgTreeNode> &tree,
                  std::vector<int32_t> &roots);

// gl-matrix mat4.scale(m, m, [x, y, z]) in place (sceneManager.ts:63: the Mandelbulb's world->local is post-scaled)
void scale_transform(float m[16], double x, double y, double z);

// camera.ts:58-69,81-88 + raymarcher.ts:62-67
void camera_from_angles(double pitch, double yaw, float rot9[9], float origin3[3]);

// phongModel.ts:15-16
void phong_light_dir(float out[3]);

}  // namespace rmh
