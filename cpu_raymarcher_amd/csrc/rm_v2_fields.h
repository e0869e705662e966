// rm_v2_fields.h -- the launch parameters of the v2 wave loop that are CONFIGURATION: the frame size, the shading model, the
// option switches, the scene's counts and grids, the tile geometry and the LDS layout the launcher derived from them.  They
// change when the host changes the scene, the canvas or an option -- not from frame to frame.  The run-time specialiser
// (rm_rtc.h) compiles the wave loop with these as literals (rm_v2_fix, generated): measured on C3 at 4K, 1 186 -> 1 330
// frames/s with frames in flight and 1.12 -> 1.01 ms alone -- the wave loop re-reads its parameters in every section of
// every trip (cold_params), and a literal needs neither the scalar load nor the register.
//
// NOT in the lists (they stay kernel arguments): every pointer, the camera (rot, origin and their widened copies), time, the
// rows of the launch (y_start, y_end, local_rows, tiles_y, the stripe fields: a host that renders a frame as tiles, or a rank
// its stripes, must not compile per tile), and what only the launcher or the v1 kernels read.
#pragma once

#define RM_V2_FIXED_SCALARS(X)                                                                                                  \
    X(width) X(height) X(accel) X(shader) X(tile_w) X(nodes_in_lds) X(filter) X(variant) X(list_cap) X(coop) X(use_grid)         \
    X(refill_threshold) X(hw_xcd) X(item_px) X(rel_boxes) X(lpt_stride) X(prim_filter) X(n0_batch) X(use_nn) X(tile_w_log2)     \
    X(tile_h_log2) X(tiles_x) X(item_wide) X(item_w_log2) X(sub_dx) X(sub_dy) X(tiles_x_magic) X(leaf_order) X(algorithm)       \
    X(general) X(uniform_radius) X(multi_step)

// The scene's counts.  As literals they are loop bounds too: worth another 2 - 3 % on the 125-sphere grid, but on a nine-sphere
// scene the optimiser unrolls the node and leaf loops completely and the instantiation that held 80 VGPRs spills 53 -- such a
// kernel is refused (rm_rtc.cpp), and the configuration is compiled again without this list.
#define RM_V2_FIXED_COUNTS(X)                                                                                                    \
    X(n_prims) X(bvh_nodes) X(oct_nodes) X(bvh_prim_count) X(oct_prim_count) X(pq_cell_count) X(pq_list_count) X(bvh_leaf_count) \
    X(nn_cell_count) X(nn_list_count)

#define RM_V2_FIXED_ARRAYS(X) \
    X(pq_dim, 3) X(lds_off, 10) X(nn_dim, 3) X(pq_origin, 3) X(pq_inv, 3) X(pq_cell, 3) X(nn_inv, 3) X(light, 3) X(light_d, 3)
