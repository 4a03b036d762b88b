// vet_kernels.hpp — map of the gfx950 (MI355X, CDNA4) device code of the viewport -> tile -> entropy path.
//
// Kernels (wave = 64 lanes everywhere), by header and by the translation unit that launches them:
//   vet_plan.hip        vet_plan_kernels.hpp   k_grid_dirs      axis tables -> rounded + normalised direction per pixel (py,px)
//                                              k_unit_dirs      the same for an explicit direction table
//                                              k_nearest_lut    direction -> nearest lattice tile (np.argmin over arccos(dot):
//                                                               FP64 arg-max of the normalised dot, first minimum of the
//                                                               distance values), one LUT per lattice
//                                              k_angular_distances   vector_angle_distance for m vectors x n tile centres
//                       vet_weight_table.hpp   k_row_stats      exact rows -> error bounds of the integer formulations
//                                              k_wtab           direction -> ELL row of (tile, FoV weight), ocml acos / pow
//                                              k_fuse_shifts, k_dirrec   fused-row shifts, per-direction records
//                                              k_wexact         direction -> ELL row of (tile, exact FP64 weight) of lattice 0:
//                                                               the rows the weights pass gathers
//                       vet_geometry.hpp       k_fb_boundaries  tile boundary edges of a Fibonacci tiling
//   vet_spatial.hip     vet_spatial_lut.hpp    k_spatial_lut    table formulation: per frame, samples -> direction ids ->
//                                                               gather of the users' rows into 64-bit integer (or FP64) LDS
//                                                               histograms (all lattices in one launch) -> Shannon entropy
//                       vet_spatial_sweep.hpp  k_spatial_w      sweep formulations (few samples per plan): lane = tile, FP64
//                                                               cone test per (user, tile), full-wave weight evaluation
//                       vet_weights_pass.hpp   k_weights_gather the tile_weights output (values at the reference's precision):
//                                                               per frame, the users' exact weight rows summed in column
//                                                               order (off the hot path; k_spatial_w<PRECISE> in weights-only
//                                                               mode where the exact rows do not fit the device)
//                       vet_spatial_u.hpp      k_spatial_u_lds  nearest-tile (unweighted) and naive lat/lon-grid mode:
//                                                               persistent stream with the nearest LUT in LDS (HBM-bound)
//                                              k_spatial_u      generic fallback (LUT gathered from global memory)
//   vet_transition.hip  vet_transition.hpp     k_transition_run per frame pair: (prior tile, current tile) pairs -> bucket
//                                                               statistics in LDS -> transition entropy; persistent workgroups
//                                              k_transition_big more than 4096 users: the bucket hash in LDS, the row cut into
//                                                               ranges of source tiles whose buckets fit it
//                                              k_transition_any fallback for lattices of thousands of tiles (hash in global scratch)
//   (several units)     vet_finalize.hpp       k_log2_table, k_finalize*   log2(k) table; mean over a plan's lattices
// Shared, kernel-free headers: vet_layout.hpp (table / histogram layout constants), vet_common.hpp (wave helpers, the
// sample -> direction-id quantiser), vet_weights.hpp (FoV weight, weighted frame entropy), vet_host.hpp (host state).
//
// No MFMA: there is no dense contraction on this path.  Reference citations are relative to
// /root/reference/src/viewport_entropy_toolkit/.  This header is documentation; the units include what they launch.
#pragma once
