// vet_kernels.hpp — gfx950 (MI355X, CDNA4) device code of the viewport -> tile -> entropy path.
//
// Kernels (wave = 64 lanes everywhere):
//   plan tables (once per plan)
//     k_grid_dirs      axis tables -> rounded + normalised direction per pixel (py,px)
//     k_nearest_lut    direction -> nearest lattice tile (FP64 arg-max of the normalised dot,
//                      lowest index on ties), one LUT per lattice
//     k_wtab           direction -> ELL row of (tile, FoV weight) pairs, exact ocml acos / pow
//     k_log2_table     log2(k), k <= 4096, for the integer-count entropies
//   spatial entropy, FoV-weighted
//     k_spatial_lut    table formulation: per frame, samples -> direction ids -> gather of the
//                      users' rows into 64-bit integer LDS histograms (all lattices in one launch)
//                      -> Shannon entropy; results are order independent, hence bit-reproducible
//     k_spatial_w      sweep formulation (few samples per plan): lane = tile, FP64 cone test per
//                      (user, tile), ballot-compacted full-wave weight evaluation
//   spatial entropy, nearest-tile (unweighted) and naive lat/lon-grid mode
//     k_spatial_u_lds  persistent stream with the nearest LUT in LDS (HBM-bound)
//     k_spatial_u      generic fallback (LUT gathered from global memory)
//   transition entropy
//     k_transition_run per frame pair: (prior tile, current tile) pairs -> bucket statistics in
//                      LDS (integer atomics + one small hash table) -> transition entropy; persistent
//                      workgroups over runs of rows, every frame quantised once
//     k_transition_any the same for any number of users (bucket hash in global scratch)
//   k_finalize         mean over the plan's lattices where they ran as separate launches
//
// No MFMA: there is no dense contraction on this path.  Reference citations are relative to
// /root/reference/src/viewport_entropy_toolkit/.
#pragma once
//
// The code lives in one header per group of kernels; this file is the umbrella vet_api.hip includes.
#pragma once
#include "vet_common.hpp"
#include "vet_plan_kernels.hpp"
#include "vet_weights.hpp"
#include "vet_spatial_sweep.hpp"
#include "vet_weight_table.hpp"
#include "vet_spatial_lut.hpp"
#include "vet_spatial_rows.hpp"
#include "vet_spatial_u.hpp"
#include "vet_transition.hpp"
#include "vet_geometry.hpp"
