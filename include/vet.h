/*
 * vet.h — C-ABI of the MI355X-native viewport -> Fibonacci-tile -> entropy engine.
 *
 * The reference (IamArmanNikkhah/viewport-entropy-toolkit) is pure Python and has
 * no FFI layer; its boundary for this path is the Python API.  This library sits
 * underneath a drop-in of that API and is bound with ctypes.
 * Each entry point names the reference code it replaces; paths are relative to
 * /root/reference/src/viewport_entropy_toolkit/.
 *
 * Conventions
 *   - plain C symbols, plain pointers and sizes; no C++ or torch types;
 *   - every function returns 0 on success or a negative VET_ERR_* code;
 *     vet_last_error() gives the thread-local message of the last failure;
 *   - the caller owns every buffer; the library keeps no caller pointer after a
 *     call returns (device inputs of an asynchronous call must stay alive until
 *     the stream has been synchronised);
 *   - sample arrays are FRAME-MAJOR: element (frame f, user u) is at [f*U + u],
 *     so one frame's users are contiguous (this is the dense form of the
 *     reference's ``vectors_df``: one row per frame, one column per user);
 *     an absent sample (reference: ``None`` cell) is NaN in mu or mv, or id -1;
 *   - "d_" parameters are device pointers, "h_" parameters are host pointers;
 *   - ``stream`` is a hipStream_t passed as void*.  NULL selects the CONTEXT'S OWN stream
 *     (hipStreamNonBlocking: not ordered against the null stream) — it is NOT the null stream.
 *     To run on the legacy default stream (what torch calls its default stream, handle 0) pass
 *     VET_STREAM_LEGACY; any other value is used as the hipStream_t it is.
 *     Device-pointer entry points only enqueue work; they do not synchronise.  A context — and
 *     every plan of it — is single-threaded and must not be in use on two streams at once: its
 *     scratch (the K > 1 workspace, the transition scratch, the resolve list of an FP table) is
 *     shared by all calls, so calls on different streams must be synchronised in between.
 *     (Batch descriptors are not part of that scratch: each batch call stages them in its own
 *     slot of an event-guarded ring, so back-to-back batch calls need no synchronisation.)  A plan's tables are complete (stream synchronised) when the
 *     call that built them returns; a result handle (vet_result) may outlive its context.
 *   - there is no CPU fallback: without a gfx950 device vet_create() fails.
 */
#ifndef VET_H_
#define VET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VET_VERSION 141 /* 0.1.4: tile_weights values at the reference's precision under every formulation
                           (+ vet_plan_set_raw_weights); batch descriptors in an event-guarded ring;
                           0.1.4.1: vet_device_pci_bus_id */
#define VET_STREAM_LEGACY ((void *)1) /* == hipStreamLegacy: the null stream with legacy ordering */
/* Policy 0 of vet_plan_set_table_policy: a weighted call gathers from the direction weight table iff it holds at least this
 * many samples per direction of the plan's direction table.  Measured (profiles/r06/first_call.txt, grid_sensitivity.txt):
 * building a direction's row costs what the sweep spends on ~20 samples; with the alias table built on the device the table's
 * FIRST call costs at most 1 ms more than the sweep's on 100x200 / 200x400 grids and 27 % more on 3840x1920, and every later
 * call of the plan is 3.7-10x cheaper than the sweep — at 2 samples per direction the table has paid for itself by the
 * plan's second (large grids) to tenth (config-2-sized videos) call.  Rounds 1-5 used 8: the alias table was a host hash map
 * then, which made a first call 1.5 s on a 3840x1920 grid. */
#define VET_TABLE_SAMPLES_PER_DIRECTION 2

enum {
    VET_OK = 0,
    VET_ERR_INVALID = -1,   /* bad argument (maps to ValueError / ValidationError) */
    VET_ERR_DEVICE = -2,    /* HIP failure / no device (maps to RuntimeError) */
    VET_ERR_RANGE = -3,     /* a 2dmu/2dmv value outside [0,1]  (reference: ValidationError,
                               utilities/data_utils.py:256-257) */
    VET_ERR_EMPTY = -4,     /* a frame without any (common) user (reference: ValidationError
                               entropy_utils.py:170 / ZeroDivisionError entropy_utils.py:299) */
    VET_ERR_UNSUPPORTED = -5
};

typedef struct vet_ctx vet_ctx;   /* one device + stream + scratch; one per thread */
typedef struct vet_plan vet_plan; /* device tables for one analyzer configuration */

/* ---- library / device ---------------------------------------------------- */
int vet_version(void);
const char *vet_last_error(void);
int vet_device_count(void);
int vet_create(int device_id, vet_ctx **out);
int vet_destroy(vet_ctx *ctx);
int vet_synchronize(vet_ctx *ctx);
/* The device a context computes on, as its PCI bus id ("0000:05:00.0"; buf of len >= 16): what a rank of a
 * multi-GPU job (one process per video, README.md:108-120) reports so that a job can show that its N ranks
 * sit on N distinct devices (bench.py's per_rank block; _dist refuses two ranks of an RCCL job on one device). */
int vet_device_pci_bus_id(vet_ctx *ctx, char *buf, int len);
/* Per-kernel timing with hipEvents on the launch stream (bench.py's roofline leg). */
int vet_profile_enable(vet_ctx *ctx, int on);
int vet_profile_reset(vet_ctx *ctx);
/* kernel ids: 0 k_grid_dirs, 1 k_nearest_lut, 2 k_spatial (any variant), 3 k_transition,
 *             4 k_finalize, 5 k_wtab (direction weight table build),
 *             6 k_weights (the weights pass: k_weights_gather over the plan's exact FP64 weight rows, or the precise sweep in
 *               weights-only mode where those do not fit: the d_weights output / fetched weight rows) */
int vet_profile_get(vet_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
const char *vet_kernel_name(int kernel_id);

/* ---- device memory helpers (for hosts without torch) --------------------- */
int vet_malloc(vet_ctx *ctx, size_t bytes, void **d_ptr);
int vet_free(vet_ctx *ctx, void *d_ptr);
int vet_memcpy_h2d(vet_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int vet_memcpy_d2h(vet_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* ---- plan: what SpatialEntropyAnalyzer.__init__ + AnalyzerConfig hold ----- */
typedef struct vet_plan_desc {
    /* Quantiser of process_viewport_data + format_trajectory_data + Vector.from_spherical
     * (utilities/data_utils.py:243-286, 390-397; data_types.py:204-216) as per-axis tables
     * built by the host with the reference's own float operations:
     *   lon axis, px = 0..W : cos(theta), sin(theta), theta = radians(lon(px))
     *   lat axis, py = 0..H : sin(phi),   cos(phi),   phi   = radians(90 - lat(py))
     * The device forms x = round6(sin(phi)*cos(theta)), y = round6(sin(phi)*sin(theta)),
     * z = round6(cos(phi)).  Leave all four NULL and set dir_table for an explicit table. */
    int video_width, video_height;
    const double *h_lon_cos, *h_lon_sin; /* [W+1] */
    const double *h_lat_sin, *h_lat_cos; /* [H+1] */
    /* Alternative to the axis tables: explicit direction table (rounded Vector xyz), used by
     * the operator-level compute_spatial_entropy / compute_transition_entropy shims
     * (utilities/entropy_utils.py:147-151, 213-219) where callers pass arbitrary Vectors. */
    const double *h_dir_table; /* [n_dirs*3] or NULL */
    int64_t n_dirs;
    /* Lattices: generate_fibonacci_lattice(tile_count) per AnalyzerConfig.tile_counts entry
     * (analyzers/spatial_entropy.py:63-66), as rounded Vector xyz. */
    int n_lattices;                 /* K >= 1 */
    const int *n_tiles;             /* [K]  n_k = 2*floor(tile_count/2)+1 */
    const double *const *h_tiles;   /* K pointers to [n_k*3] */
    const double *h_max_entropy;    /* [K]  -n*(1/n)*log2(1/n), entropy_utils.py:201-203 */
    /* EntropyConfig (utilities/entropy_utils.py:20-38) */
    double fov_angle;               /* degrees, (0,360] */
    double max_angular_distance;    /* np.radians(fov_angle/2), entropy_utils.py:124 */
    double power_factor;            /* > 0 */
    int use_weight_distribution;
    /* Optional "binned" lattices (NULL = none): h_bin_lut[k] != NULL replaces lattice k's
     * nearest-tile search by a caller-supplied direction -> bin table [n_dirs] with n_tiles[k]
     * bins (h_tiles[k] is then ignored).  This is the naive lat/lon tiling of
     * compute_naive_spatial_entropy (utilities/entropy_utils.py:383-452): every user adds 1 to
     * its bin; use_weight_distribution only selects the normaliser (log2 of h_max_entropy's n
     * always, vs. log2(users) when users <= n). */
    const uint16_t *const *h_bin_lut;
    /* [K] or NULL: the tile count the normaliser compares the user count with when it differs from
     * the number of histogram bins (naive tiling: (180/h)*(360/w) tiles, but lon = 180 / lat = 90
     * open one more bin column / row). */
    const int *n_norm_tiles;
} vet_plan_desc;

int vet_plan_create(vet_ctx *ctx, const vet_plan_desc *desc, vet_plan **out);
int vet_plan_destroy(vet_plan *plan);
int64_t vet_plan_n_dirs(const vet_plan *plan);
/* Weighted spatial mode (calculate_tile_weights, entropy_utils.py:108-144) has three formulations:
 *   0 table    direction weight table, built once per plan and gathered per distinct direction of a
 *              frame (memory bound); 32-bit block-floating-point weights, 64-bit integer histograms
 *   1 sweep    every sample sweeps every tile (FP64 VALU bound); 2^-52 fixed-point integer histograms
 *   2 precise  the sweep with the exact weights in FP64 histograms, in the reference's summation order
 *   3 ftable   the table with FP32 weights (2^-24 relative each, scaled by their row's exponent) in FP64
 *              histograms: |dH|/H <= 1.2e-7 for every frame whatever the weights' dynamic range
 * The integer formulations (0, 1) are order independent (bit-identical run to run, under any user
 * permutation, frame split or GPU count).  They are only used where a bound computed from the plan's
 * own rows proves their deviation from exact arithmetic <= 1e-7 relative for EVERY possible frame
 * (k_row_stats; the contract is 1e-6); plans outside that — FoV cones narrower than the lattice
 * spacing, large power factors — run `ftable` (calls large enough for a table) or `precise`; those two
 * sum in FP64 in a fixed order: `ftable` sorts every frame's list of distinct rows, gives every wave a histogram
 * of its own and adds those in wave order (bit-identical run to run, under any user permutation of frames of
 * <= 2048 users, frame split or GPU count — as measured on gfx950: where two rows of one wave instruction hit the
 * same tile the result also rests on the LDS servicing the lanes of a ds_add_f64 in lane order, which the hardware
 * does and the ISA does not promise; tests/test_hip_contract.py and tools/repeat_check.py are the guard);
 * `precise` sums the users in column order, as the reference does
 * (as the resolver of `ftable`: in ascending direction order).
 * The reference's NaN frames: every tile with distance < fov/2 is a key of the reference's per-frame dict,
 * also when ((max - d) / max) ** power_factor underflows to exactly 0.0 (entropy_utils.py:131-135), and a key
 * whose summed weight is 0.0 (or underflows against the frame total) makes the entropy NaN = 0 * log2 0
 * (:195-198).  Plans with such weights (below 2^-1048; power_factor >~ 80 on the default grids) never use an
 * integer formulation; `precise` keeps the exact key set, and `ftable` keeps every in-FoV tile without an FP32
 * value as a marker entry and hands the frames those markers decide to `precise` inside the same call.
 * Which formulation a call uses is a pure function of the plan and the call's shape, never of the
 * plan's history: policy 0 = table iff the call (or batch) holds >= VET_TABLE_SAMPLES_PER_DIRECTION samples per direction of the
 * plan's direction table, +1 = table whenever it is inside the contract and fits, -1 = never table.
 * vet_plan_table_stride: row length of lattice k's table (of the plan's fused table — one row per direction over
 * all lattices — where that is the one in use), 0 = not built (yet), -1 = does not fit.
 * vet_plan_last_formulation: formulation lattice k used in the plan's last weighted call (-1 none).
 * vet_plan_error_bounds: the proven relative entropy error bounds of lattice k for the table and the
 * sweep (at <= 1024 users) formulations; inf = a frame exists whose entropy no fixed point resolves. */
int vet_plan_set_table_policy(vet_plan *plan, int policy);
/* tile_weights VALUES (the d_weights / h_weights outputs and the weight rows of a vet_result; calculate_tile_weights and
 * the accumulation of compute_spatial_entropy, utilities/entropy_utils.py:131-136, 190-192).  Whatever formulation
 * produces the entropy, they are the reference's: exact FP64 weights (ocml acos / pow), summed over the users in column
 * order within each of NW contiguous shares of the users (NW = 4, 2 or 1 by lattice size and LDS), the shares added in order —
 * deterministic, within 1e-9 relative of the reference's single sequential sum, not bit-equal to it —, by a weights pass of its own: a gather of
 * exact FP64 weight rows built once per plan on the first request (the `precise` sweep in weights-only mode where those
 * rows do not fit the device).  Only calls that ask for the weights pay for it (config-3 shape: 2.8 ms for all 30 000
 * frames; a vet_result computes the rows of a fetched block, 256 frames 0.2 ms), and the entropy path is untouched.  on != 0 returns the formulation's own histogram instead (block-floating-point / FP32 / 2^-52 fixed-point
 * weight sums: at most n_users * 2^-33 of the row's scale off under the table): a diagnostic for the table layouts. */
int vet_plan_set_raw_weights(vet_plan *plan, int on);
int vet_plan_table_stride(const vet_plan *plan, int lattice);
/* rows of a weight table = distinct directions up to the lattice's mirror symmetry (0 before the first table exists):
 * a table takes (rows + 1) * stride * 6 bytes */
int64_t vet_plan_table_rows(const vet_plan *plan);
int vet_plan_last_formulation(const vet_plan *plan, int lattice);
int vet_plan_error_bounds(vet_plan *plan, int lattice, double *table_bound, double *sweep_bound);
/* Parity hooks: read back the device-built tables (synchronous). */
int vet_plan_read_dirs(vet_plan *plan, double *h_xyz /* [n_dirs*3] rounded Vector xyz */);
int vet_plan_read_nearest(vet_plan *plan, int lattice, int32_t *h_nearest /* [n_dirs] */);

/* ---- hot path: SpatialEntropyAnalyzer.compute_entropy ---------------------
 * (analyzers/spatial_entropy.py:107-164 -> entropy_utils.py:147-211, 108-144, 89-106, 41-87)
 *   d_entropy [T]      mean over the plan's lattices of the normalised spatial entropy
 *   d_assign  [T*U]    nearest tile of lattice 0 per sample, -1 absent      (nullable)
 *   d_weights [T*n_0]  per-frame tile weight sums of lattice 0              (nullable)
 *                      at the reference's precision under every formulation (vet_plan_set_raw_weights);
 *                      -0.0 = the tile is a key of the reference's dict with the value 0.0 (in some
 *                      user's FoV, weight underflowed), +0.0 = no key
 *   d_present [T]      users present per frame                              (nullable)
 *   d_status  [2]      {#samples outside [0,1], #frames without a user}; the call ADDS to
 *                      it, the caller zeroes it                             (nullable)   */
int vet_spatial_entropy(vet_plan *plan, const double *d_mu, const double *d_mv,
                        int n_users, int n_frames,
                        double *d_entropy, int32_t *d_assign, double *d_weights,
                        int32_t *d_present, int32_t *d_status, void *stream);
/* Same, samples given as direction ids into the plan's direction table (-1 absent). */
int vet_spatial_entropy_ids(vet_plan *plan, const int32_t *d_ids, int n_users, int n_frames,
                            double *d_entropy, int32_t *d_assign, double *d_weights,
                            int32_t *d_present, int32_t *d_status, void *stream);

/* ---- hot path: TransitionEntropyAnalyzer.compute_entropy ------------------
 * (analyzers/transition_entropy.py:107-175 -> entropy_utils.py:213-332)
 * Output row r compares frame r (prior) with frame r+1 (current), r = 0..T-2.
 *   d_entropy  [T-1]
 *   d_pairs    [(T-1)*U*2]  (prior tile, current tile) of lattice 0, -1 if not in both (nullable)
 *   d_srccount [(T-1)*n_0]  users per source tile of lattice 0                          (nullable)
 *   d_common   [T-1]        users present in both frames                                (nullable)
 *   d_status   [2]          {#samples outside [0,1], #rows without a common user}       (nullable) */
int vet_transition_entropy(vet_plan *plan, const double *d_mu, const double *d_mv,
                           int n_users, int n_frames,
                           double *d_entropy, int32_t *d_pairs, int32_t *d_srccount,
                           int32_t *d_common, int32_t *d_status, void *stream);
int vet_transition_entropy_ids(vet_plan *plan, const int32_t *d_ids, int n_users, int n_frames,
                               double *d_entropy, int32_t *d_pairs, int32_t *d_srccount,
                               int32_t *d_common, int32_t *d_status, void *stream);

/* ---- many videos, one launch ---------------------------------------------------
 * Short videos are launch-bound one call at a time (the reference's scale-out unit is "many short videos",
 * README.md:108-120); a batch shares one grid: the weighted table formulation (k_spatial_lut over the plan's fused
 * table), the nearest-tile / binned modes (k_spatial_u_lds, one launch per lattice) and transition mode
 * (k_transition_run, one launch per lattice, every video with its own workgroups).  Batches outside those kernels'
 * limits run video by video inside the call.  Lattice 0's tile weights / source counts are not produced by the
 * batched forms.  Asynchronous on ``stream`` like vet_spatial_entropy. */
typedef struct vet_video {
    const double *d_mu, *d_mv;   /* [n_frames * n_users], frame-major */
    int n_users, n_frames;
    double *d_entropy;           /* [n_frames] */
    int32_t *d_assign;           /* [n_frames * n_users] or NULL */
    int32_t *d_present;          /* [n_frames] or NULL */
} vet_video;
int vet_spatial_entropy_batch(vet_plan *plan, int n_videos, const vet_video *videos, int32_t *d_status,
                              void *stream);
/* Same with concatenated host buffers (video v starts where video v-1 ends); synchronous. */
int vet_spatial_entropy_batch_host(vet_plan *plan, int n_videos, const int *n_users, const int *n_frames,
                                   const double *h_mu, const double *h_mv, double *h_entropy,
                                   int32_t *h_assign, int32_t *h_present);
/* Transition mode (TransitionEntropyAnalyzer.compute_entropy per video, analyzers/transition_entropy.py:107-175):
 * d_entropy [n_frames-1], d_assign = the (prior, current) tile pairs [(n_frames-1) * n_users * 2] or NULL,
 * d_present = users present in both frames [n_frames-1] or NULL; every video needs at least two frames. */
int vet_transition_entropy_batch(vet_plan *plan, int n_videos, const vet_video *videos, int32_t *d_status,
                                 void *stream);
int vet_transition_entropy_batch_host(vet_plan *plan, int n_videos, const int *n_users, const int *n_frames,
                                      const double *h_mu, const double *h_mv, double *h_entropy,
                                      int32_t *h_pairs, int32_t *h_common);

/* ---- host-buffer convenience: H2D, run, D2H, synchronous ------------------
 * Return VET_ERR_RANGE / VET_ERR_EMPTY when the status words are non-zero (outputs are still
 * written).  h_mu/h_mv may be NULL when h_ids is given and vice versa. */
int vet_spatial_entropy_host(vet_plan *plan, const double *h_mu, const double *h_mv,
                             const int32_t *h_ids, int n_users, int n_frames,
                             double *h_entropy, int32_t *h_assign, double *h_weights,
                             int32_t *h_present);
int vet_transition_entropy_host(vet_plan *plan, const double *h_mu, const double *h_mv,
                                const int32_t *h_ids, int n_users, int n_frames,
                                double *h_entropy, int32_t *h_pairs, int32_t *h_srccount,
                                int32_t *h_common);

/* ---- host-buffer runs whose optional outputs stay on the device -----------------------------
 * SpatialEntropyAnalyzer.compute_entropy returns per frame a dict of tile weights and a dict of tile
 * assignments (analyzers/spatial_entropy.py:152-163) which most callers never read (the CSV holds time and
 * entropy only, :216-219).  These variants bring back the entropy series (and the per-frame user counts)
 * and keep assign [T][U] i32 + weights [T][n_0] f64 (transition: pairs [(T-1)][U][2] i32 + srccount
 * [(T-1)][n_0] i32) in device memory owned by a vet_result, from which rows are fetched on demand.
 * Weighted spatial results hold the samples' direction ids [T][U] i32 instead of the weights when those are not larger
 * (U * 4 <= n_0 * 8 bytes per frame; otherwise the weight rows are stored) and compute the weight rows of a fetched block
 * when it is fetched (vet_plan_set_raw_weights: the reference's values, off the hot path; same bits as the eager output).
 * A result handle is returned also with VET_ERR_RANGE / VET_ERR_EMPTY.  vet_result_fetch may be called from several
 * threads on one result (fetches of lazily computed weight rows take turns on the result's staging buffer). */
typedef struct vet_result vet_result;
int vet_spatial_entropy_host_resident(vet_plan *plan, const double *h_mu, const double *h_mv,
                                      const int32_t *h_ids, int n_users, int n_frames,
                                      double *h_entropy, int32_t *h_present, vet_result **out);
int vet_transition_entropy_host_resident(vet_plan *plan, const double *h_mu, const double *h_mv,
                                         const int32_t *h_ids, int n_users, int n_frames,
                                         double *h_entropy, int32_t *h_common, vet_result **out);
/* which: 0 = assignments / pairs, 1 = weights / source counts; rows [row0, row0 + n_rows) -> h_dst */
int vet_result_fetch(vet_result *result, int which, int64_t row0, int64_t n_rows, void *h_dst);
int vet_result_free(vet_result *result);

/* ---- angular distances ------------------------------------------------------------------------
 * vector_angle_distance / find_angular_distances (utilities/entropy_utils.py:41-87) for m vectors x n tile centres:
 *   h_out[i*n + j] = arccos(clip(dot(v_i / |v_i|, t_j / |t_j|), -1, 1))   radians, [0, pi]
 * h_vectors [m*3] and h_tiles [n*3] are raw (not normalised) xyz, as Vector holds them; a zero-length vector gives NaN
 * (numpy's 0/0), no error.  Same arithmetic as the nearest-tile and weight kernels (find_nearest_tile is the first minimum
 * of a row of this matrix).  Synchronous. */
int vet_angular_distances(vet_ctx *ctx, const double *h_vectors, int64_t n_vectors, const double *h_tiles, int n_tiles,
                          double *h_out);

/* ---- tile boundary geometry of a Fibonacci tiling ---------------------------------------------
 * get_fb_tile_boundaries (utilities/data_utils.py:58-189): for every tile the boundary edges (pairs of points on
 * the unit sphere) in the order the reference appends them.  h_tiles [n*3] are the lattice Vectors as
 * generate_fibonacci_lattice returns them; h_edges [n][max_edges][2][3] (NaN padded), h_count [n] edges per tile.
 * VET_ERR_UNSUPPORTED when a tile has more than 32 neighbours within 1.7 x its nearest one or more than
 * max_edges edges.  Synchronous.  The corner walk and the spherical-excess areas built on the edges
 * (get_tile_corners :530-575, compute_spherical_polygon_area :657-678) stay on the host. */
int vet_fb_tile_boundaries(vet_ctx *ctx, const double *h_tiles, int n_tiles, int max_edges,
                           double *h_edges, int32_t *h_count);

/* ---- host-side track loader (no GPU involved) -------------------------------------------
 * Replaces the per-file `pd.read_csv(filepath)` + column selection of process_viewport_data
 * (utilities/data_utils.py:305-316) for a whole directory: the files are parsed on n_threads host
 * threads (0 = all cores) into FP64 columns `time`, `2dmu`, `2dmv`, one entry per data row, NaN for
 * a missing value (rows are NOT dropped here; the caller applies dropna and the range checks).
 * The decimal conversion reproduces pandas' default C-engine converter bit for bit; a file that is
 * not plain unquoted numeric CSV gets status VET_CSV_FALLBACK and must be parsed by pandas. */
#define VET_CSV_OK 0
#define VET_CSV_FALLBACK 1   /* syntax outside the fast path: parse this file with pandas */
#define VET_CSV_IO 2         /* cannot open / read */
typedef struct vet_track {
    double *time, *mu, *mv;  /* malloc'ed by the library, n_rows each; release with vet_csv_free_tracks */
    int64_t n_rows;
    int status;              /* VET_CSV_* */
} vet_track;
int vet_csv_read_tracks(int n_files, const char *const *paths, vet_track *tracks, int n_threads);
void vet_csv_free_tracks(int n_files, vet_track *tracks);

#ifdef __cplusplus
}
#endif
#endif /* VET_H_ */
