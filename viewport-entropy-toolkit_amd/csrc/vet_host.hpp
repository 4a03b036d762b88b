// vet_host.hpp — host-side state shared by the translation units of libvet_hip.so (not part of the C-ABI).
//
//   vet_context.hip     library / context / profiling / device-memory helpers, the tuning knobs (parsed once)
//   vet_plan.hip        device tables of a plan: direction table, lattices, nearest-tile LUTs, alias and weight tables,
//                       error bounds; parity read-back hooks; angular distances; tile boundary geometry
//   vet_spatial.hip     launch logic of the spatial-entropy kernels (single videos and batches)
//   vet_transition.hip  launch logic of the transition-entropy kernels (single videos and batches)
//   vet_hostapi.hip     host-buffer entry points and device-resident results (no kernels of their own)
// Every kernel header is included by exactly one of them.  There is no CPU compute path anywhere.
#pragma once
#include "../../include/vet.h"
#include "vet_layout.hpp"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace vh {

int fail(int code, const char* fmt, ...);      // records the thread's error message, returns code
const char* last_error();

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return vh::fail(VET_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                                  \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t b) { return hipMalloc(&p, b ? b : 8); }
};

enum { KID_GRID = 0, KID_NEAREST = 1, KID_SPATIAL = 2, KID_TRANSITION = 3, KID_FINALIZE = 4, KID_WTAB = 5, KID_WEIGHTS = 6, KID_COUNT = 7 };

struct EventPair {
    int kid;
    hipEvent_t a, b;
};

// What the weights-only pass of the precise sweep needs of a plan (tile_weights VALUES at the reference's precision:
// exact FP64 weights summed in the reference's column order, calculate_tile_weights / compute_spatial_entropy,
// utilities/entropy_utils.py:108-144, 179-192).  The two device tables are shared with the plan, so a device-resident
// result (vet_result) that recomputes its weight rows on fetch may outlive the plan and its context.
struct WeightsCore {
    int device = 0;
    size_t lds_max = 64 * 1024;
    int n_cu = 256;
    std::shared_ptr<void> dir_unit;    // [n_dirs][3] f64 unit directions
    std::shared_ptr<void> tiles0;      // [n0][3] f64 unit tile centres of lattice 0
    int n0 = 0;
    int64_t n_dirs = 0;
    double cos_cull = 0.0, max_ang = 0.0, power = 2.0;
    // Exact weight rows of lattice 0 (k_wexact, built on the first request for weights): one ELL row per canonical
    // direction holding the tile and the exact FP64 weight of every tile with distance < fov/2 — zero-valued keys included.
    // The weights pass then gathers rows (k_weights_gather) instead of sweeping every tile with acos / pow per sample.
    struct Exact {
        int state = 0;                 // 0 not decided, 1 ready, -1 not usable (too large for the device, or no memory at the
                                       // first request): precise sweep instead.  Decided once (ensure_exact_weights, on the
                                       // plan's single thread) and never changed afterwards: results that share this core
                                       // read it from other threads
        int stride = 0, n_rows = 0;
        std::shared_ptr<void> alias;   // [n_dirs] u32  direction -> row | mirrored << 31
        std::shared_ptr<void> idx;     // [n_rows][stride] u16 tiles
        std::shared_ptr<void> w;       // [n_rows][stride] f64 weights
        std::shared_ptr<void> len;     // [n_rows] u32 entries in use
    } ex;
};

// One slot of the batch-descriptor ring (vet_ctx::stage)
struct BatchStage {
    void* h = nullptr;          // pinned host copy
    void* d = nullptr;          // device copy
    size_t cap = 0;
    hipEvent_t done = nullptr;  // recorded behind the last launch that reads d
    bool pending = false;
};
constexpr int kBatchStages = 4;

// Tuning knobs.  The environment is read ONCE, in vet_create; nothing between a C-ABI entry point and
// its kernel launches calls getenv.  A value outside its range is ignored (the built-in default stays).
struct Tuning {
    int gs_log2 = 0;            // VET_GS_LOG2 1..4: lanes per gather group (0: by row length)
    int tab_interleave = 1;     // VET_TAB_INTERLEAVE
    int lut_threads = 256;      // VET_LUT_THREADS
    int lut_fpw = 0;            // VET_LUT_FPW (0: by shape)
    int stride_align = 64;      // VET_STRIDE_ALIGN
    int no_dedup = 0;           // VET_NO_DEDUP
    int dedup_min_users = 128;  // VET_DEDUP_MIN_USERS
    int no_mirror = 0;          // VET_NO_MIRROR
    int u_wgs_per_cu = 2;       // VET_U_WGS_PER_CU
    int u_no_lds = 0;           // VET_U_NO_LDS
    int u_fpw = 0;              // VET_U_FPW (0: by shape)
    int u_waves = 4;            // VET_U_WAVES
    int t_threads = 0;          // VET_T_THREADS (0: by shape)
    int t_wgs_per_cu = 0;       // VET_T_WGS_PER_CU (0: by LDS)
    int t_global = 0;           // VET_T_GLOBAL
    int t_hs_pct = 200;         // VET_T_HS_PCT
    int no_fused = 0;           // VET_NO_FUSED
    int fused_single = 0;       // VET_FUSED: fused table also for one-lattice plans
    int lut_occ8 = -1;          // VET_LUT_OCC8 (fused table kernel: -1 by shape)
    int fused_narrow = 1;       // VET_FUSED_NARROW: 8-lane rows for fused rows of 65..96 entries
    int narrow_deal = 1;        // VET_NARROW_DEAL: class-dealt blocks for those 8-lane rows (0: plain order, round 4)
    std::string lut_timeline;   // VET_LUT_TIMELINE=path (development builds only): per-workgroup wall-clock timeline of the fused table kernel
    int no_exact_rows = 0;      // VET_NO_EXACT_ROWS: the weights pass never builds the exact weight rows (as if they did not fit
                                // the device): the precise sweep in weights-only mode serves — the fallback's test switch
    void from_environment();
};

}  // namespace vh

struct vet_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    size_t lds_max = 64 * 1024;
    vh::Tuning tune;
    // grow-only workspace for per-lattice entropies + status words of the host variants
    void* ws = nullptr;
    size_t ws_bytes = 0;
    double* d_log2 = nullptr;      // log2(k), k = 0..4096
    bool attrs_set = false;        // dynamic-LDS limits of the run kernels raised (first plan)
    // grow-only device staging buffers (no hipMalloc per call): 0-6 host-buffer entry points, (7 unused: batch descriptors live in the blob ring below),
    // 8 transition scratch, 9 resolve list
    void* pool[12] = {};
    size_t pool_cap[12] = {};
    // descriptor blobs of the batch entry points: a ring of (pinned host, device) buffer pairs, each guarded by an event
    // recorded behind the last launch that reads it (vh::BatchBlob) — the calls only enqueue work, so neither the host
    // copy nor the device copy of one batch may be reused while an earlier batch (possibly on another stream) is pending
    vh::BatchStage stage[vh::kBatchStages];
    int stage_next = 0;
    // profiling
    bool profiling = false;
    std::vector<vh::EventPair> pending;
    std::vector<hipEvent_t> free_events;
    double prof_ms[vh::KID_COUNT] = {};
    int64_t prof_n[vh::KID_COUNT] = {};
};

namespace vh {

struct Lattice {
    int n = 0;
    double* d_tiles = nullptr;     // [n][3] unit
    std::vector<double> h_unit;    // host copy of the unit tiles
    uint16_t* d_nearest = nullptr; // [n_dirs]
    double hmax = 0.0;
    // direction weight table (ELL), built on first use when the video has more samples than the
    // plan has directions
    uint32_t* d_tab_w = nullptr;   // [n_rows+1][stride] u32 mantissas (block floating point per row)
    uint16_t* d_tab_i = nullptr;   // [n_rows+1][stride]
    uint32_t* d_tab_meta = nullptr;// [n_rows+1] entries in use | row shift << 16
    uint8_t* d_row_s = nullptr;    // [n_dirs+1] row shift (k_row_stats)
    uint16_t* d_row_e = nullptr;   // [n_dirs+1] unclamped row exponent (FP table)
    bool fp_table = false;         // the table holds FP32 weights (plans whose integer bound is outside the contract)
    // k_row_stats: worst-case relative entropy error of integer histograms over every possible frame
    bool stats_done = false;
    double crit_tab = 0.0;         // table formulation (step 2^(e_row - 33) per entry)
    double crit_base = 0.0;        // times the step of the sweep formulation
    long ultra = 0;                // in-FoV (direction, tile) pairs whose weight is below 2^-1048 (k_row_stats): the
                                   // reference's NaN frames; such plans never use an integer formulation
    int markers = 0;               // marker entries of the FP table (k_wtab): frames they decide go to the precise sweep
    int last_form = -1;            // formulation of the last weighted call (parity / bench introspection)
    int stride = 0;                // 0 = not built, -1 = not usable (too large)
    int gs_log2 = 4;               // lanes per gather group (log2); fixed when the table is built
    bool interleaved = false;      // well-filled row blocks are dealt by LDS bank class (k_wtab)
    bool binned = false;           // caller-supplied direction -> bin table (naive lat/lon tiling)
    int norm_n = 0;                // tile count used by the normaliser rule
};

}  // namespace vh

struct vet_plan {
    vet_ctx* ctx = nullptr;
    int W = 0, H = 0;
    bool grid = false;
    int64_t n_dirs = 0;
    double* d_dir_raw = nullptr;
    double* d_dir_unit = nullptr;
    std::vector<vh::Lattice> lat;
    double fov = 120.0, max_ang = 0.0, power = 2.0;
    int weighted = 1;
    double cos_cull = 0.0;
    int table_policy = 0;          // 0 by call size, 1 table whenever it is inside the contract, -1 never
    bool raw_weights = false;      // tile_weights = the formulation's own histogram (diagnostic) instead of the exact pass
    uint32_t* d_alias = nullptr;   // [n_dirs] direction id -> table row (dense) | mirrored << 31 (ensure_alias)
    bool mirror = false;           // rows are shared between mirror-image directions
    uint2* d_dirrec = nullptr;     // [n_dirs] alias | nearest tile | lattice-0 row meta (k_dirrec), dedup-capable plans
    int n_rows = 0;                // table rows in use = canonical directions, densely numbered (ensure_alias)
    int* d_canon = nullptr;        // [n_rows] table row -> its direction
    // fused table: one row per distinct direction over ALL lattices (vet_layout.hpp)
    struct Fused {
        int state = 0;             // 0 not built, 1 ready, -1 not usable for this plan
        int R = 0, stride = 0, gs_log2 = 4;
        bool interleaved = false;
        vet::FusedLayout lay;
        uint32_t* d_meta = nullptr;// [R+1] entries in use | row shift << 16
        uint2* d_dirrec = nullptr; // [n_dirs] k_spatial_lut's per-direction record over the fused rows (k_dirrec)
        uint8_t* d_row_s = nullptr;// [R+1] fused row shifts
        uint32_t* d_w = nullptr;   // [R+1][stride]
        uint16_t* d_i = nullptr;   // [R+1][stride]
    } fused;
    std::shared_ptr<vh::WeightsCore> wcore;   // owner of d_dir_unit and lat[0].d_tiles (shared with device-resident results)
    bool stats_all = false;        // k_row_stats has run for every weighted lattice
    bool ultra = false;            // some lattice has ultra-tiny in-FoV weights: FP64 formulations only (plan-wide)
};

namespace vh {

// hipEvent pair around the launches of a scope, on the launch stream (vet_profile_*)
struct ProfScope {
    vet_ctx* c;
    hipStream_t s;
    int kid;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(vet_ctx* c_, hipStream_t s_, int kid_) : c(c_), s(s_), kid(kid_) {
        if (!c->profiling) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!c->free_events.empty()) { e = c->free_events.back(); c->free_events.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get(); b = get();
        (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, s);
        c->pending.push_back({kid, a, b});
    }
};

// The descriptor blob of ONE batch call: acquire() takes the ring's next slot (waiting only if the batch that used it
// kBatchStages calls ago has not finished), the caller fills host(), upload() enqueues the copy, the launches follow, and
// the destructor records the slot's event on the launch stream.
struct BatchBlob {
    vet_ctx* c = nullptr;
    BatchStage* st = nullptr;
    hipStream_t s = nullptr;
    size_t bytes = 0;
    bool uploaded = false;
    int acquire(vet_ctx* ctx, size_t nbytes);
    void* host() const { return st->h; }
    void* dev() const { return st->d; }
    int upload(hipStream_t stream);
    ~BatchBlob();
};

int ensure_ws(vet_ctx* c, size_t bytes);                          // grow-only workspace (c->ws)
int pooled(vet_ctx* c, int slot, size_t bytes, void** out);       // slot-indexed grow-only device buffer
int grid_for(long work, int block, int n_cu);
int check_run_args(const vet_plan* pl, int U, int T, const void* out);

constexpr size_t kMaxTableBytes = (size_t)24 << 30;   // per lattice; HBM is 288 GB
constexpr double kContractMargin = 1e-7;              // bound on |dH|/H an integer formulation may have (contract: 1e-6)
constexpr size_t kWholeLds = 160 * 1024 - 512;        // a single workgroup per CU may take the whole LDS

// vet_plan.hip: tables built on first use (each synchronises once)
int ensure_alias(vet_plan* pl);
int ensure_all_stats(vet_plan* pl, hipStream_t s);
int ensure_wtab(vet_plan* pl, int k, hipStream_t s);
int ensure_fused(vet_plan* pl, hipStream_t s);
bool any_binned(const vet_plan* pl);

// vet_spatial.hip: tile_weights of lattice 0 for frames [0, T) of a sample array given as direction ids, by the precise
// sweep in weights-only mode (users in column order); prof = context to attribute the launch to, or null
int weights_pass_ids(const WeightsCore& w, const int32_t* d_ids, int U, int T, double* d_weights, hipStream_t s, vet_ctx* prof);
// vet_plan.hip: the exact weight rows of lattice 0 (first use; synchronises once).  Leaves ex.state = -1 when they do not fit.
int ensure_exact_weights(vet_plan* pl, hipStream_t s);
// (mu, mv) -> direction ids [n] (-1 absent or out of range) on the plan's pixel grid
int sample_ids(const vet_plan* pl, const double* d_mu, const double* d_mv, long n, int32_t* d_out, hipStream_t s);

// dynamic-LDS limits of the run kernels (once per context, from vet_plan_create)
int spatial_set_attrs(vet_ctx* c);
int transition_set_attrs(vet_ctx* c);

}  // namespace vh
