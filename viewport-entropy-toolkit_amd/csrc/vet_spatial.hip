// vet_spatial.hip — launch logic of the spatial-entropy kernels behind vet_spatial_entropy* (include/vet.h):
// formulation choice (table / sweep / precise / FP table), launch geometry, single videos and batches.
// No CPU compute path; nothing here reads the environment (the context's Tuning was parsed in vet_create).
#include "vet_host.hpp"
#include "vet_finalize.hpp"
#include "vet_spatial_sweep.hpp"
#include "vet_weights_pass.hpp"
#include "vet_spatial_lut.hpp"
#include "vet_spatial_u.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace vh {

namespace {

struct Geometry {
    int NW, FPW, G, UC, R;
    size_t lds;
};

// launch geometry of the weighted sweep kernels (k_spatial_w) for a lattice of n tiles and U users
int sweep_geometry(size_t lds_max, int n, int U, Geometry* g, bool one_frame = false) {
    g->R = n > vet::WAVE ? 2 : 1;
    g->G = (n + vet::WAVE * g->R - 1) / (vet::WAVE * g->R);
    if (g->G >= 4) { g->NW = g->G > 16 ? 16 : g->G; g->FPW = 1; }
    else { g->NW = 4; g->FPW = 4 / g->G; }
    if (one_frame) g->FPW = 1;
    g->UC = U < 1024 ? U : 1024;
    auto lds_of = [&](int fpw, int uc) {
        size_t b = (size_t)fpw * n * 8;
        b += (size_t)fpw * uc * 24;
        b += (size_t)g->NW * vet::WAVE * (g->R + 1) * (8 + 2);
        b += (size_t)2 * fpw * 4 + 64;
        return b;
    };
    while (lds_of(g->FPW, g->UC) > lds_max && g->FPW > 1) g->FPW /= 2;
    while (lds_of(g->FPW, g->UC) > lds_max && g->UC > 64) g->UC /= 2;
    g->lds = lds_of(g->FPW, g->UC);
    if (g->lds > lds_max)
        return fail(VET_ERR_UNSUPPORTED, "lattice of %d tiles does not fit the LDS histogram (%zu B)", n, g->lds);
    return VET_OK;
}

// launch geometry of the spatial kernels for a lattice of n tiles and U users
int spatial_geometry(const vet_ctx* c, int n, int U, bool weighted, Geometry* g, bool one_frame = false) {
    if (!weighted) {
        // k_spatial_u: one wave per frame in the entropy phase; keep >= 2048 samples per workgroup
        g->R = 1; g->G = 1; g->UC = 0;
        g->NW = c->tune.u_waves;
        g->FPW = U >= 2048 ? 2 : (U >= 512 ? 4 : (U >= 128 ? 8 : 32));
        if (c->tune.u_fpw) g->FPW = c->tune.u_fpw;
        while ((size_t)g->FPW * n * 4 > c->lds_max && g->FPW > 1) g->FPW /= 2;
        g->lds = (size_t)g->FPW * n * 4;
        if (g->lds > c->lds_max)
            return fail(VET_ERR_UNSUPPORTED, "lattice of %d tiles does not fit the LDS histogram (%zu B)", n, g->lds);
        return VET_OK;
    }
    return sweep_geometry(c->lds_max, n, U, g, one_frame);
}

// weight-evaluation variant of k_spatial_w (see fov_weight_fx)
int weight_mode(const vet_plan* pl) {
    const bool fast_acos = pl->max_ang <= 1.0471975511965979;   // fov <= 120 deg: z <= 0.2502
    if (fast_acos && pl->power == 2.0) return 1;
    if (fast_acos && pl->power == 1.0) return 2;
    return 0;
}

template <bool FROM_IDS>
const void* spatial_w_kernel(int wmode, int R, bool precise = false) {
    if (precise) return R == 1 ? (const void*)vet::k_spatial_w<FROM_IDS, 0, 1, true> : (const void*)vet::k_spatial_w<FROM_IDS, 0, 2, true>;
#define VET_PICK(W, RR) if (wmode == W && R == RR) return (const void*)vet::k_spatial_w<FROM_IDS, W, RR, false>
    VET_PICK(0, 1); VET_PICK(0, 2); VET_PICK(1, 1); VET_PICK(1, 2); VET_PICK(2, 1); VET_PICK(2, 2);
#undef VET_PICK
    return nullptr;
}

// fused rows are short (config 4: 94 entries = 2 blocks): four rows in flight per lane group and 7 workgroups per CU
// (72 VGPRs, one spilled register) measured 3-5 % faster than two rows and 8 workgroups (config 4 0.165 -> 0.160 ms, 64 x config 2
// 0.907 -> 0.882, defaults 0.531 -> 0.519); single-lattice tables keep two (profiles/r01/v3_*)
#ifndef VET_FUSED_UN
#define VET_FUSED_UN 4
#endif
// k_spatial_lut<FUSED, UN == 2> IS the narrow (8-lane rows) kernel: the row width of the class-dealt layout is inferred from
// UN there (GSL_IL), so the 16-lane fused kernel must not be built with two rows in flight
static_assert(VET_FUSED_UN != 2, "VET_FUSED_UN=2 would alias the 16-lane fused kernel onto the 8-lane (narrow) instantiation");
template <bool FROM_IDS>
const void* lut_kernel_fused(bool il, bool occ8, bool dedup, bool narrow = false) {
    // narrow: rows of 8-lane groups (32-entry blocks; fused rows of 65..96 entries fill three of them instead of two half-empty
    // 64-entry ones): two rows in flight per lane group keep a wave at 16 rows per step, as four do with 16-lane groups
    if (narrow) {
#define VET_PICKN(I, O, D) if (il == I && occ8 == O && dedup == D) return (const void*)vet::k_spatial_lut<FROM_IDS, 2, I, O, D, false, true>
        VET_PICKN(false, false, false); VET_PICKN(false, true, false); VET_PICKN(false, false, true); VET_PICKN(false, true, true);
        VET_PICKN(true, false, false); VET_PICKN(true, true, false); VET_PICKN(true, false, true); VET_PICKN(true, true, true);
#undef VET_PICKN
    }
#define VET_PICK(I, O, D) if (il == I && occ8 == O && dedup == D) return (const void*)vet::k_spatial_lut<FROM_IDS, VET_FUSED_UN, I, O, D, false, true>
    VET_PICK(false, false, false); VET_PICK(false, true, false); VET_PICK(true, false, false); VET_PICK(true, true, false);
    VET_PICK(false, false, true); VET_PICK(false, true, true); VET_PICK(true, false, true); VET_PICK(true, true, true);
#undef VET_PICK
    return nullptr;
}

template <bool FROM_IDS>
const void* lut_kernel(bool il, bool occ8, bool dedup, bool fpt = false) {
    if (fpt) {      // FP table: 7 workgroups per CU (FP64 scale registers)
#define VET_PICKF(I, D) if (il == I && dedup == D) return (const void*)vet::k_spatial_lut<FROM_IDS, 2, I, false, D, true>
        VET_PICKF(false, false); VET_PICKF(true, false); VET_PICKF(false, true); VET_PICKF(true, true);
#undef VET_PICKF
        return nullptr;
    }
#define VET_PICK(I, O, D) if (il == I && occ8 == O && dedup == D) return (const void*)vet::k_spatial_lut<FROM_IDS, 2, I, O, D, false>
    VET_PICK(false, false, false); VET_PICK(false, true, false); VET_PICK(true, false, false); VET_PICK(true, true, false);
    VET_PICK(false, false, true); VET_PICK(false, true, true); VET_PICK(true, false, true); VET_PICK(true, true, true);
#undef VET_PICK
    return nullptr;
}

// Frames per workgroup of the table kernel: about 1024 samples per workgroup, at most one frame per
// wave (the epilogue reduces a frame per wave, and every frame costs n_sum * 8 B of LDS), and never
// so many that the launch has fewer than ~4 workgroups per CU (measured: config 4 best at 4 frames
// x 256 users, config 2 at 2 x 64 with only 3000 frames; the reference's five default lattices, 1425
// tiles, at 1 x 256: profiles/r01/v6_table_geometry_sweep.log).
int lut_frames_per_wg(int U, long total_frames, int n_cu, int n_sum) {
    long f = 1024 / (U > 0 ? U : 1);
    const long by_grid = total_frames / (4L * n_cu);
    if (f > by_grid) f = by_grid;
    if (f > 4) f = 4;
    int fpw = 1;
    while (2 * fpw <= f) fpw *= 2;
    // keep ~7 workgroups per CU resident: at most ~20 KB of LDS histograms per workgroup
    while (fpw > 1 && (size_t)fpw * n_sum * 8 > 20 * 1024) fpw /= 2;
    // the launch runs in waves of 8 workgroups per CU: fewer frames per workgroup where that shortens the tail
    // (config 4: 2 500 workgroups of 4 frames = 2 waves x 4 frames; 5 000 of 2 frames = 3 x 2; 10 000 of 1 frame = 5 x 1;
    // measured 0.175 / 0.159 / 0.163 ms: the constant charges a workgroup's fixed cost)
    const long slots = 8L * n_cu;
    auto cost = [&](int f) { const long wgs = (total_frames + f - 1) / f; return (double)((wgs + slots - 1) / slots) * (f + 0.6); };
    for (int f = fpw / 2; f >= 1; f /= 2)
        if (cost(f) < cost(fpw)) fpw = f;
    return fpw;
}

// The formulation of a weighted call is a function of the plan, the call's shape and (table does not fit the free
// device memory -> sweep) the memory left on the device — never of what the plan has processed before — and every
// formulation adds in a fixed order, so the same input gives the same floats:
//   table    policy +1, or policy 0 and the call holds at least VET_TABLE_SAMPLES_PER_DIRECTION samples per direction
//            of the table (include/vet.h: building a row costs about what the sweep spends on 20 samples; a gathered
//            sample is 4-10x cheaper than a swept one), if the table fits and its error bound is inside the contract;
//   sweep    integer (2^-52) histogram, if its error bound is inside the contract;
//   precise  FP64 histogram and exact weights otherwise.
enum { F_TABLE = 0, F_SWEEP = 1, F_PRECISE = 2, F_FTABLE = 3 };

bool table_requested(const vet_plan* pl, long samples, int U) {
    if (!pl->weighted || pl->table_policy < 0 || any_binned(pl) || U >= 65536) return false;
    if ((int)pl->lat.size() > vet::MAX_LATTICES) return false;
    return pl->table_policy > 0 || samples >= (long)VET_TABLE_SAMPLES_PER_DIRECTION * (long)pl->n_dirs;
}

int sweep_shift(int U) {
    int ubits = 0;
    while ((1L << ubits) < (long)U) ++ubits;
    return ubits > 10 ? ubits - 10 : 0;      // per-tile sums of U weights stay below 2^62
}

// integer sweep if its error bound is inside the contract, FP64 sweep otherwise
int sweep_formulation(const vet_plan* pl, const Lattice& L, int U) {
    if (pl->ultra) return F_PRECISE;          // the reference's NaN frames need the exact key set
    // the sweep truncates at 2^(shift-52); its fast arc cosine (fov <= 120, power 1 or 2) is good to 4e-14
    double step = std::ldexp(1.0, sweep_shift(U) - 52);
    if (weight_mode(pl) != 0 && step < 4e-14) step = 4e-14;
    return L.crit_base * step <= kContractMargin ? F_SWEEP : F_PRECISE;
}

// formulation of lattice k for a call; builds the statistics (and the table) on first use
int choose_formulation(vet_plan* pl, int k, bool want_table, int U, hipStream_t s, int* out) {
    Lattice& L = pl->lat[k];
    int rc = ensure_all_stats(pl, s);
    if (rc) return rc;
    if (want_table) {
        rc = ensure_wtab(pl, k, s);
        if (rc) return rc;
        if (L.stride > 0) { *out = L.fp_table ? F_FTABLE : F_TABLE; return VET_OK; }
    }
    *out = sweep_formulation(pl, L, U);
    return VET_OK;
}

// one launch of the table kernel over lattices lat_idx[0..K) of the plan (single video or a batch)
template <bool FROM_IDS>
int launch_lut(vet_plan* pl, const int* lat_idx, int K, const vet::SampleSrc& src, int U, int T,
               const vet::VideoDesc* d_videos, int n_videos, int blocks_batch, size_t lds_batch, int batch_max_users,
               double* d_entropy, int32_t* d_assign, double* d_weights, int32_t* d_present, int32_t* d_status,
               hipStream_t s, bool* launched, uint32_t* d_resolve = nullptr) {
    vet_ctx* c = pl->ctx;
    *launched = false;
    vet::LutParams q{};
    q.resolve = d_resolve;
    q.dedup_min_users = c->tune.dedup_min_users;
    q.videos = d_videos; q.n_videos = n_videos;
    q.src = src; q.U = U; q.T = T;
    q.nearest = pl->lat[lat_idx[0]].d_nearest;
    q.alias = pl->d_alias;
    q.dirrec = pl->d_dirrec;
    q.rec_meta = lat_idx[0] == 0 ? 1 : 0;
    q.K = K; q.n_sum = 0;
    bool il = false;
    const bool fpt = pl->lat[lat_idx[0]].fp_table;       // the caller passes lattices of one kind
    for (int k = 0; k < K; ++k) {
        const Lattice& L = pl->lat[lat_idx[k]];
        q.lat[k].tab_w = L.d_tab_w; q.lat[k].tab_i = L.d_tab_i; q.lat[k].tab_meta = L.d_tab_meta;
        q.lat[k].stride = L.stride;
        q.lat[k].gs_log2 = L.gs_log2; q.lat[k].interleaved = L.interleaved ? 1 : 0;
        q.lat[k].n = L.n; q.lat[k].hmax = L.hmax;
        q.lat[k].zrow = (uint32_t)pl->n_rows;
        q.n_sum += L.n;
        il = il || L.interleaved;
    }
    q.entropy = d_entropy; q.assign = d_assign; q.weights = d_weights; q.present = d_present;
    q.status = d_status;
    // the per-frame set of distinct rows pays for itself from ~128 users per frame on (measured: config 2, 64
    // users, 0.0332 ms without vs 0.0366 ms with; config 4, 256 users, equal; config 3, 1024 users, 1.60 -> 1.53 ms)
    const bool dedup = (uint64_t)pl->n_rows <= vet::DEDUP_MAX_DIRS && pl->d_dirrec && !c->tune.no_dedup &&
                       (d_videos ? batch_max_users : U) >= c->tune.dedup_min_users;
    int blocks = blocks_batch, threads = 256;
    size_t lds = lds_batch;
    bool occ8 = true;
    // FP table: canonical row order through a bitmap over (row, mirrored) where that is small (<= 8 KB of LDS)
    q.sort_words = (fpt && dedup && 2 * pl->n_rows <= 65536) ? (int)((2 * pl->n_rows + 31) / 32) : 0;
    if (!d_videos) {
        q.UC = U < 2048 ? U : 2048;
        threads = fpt ? 256 : c->tune.lut_threads;     // the FP table's row sort counts on 256 threads
        int fpw = lut_frames_per_wg(U, T, c->n_cu, q.n_sum);
        if (c->tune.lut_fpw) fpw = c->tune.lut_fpw;
        for (;; fpw /= 2) {
            lds = vet::lut_lds_bytes(U, q.UC, fpw, q.n_sum, dedup, d_resolve != nullptr, fpt ? threads / 64 : 1, q.sort_words);
            if (lds <= c->lds_max || fpw == 1) break;
        }
        if (lds > c->lds_max) return VET_OK;      // not launched: caller falls back to the sweep
        q.FPW = fpw;
        blocks = (T + fpw - 1) / fpw;
        occ8 = K == 1 && threads == 256;
    } else {
        q.FPW = 1; q.UC = 1;
    }
    ProfScope ps(c, s, KID_SPATIAL);
    void* args[] = {(void*)&q};
    // 2 rows in flight per lane group measured best (4 and 8 were tried, profiles/r01/v3_*)
    HIP_TRY(hipLaunchKernel(lut_kernel<FROM_IDS>(il, occ8 && !fpt, dedup, fpt), dim3((unsigned)blocks), dim3(threads), args, lds, s));
    HIP_TRY(hipGetLastError());
    *launched = true;
    return VET_OK;
}

// k_spatial_lut over the plan's fused table (single video, or a batch): the kernel sees ONE lattice of N slots
template <bool FROM_IDS>
int launch_lut_fused(vet_plan* pl, const vet::SampleSrc& src, int U, int T, const vet::VideoDesc* d_videos, int n_videos,
                     int blocks_batch, size_t lds_batch, int batch_max_users, double* d_entropy, int32_t* d_assign,
                     double* d_weights, int32_t* d_present, int32_t* d_status, hipStream_t s, bool* launched) {
    vet_ctx* c = pl->ctx;
    const auto& F = pl->fused;
    *launched = false;
    vet::LutParams q{};
    q.dedup_min_users = c->tune.dedup_min_users;
    q.videos = d_videos; q.n_videos = n_videos;
    q.src = src; q.U = U; q.T = T;
    q.nearest = pl->lat[0].d_nearest; q.alias = nullptr; q.dirrec = F.d_dirrec; q.rec_meta = 1;
    q.K = 1; q.n_sum = F.lay.N;
    q.lat[0].tab_w = F.d_w; q.lat[0].tab_i = F.d_i; q.lat[0].tab_meta = F.d_meta;
    q.lat[0].stride = F.stride; q.lat[0].gs_log2 = F.gs_log2; q.lat[0].interleaved = F.interleaved ? 1 : 0;
    q.lat[0].n = F.lay.N; q.lat[0].hmax = 0.0; q.lat[0].zrow = (uint32_t)F.R;
    q.lay = F.lay;
    q.entropy = d_entropy; q.assign = d_assign; q.weights = d_weights; q.present = d_present; q.status = d_status;
    const bool dedup = !c->tune.no_dedup && (d_videos ? batch_max_users : U) >= c->tune.dedup_min_users;
    int blocks = blocks_batch, threads = 256;
    size_t lds = lds_batch;
    if (!d_videos) {
        q.UC = U < 2048 ? U : 2048;
        int fpw = lut_frames_per_wg(U, T, c->n_cu, q.n_sum);
        if (c->tune.lut_fpw) fpw = c->tune.lut_fpw;
        for (;; fpw /= 2) {
            lds = vet::lut_lds_bytes(U, q.UC, fpw, q.n_sum, dedup);
            if (lds <= c->lds_max || fpw == 1) break;
        }
        if (lds > c->lds_max) return VET_OK;      // not launched: the caller falls back
        q.FPW = fpw;
        blocks = (T + fpw - 1) / fpw;
    } else {
        q.FPW = 1; q.UC = 1;
    }
    // 8 workgroups per CU (64 VGPRs) pay for the narrow kernel on a single video of many users (two rows in flight: few
    // registers; config 4 0.146 -> 0.142 ms), not for the 16-lane kernel with four rows in flight (defaults +5 %), for
    // batches (+1 %) or for frames without the set (config 2 +2 %)
    const bool occ8 = c->tune.lut_occ8 >= 0 ? c->tune.lut_occ8 != 0 : (F.gs_log2 == 3 && !d_videos && dedup);
#if VET_STAGE_CYCLES
    DevBuf dbg, tl;                           // development builds: cycles per stage (thread 0 of every workgroup), synchronous
    HIP_TRY(dbg.alloc(64));
    HIP_TRY(hipMemsetAsync(dbg.p, 0, 64, s));
    // VET_LUT_TIMELINE=path: per-workgroup wall-clock timeline instead of the (intrusive: waits on vmcnt) sub-stage counters
    const char* tl_path = c->tune.lut_timeline.empty() ? nullptr : c->tune.lut_timeline.c_str();
    const long n_items_tl = blocks;
    if (tl_path) {
        HIP_TRY(tl.alloc((size_t)n_items_tl * 48));
        HIP_TRY(hipMemsetAsync(tl.p, 0, (size_t)n_items_tl * 48, s));
        q.timeline = (unsigned long long*)tl.p;
    } else {
        q.dbg = (unsigned long long*)dbg.p;
    }
#endif
    {
        ProfScope ps(c, s, KID_SPATIAL);
        void* args[] = {(void*)&q};
        HIP_TRY(hipLaunchKernel(lut_kernel_fused<FROM_IDS>(F.interleaved, occ8, dedup, F.gs_log2 == 3), dim3((unsigned)blocks), dim3(threads), args, lds, s));
        HIP_TRY(hipGetLastError());
    }
#if VET_STAGE_CYCLES
    if (tl_path) {
        std::vector<unsigned long long> h((size_t)n_items_tl * 6);
        HIP_TRY(hipMemcpyAsync(h.data(), tl.p, h.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (FILE* f = fopen(tl_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    } else
    {
        unsigned long long t[8] = {};
        HIP_TRY(hipMemcpyAsync(t, dbg.p, 64, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        fprintf(stderr, "[k_spatial_lut fused] blocks %d FPW %d lds %zu | cycles per workgroup (thread 0): samples->set %.0f  lists %.0f  walk %.0f  entropy %.0f"
                        " | samples->set = init %.0f + sample loads %.0f + record gathers %.0f + inserts %.0f (+ barrier)\n",
                blocks, q.FPW, lds, (double)t[0] / blocks, (double)t[1] / blocks, (double)t[2] / blocks, (double)t[3] / blocks,
                (double)t[4] / blocks, (double)t[5] / blocks, (double)t[6] / blocks, (double)t[7] / blocks);
    }
#endif
    *launched = true;
    return VET_OK;
}

// LDS bytes and frames per workgroup of one video of a batch (0 = does not fit)
size_t batch_video_geometry(const vet_ctx* c, int U, long total_frames, int n_sum, bool dedup, int* fpw_out, int* uc_out,
                            int priv = 1, int sort_words = 0) {
    const int UC = U < 2048 ? U : 2048;
    int fpw = lut_frames_per_wg(U, total_frames, c->n_cu, n_sum);
    size_t lds = 0;
    for (;; fpw /= 2) {
        lds = vet::lut_lds_bytes(U, UC, fpw, n_sum, dedup, false, priv, sort_words);
        if (lds <= c->lds_max || fpw == 1) break;
    }
    *fpw_out = fpw; *uc_out = UC;
    return lds <= c->lds_max ? lds : 0;
}

// markers in the FP tables of lattices lat_idx[0..K)?  Then the launch needs a resolve list (pool slot 9): [0] = count
int resolve_list_for(vet_plan* pl, const int* lat_idx, int K, int T, hipStream_t s, uint32_t** out) {
    *out = nullptr;
    bool any = false;
    for (int k = 0; k < K; ++k) any = any || (pl->lat[lat_idx[k]].fp_table && pl->lat[lat_idx[k]].markers > 0);
    if (!any) return VET_OK;
    void* buf = nullptr;
    int rc = pooled(pl->ctx, 9, ((size_t)T + 1) * sizeof(uint32_t), &buf);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(buf, 0, sizeof(uint32_t), s));
    *out = (uint32_t*)buf;
    return VET_OK;
}

// The frames a table launch could not decide (a key of the reference's dict whose table weight sum is 0.0) run
// through the precise sweep — exact weights, exact key set — which overwrites their entropy (and weights row):
// NaN where the reference's 0 * log2 0 gives NaN (entropy_utils.py:195-198).  out = the launch's entropy output
// ([T]: one lattice's row, or the mean over the K fused lattices; ws then holds the per-lattice values).
template <bool FROM_IDS>
int resolve_frames(vet_plan* pl, const int* lat_idx, int K, const vet::SampleSrc& src, int U, int T, double* out,
                   double* d_weights, const uint32_t* d_list, double* ws, hipStream_t s) {
    vet_ctx* c = pl->ctx;
    for (int k = 0; k < K; ++k) {
        const Lattice& L = pl->lat[lat_idx[k]];
        Geometry g;
        int rc = spatial_geometry(c, L.n, U, true, &g, true);
        if (rc) return rc;
        vet::SpatialParams p{};
        p.src = src; p.U = U; p.T = T;
        p.dir_unit = pl->d_dir_unit; p.nearest = L.d_nearest; p.tiles = L.d_tiles; p.n = L.n;
        p.cos_cull = pl->cos_cull;
        p.wc.max_ang = pl->max_ang; p.wc.inv_max = 1.0 / pl->max_ang; p.wc.power = pl->power; p.wc.shift = 0;
        p.hmax = L.hmax;
        p.ent_k = K == 1 ? out : ws + (size_t)k * T;
        p.assign = nullptr; p.present = nullptr; p.status = nullptr;       // written by the table launch
        p.weights = (k == 0 && lat_idx[0] == 0) ? d_weights : nullptr;
        p.FPW = 1; p.G = g.G; p.UC = g.UC;
        p.log2_tab = c->d_log2; p.full_norm = 0; p.norm_n = L.norm_n;
        p.frame_list = d_list;
        long grid = (long)c->n_cu * 4;
        if (grid > T) grid = T;
        void* args[] = {(void*)&p};
        ProfScope ps(c, s, KID_SPATIAL);
        // + the users' direction ids (canonical order of the sums): the chunk of users shrinks until both fit
        auto lds_of = [&](int uc) { return g.lds - (size_t)g.UC * 24 + (size_t)uc * 28 + 16; };
        int uc = g.UC;
        while (lds_of(uc) > c->lds_max && uc > 64) uc /= 2;
        const size_t lds = lds_of(uc);
        if (lds > c->lds_max) return fail(VET_ERR_UNSUPPORTED, "resolver: %zu B of LDS", lds);
        p.UC = uc;
        HIP_TRY(hipLaunchKernel(spatial_w_kernel<FROM_IDS>(0, g.R, true), dim3((unsigned)grid), dim3(g.NW * vet::WAVE), args, lds, s));
    }
    if (K > 1) {
        ProfScope ps(c, s, KID_FINALIZE);
        hipLaunchKernelGGL(vet::k_finalize_list, dim3(grid_for(T, 256, c->n_cu)), dim3(256), 0, s, (const double*)ws, K, (long)T,
                           d_list, out);
        HIP_TRY(hipGetLastError());
    }
    return VET_OK;
}

// tile_weights of lattice 0, frames [0, T) — the weights pass: k_weights_gather over the plan's exact FP64 weight rows
// (vet_weights_pass.hpp), or, where those rows do not exist (too large for the device), the precise sweep in weights-only
// mode (exact ocml weights per sample, every tile owned by one wave, users in column order).  -0.0 = key with the value 0.0.
template <bool FROM_IDS>
int launch_weights_pass(const WeightsCore& w, const vet::SampleSrc& src, int U, int T, double* d_weights, hipStream_t s,
                        vet_ctx* prof) {
    if (T <= 0) return VET_OK;
    if (w.ex.state == 1) {
        // the exact weight rows exist (ensure_exact_weights): gather them, one workgroup per frame
        int nw = 4;
        while (nw > 1 && (size_t)nw * w.n0 * 8 > w.lds_max) nw /= 2;
        if ((size_t)nw * w.n0 * 8 <= w.lds_max) {
            vet::WeightsGatherParams q{};
            q.src = src; q.U = U; q.T = T;
            q.alias = (const uint32_t*)w.ex.alias.get(); q.idx = (const uint16_t*)w.ex.idx.get();
            q.w = (const double*)w.ex.w.get(); q.len = (const uint32_t*)w.ex.len.get();
            q.stride = w.ex.stride; q.n = w.n0; q.out = d_weights;
            const int chunks = w.ex.stride / vet::WAVE;
            const void* fn = chunks <= 1 ? (const void*)vet::k_weights_gather<FROM_IDS, 1>
                           : chunks <= 2 ? (const void*)vet::k_weights_gather<FROM_IDS, 2>
                           : chunks <= 4 ? (const void*)vet::k_weights_gather<FROM_IDS, 4> : (const void*)vet::k_weights_gather<FROM_IDS, 0>;
            void* args[] = {(void*)&q};
            const size_t lds = (size_t)nw * w.n0 * 8;
            if (prof) {
                ProfScope ps(prof, s, KID_WEIGHTS);
                HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)T), dim3(nw * vet::WAVE), args, lds, s));
            } else {
                HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)T), dim3(nw * vet::WAVE), args, lds, s));
            }
            HIP_TRY(hipGetLastError());
            return VET_OK;
        }
    }
    Geometry g;
    int rc = sweep_geometry(w.lds_max, w.n0, U, &g);
    if (rc) return rc;
    vet::SpatialParams p{};
    p.src = src; p.U = U; p.T = T;
    p.dir_unit = (const double*)w.dir_unit.get(); p.nearest = nullptr; p.tiles = (const double*)w.tiles0.get(); p.n = w.n0;
    p.cos_cull = w.cos_cull;
    p.wc.max_ang = w.max_ang; p.wc.inv_max = 1.0 / w.max_ang; p.wc.power = w.power; p.wc.shift = 0;
    p.hmax = 1.0;
    p.ent_k = nullptr; p.assign = nullptr; p.present = nullptr; p.status = nullptr;      // weights only
    p.weights = d_weights;
    p.FPW = g.FPW; p.G = g.G; p.UC = g.UC;
    p.norm_n = w.n0; p.frame_list = nullptr;
    const int blocks = (T + g.FPW - 1) / g.FPW;
    void* args[] = {(void*)&p};
    if (prof) {
        ProfScope ps(prof, s, KID_WEIGHTS);
        HIP_TRY(hipLaunchKernel(spatial_w_kernel<FROM_IDS>(0, g.R, true), dim3(blocks), dim3(g.NW * vet::WAVE), args, g.lds, s));
    } else {
        HIP_TRY(hipLaunchKernel(spatial_w_kernel<FROM_IDS>(0, g.R, true), dim3(blocks), dim3(g.NW * vet::WAVE), args, g.lds, s));
    }
    HIP_TRY(hipGetLastError());
    return VET_OK;
}

template <bool FROM_IDS>
int launch_spatial_main(vet_plan* pl, const vet::SampleSrc& src, int U, int T, double* d_entropy, int32_t* d_assign,
                        double* d_weights, int32_t* d_present, int32_t* d_status, hipStream_t s);

// tile_weights VALUES carry the reference's precision under every formulation (utilities/entropy_utils.py:131-136,
// 190-192).  The formulations' histograms hold block-floating-point, FP32-rounded or 2^-52 fixed-point weights, good for
// the ENTROPY contract only, so on weighted plans the main launch writes no weights at all and the weights output is
// left to the weights pass over the same samples — ONE producer for every weights output (the eager d_weights and the
// rows a device-resident result computes per fetched block are the same bits) and off the hot path: only calls that ask
// for d_weights pay for it.  Unweighted / binned plans count users per tile: integers, exact, written by the main launch.
template <bool FROM_IDS>
int launch_spatial(vet_plan* pl, const vet::SampleSrc& src, int U, int T, double* d_entropy, int32_t* d_assign,
                   double* d_weights, int32_t* d_present, int32_t* d_status, hipStream_t s) {
    const bool exact = d_weights && pl->weighted && !pl->lat[0].binned && !pl->raw_weights;
    int rc = launch_spatial_main<FROM_IDS>(pl, src, U, T, d_entropy, d_assign, exact ? nullptr : d_weights, d_present, d_status, s);
    if (rc || !exact) return rc;
    rc = ensure_exact_weights(pl, s);          // first request for weights: the exact rows of lattice 0 (or not, if too large)
    if (rc) return rc;
    return launch_weights_pass<FROM_IDS>(*pl->wcore, src, U, T, d_weights, s, pl->ctx);
}

// d_weights: the formulation's own histogram of lattice 0 (unweighted / binned plans: the exact integer counts; weighted
// plans only with vet_plan_set_raw_weights)
template <bool FROM_IDS>
int launch_spatial_main(vet_plan* pl, const vet::SampleSrc& src, int U, int T, double* d_entropy, int32_t* d_assign,
                        double* d_weights, int32_t* d_present, int32_t* d_status, hipStream_t s) {
    vet_ctx* c = pl->ctx;
    const int K = (int)pl->lat.size();
    double* ent_k = d_entropy;
    // ---- formulation per lattice (weighted Fibonacci lattices only)
    int form[64];
    if (K > 64) return fail(VET_ERR_UNSUPPORTED, "more than 64 lattices in one plan");
    const bool want_table = table_requested(pl, (long)U * T, U);
    // ---- weighted, integer table formulation over the plan's fused table (one row per distinct direction over all
    // lattices) where the plan allows (ensure_fused)
    if (want_table && pl->weighted) {
        int rc = ensure_fused(pl, s);
        if (rc) return rc;
        if (pl->fused.state == 1) {
            bool launched = false;
            rc = launch_lut_fused<FROM_IDS>(pl, src, U, T, nullptr, 0, 0, 0, 0, d_entropy, d_assign, d_weights, d_present,
                                            d_status, s, &launched);
            if (launched) for (int k = 0; k < K; ++k) pl->lat[k].last_form = F_TABLE;
            if (rc || launched) return rc;
        }
    }
    bool all_table = pl->weighted != 0;
    for (int k = 0; k < K; ++k) {
        form[k] = F_SWEEP;
        if (pl->weighted && !pl->lat[k].binned) {
            int rc = choose_formulation(pl, k, want_table, U, s, &form[k]);
            if (rc) return rc;
        }
        all_table = all_table && (form[k] == F_TABLE || form[k] == F_FTABLE) && form[k] == form[0];
    }
    // ---- weighted, table formulation: every lattice in one launch
    if (all_table) {
        int idx[vet::MAX_LATTICES];
        for (int k = 0; k < K; ++k) idx[k] = k;
        bool launched = false;
        uint32_t* d_list = nullptr;
        int rc = resolve_list_for(pl, idx, K, T, s, &d_list);
        if (rc) return rc;
        if (d_list && K > 1) {
            rc = ensure_ws(c, (size_t)K * T * sizeof(double));
            if (rc) return rc;
        }
        rc = launch_lut<FROM_IDS>(pl, idx, K, src, U, T, nullptr, 0, 0, 0, 0, d_entropy, d_assign, d_weights, d_present,
                                  d_status, s, &launched, d_list);
        if (launched) for (int k = 0; k < K; ++k) pl->lat[k].last_form = form[k];
        if (!rc && launched && d_list)
            rc = resolve_frames<FROM_IDS>(pl, idx, K, src, U, T, d_entropy, d_weights, d_list, (double*)c->ws, s);
        if (rc || launched) return rc;
        for (int k = 0; k < K; ++k) form[k] = sweep_formulation(pl, pl->lat[k], U);   // histograms do not fit the LDS
    }
    if (K > 1) {
        int rc = ensure_ws(c, (size_t)K * T * sizeof(double));
        if (rc) return rc;
        ent_k = (double*)c->ws;
    }
    for (int k = 0; k < K; ++k) {
        const Lattice& L = pl->lat[k];
        if (form[k] == F_TABLE || form[k] == F_FTABLE) {
            bool launched = false;
            uint32_t* d_list = nullptr;
            int rc = resolve_list_for(pl, &k, 1, T, s, &d_list);
            if (rc) return rc;
            rc = launch_lut<FROM_IDS>(pl, &k, 1, src, U, T, nullptr, 0, 0, 0, 0, ent_k + (size_t)k * T,
                                      k == 0 ? d_assign : nullptr, k == 0 ? d_weights : nullptr,
                                      k == 0 ? d_present : nullptr, k == 0 ? d_status : nullptr, s, &launched, d_list);
            if (!rc && launched && d_list)
                rc = resolve_frames<FROM_IDS>(pl, &k, 1, src, U, T, ent_k + (size_t)k * T, d_weights, d_list, nullptr, s);
            if (rc) return rc;
            if (launched) { pl->lat[k].last_form = form[k]; continue; }
            form[k] = sweep_formulation(pl, L, U);
        }
        // binned lattices (naive tiling) are always integer counts; the flag picks the normaliser
        const bool hist_weighted = pl->weighted != 0 && !L.binned;
        const bool precise = hist_weighted && form[k] == F_PRECISE;
        if (hist_weighted) pl->lat[k].last_form = form[k];
        Geometry g;
        int rc = spatial_geometry(c, L.n, U, hist_weighted, &g);
        if (rc) return rc;
        vet::SpatialParams p{};
        p.src = src;
        p.U = U; p.T = T;
        p.dir_unit = pl->d_dir_unit;
        p.nearest = L.d_nearest;
        p.tiles = L.d_tiles;
        p.n = L.n;
        p.cos_cull = pl->cos_cull;
        p.wc.max_ang = pl->max_ang;
        p.wc.inv_max = 1.0 / pl->max_ang;
        p.wc.power = pl->power;
        p.wc.shift = sweep_shift(U);
        p.hmax = L.hmax;
        p.ent_k = ent_k + (size_t)k * T;
        p.assign = k == 0 ? d_assign : nullptr;
        p.weights = k == 0 ? d_weights : nullptr;
        p.present = k == 0 ? d_present : nullptr;
        p.status = k == 0 ? d_status : nullptr;
        p.FPW = g.FPW; p.G = g.G; p.UC = g.UC;
        p.log2_tab = c->d_log2;
        p.full_norm = (L.binned && pl->weighted) ? 1 : 0;
        p.norm_n = L.norm_n;
        p.frame_list = nullptr;
        if (!hist_weighted && !FROM_IDS && U <= 4096 && !c->tune.u_no_lds) {
            // persistent variant with the nearest LUT in LDS: a round is 4096 users = FB frames
            // (2048 pairs with 16-byte loads when U is even, 4096 single users otherwise)
            constexpr int THREADS = 1024;
            const bool pairs = (U & 1) == 0;
            int FB = 4096 / U;
            if (FB > 64) FB = 64;
            auto lds_of = [&](int fb) {
                return (((size_t)pl->n_dirs * 2 + 15) & ~(size_t)15) + (size_t)(U + 1) * 8 +
                       ((((size_t)fb * L.n + 1) & ~(size_t)1) * 4) + (size_t)fb * (THREADS / 64) * 8 + fb * 4 + 16;
            };
            while (FB > 1 && lds_of(FB) > c->lds_max) FB /= 2;      // few users x many tiles: fewer frames per round
            const size_t lds = lds_of(FB);
            if (lds <= c->lds_max) {
                vet::SpatialParams q = p;
                q.FPW = FB;
                const long nblk = ((long)T + FB - 1) / FB;
                long grid = (long)c->n_cu * c->tune.u_wgs_per_cu;
                if (grid > nblk) grid = nblk;
                // even rounds: every persistent workgroup walks the same number of blocks (no tail)
                const long rounds = (nblk + grid - 1) / grid;
                grid = (nblk + rounds - 1) / rounds;
                ProfScope ps(c, s, KID_SPATIAL);
                const void* fn = q.weights ? (pairs ? (const void*)vet::k_spatial_u_lds<true, true> : (const void*)vet::k_spatial_u_lds<true, false>)
                                           : (pairs ? (const void*)vet::k_spatial_u_lds<false, true> : (const void*)vet::k_spatial_u_lds<false, false>);
                void* args[] = {(void*)&q};
                HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(THREADS), args, lds, s));
                HIP_TRY(hipGetLastError());
                continue;
            }
        }
        const int blocks = (T + g.FPW - 1) / g.FPW;
        const void* fn = hist_weighted ? spatial_w_kernel<FROM_IDS>(weight_mode(pl), g.R, precise)
                                       : (const void*)vet::k_spatial_u<FROM_IDS>;
        void* args[] = {(void*)&p};
        ProfScope ps(c, s, KID_SPATIAL);
        HIP_TRY(hipLaunchKernel(fn, dim3(blocks), dim3(g.NW * vet::WAVE), args, g.lds, s));
    }
    if (K > 1) {
        ProfScope ps(c, s, KID_FINALIZE);
        hipLaunchKernelGGL(vet::k_finalize, dim3(grid_for(T, 256, c->n_cu)), dim3(256), 0, s, ent_k, K, (long)T,
                           d_entropy);
        HIP_TRY(hipGetLastError());
    }
    return VET_OK;
}

}  // namespace

__global__ void k_sample_ids(const vet::SampleSrc src, long n, int32_t* __restrict__ out) {
    bool bad = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(vet::sample_dir<false>(src, i, bad), out + i);
}

int weights_pass_ids(const WeightsCore& w, const int32_t* d_ids, int U, int T, double* d_weights, hipStream_t s, vet_ctx* prof) {
    const vet::SampleSrc src{nullptr, nullptr, d_ids, 0, 0, (long)w.n_dirs};
    return launch_weights_pass<true>(w, src, U, T, d_weights, s, prof);
}

int sample_ids(const vet_plan* pl, const double* d_mu, const double* d_mv, long n, int32_t* d_out, hipStream_t s) {
    const vet::SampleSrc src{d_mu, d_mv, nullptr, pl->W, pl->H, (long)pl->n_dirs};
    hipLaunchKernelGGL(k_sample_ids, dim3(grid_for(n, 256, pl->ctx->n_cu)), dim3(256), 0, s, src, n, d_out);
    HIP_TRY(hipGetLastError());
    return VET_OK;
}

int spatial_set_attrs(vet_ctx* c) {
#define ATTR_TRY(fn, bytes) HIP_TRY(hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
    for (int R = 1; R <= 2; ++R) {
        for (int wm = 0; wm < 3; ++wm) {
            ATTR_TRY(spatial_w_kernel<false>(wm, R), c->lds_max);
            ATTR_TRY(spatial_w_kernel<true>(wm, R), c->lds_max);
        }
        ATTR_TRY(spatial_w_kernel<false>(0, R, true), c->lds_max);
        ATTR_TRY(spatial_w_kernel<true>(0, R, true), c->lds_max);
    }
    for (int v = 0; v < 8; ++v) {
        ATTR_TRY(lut_kernel_fused<false>(v & 1, v & 2, v & 4), c->lds_max);
        ATTR_TRY(lut_kernel_fused<true>(v & 1, v & 2, v & 4), c->lds_max);
        ATTR_TRY(lut_kernel_fused<false>(v & 1, v & 2, v & 4, true), c->lds_max);
        ATTR_TRY(lut_kernel_fused<true>(v & 1, v & 2, v & 4, true), c->lds_max);
        ATTR_TRY(lut_kernel<false>(v & 1, v & 2, v & 4), c->lds_max);
        ATTR_TRY(lut_kernel<true>(v & 1, v & 2, v & 4), c->lds_max);
        if (!(v & 2)) {
            ATTR_TRY(lut_kernel<false>(v & 1, false, v & 4, true), c->lds_max);
            ATTR_TRY(lut_kernel<true>(v & 1, false, v & 4, true), c->lds_max);
        }
    }
    ATTR_TRY((vet::k_weights_gather<false, 0>), c->lds_max); ATTR_TRY((vet::k_weights_gather<true, 0>), c->lds_max);
    ATTR_TRY((vet::k_weights_gather<false, 1>), c->lds_max); ATTR_TRY((vet::k_weights_gather<true, 1>), c->lds_max);
    ATTR_TRY((vet::k_weights_gather<false, 2>), c->lds_max); ATTR_TRY((vet::k_weights_gather<true, 2>), c->lds_max);
    ATTR_TRY((vet::k_weights_gather<false, 4>), c->lds_max); ATTR_TRY((vet::k_weights_gather<true, 4>), c->lds_max);
    ATTR_TRY((vet::k_spatial_u_lds<false, true, true>), c->lds_max);
    ATTR_TRY((vet::k_spatial_u_lds<false, false, true>), c->lds_max);
    ATTR_TRY((vet::k_spatial_u_lds<false, true>), c->lds_max);
    ATTR_TRY((vet::k_spatial_u_lds<true, true>), c->lds_max);
    ATTR_TRY((vet::k_spatial_u_lds<false, false>), c->lds_max);
    ATTR_TRY((vet::k_spatial_u_lds<true, false>), c->lds_max);
    ATTR_TRY(vet::k_spatial_u<false>, c->lds_max);
    ATTR_TRY(vet::k_spatial_u<true>, c->lds_max);
#undef ATTR_TRY
    return VET_OK;
}

}  // namespace vh

using namespace vh;

extern "C" {

int vet_spatial_entropy(vet_plan* pl, const double* d_mu, const double* d_mv, int U, int T, double* d_entropy,
                        int32_t* d_assign, double* d_weights, int32_t* d_present, int32_t* d_status, void* stream) {
    int rc = check_run_args(pl, U, T, d_entropy);
    if (rc) return rc;
    if (!pl->grid) return fail(VET_ERR_INVALID, "plan has no pixel grid; use vet_spatial_entropy_ids");
    if (!d_mu || !d_mv) return fail(VET_ERR_INVALID, "d_mu / d_mv is NULL");
    vet::SampleSrc src{d_mu, d_mv, nullptr, pl->W, pl->H, (long)pl->n_dirs};
    return launch_spatial<false>(pl, src, U, T, d_entropy, d_assign, d_weights, d_present, d_status,
                                 stream ? (hipStream_t)stream : pl->ctx->stream);
}

int vet_spatial_entropy_ids(vet_plan* pl, const int32_t* d_ids, int U, int T, double* d_entropy, int32_t* d_assign,
                            double* d_weights, int32_t* d_present, int32_t* d_status, void* stream) {
    int rc = check_run_args(pl, U, T, d_entropy);
    if (rc) return rc;
    if (!d_ids) return fail(VET_ERR_INVALID, "d_ids is NULL");
    vet::SampleSrc src{nullptr, nullptr, d_ids, pl->W, pl->H, (long)pl->n_dirs};
    return launch_spatial<true>(pl, src, U, T, d_entropy, d_assign, d_weights, d_present, d_status,
                                stream ? (hipStream_t)stream : pl->ctx->stream);
}


// ------------------------------------------------------------------------------------------------
// Batch of videos in ONE launch (weighted table formulation): short videos are launch-bound one at
// a time (config 2: 43 us of kernel per call), so their frame blocks share a grid.  Falls back to
// one call per video when the table formulation does not apply.
// the descriptors of a batch -> device, through a slot of the context's blob ring (vh::BatchBlob: neither copy is reused
// while an earlier batch is pending; the blob's destructor marks the slot behind the launches of this call)
static int upload_descriptors(vet_ctx* c, const std::vector<vet::VideoDesc>& desc, hipStream_t s, BatchBlob& blob) {
    const size_t bytes = desc.size() * sizeof(vet::VideoDesc);
    int rc = blob.acquire(c, bytes);
    if (rc) return rc;
    memcpy(blob.host(), desc.data(), bytes);
    return blob.upload(s);
}

// Unweighted (nearest-tile) batch: every video's frame blocks in ONE k_spatial_u_lds launch per lattice; with several
// lattices the per-lattice values go through the workspace and k_finalize_batch forms the means.  Returns launched =
// false when the batch does not fit the kernel (odd shapes, LUT too large for LDS): the caller loops over the videos.
static int batch_unweighted(vet_plan* pl, int n_videos, const vet_video* videos, int32_t* d_status, hipStream_t s, bool* launched) {
    *launched = false;
    vet_ctx* c = pl->ctx;
    const int K = (int)pl->lat.size();
    if (c->tune.u_no_lds) return VET_OK;
    int max_users = 0;
    bool pairs = true;
    long frames = 0;
    for (int v = 0; v < n_videos; ++v) {
        max_users = std::max(max_users, videos[v].n_users);
        pairs = pairs && (videos[v].n_users & 1) == 0;
        frames += videos[v].n_frames;
    }
    if (max_users > 4096) return VET_OK;
    constexpr int THREADS = 1024;
    std::vector<vet::VideoDesc> desc((size_t)n_videos * K);
    std::vector<long> frame0((size_t)n_videos + 1);
    std::vector<double*> outs(n_videos);
    int n_max = 0;
    for (const auto& L : pl->lat) n_max = std::max(n_max, L.n);
    auto lds_of = [&](int fb) {
        return (((size_t)pl->n_dirs * 2 + 15) & ~(size_t)15) + (size_t)(max_users + 1) * 8 +
               ((((size_t)fb * n_max + 1) & ~(size_t)1) * 4) + (size_t)fb * (THREADS / 64) * 8 + fb * 4 + 16;
    };
    int fb_cap = 64;
    while (fb_cap > 1 && lds_of(fb_cap) > c->lds_max) fb_cap /= 2;
    int fb_max = 1, block = 0;
    for (int v = 0; v < n_videos; ++v) {
        const vet_video& x = videos[v];
        int fb = 4096 / x.n_users;
        if (fb > fb_cap) fb = fb_cap;
        if (fb < 1) fb = 1;
        fb_max = std::max(fb_max, fb);
        frame0[v] = v ? frame0[v - 1] + videos[v - 1].n_frames : 0;
        outs[v] = x.d_entropy;
        vet::VideoDesc& d = desc[v];
        d.mu = x.d_mu; d.mv = x.d_mv; d.U = x.n_users; d.T = x.n_frames;
        d.entropy = x.d_entropy; d.assign = x.d_assign; d.present = x.d_present;
        d.FPW = fb; d.UC = 0; d.block0 = block; d.pad_ = 0;
        block += (x.n_frames + fb - 1) / fb;
    }
    frame0[n_videos] = frames;
    const size_t lds = lds_of(fb_max);
    if (lds > c->lds_max) return VET_OK;
    double* ws = nullptr;
    if (K > 1) {
        int rc = ensure_ws(c, (size_t)K * frames * sizeof(double));
        if (rc) return rc;
        ws = (double*)c->ws;
        for (int k = 0; k < K; ++k)
            for (int v = 0; v < n_videos; ++v) {
                vet::VideoDesc& d = desc[(size_t)k * n_videos + v];
                d = desc[v];
                d.entropy = ws + (size_t)k * frames + frame0[v];
                if (k) { d.assign = nullptr; d.present = nullptr; }
            }
        // (desc[v] of lattice 0 was overwritten last: its entropy now points into the workspace too)
    }
    // descriptors, frame offsets and output pointers go to the device as ONE blob whose host copy the context keeps
    // alive (no synchronisation here: the call only enqueues work, include/vet.h)
    const size_t desc_b = desc.size() * sizeof(vet::VideoDesc), f0_b = frame0.size() * 8, outs_b = outs.size() * 8;
    BatchBlob blob;
    int rc = blob.acquire(c, desc_b + f0_b + outs_b);
    if (rc) return rc;
    memcpy(blob.host(), desc.data(), desc_b);
    memcpy((char*)blob.host() + desc_b, frame0.data(), f0_b);
    memcpy((char*)blob.host() + desc_b + f0_b, outs.data(), outs_b);
    char* base = (char*)blob.dev();
    long* d_frame0 = (long*)(base + desc_b);
    double** d_outs = (double**)(base + desc_b + f0_b);
    rc = blob.upload(s);
    if (rc) return rc;
    for (int k = 0; k < K; ++k) {
        const Lattice& L = pl->lat[k];
        vet::SpatialParams q{};
        q.src = vet::SampleSrc{nullptr, nullptr, nullptr, pl->W, pl->H, (long)pl->n_dirs};
        q.U = max_users; q.T = 0;
        q.nearest = L.d_nearest; q.n = L.n; q.hmax = L.hmax;
        q.status = k == 0 ? d_status : nullptr;
        q.FPW = fb_max;
        q.log2_tab = c->d_log2;
        q.full_norm = (L.binned && pl->weighted) ? 1 : 0;
        q.norm_n = L.norm_n;
        q.videos = (const vet::VideoDesc*)base + (size_t)k * n_videos;
        q.n_videos = n_videos; q.n_blocks = block;
        long grid = (long)c->n_cu * c->tune.u_wgs_per_cu;
        if (grid > block) grid = block;
        const long rounds = (block + grid - 1) / grid;
        grid = (block + rounds - 1) / rounds;
        ProfScope ps(c, s, KID_SPATIAL);
        const void* fn = pairs ? (const void*)vet::k_spatial_u_lds<false, true, true> : (const void*)vet::k_spatial_u_lds<false, false, true>;
        void* args[] = {(void*)&q};
        HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(THREADS), args, lds, s));
        HIP_TRY(hipGetLastError());
    }
    if (K > 1) {
        ProfScope ps(c, s, KID_FINALIZE);
        hipLaunchKernelGGL(vet::k_finalize_batch, dim3(grid_for(frames, 256, c->n_cu)), dim3(256), 0, s, (const double*)ws, K, frames,
                           (const long*)d_frame0, (double* const*)d_outs, n_videos);
        HIP_TRY(hipGetLastError());
    }
    *launched = true;
    return VET_OK;
}

int vet_spatial_entropy_batch(vet_plan* pl, int n_videos, const vet_video* videos, int32_t* d_status, void* stream) {
    if (!pl) return fail(VET_ERR_INVALID, "plan is NULL");
    if (n_videos <= 0 || !videos) return fail(VET_ERR_INVALID, "need at least one video");
    if (!pl->grid) return fail(VET_ERR_INVALID, "plan has no pixel grid");
    vet_ctx* c = pl->ctx;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const int K = (int)pl->lat.size();
    long total = 0, total_frames = 0;
    for (int v = 0; v < n_videos; ++v) {
        const vet_video& x = videos[v];
        if (x.n_users <= 0 || x.n_frames <= 0 || !x.d_mu || !x.d_mv || !x.d_entropy)
            return fail(VET_ERR_INVALID, "video %d: bad shape or NULL pointer", v);
        total += (long)x.n_users * x.n_frames;
        total_frames += x.n_frames;
    }
    int max_users = 0;
    for (int v = 0; v < n_videos; ++v) max_users = videos[v].n_users > max_users ? videos[v].n_users : max_users;
    if (!pl->weighted || any_binned(pl)) {
        // nearest-tile counts (unweighted mode, binned lattices): one k_spatial_u_lds launch per lattice for all videos
        bool all_counts = true;
        for (const auto& L : pl->lat) all_counts = all_counts && (!pl->weighted || L.binned);
        if (all_counts) {
            bool launched = false;
            int rc = batch_unweighted(pl, n_videos, videos, d_status, s, &launched);
            if (rc || launched) return rc;
        }
    }
    bool table = table_requested(pl, total, max_users);
    if (table) {
        // the plan's fused table: every video's frame blocks in one k_spatial_lut launch
        int rc = ensure_fused(pl, s);
        if (rc) return rc;
        if (pl->fused.state == 1) {
            const int N = pl->fused.lay.N;
            const bool dedup = !c->tune.no_dedup && max_users >= c->tune.dedup_min_users;
            std::vector<vet::VideoDesc> desc(n_videos);
            int block = 0;
            size_t lds_max = 0;
            bool fits = true;
            for (int v = 0; v < n_videos && fits; ++v) {
                const vet_video& x = videos[v];
                vet::VideoDesc& d = desc[v];
                d.mu = x.d_mu; d.mv = x.d_mv; d.U = x.n_users; d.T = x.n_frames;
                d.entropy = x.d_entropy; d.assign = x.d_assign; d.present = x.d_present;
                const size_t lds = batch_video_geometry(c, d.U, total_frames, N, dedup, &d.FPW, &d.UC);
                if (lds == 0) fits = false;
                d.block0 = block; d.pad_ = 0;
                block += (d.T + d.FPW - 1) / d.FPW;
                lds_max = lds > lds_max ? lds : lds_max;
            }
            if (fits) {
                BatchBlob blob;
                rc = upload_descriptors(c, desc, s, blob);
                if (rc) return rc;
                void* d_desc = blob.dev();
                const vet::SampleSrc src{nullptr, nullptr, nullptr, pl->W, pl->H, (long)pl->n_dirs};
                bool launched = false;
                rc = launch_lut_fused<false>(pl, src, 0, 0, (const vet::VideoDesc*)d_desc, n_videos, block, lds_max, max_users,
                                             nullptr, nullptr, nullptr, nullptr, d_status, s, &launched);
                if (launched) for (int k = 0; k < K; ++k) pl->lat[k].last_form = F_TABLE;
                if (rc || launched) return rc;
            }
        }
    }
    int form0 = F_SWEEP;
    for (int k = 0; k < K && table; ++k) {
        int form = F_SWEEP;
        int rc = choose_formulation(pl, k, true, max_users, s, &form);
        if (rc) return rc;
        if (k == 0) form0 = form;
        table = (form == F_TABLE || form == F_FTABLE) && form == form0;      // one launch: tables of one kind
        table = table && pl->lat[k].markers == 0;                            // marker tables need the per-video resolver
    }
    int n_sum = 0;
    for (int k = 0; k < K; ++k) n_sum += pl->lat[k].n;
    std::vector<vet::VideoDesc> desc;
    size_t lds_max = 0;
    if (table) {
        const bool dedup = (uint64_t)pl->n_rows <= vet::DEDUP_MAX_DIRS && pl->d_dirrec && !c->tune.no_dedup &&
                           max_users >= c->tune.dedup_min_users;
        desc.resize(n_videos);
        int block = 0;
        for (int v = 0; v < n_videos && table; ++v) {
            const vet_video& x = videos[v];
            vet::VideoDesc& d = desc[v];
            d.mu = x.d_mu; d.mv = x.d_mv; d.U = x.n_users; d.T = x.n_frames;
            d.entropy = x.d_entropy; d.assign = x.d_assign; d.present = x.d_present;
            const size_t lds = batch_video_geometry(c, d.U, total_frames, n_sum, dedup, &d.FPW, &d.UC, form0 == F_FTABLE ? 4 : 1,
                                                    (form0 == F_FTABLE && dedup && 2 * pl->n_rows <= 65536) ? (int)((2 * pl->n_rows + 31) / 32) : 0);
            if (lds == 0) table = false;
            d.block0 = block; d.pad_ = 0;
            block += (d.T + d.FPW - 1) / d.FPW;
            lds_max = lds > lds_max ? lds : lds_max;
        }
        if (table) {
            BatchBlob blob;
            int rc = upload_descriptors(c, desc, s, blob);
            if (rc) return rc;
            void* d_desc = blob.dev();
            int idx[vet::MAX_LATTICES];
            for (int k = 0; k < K; ++k) idx[k] = k;
            const vet::SampleSrc src{nullptr, nullptr, nullptr, pl->W, pl->H, (long)pl->n_dirs};
            bool launched = false;
            for (int k = 0; k < K; ++k) pl->lat[k].last_form = form0;
            return launch_lut<false>(pl, idx, K, src, 0, 0, (const vet::VideoDesc*)d_desc, n_videos, block, lds_max, max_users, nullptr,
                                     nullptr, nullptr, nullptr, d_status, s, &launched);
        }
    }
    for (int v = 0; v < n_videos; ++v) {
        const vet_video& x = videos[v];
        int rc = vet_spatial_entropy(pl, x.d_mu, x.d_mv, x.n_users, x.n_frames, x.d_entropy, x.d_assign, nullptr,
                                     x.d_present, d_status, s);
        if (rc) return rc;
    }
    return VET_OK;
}

}  // extern "C"
