// vet_transition.hip — launch logic of the transition-entropy kernels behind vet_transition_entropy* (include/vet.h):
// k_transition_run (everything in LDS, up to 4096 users) / k_transition_any (bucket hash in global scratch), single
// videos and batches.  No CPU compute path; nothing here reads the environment.
#include "vet_host.hpp"
#include "vet_finalize.hpp"
#include "vet_transition.hpp"

#include <algorithm>
#include <cstring>

namespace vh {

namespace {

template <bool FROM_IDS>
const void* transition_run_kernel(int upt, bool exact, int threads) {
    // the default workgroup of up to 512 users (128 threads) has its size compiled in
#define VET_PICK(N) if (upt == N) return threads == 128 ? (exact ? (const void*)vet::k_transition_run<FROM_IDS, N, true, 128> : (const void*)vet::k_transition_run<FROM_IDS, N, false, 128>) \
                                                        : (exact ? (const void*)vet::k_transition_run<FROM_IDS, N, true, 0> : (const void*)vet::k_transition_run<FROM_IDS, N, false, 0>)
    VET_PICK(1); VET_PICK(2); VET_PICK(4); VET_PICK(8);
#undef VET_PICK
    return nullptr;
}

const void* transition_batch_kernel(int upt, bool exact) {       // batched launches: the workgroup size is read from blockDim
#define VET_PICK(N) if (upt == N) return exact ? (const void*)vet::k_transition_run<false, N, true, 0, true> : (const void*)vet::k_transition_run<false, N, false, 0, true>
    VET_PICK(1); VET_PICK(2); VET_PICK(4); VET_PICK(8);
#undef VET_PICK
    return nullptr;
}

template <bool FROM_IDS>
int launch_transition(vet_plan* pl, const vet::SampleSrc& src, int U, int T, double* d_entropy, int32_t* d_pairs,
                      int32_t* d_srccount, int32_t* d_common, int32_t* d_status, hipStream_t s) {
    vet_ctx* c = pl->ctx;
    const int K = (int)pl->lat.size();
    const int R = T - 1;
    if (R <= 0) return VET_OK;
    double* ent_k = d_entropy;
    if (K > 1) {
        int rc = ensure_ws(c, (size_t)K * R * sizeof(double));
        if (rc) return rc;
        ent_k = (double*)c->ws;
    }
    int HS = 64, lg = 6;
    const int hs_pct = c->tune.t_hs_pct;      // bucket-hash slots per 100 users
    while ((long)HS * 100 < (long)hs_pct * U) { HS <<= 1; ++lg; }
    const size_t U4 = ((size_t)U + 3) & ~(size_t)3;
    for (int k = 0; k < K; ++k) {
        const Lattice& L = pl->lat[k];
        const size_t n4 = ((size_t)L.n + 3) & ~(size_t)3;
        const size_t lds_tiles = 2 * 20 * 8 + 4 * n4 * 4;
        const size_t lds_run = lds_tiles + (size_t)3 * HS * 4 + ((size_t)U + 2) * 8;     // + log2(k), k <= U
        const size_t lds_cap = 160 * 1024 - 512;     // a single workgroup may take the whole LDS
        if (lds_tiles > lds_cap)
            return fail(VET_ERR_UNSUPPORTED, "transition kernel: %d tiles need %zu B of LDS (max %zu)", L.n, lds_tiles, lds_cap);
        vet::TransParams p{};
        p.src = src;
        p.U = U; p.T = T;
        p.nearest = L.d_nearest;
        p.n = L.n;
        p.hmax = L.hmax;
        p.ent_k = ent_k + (size_t)k * R;
        p.pairs = k == 0 ? d_pairs : nullptr;
        p.srccount = k == 0 ? d_srccount : nullptr;
        p.common = k == 0 ? d_common : nullptr;
        p.status = k == 0 ? d_status : nullptr;
        p.HS = HS; p.hs_shift = 32 - lg;
        p.log2_tab = c->d_log2;
        p.scratch = nullptr;
        p.run_q = 0; p.run_r = 0;
        ProfScope ps(c, s, KID_TRANSITION);
        if (U <= 4096 && lds_run <= lds_cap && !c->tune.t_global) {
            // persistent workgroups over contiguous runs of rows (k_transition_run); users per thread 1, 2, 4 or 8:
            // two waves per row up to 512 users (measured: 46 us vs 51 us with four, profiles/r02)
            int threads = U <= 512 ? 128 : (U <= 2048 ? 512 : 1024);
            if (c->tune.t_threads) threads = c->tune.t_threads;
            int upt = (U + threads - 1) / threads;
            upt = upt <= 1 ? 1 : (upt <= 2 ? 2 : (upt <= 4 ? 4 : 8));
            while ((long)upt * threads < U) threads *= 2;
            long per_cu = (long)(lds_cap / lds_run);
            const long by_waves = 32 / (threads / 64);
            if (per_cu > by_waves) per_cu = by_waves;
            per_cu = c->tune.t_wgs_per_cu ? c->tune.t_wgs_per_cu : (per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu));
            long grid = (long)c->n_cu * per_cu;
            if (grid > R) grid = R;
            p.run_q = (int)(R / grid); p.run_r = (int)(R % grid);
            const void* fn = transition_run_kernel<FROM_IDS>(upt, (long)upt * threads == U, threads);
            void* args[] = {(void*)&p};
            HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(threads), args, lds_run, s));
        } else if (U < (1 << 19) && L.n <= vet::TRANS_BIG_MAX_TILES && !c->tune.t_global) {
            // more users than the register kernel holds: the bucket hash stays in LDS, the row is cut into ranges of
            // source tiles whose buckets fit it (k_transition_big); one persistent workgroup per CU
            long grid = c->n_cu;
            if (grid > R) grid = R;
            void* scratch = nullptr;
            int rc = pooled(c, 8, U4 * 4 * (size_t)grid, &scratch);
            if (rc) return rc;
            p.scratch = (uint32_t*)scratch;
            void* args[] = {(void*)&p};
            const void* fn = FROM_IDS ? (const void*)vet::k_transition_big<true> : (const void*)vet::k_transition_big<false>;
            HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(vet::TRANS_BIG_THREADS), args, vet::trans_big_lds_bytes(L.n), s));
        } else {
            // lattices of thousands of tiles (or 2^19 users): bucket hash and per-user words in global scratch, one slice
            // per persistent workgroup (the reference accepts any number of users, entropy_utils.py:259-332)
            const size_t slice = ((size_t)3 * HS + 2 * U4) * 4;
            long grid = (long)c->n_cu * 2;
            if (grid > R) grid = R;
            while (grid > 1 && slice * (size_t)grid > ((size_t)2 << 30)) grid /= 2;
            void* scratch = nullptr;
            int rc = pooled(c, 8, slice * (size_t)grid, &scratch);
            if (rc) return rc;
            p.scratch = (uint32_t*)scratch;
            hipLaunchKernelGGL((vet::k_transition_any<FROM_IDS>), dim3((unsigned)grid), dim3(1024), lds_tiles, s, p);
        }
        HIP_TRY(hipGetLastError());
    }
    if (K > 1) {
        ProfScope ps(c, s, KID_FINALIZE);
        hipLaunchKernelGGL(vet::k_finalize, dim3(grid_for(R, 256, c->n_cu)), dim3(256), 0, s, ent_k, K, (long)R,
                           d_entropy);
        HIP_TRY(hipGetLastError());
    }
    return VET_OK;
}

}  // namespace

int transition_set_attrs(vet_ctx* c) {
    std::vector<const void*> tk = {(const void*)vet::k_transition_any<false>, (const void*)vet::k_transition_any<true>,
                                   (const void*)vet::k_transition_big<false>, (const void*)vet::k_transition_big<true>};
    for (int upt : {1, 2, 4, 8})
        for (int ex = 0; ex < 2; ++ex)
            for (int threads : {128, 0}) {
                tk.push_back(transition_run_kernel<false>(upt, ex != 0, threads));
                tk.push_back(transition_run_kernel<true>(upt, ex != 0, threads));
            }
    for (int upt : {1, 2, 4, 8})
        for (int ex = 0; ex < 2; ++ex) tk.push_back(transition_batch_kernel(upt, ex != 0));
    for (const void* f : tk) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWholeLds));
    return VET_OK;
}

}  // namespace vh

using namespace vh;

extern "C" {

int vet_transition_entropy(vet_plan* pl, const double* d_mu, const double* d_mv, int U, int T, double* d_entropy,
                           int32_t* d_pairs, int32_t* d_srccount, int32_t* d_common, int32_t* d_status,
                           void* stream) {
    int rc = check_run_args(pl, U, T, d_entropy);
    if (rc) return rc;
    if (!pl->grid) return fail(VET_ERR_INVALID, "plan has no pixel grid; use vet_transition_entropy_ids");
    if (!d_mu || !d_mv) return fail(VET_ERR_INVALID, "d_mu / d_mv is NULL");
    vet::SampleSrc src{d_mu, d_mv, nullptr, pl->W, pl->H, (long)pl->n_dirs};
    return launch_transition<false>(pl, src, U, T, d_entropy, d_pairs, d_srccount, d_common, d_status,
                                    stream ? (hipStream_t)stream : pl->ctx->stream);
}

int vet_transition_entropy_ids(vet_plan* pl, const int32_t* d_ids, int U, int T, double* d_entropy,
                               int32_t* d_pairs, int32_t* d_srccount, int32_t* d_common, int32_t* d_status,
                               void* stream) {
    int rc = check_run_args(pl, U, T, d_entropy);
    if (rc) return rc;
    if (!d_ids) return fail(VET_ERR_INVALID, "d_ids is NULL");
    vet::SampleSrc src{nullptr, nullptr, d_ids, pl->W, pl->H, (long)pl->n_dirs};
    return launch_transition<true>(pl, src, U, T, d_entropy, d_pairs, d_srccount, d_common, d_status,
                                   stream ? (hipStream_t)stream : pl->ctx->stream);
}

// Transition mode over a batch of videos: ONE k_transition_run launch per lattice, every video with its own
// workgroups (in proportion to its rows).  d_entropy [T-1], d_assign = pairs [(T-1)*U*2] (nullable), d_present =
// users present in both frames [T-1] (nullable).  Batches that do not fit the LDS kernel run video by video.
int vet_transition_entropy_batch(vet_plan* pl, int n_videos, const vet_video* videos, int32_t* d_status, void* stream) {
    if (!pl) return fail(VET_ERR_INVALID, "plan is NULL");
    if (n_videos <= 0 || !videos) return fail(VET_ERR_INVALID, "need at least one video");
    if (!pl->grid) return fail(VET_ERR_INVALID, "plan has no pixel grid");
    vet_ctx* c = pl->ctx;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const int K = (int)pl->lat.size();
    int max_users = 0, min_users = 1 << 30;
    long rows = 0;
    for (int v = 0; v < n_videos; ++v) {
        const vet_video& x = videos[v];
        if (x.n_users <= 0 || x.n_frames <= 0 || !x.d_mu || !x.d_mv || !x.d_entropy)
            return fail(VET_ERR_INVALID, "video %d: bad shape or NULL pointer", v);
        max_users = std::max(max_users, x.n_users);
        min_users = std::min(min_users, x.n_users);
        rows += x.n_frames > 1 ? x.n_frames - 1 : 0;
    }
    int HS = 64, lg = 6;
    while (HS < 2 * max_users) { HS <<= 1; ++lg; }
    size_t n4_max = 0;
    for (const auto& L : pl->lat) n4_max = std::max(n4_max, ((size_t)L.n + 3) & ~(size_t)3);
    const size_t lds_cap = 160 * 1024 - 512;
    const size_t lds_run = 2 * 20 * 8 + 4 * n4_max * 4 + (size_t)3 * HS * 4 + ((size_t)max_users + 2) * 8;
    bool one_launch = rows > 0 && max_users <= 4096 && lds_run <= lds_cap && !c->tune.t_global;
    for (int v = 0; v < n_videos; ++v) one_launch = one_launch && videos[v].n_frames > 1;
    if (!one_launch) {
        for (int v = 0; v < n_videos; ++v) {
            const vet_video& x = videos[v];
            int rc = vet_transition_entropy(pl, x.d_mu, x.d_mv, x.n_users, x.n_frames, x.d_entropy, x.d_assign, nullptr,
                                            x.d_present, d_status, s);
            if (rc) return rc;
        }
        return VET_OK;
    }
    int threads = max_users <= 512 ? 128 : (max_users <= 2048 ? 512 : 1024);
    if (c->tune.t_threads) threads = c->tune.t_threads;
    int upt = (max_users + threads - 1) / threads;
    upt = upt <= 1 ? 1 : (upt <= 2 ? 2 : (upt <= 4 ? 4 : 8));
    while ((long)upt * threads < max_users) threads *= 2;
    long per_cu = (long)(lds_cap / lds_run);
    const long by_waves = 32 / (threads / 64);
    if (per_cu > by_waves) per_cu = by_waves;
    per_cu = c->tune.t_wgs_per_cu ? c->tune.t_wgs_per_cu : (per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu));
    long grid_want = (long)c->n_cu * per_cu;
    if (grid_want > rows) grid_want = rows;
    if (grid_want < n_videos) grid_want = n_videos;
    std::vector<vet::TransVideo> tv((size_t)n_videos * K);
    std::vector<long> row0((size_t)n_videos + 1);
    std::vector<double*> outs(n_videos);
    int wg = 0;
    for (int v = 0; v < n_videos; ++v) {
        const vet_video& x = videos[v];
        const long R = x.n_frames - 1;
        long n_wgs = (grid_want * R + rows / 2) / rows;
        if (n_wgs < 1) n_wgs = 1;
        if (n_wgs > R) n_wgs = R;
        row0[v] = v ? row0[v - 1] + (videos[v - 1].n_frames - 1) : 0;
        outs[v] = x.d_entropy;
        vet::TransVideo& d = tv[v];
        d.mu = x.d_mu; d.mv = x.d_mv; d.U = x.n_users; d.T = x.n_frames;
        d.ent = x.d_entropy; d.pairs = x.d_assign; d.common = x.d_present;
        d.wg0 = wg; d.n_wgs = (int)n_wgs; d.run_q = (int)(R / n_wgs); d.run_r = (int)(R % n_wgs);
        wg += (int)n_wgs;
    }
    row0[n_videos] = rows;
    double* ws = nullptr;
    if (K > 1) {
        int rc = ensure_ws(c, (size_t)K * rows * sizeof(double));
        if (rc) return rc;
        ws = (double*)c->ws;
        for (int k = K - 1; k >= 0; --k)
            for (int v = 0; v < n_videos; ++v) {
                vet::TransVideo& d = tv[(size_t)k * n_videos + v];
                d = tv[v];
                d.ent = ws + (size_t)k * rows + row0[v];
                if (k) { d.pairs = nullptr; d.common = nullptr; }
            }
    }
    // descriptors, row offsets and output pointers go to the device as ONE blob whose host copy the context keeps alive
    // (no synchronisation here: the call only enqueues work, include/vet.h)
    const size_t tv_b = tv.size() * sizeof(vet::TransVideo), r0_b = row0.size() * 8, outs_b = outs.size() * 8;
    BatchBlob blob;
    int rc = blob.acquire(c, tv_b + r0_b + outs_b);
    if (rc) return rc;
    memcpy(blob.host(), tv.data(), tv_b);
    memcpy((char*)blob.host() + tv_b, row0.data(), r0_b);
    memcpy((char*)blob.host() + tv_b + r0_b, outs.data(), outs_b);
    char* base = (char*)blob.dev();
    long* d_row0 = (long*)(base + tv_b);
    double** d_outs = (double**)(base + tv_b + r0_b);
    rc = blob.upload(s);
    if (rc) return rc;
    const bool exact = min_users == max_users && (long)upt * threads == max_users;
    for (int k = 0; k < K; ++k) {
        const Lattice& L = pl->lat[k];
        vet::TransParams p{};
        p.src = vet::SampleSrc{nullptr, nullptr, nullptr, pl->W, pl->H, (long)pl->n_dirs};
        p.U = max_users; p.T = 0;
        p.nearest = L.d_nearest; p.n = L.n; p.hmax = L.hmax;
        p.status = k == 0 ? d_status : nullptr;
        p.HS = HS; p.hs_shift = 32 - lg;
        p.log2_tab = c->d_log2;
        p.videos = (const vet::TransVideo*)base + (size_t)k * n_videos; p.n_videos = n_videos;
        const size_t n4 = ((size_t)L.n + 3) & ~(size_t)3;
        const size_t lds = 2 * 20 * 8 + 4 * n4 * 4 + (size_t)3 * HS * 4 + ((size_t)max_users + 2) * 8;
        ProfScope ps(c, s, KID_TRANSITION);
        const void* fn = transition_batch_kernel(upt, exact);
        void* args[] = {(void*)&p};
        HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)wg), dim3(threads), args, lds, s));
        HIP_TRY(hipGetLastError());
    }
    if (K > 1) {
        ProfScope ps(c, s, KID_FINALIZE);
        hipLaunchKernelGGL(vet::k_finalize_batch, dim3(grid_for(rows, 256, c->n_cu)), dim3(256), 0, s, (const double*)ws, K, rows,
                           (const long*)d_row0, (double* const*)d_outs, n_videos);
        HIP_TRY(hipGetLastError());
    }
    return VET_OK;
}

}  // extern "C"
