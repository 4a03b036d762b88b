// vet_hostapi.hip — host-buffer entry points of the C-ABI (include/vet.h): stage through the context's grow-only device
// buffers, run the device-pointer entry points, copy back (synchronous); device-resident results (vet_result).
// No kernels of its own and no CPU compute path.
#include "vet_host.hpp"

#include <mutex>
#include <vector>

using namespace vh;

extern "C" {

#define POOL(slot, bytes, var) do { int rc_ = pooled(c, slot, bytes, &var); if (rc_) return rc_; } while (0)

struct vet_result {
    int device = 0;                      // the result may outlive its context: only the device id is kept
    void* d[2] = {nullptr, nullptr};     // 0: assign / pairs, 1: weights / srccount (null when the weights are lazy)
    size_t row_bytes[2] = {0, 0};
    int64_t rows = 0;
    // Weighted spatial results whose direction ids [T][U] i32 are not larger than the weight rows [T][n_0] f64 do not store
    // tile_weights: they keep the ids and the plan's shared tables, and a fetched block of weight rows is computed by the
    // weights pass (k_weights_gather over the exact FP64 rows of lattice 0; the precise sweep in weights-only mode where
    // those do not fit) — the reference's values whatever formulation produced the entropy, and no 120 MB weights pass on
    // the hot path of BASELINE config 3 (1024 users, 501 tiles: the two footprints are equal).  Audiences with
    // U * 4 > n_0 * 8 (9000 users on 51 tiles: ids would be 88 x the weights) store the weight rows instead.
    // The weights path (exact rows or precise sweep) was decided once for the plan before the result was created
    // (ensure_exact_weights never revisits it), and `core` is immutable from then on: eager == fetched, same bits.
    bool lazy_weights = false;
    std::shared_ptr<WeightsCore> core;
    int32_t* d_ids = nullptr;
    int U = 0;
    void* d_tmp = nullptr;               // grow-only staging of the fetched weight rows
    size_t tmp_cap = 0;
    std::mutex fetch_mu;                 // d_tmp is one buffer: concurrent fetches of one result take turns
};

static int run_host(vet_plan* pl, bool transition, const double* h_mu, const double* h_mv, const int32_t* h_ids,
                    int U, int T, double* h_entropy, int32_t* h_a, void* h_b, int32_t* h_c, vet_result** keep = nullptr) {
    int rc = check_run_args(pl, U, T, h_entropy);
    if (rc) return rc;
    const bool ids = h_ids != nullptr;
    if (!ids && (!h_mu || !h_mv)) return fail(VET_ERR_INVALID, "need h_mu and h_mv, or h_ids");
    if (!ids && !pl->grid) return fail(VET_ERR_INVALID, "plan has no pixel grid; pass h_ids");
    vet_ctx* c = pl->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t S = (size_t)U * T;
    const int R = transition ? T - 1 : T;
    const int n0 = pl->lat[0].n;
    void *mu = nullptr, *mv = nullptr, *id = nullptr, *ent = nullptr, *a = nullptr, *b = nullptr, *cc = nullptr,
         *st = nullptr;
    if (ids) {
        POOL(0, S * 4, id);
        HIP_TRY(hipMemcpyAsync(id, h_ids, S * 4, hipMemcpyHostToDevice, s));
    } else {
        POOL(0, S * 8, mu);
        POOL(1, S * 8, mv);
        HIP_TRY(hipMemcpyAsync(mu, h_mu, S * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(mv, h_mv, S * 8, hipMemcpyHostToDevice, s));
    }
    POOL(2, (size_t)(R > 0 ? R : 1) * 8, ent);
    const size_t a_bytes = transition ? (size_t)(R > 0 ? R : 0) * U * 2 * 4 : S * 4;
    const size_t b_bytes = transition ? (size_t)(R > 0 ? R : 0) * n0 * 4 : (size_t)T * n0 * 8;
    vet_result* res = nullptr;
    if (keep) {
        // the optional outputs stay in device memory of their own, owned by the result handle
        *keep = nullptr;
        res = new vet_result();
        res->device = c->device;
        res->rows = R > 0 ? R : 0;
        res->row_bytes[0] = transition ? (size_t)U * 2 * 4 : (size_t)U * 4;
        res->row_bytes[1] = transition ? (size_t)n0 * 4 : (size_t)n0 * 8;
        res->lazy_weights = !transition && pl->weighted && !pl->lat[0].binned && !pl->raw_weights &&
                            (size_t)U * 4 <= (size_t)n0 * 8;
        bool ok = hipMalloc(&res->d[0], a_bytes ? a_bytes : 8) == hipSuccess;
        if (res->lazy_weights) {
            rc = ensure_exact_weights(pl, s);      // the rows a fetched block gathers (precise sweep if they do not fit)
            if (rc) { vet_result_free(res); return rc; }
        }
        if (ok && res->lazy_weights) {
            res->core = pl->wcore; res->U = U;
            ok = hipMalloc((void**)&res->d_ids, S * 4 ? S * 4 : 8) == hipSuccess;
        } else if (ok) {
            ok = hipMalloc(&res->d[1], b_bytes ? b_bytes : 8) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            vet_result_free(res);
            return fail(VET_ERR_DEVICE, "out of device memory for the resident outputs (%zu B)", a_bytes + b_bytes);
        }
        a = res->d[0]; b = res->d[1];
    } else {
        if (h_a) POOL(3, a_bytes, a);
        if (h_b) POOL(4, b_bytes, b);
    }
    struct Guard { vet_result* r; ~Guard() { if (r) vet_result_free(r); } } guard{res};
    POOL(5, (size_t)(R > 0 ? R : 1) * 4, cc);
    POOL(6, 8, st);
    HIP_TRY(hipMemsetAsync(st, 0, 8, s));
    if (transition) {
        rc = ids ? vet_transition_entropy_ids(pl, (const int32_t*)id, U, T, (double*)ent, (int32_t*)a, (int32_t*)b,
                                              (int32_t*)cc, (int32_t*)st, s)
                 : vet_transition_entropy(pl, (const double*)mu, (const double*)mv, U, T, (double*)ent,
                                          (int32_t*)a, (int32_t*)b, (int32_t*)cc, (int32_t*)st, s);
    } else {
        rc = ids ? vet_spatial_entropy_ids(pl, (const int32_t*)id, U, T, (double*)ent, (int32_t*)a, (double*)b,
                                           (int32_t*)cc, (int32_t*)st, s)
                 : vet_spatial_entropy(pl, (const double*)mu, (const double*)mv, U, T, (double*)ent, (int32_t*)a,
                                       (double*)b, (int32_t*)cc, (int32_t*)st, s);
    }
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    if (res && res->lazy_weights) {
        // the direction ids the weight rows are recomputed from on fetch
        if (ids) HIP_TRY(hipMemcpyAsync(res->d_ids, id, S * 4, hipMemcpyDeviceToDevice, s));
        else {
            rc = sample_ids(pl, (const double*)mu, (const double*)mv, (long)S, res->d_ids, s);
            if (rc) { (void)hipStreamSynchronize(s); return rc; }
        }
    }
    int32_t status[2] = {0, 0};
    if (R > 0) {
        HIP_TRY(hipMemcpyAsync(h_entropy, ent, (size_t)R * 8, hipMemcpyDeviceToHost, s));
        if (h_a) HIP_TRY(hipMemcpyAsync(h_a, a, a_bytes, hipMemcpyDeviceToHost, s));
        if (h_b) HIP_TRY(hipMemcpyAsync(h_b, b, b_bytes, hipMemcpyDeviceToHost, s));
        if (h_c) HIP_TRY(hipMemcpyAsync(h_c, cc, (size_t)R * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(status, st, 8, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (keep) { *keep = res; guard.r = nullptr; }       // outputs are written also when a status word is set
    if (status[0]) return fail(VET_ERR_RANGE, "Normalized coordinates must be between 0 and 1 (%d samples)", status[0]);
    if (status[1])
        return fail(VET_ERR_EMPTY, transition ? "%d frame pair(s) without a user present in both frames"
                                              : "%d frame(s) without any user (Empty vector dictionary)", status[1]);
    return VET_OK;
}

int vet_spatial_entropy_host_resident(vet_plan* pl, const double* h_mu, const double* h_mv, const int32_t* h_ids, int U,
                                      int T, double* h_entropy, int32_t* h_present, vet_result** out) {
    if (!out) return fail(VET_ERR_INVALID, "out is NULL");
    return run_host(pl, false, h_mu, h_mv, h_ids, U, T, h_entropy, nullptr, nullptr, h_present, out);
}

int vet_transition_entropy_host_resident(vet_plan* pl, const double* h_mu, const double* h_mv, const int32_t* h_ids,
                                         int U, int T, double* h_entropy, int32_t* h_common, vet_result** out) {
    if (!out) return fail(VET_ERR_INVALID, "out is NULL");
    return run_host(pl, true, h_mu, h_mv, h_ids, U, T, h_entropy, nullptr, nullptr, h_common, out);
}

int vet_result_fetch(vet_result* r, int which, int64_t row0, int64_t n_rows, void* h_dst) {
    if (!r || !h_dst) return fail(VET_ERR_INVALID, "result or destination is NULL");
    if (which < 0 || which > 1) return fail(VET_ERR_INVALID, "which must be 0 (assignments / pairs) or 1 (weights / source counts)");
    if (row0 < 0 || n_rows < 0 || row0 + n_rows > r->rows)
        return fail(VET_ERR_INVALID, "rows [%lld, %lld) outside the result's %lld rows", (long long)row0,
                    (long long)(row0 + n_rows), (long long)r->rows);
    if (n_rows == 0) return VET_OK;
    HIP_TRY(hipSetDevice(r->device));
    if (which == 1 && r->lazy_weights) {
        // tile_weights rows [row0, row0 + n_rows): computed now, from the resident direction ids (null stream: the
        // call that made the result has synchronised its stream, and the result may have outlived its context)
        const size_t bytes = (size_t)n_rows * r->row_bytes[1];
        std::lock_guard<std::mutex> lock(r->fetch_mu);
        if (r->tmp_cap < bytes) {
            if (r->d_tmp) { HIP_TRY(hipFree(r->d_tmp)); r->d_tmp = nullptr; r->tmp_cap = 0; }
            HIP_TRY(hipMalloc(&r->d_tmp, bytes));
            r->tmp_cap = bytes;
        }
        int rc = weights_pass_ids(*r->core, r->d_ids + (size_t)row0 * r->U, r->U, (int)n_rows, (double*)r->d_tmp, nullptr, nullptr);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(h_dst, r->d_tmp, bytes, hipMemcpyDeviceToHost));
        return VET_OK;
    }
    HIP_TRY(hipMemcpy(h_dst, (const char*)r->d[which] + (size_t)row0 * r->row_bytes[which], (size_t)n_rows * r->row_bytes[which],
                      hipMemcpyDeviceToHost));
    return VET_OK;
}

int vet_result_free(vet_result* r) {
    if (!r) return VET_OK;
    (void)hipSetDevice(r->device);
    for (void* q : r->d) if (q) (void)hipFree(q);
    if (r->d_ids) (void)hipFree(r->d_ids);
    if (r->d_tmp) (void)hipFree(r->d_tmp);
    delete r;
    return VET_OK;
}

int vet_spatial_entropy_host(vet_plan* pl, const double* h_mu, const double* h_mv, const int32_t* h_ids, int U,
                             int T, double* h_entropy, int32_t* h_assign, double* h_weights, int32_t* h_present) {
    return run_host(pl, false, h_mu, h_mv, h_ids, U, T, h_entropy, h_assign, h_weights, h_present);
}

int vet_transition_entropy_host(vet_plan* pl, const double* h_mu, const double* h_mv, const int32_t* h_ids, int U,
                                int T, double* h_entropy, int32_t* h_pairs, int32_t* h_srccount,
                                int32_t* h_common) {
    return run_host(pl, true, h_mu, h_mv, h_ids, U, T, h_entropy, h_pairs, h_srccount, h_common);
}

// Concatenated host buffers: video v's samples start at element sum_{w<v} U_w*T_w of h_mu / h_mv /
// h_assign and its entropies at sum_{w<v} T_w of h_entropy / h_present.  Two H2D copies, one launch
// (when the table formulation applies), two or three D2H copies.
int vet_spatial_entropy_batch_host(vet_plan* pl, int n_videos, const int* n_users, const int* n_frames,
                                   const double* h_mu, const double* h_mv, double* h_entropy, int32_t* h_assign,
                                   int32_t* h_present) {
    if (!pl || n_videos <= 0 || !n_users || !n_frames || !h_mu || !h_mv || !h_entropy)
        return fail(VET_ERR_INVALID, "bad batch arguments");
    vet_ctx* c = pl->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    size_t S = 0, R = 0;
    for (int v = 0; v < n_videos; ++v) {
        if (n_users[v] <= 0 || n_frames[v] <= 0) return fail(VET_ERR_INVALID, "video %d: bad shape", v);
        S += (size_t)n_users[v] * n_frames[v];
        R += (size_t)n_frames[v];
    }
    void *mu = nullptr, *mv = nullptr, *ent = nullptr, *as = nullptr, *pr = nullptr, *st = nullptr;
    POOL(0, S * 8, mu); POOL(1, S * 8, mv); POOL(2, R * 8, ent);
    if (h_assign) POOL(3, S * 4, as);
    if (h_present) POOL(5, R * 4, pr);
    POOL(6, 8, st);
    HIP_TRY(hipMemcpyAsync(mu, h_mu, S * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(mv, h_mv, S * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(st, 0, 8, s));
    std::vector<vet_video> vids(n_videos);
    size_t so = 0, ro = 0;
    for (int v = 0; v < n_videos; ++v) {
        vids[v].d_mu = (const double*)mu + so; vids[v].d_mv = (const double*)mv + so;
        vids[v].n_users = n_users[v]; vids[v].n_frames = n_frames[v];
        vids[v].d_entropy = (double*)ent + ro;
        vids[v].d_assign = as ? (int32_t*)as + so : nullptr;
        vids[v].d_present = pr ? (int32_t*)pr + ro : nullptr;
        so += (size_t)n_users[v] * n_frames[v];
        ro += (size_t)n_frames[v];
    }
    int rc = vet_spatial_entropy_batch(pl, n_videos, vids.data(), (int32_t*)st, s);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    int32_t status[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h_entropy, ent, R * 8, hipMemcpyDeviceToHost, s));
    if (h_assign) HIP_TRY(hipMemcpyAsync(h_assign, as, S * 4, hipMemcpyDeviceToHost, s));
    if (h_present) HIP_TRY(hipMemcpyAsync(h_present, pr, R * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(status, st, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (status[0]) return fail(VET_ERR_RANGE, "Normalized coordinates must be between 0 and 1 (%d samples)", status[0]);
    if (status[1]) return fail(VET_ERR_EMPTY, "%d frame(s) without any user (Empty vector dictionary)", status[1]);
    return VET_OK;
}

// Transition batch with concatenated host buffers: video v's samples start at element sum_{w<v} U_w*T_w of h_mu / h_mv,
// its rows at sum_{w<v} (T_w-1) of h_entropy / h_common and its pairs at 2 * sum_{w<v} U_w*(T_w-1) of h_pairs.  Synchronous.
int vet_transition_entropy_batch_host(vet_plan* pl, int n_videos, const int* n_users, const int* n_frames,
                                      const double* h_mu, const double* h_mv, double* h_entropy, int32_t* h_pairs,
                                      int32_t* h_common) {
    if (!pl || n_videos <= 0 || !n_users || !n_frames || !h_mu || !h_mv || !h_entropy)
        return fail(VET_ERR_INVALID, "bad batch arguments");
    vet_ctx* c = pl->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    size_t S = 0, R = 0, P = 0;
    for (int v = 0; v < n_videos; ++v) {
        if (n_users[v] <= 0 || n_frames[v] <= 1) return fail(VET_ERR_INVALID, "video %d: need users and at least two frames", v);
        S += (size_t)n_users[v] * n_frames[v];
        R += (size_t)n_frames[v] - 1;
        P += (size_t)n_users[v] * (n_frames[v] - 1) * 2;
    }
    void *mu = nullptr, *mv = nullptr, *ent = nullptr, *pr = nullptr, *cm = nullptr, *st = nullptr;
    POOL(0, S * 8, mu); POOL(1, S * 8, mv); POOL(2, R * 8, ent);
    if (h_pairs) POOL(3, P * 4, pr);
    if (h_common) POOL(5, R * 4, cm);
    POOL(6, 8, st);
    HIP_TRY(hipMemcpyAsync(mu, h_mu, S * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(mv, h_mv, S * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(st, 0, 8, s));
    std::vector<vet_video> vids(n_videos);
    size_t so = 0, ro = 0, po = 0;
    for (int v = 0; v < n_videos; ++v) {
        vids[v].d_mu = (const double*)mu + so; vids[v].d_mv = (const double*)mv + so;
        vids[v].n_users = n_users[v]; vids[v].n_frames = n_frames[v];
        vids[v].d_entropy = (double*)ent + ro;
        vids[v].d_assign = pr ? (int32_t*)pr + po : nullptr;
        vids[v].d_present = cm ? (int32_t*)cm + ro : nullptr;
        so += (size_t)n_users[v] * n_frames[v];
        ro += (size_t)n_frames[v] - 1;
        po += (size_t)n_users[v] * (n_frames[v] - 1) * 2;
    }
    int rc = vet_transition_entropy_batch(pl, n_videos, vids.data(), (int32_t*)st, s);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    int32_t status[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h_entropy, ent, R * 8, hipMemcpyDeviceToHost, s));
    if (h_pairs) HIP_TRY(hipMemcpyAsync(h_pairs, pr, P * 4, hipMemcpyDeviceToHost, s));
    if (h_common) HIP_TRY(hipMemcpyAsync(h_common, cm, R * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(status, st, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (status[0]) return fail(VET_ERR_RANGE, "Normalized coordinates must be between 0 and 1 (%d samples)", status[0]);
    if (status[1]) return fail(VET_ERR_EMPTY, "%d frame pair(s) without a user present in both frames", status[1]);
    return VET_OK;
}


}  // extern "C"
