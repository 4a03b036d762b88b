// vet_api.hip — C-ABI (include/vet.h) over the gfx950 kernels in vet_kernels.hpp.
//
// Host responsibilities: device tables of a plan (direction table, unit lattices, one
// nearest-tile LUT per lattice), launch geometry, a grow-only workspace (no hipMalloc in the
// steady state), optional hipEvent timing per kernel.  There is no CPU compute path here.
#include "../../include/vet.h"
#include "vet_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <algorithm>
#include <array>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(VET_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t b) { return hipMalloc(&p, b ? b : 8); }
};

enum { KID_GRID = 0, KID_NEAREST = 1, KID_SPATIAL = 2, KID_TRANSITION = 3, KID_FINALIZE = 4, KID_WTAB = 5, KID_COUNT = 6 };
const char* const kKernelNames[KID_COUNT] = {"k_grid_dirs", "k_nearest_lut", "k_spatial", "k_transition",
                                             "k_finalize", "k_wtab"};

struct EventPair {
    int kid;
    hipEvent_t a, b;
};

}  // namespace

struct vet_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    size_t lds_max = 64 * 1024;
    // grow-only workspace for per-lattice entropies + status words of the host variants
    void* ws = nullptr;
    size_t ws_bytes = 0;
    double* d_log2 = nullptr;      // log2(k), k = 0..4096
    bool attrs_set = false;        // dynamic-LDS limits of the run kernels raised (first plan)
    // grow-only device staging buffers of the host-buffer entry points (no hipMalloc per call)
    void* pool[10] = {};
    size_t pool_cap[10] = {};
    std::vector<vet::VideoDesc> batch_desc;   // host copy of the last batch's descriptors (kept alive)
    // profiling
    bool profiling = false;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> free_events;
    double prof_ms[KID_COUNT] = {};
    int64_t prof_n[KID_COUNT] = {};
};

extern "C" {
static int pooled(vet_ctx* c, int slot, size_t bytes, void** out);   // slot-indexed grow-only device buffer
}

struct Lattice {
    int n = 0;
    double* d_tiles = nullptr;     // [n][3] unit
    std::vector<double> h_unit;    // host copy of the unit tiles
    uint16_t* d_nearest = nullptr; // [n_dirs]
    double hmax = 0.0;
    // direction weight table (ELL), built on first use when the video has more samples than the
    // plan has directions
    uint32_t* d_tab_w = nullptr;   // [n_dirs+1][stride] u32 mantissas (block floating point per row)
    uint16_t* d_tab_i = nullptr;   // [n_dirs+1][stride]
    uint32_t* d_tab_meta = nullptr;// [n_dirs+1] entries in use | row shift << 16
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

struct vet_plan {
    vet_ctx* ctx = nullptr;
    int W = 0, H = 0;
    bool grid = false;
    int64_t n_dirs = 0;
    double* d_dir_raw = nullptr;
    double* d_dir_unit = nullptr;
    std::vector<Lattice> lat;
    double fov = 120.0, max_ang = 0.0, power = 2.0;
    int weighted = 1;
    double cos_cull = 0.0;
    int table_policy = 0;          // 0 by call size, 1 table whenever it is inside the contract, -1 never
    uint32_t* d_alias = nullptr;   // [n_dirs] direction id -> table row (dense) | mirrored << 31 (ensure_alias)
    bool mirror = false;           // rows are shared between mirror-image directions
    uint2* d_dirrec = nullptr;     // [n_dirs] alias | nearest tile | lattice-0 row meta (k_dirrec), dedup-capable plans
    std::vector<uint32_t> h_alias; // direction id -> canonical DIRECTION | mirrored << 31 (host only)
    int n_rows = 0;                // table rows in use = canonical directions, densely numbered (ensure_alias)
    int* d_canon = nullptr;        // [n_rows] table row -> its direction
    // fused table of k_spatial_rows (vet_spatial_rows.hpp): one row per distinct direction over ALL lattices
    struct Fused {
        int state = 0;             // 0 not built, 1 ready, -1 not usable for this plan
        int R = 0, stride = 0, gs_log2 = 4;
        bool interleaved = false;
        vet::FusedLayout lay;
        int* d_canon = nullptr;    // [R] row -> direction
        uint32_t* d_rec = nullptr; // [n_dirs] direction -> row | mirrored << 15 | nearest tile << 16
        uint16_t* d_lens = nullptr;// [R+2]
        uint32_t* d_meta = nullptr;// [R+1] entries in use | row shift << 16 (k_spatial_lut's form of lens)
        uint2* d_dirrec = nullptr; // [n_dirs] k_spatial_lut's per-direction record over the fused rows (k_dirrec)
        uint8_t* d_row_s = nullptr;// [R+1] fused row shifts
        uint32_t* d_w = nullptr;   // [R+1][stride]
        uint16_t* d_i = nullptr;   // [R+1][stride]
    } fused;
    bool stats_all = false;        // k_row_stats has run for every weighted lattice
    bool ultra = false;            // some lattice has ultra-tiny in-FoV weights: FP64 formulations only (plan-wide)
};

namespace {

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

int collect_profile(vet_ctx* c) {
    for (auto& ep : c->pending) {
        HIP_TRY(hipEventSynchronize(ep.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
        c->prof_ms[ep.kid] += ms;
        c->prof_n[ep.kid] += 1;
        c->free_events.push_back(ep.a);
        c->free_events.push_back(ep.b);
    }
    c->pending.clear();
    return VET_OK;
}

int ensure_ws(vet_ctx* c, size_t bytes) {
    if (bytes <= c->ws_bytes) return VET_OK;
    // earlier calls may still use the old workspace on a caller's stream: wait for the device
    HIP_TRY(hipDeviceSynchronize());
    if (c->ws) HIP_TRY(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_bytes = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipMalloc(&c->ws, want));
    c->ws_bytes = want;
    return VET_OK;
}

// Tuning knobs come from the environment; a value outside [lo, hi] is ignored.
int env_int(const char* name, int lo, int hi, int fallback) {
    const char* e = getenv(name);
    if (!e || !*e) return fallback;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < lo || v > hi) return fallback;
    return (int)v;
}
int env_threads(const char* name, int fallback) {       // workgroup size: whole waves, at most 1024 threads
    const int v = env_int(name, 64, 1024, fallback);
    return v % 64 == 0 ? v : fallback;
}

int grid_for(long work, int block, int n_cu) {
    long b = (work + block - 1) / block;
    long cap = (long)n_cu * 8;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

struct Geometry {
    int NW, FPW, G, UC, R;
    size_t lds;
};

// launch geometry of the spatial kernels for a lattice of n tiles and U users
int spatial_geometry(const vet_ctx* c, int n, int U, bool weighted, Geometry* g, bool one_frame = false) {
    if (!weighted) {
        // k_spatial_u: one wave per frame in the entropy phase; keep >= 2048 samples per workgroup
        g->R = 1; g->G = 1; g->UC = 0;
        g->NW = 4;
        g->NW = env_int("VET_U_WAVES", 1, 16, g->NW);
        g->FPW = U >= 2048 ? 2 : (U >= 512 ? 4 : (U >= 128 ? 8 : 32));
        g->FPW = env_int("VET_U_FPW", 1, 64, g->FPW);
        while ((size_t)g->FPW * n * 4 > c->lds_max && g->FPW > 1) g->FPW /= 2;
        g->lds = (size_t)g->FPW * n * 4;
        if (g->lds > c->lds_max)
            return fail(VET_ERR_UNSUPPORTED, "lattice of %d tiles does not fit the LDS histogram (%zu B)", n, g->lds);
        return VET_OK;
    }
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
    while (lds_of(g->FPW, g->UC) > c->lds_max && g->FPW > 1) g->FPW /= 2;
    while (lds_of(g->FPW, g->UC) > c->lds_max && g->UC > 64) g->UC /= 2;
    g->lds = lds_of(g->FPW, g->UC);
    if (g->lds > c->lds_max)
        return fail(VET_ERR_UNSUPPORTED, "lattice of %d tiles does not fit the LDS histogram (%zu B)", n, g->lds);
    return VET_OK;
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
template <bool FROM_IDS>
const void* lut_kernel_fused(bool il, bool occ8, bool dedup) {
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

constexpr size_t kMaxTableBytes = (size_t)24 << 30;   // per lattice; HBM is 288 GB
constexpr double kContractMargin = 1e-7;              // bound on |dH|/H an integer formulation may have (contract: 1e-6)

// direction id -> table row.  Directions with the same Vector (value equality, -0.0 == 0.0) share a row:
// the pole row of a pixel grid, the -180 -> 0 / -90 -> 0 remaps (utilities/data_utils.py:394-397) and
// 6-decimal collisions make different pixels the same direction.  And the Fibonacci lattice is symmetric
// under (x,y,z) -> (x,-y,-z) (tile i <-> tile n-1-i, utilities/data_utils.py:40-50: lat is odd in i, lon of -i
// is 360 - lon of i), so when every lattice of the plan and the direction table have that symmetry BIT FOR
// BIT, a direction and its mirror image share one row too, the mirrored one adding into tiles n-1-t: the
// dot products d.t are then identical bit for bit, hence the weights.  Halves the table's cache footprint.
// alias[d] = row | mirrored << 31.
int ensure_alias(vet_plan* pl) {
    if (pl->d_alias) return VET_OK;
    const size_t D = (size_t)pl->n_dirs;
    std::vector<double> raw(D * 3);
    HIP_TRY(hipMemcpy(raw.data(), pl->d_dir_raw, D * 24, hipMemcpyDeviceToHost));
    struct KeyHash {
        size_t operator()(const std::array<uint64_t, 3>& k) const {
            uint64_t h = k[0] * 0x9E3779B97F4A7C15ull;
            h ^= (k[1] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
            h ^= (k[2] * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2));
            return (size_t)h;
        }
    };
    auto key_of = [](double x, double y, double z) {
        std::array<uint64_t, 3> key;
        const double v[3] = {x + 0.0, y + 0.0, z + 0.0};    // -0.0 -> +0.0
        memcpy(key.data(), v, 24);
        return key;
    };
    std::unordered_map<std::array<uint64_t, 3>, uint32_t, KeyHash> first;
    first.reserve(D * 2);
    std::vector<uint32_t> alias(D);
    for (size_t d = 0; d < D; ++d)
        alias[d] = first.emplace(key_of(raw[3 * d], raw[3 * d + 1], raw[3 * d + 2]), (uint32_t)d).first->second;
    // mirror symmetry of every lattice, bit for bit on the unit vectors the kernels use
    bool mirror = !getenv("VET_NO_MIRROR") && pl->weighted;
    for (const auto& L : pl->lat) {
        if (L.binned || L.h_unit.empty()) { mirror = false; break; }
        for (int i = 0; i < L.n && mirror; ++i) {
            const double* a = &L.h_unit[3 * (size_t)i];
            const double* b = &L.h_unit[3 * (size_t)(L.n - 1 - i)];
            mirror = a[0] == b[0] && a[1] == -b[1] && a[2] == -b[2];
        }
        if (!mirror) break;
    }
    pl->mirror = mirror;
    if (mirror) {
        for (size_t d = 0; d < D; ++d) {
            if (alias[d] != d) continue;                      // canonical rows only
            const auto it = first.find(key_of(raw[3 * d], -raw[3 * d + 1], -raw[3 * d + 2]));
            if (it != first.end() && it->second < d) alias[d] = it->second | 0x80000000u;
        }
        for (size_t d = 0; d < D; ++d) {                      // ids aliased to a mirrored row
            const uint32_t a = alias[d];
            if (!(a & 0x80000000u) && a != d) alias[d] = alias[a];
        }
    }
    // table rows = canonical directions, densely numbered: a table holds n_rows + 1 rows instead of n_dirs + 1
    // (100 x 200 grid: 9 951 of 20 301 — half the memory and half the build time; 3840 x 1920: 4.3 instead of 8.5 GB)
    std::vector<int> canon;
    std::vector<uint32_t> rowid(D, 0u), rowsel(D);
    for (size_t d = 0; d < D; ++d)
        if (alias[d] == (uint32_t)d) { rowid[d] = (uint32_t)canon.size(); canon.push_back((int)d); }
    for (size_t d = 0; d < D; ++d) rowsel[d] = rowid[alias[d] & 0x7FFFFFFFu] | (alias[d] & 0x80000000u);
    uint32_t* d_alias = nullptr;
    int* d_canon = nullptr;
    hipError_t e = hipMalloc((void**)&d_alias, D * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_canon, canon.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_alias, rowsel.data(), D * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_canon, canon.data(), canon.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (d_alias) (void)hipFree(d_alias);
        if (d_canon) (void)hipFree(d_canon);
        return fail(VET_ERR_DEVICE, "alias table upload failed: %s", hipGetErrorString(e));
    }
    pl->d_alias = d_alias;
    pl->d_canon = d_canon;
    pl->n_rows = (int)canon.size();
    pl->h_alias = std::move(alias);
    return VET_OK;
}

// k_row_stats of lattice k (first weighted run only; synchronises once)
int ensure_stats(vet_plan* pl, int k, hipStream_t s) {
    Lattice& L = pl->lat[k];
    if (L.stats_done) return VET_OK;
    vet_ctx* c = pl->ctx;
    unsigned long long* d_crit = nullptr;
    HIP_TRY(hipMalloc((void**)&d_crit, 24));
    if ((!L.d_row_s && hipMalloc((void**)&L.d_row_s, (size_t)pl->n_dirs + 1) != hipSuccess) ||
        (!L.d_row_e && hipMalloc((void**)&L.d_row_e, ((size_t)pl->n_dirs + 1) * 2) != hipSuccess)) {
        (void)hipFree(d_crit);
        return fail(VET_ERR_DEVICE, "hipMalloc of the row shift table failed");
    }
    hipError_t e = hipMemsetAsync(d_crit, 0, 24, s);
    vet::StatsParams p{};
    p.dir_unit = pl->d_dir_unit; p.D = (long)pl->n_dirs;
    p.tiles = L.d_tiles; p.n = L.n;
    p.cos_cull = pl->cos_cull;
    p.wc.max_ang = pl->max_ang; p.wc.inv_max = 1.0 / pl->max_ang; p.wc.power = pl->power; p.wc.shift = 0;
    p.row_s = L.d_row_s; p.row_e = L.d_row_e; p.crit = d_crit;
    const int blocks = grid_for((long)pl->n_dirs * vet::WAVE, 256, c->n_cu * 2);
    {
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_row_stats, dim3(blocks), dim3(256), 0, s, p);
    }
    unsigned long long bits[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(bits, d_crit, 24, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_crit);
    if (e != hipSuccess) return fail(VET_ERR_DEVICE, "k_row_stats failed: %s", hipGetErrorString(e));
    memcpy(&L.crit_tab, &bits[0], 8);
    memcpy(&L.crit_base, &bits[1], 8);
    L.ultra = (long)bits[2];
    L.stats_done = true;
    return VET_OK;
}

// statistics of every weighted lattice; whether the plan has ultra-tiny weights is a plan-wide fact (the lattices
// of a fused table launch must be of one kind)
int ensure_all_stats(vet_plan* pl, hipStream_t s) {
    if (pl->stats_all) return VET_OK;
    bool ultra = false;
    for (int k = 0; k < (int)pl->lat.size(); ++k) {
        if (!pl->weighted || pl->lat[k].binned) continue;
        int rc = ensure_stats(pl, k, s);
        if (rc) return rc;
        ultra = ultra || pl->lat[k].ultra > 0;
    }
    pl->ultra = ultra;
    pl->stats_all = true;
    return VET_OK;
}

// Builds lattice k's direction weight table on stream s (first use only; synchronises once).
// A table that does not fit (size cap, allocation failure) marks the lattice stride = -1: the plan then
// stays on the sweep formulation.
int ensure_wtab(vet_plan* pl, int k, hipStream_t s) {
    Lattice& L = pl->lat[k];
    if (L.stride != 0) return VET_OK;
    vet_ctx* c = pl->ctx;
    int rc = ensure_all_stats(pl, s);
    if (rc) return rc;
    rc = ensure_alias(pl);
    if (rc) return rc;
    int* d_max = nullptr;                      // [0] longest row (count pass), [1] marker entries (fill pass)
    HIP_TRY(hipMalloc((void**)&d_max, 2 * sizeof(int)));
    struct FreeMax { int* p; ~FreeMax() { (void)hipFree(p); } } free_max{d_max};
    hipError_t e = hipMemsetAsync(d_max, 0, 2 * sizeof(int), s);
    vet::WtabParams p{};
    const long R = pl->n_rows;                 // rows = canonical directions (ensure_alias)
    p.dir_unit = pl->d_dir_unit; p.D = R;
    p.canon = pl->d_canon; p.shift_by_dir = 1; p.nl = 0;
    p.tiles = L.d_tiles; p.n = L.n;
    p.cos_cull = pl->cos_cull;
    p.wc.max_ang = pl->max_ang; p.wc.inv_max = 1.0 / pl->max_ang; p.wc.power = pl->power; p.wc.shift = 0;
    // integer mantissas where their error bound is inside the contract, FP32 weights otherwise
    L.fp_table = pl->ultra || !(L.crit_tab <= kContractMargin);
    p.stride = 0; p.w = nullptr; p.idx = nullptr; p.meta = nullptr; p.row_s = L.d_row_s; p.row_e = L.d_row_e; p.fp = L.fp_table ? 1 : 0;
    p.maxcount = d_max; p.markers = nullptr; p.gs_log2 = -1;
    const int blocks = grid_for(R * vet::WAVE, 256, c->n_cu * 2);
    {
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_wtab<false>, dim3(blocks), dim3(256), 0, s, p);
    }
    int longest = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&longest, d_max, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(VET_ERR_DEVICE, "k_wtab<count> failed: %s", hipGetErrorString(e));
    int align = 64;      // rows start on 128-byte lines (u16 tile rows) / 256 bytes (u32 weight rows)
    align = (env_int("VET_STRIDE_ALIGN", 64, 1024, align) + 63) / 64 * 64;   // whole 64-entry blocks: the walk reads whole blocks
    int stride = ((longest > 0 ? longest : 1) + align - 1) / align * align;
    const size_t rows = (size_t)R + 1;            // one extra, all-zero row (index n_rows) for the gather's idle lanes
    const size_t bytes = rows * stride * 6 + rows * 4;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = kMaxTableBytes;
    // (the gather addresses entries with 32-bit offsets: fewer than 2^32 of them)
    if (stride > 65535 || bytes > kMaxTableBytes || rows * (size_t)stride >= ((size_t)1 << 32) || bytes + ((size_t)64 << 20) > free_b) { L.stride = -1; return VET_OK; }
    // a failed earlier attempt may have left buffers behind
    auto drop = [&]() {
        if (L.d_tab_w) { (void)hipFree(L.d_tab_w); L.d_tab_w = nullptr; }
        if (L.d_tab_i) { (void)hipFree(L.d_tab_i); L.d_tab_i = nullptr; }
        if (L.d_tab_meta) { (void)hipFree(L.d_tab_meta); L.d_tab_meta = nullptr; }
    };
    drop();
    if (hipMalloc((void**)&L.d_tab_w, rows * stride * 4) != hipSuccess ||
        hipMalloc((void**)&L.d_tab_i, rows * stride * 2) != hipSuccess ||
        hipMalloc((void**)&L.d_tab_meta, rows * 4) != hipSuccess) {
        (void)hipGetLastError();                  // out of memory is not sticky: the sweep still works
        drop();
        L.stride = -1;
        return VET_OK;
    }
    // 4 entries per lane and 2 rows in flight per group; lanes per row (part of the row layout) = the
    // smallest power of two whose 4-entry chunks cover the longest row, at most 16 (measured best for
    // long rows, profiles/r01/v4_table_vs_xcd_partition_sweep.log), so the short rows of small
    // lattices do not idle most of a group
    L.gs_log2 = 1;
    while (L.gs_log2 < 4 && (4 << L.gs_log2) < longest) ++L.gs_log2;
    L.gs_log2 = env_int("VET_GS_LOG2", 1, 4, L.gs_log2);
    // 16-lane rows with at least one block that is 3/4 full get the class-dealt layout (k_wtab)
    L.interleaved = L.gs_log2 == 4 && stride % 64 == 0 && 4 * longest >= 3 * 64;
    L.interleaved = L.interleaved && env_int("VET_TAB_INTERLEAVE", 0, 1, 1) != 0;
    p.stride = stride; p.w = L.d_tab_w; p.idx = L.d_tab_i; p.meta = L.d_tab_meta; p.maxcount = nullptr;
    p.markers = L.fp_table ? d_max + 1 : nullptr;
    p.gs_log2 = L.interleaved ? L.gs_log2 : -1;
    {
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_wtab<true>, dim3(blocks), dim3(256), 0, s, p);
    }
    HIP_TRY(hipGetLastError());
    if (k == 0 && (uint64_t)pl->n_rows <= vet::DEDUP_MAX_DIRS) {
        if (!pl->d_dirrec) HIP_TRY(hipMalloc((void**)&pl->d_dirrec, (size_t)pl->n_dirs * sizeof(uint2)));
        hipLaunchKernelGGL(vet::k_dirrec, dim3(grid_for(pl->n_dirs, 256, c->n_cu)), dim3(256), 0, s, pl->d_alias,
                           L.d_nearest, L.d_tab_meta, (long)pl->n_dirs, pl->d_dirrec);
        HIP_TRY(hipGetLastError());
    }
    // the table is complete before this returns: a later call may run on another stream (first use only)
    int markers = 0;
    HIP_TRY(hipMemcpyAsync(&markers, d_max + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    L.markers = markers;
    L.stride = stride;
    return VET_OK;
}

bool any_binned(const vet_plan* pl);
constexpr size_t kRowsLdsCap = 160 * 1024 - 512;      // k_spatial_rows: one workgroup per CU takes the whole LDS

// Builds the plan's fused table (first use; synchronises).  state = -1: the plan stays on k_spatial_lut.
int ensure_fused(vet_plan* pl, hipStream_t s) {
    auto& F = pl->fused;
    if (F.state != 0) return VET_OK;
    F.state = -1;
    vet_ctx* c = pl->ctx;
    const int K = (int)pl->lat.size();
    // one lattice: a fused row is the lattice's own row — nothing to share, and the per-lattice epilogue is a little
    // cheaper (clustered audience, config-3 shape: 0.48 vs 0.51 ms); VET_FUSED=1 / VET_ROWS=1 fuse such plans too
    if (K == 1 && !getenv("VET_FUSED") && !getenv("VET_ROWS")) return VET_OK;
    if (!pl->weighted || K > vet::MAX_LATTICES || any_binned(pl) || getenv("VET_NO_FUSED")) return VET_OK;
    if ((uint64_t)pl->n_dirs > vet::DEDUP_MAX_DIRS) return VET_OK;
    int rc = ensure_all_stats(pl, s);
    if (rc) return rc;
    if (pl->ultra) return VET_OK;
    for (const auto& L : pl->lat)
        if (!(L.crit_tab <= kContractMargin) || !L.d_row_s) return VET_OK;
    rc = ensure_alias(pl);
    if (rc) return rc;
    const size_t D = (size_t)pl->n_dirs;
    const int R = pl->n_rows;                  // canonical directions, densely numbered (ensure_alias)
    if (R == 0) return VET_OK;
    vet::FusedLayout& lay = F.lay;
    lay.K = K; lay.Hs = 0; lay.CF = 0;
    for (int k = 0; k < K; ++k) {
        lay.n[k] = pl->lat[k].n; lay.off[k] = 2 * K + lay.Hs; lay.Hs += pl->lat[k].n >> 1; lay.hmax[k] = pl->lat[k].hmax;
        lay.CF += (pl->lat[k].n + vet::WAVE - 1) / vet::WAVE;
    }
    lay.N = 2 * (lay.Hs + K) + 4 * K;
    lay.totals = getenv("VET_ROWS") ? 1 : 0;      // k_spatial_rows (experimental) wants the total slots; same-address LDS atomics cost k_spatial_lut 12 %
    if (lay.N > 65535 || lay.N < 32) return VET_OK;

    DevBuf ptrs_d, delta_d, max_d;
    HIP_TRY(ptrs_d.alloc(sizeof(void*) * vet::MAX_LATTICES));
    HIP_TRY(delta_d.alloc(sizeof(int) * vet::MAX_LATTICES));
    HIP_TRY(max_d.alloc(sizeof(int)));
    F.d_canon = nullptr;                       // (the plan's own: pl->d_canon)
    HIP_TRY(hipMalloc((void**)&F.d_rec, D * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&F.d_row_s, (size_t)R + 1));
    const uint8_t* ptrs[vet::MAX_LATTICES] = {};
    for (int k = 0; k < K; ++k) ptrs[k] = pl->lat[k].d_row_s;
    HIP_TRY(hipMemcpyAsync(ptrs_d.p, ptrs, sizeof(ptrs), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(delta_d.p, 0, sizeof(int) * vet::MAX_LATTICES, s));
    HIP_TRY(hipMemsetAsync(max_d.p, 0, sizeof(int), s));
    HIP_TRY(hipMemsetAsync(F.d_row_s + R, vet::TAB_X, 1, s));
    hipLaunchKernelGGL(vet::k_rowrec, dim3(grid_for((long)D, 256, c->n_cu)), dim3(256), 0, s, (const uint32_t*)pl->d_alias,
                       (const uint16_t*)pl->lat[0].d_nearest, (long)D, F.d_rec);
    hipLaunchKernelGGL(vet::k_fuse_shifts, dim3(grid_for(R, 256, c->n_cu)), dim3(256), 0, s, (const int*)pl->d_canon, R, K,
                       (const uint8_t* const*)ptrs_d.p, F.d_row_s, (int*)delta_d.p);
    vet::WtabParams p{};
    p.dir_unit = pl->d_dir_unit; p.D = R;
    p.tiles = nullptr; p.n = 0;
    p.cos_cull = pl->cos_cull;
    p.wc.max_ang = pl->max_ang; p.wc.inv_max = 1.0 / pl->max_ang; p.wc.power = pl->power; p.wc.shift = 0;
    p.stride = 0; p.w = nullptr; p.idx = nullptr; p.meta = nullptr; p.row_s = F.d_row_s; p.row_e = nullptr; p.fp = 0;
    p.markers = nullptr; p.maxcount = (int*)max_d.p; p.gs_log2 = -1;
    p.canon = pl->d_canon; p.shift_by_dir = 0; p.nl = K; p.Hs = lay.Hs; p.N = lay.N; p.totals = lay.totals; p.lens = nullptr;
    for (int k = 0; k < 8; ++k) { p.tiles_v[k] = k < K ? pl->lat[k].d_tiles : nullptr; p.n_v[k] = k < K ? lay.n[k] : 0; p.off_v[k] = k < K ? lay.off[k] : 0; }
    const int blocks = grid_for((long)R * vet::WAVE, 256, c->n_cu * 2);
    {
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_wtab<false>, dim3(blocks), dim3(256), 0, s, p);
    }
    int longest = 0, delta[vet::MAX_LATTICES] = {};
    HIP_TRY(hipMemcpyAsync(&longest, max_d.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(delta, delta_d.p, sizeof(delta), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // a shared shift is coarser than a lattice's own where delta[k] > 0: that lattice's error bound is evaluated
    // again with the rows' shared shifts (k_row_stats, fused pass)
    for (int k = 0; k < K; ++k) {
        if (delta[k] == 0) continue;
        DevBuf crit;
        HIP_TRY(crit.alloc(24));
        HIP_TRY(hipMemsetAsync(crit.p, 0, 24, s));
        vet::StatsParams sp{};
        sp.dir_unit = pl->d_dir_unit; sp.D = R;
        sp.tiles = pl->lat[k].d_tiles; sp.n = pl->lat[k].n;
        sp.cos_cull = pl->cos_cull;
        sp.wc.max_ang = pl->max_ang; sp.wc.inv_max = 1.0 / pl->max_ang; sp.wc.power = pl->power; sp.wc.shift = 0;
        sp.row_s = nullptr; sp.row_e = nullptr; sp.crit = (unsigned long long*)crit.p;
        sp.canon = pl->d_canon; sp.shift_in = F.d_row_s;
        hipLaunchKernelGGL(vet::k_row_stats, dim3(grid_for((long)R * vet::WAVE, 256, c->n_cu * 2)), dim3(256), 0, s, sp);
        unsigned long long bits = 0;
        HIP_TRY(hipMemcpyAsync(&bits, crit.p, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        double bound = 0.0;
        memcpy(&bound, &bits, 8);
        if (!(bound <= kContractMargin)) return VET_OK;
    }
    if (longest >= (1 << vet::ROWS_LEN_BITS)) return VET_OK;
    const int stride = ((longest > 0 ? longest : 1) + 63) / 64 * 64;
    const size_t rows = (size_t)R + 1;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = kMaxTableBytes;
    const size_t bytes = rows * stride * 6;
    if (bytes > kMaxTableBytes || rows * (size_t)stride >= ((size_t)1 << 32) || bytes + ((size_t)64 << 20) > free_b) return VET_OK;
    if (hipMalloc((void**)&F.d_w, rows * stride * 4) != hipSuccess || hipMalloc((void**)&F.d_i, rows * stride * 2) != hipSuccess ||
        hipMalloc((void**)&F.d_lens, (rows + 1) * 2) != hipSuccess || hipMalloc((void**)&F.d_meta, rows * 4) != hipSuccess ||
        hipMalloc((void**)&F.d_dirrec, D * sizeof(uint2)) != hipSuccess) {
        (void)hipGetLastError();
        return VET_OK;
    }
    F.gs_log2 = 1;
    while (F.gs_log2 < 4 && (4 << F.gs_log2) < longest) ++F.gs_log2;
    F.interleaved = F.gs_log2 == 4 && 4 * longest >= 3 * 64 && env_int("VET_TAB_INTERLEAVE", 0, 1, 1) != 0;
    p.stride = stride; p.w = F.d_w; p.idx = F.d_i; p.lens = F.d_lens; p.meta = F.d_meta; p.maxcount = nullptr;
    p.gs_log2 = F.interleaved ? F.gs_log2 : -1;
    {
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_wtab<true>, dim3(blocks), dim3(256), 0, s, p);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(vet::k_dirrec, dim3(grid_for((long)D, 256, c->n_cu)), dim3(256), 0, s, (const uint32_t*)pl->d_alias,
                       (const uint16_t*)pl->lat[0].d_nearest, (const uint32_t*)F.d_meta, (long)D, F.d_dirrec);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    F.R = R; F.stride = stride;
    F.state = 1;
    return VET_OK;
}

// frames per round of k_spatial_rows for frames of U users (0: does not fit); rounds evened out over the CUs
int rows_frames_per_round(const vet_plan* pl, int U, long T, int n_cu, int* hs_out) {
    const auto& F = pl->fused;
    if (F.state != 1 || U > 2048 || !pl->grid || F.R > 32767 || !getenv("VET_ROWS")) return 0;   // experimental
    int HS = 64;
    while (HS < 2 * U) HS <<= 1;
    *hs_out = HS;
    int fb = (vet::ROWS_SPT * vet::ROWS_THREADS) / U;
    if (fb > 16) fb = 16;
    const size_t fixed = vet::rows_lds_static((int)pl->n_dirs, F.R);
    while (fb > 0 && fixed + vet::rows_lds_round(fb, HS, U, F.lay.N, F.lay.CF) > kRowsLdsCap) --fb;
    if (fb <= 0) return 0;
    fb = env_int("VET_ROWS_FB", 1, fb, fb);
    if (T > 0) {                         // same number of rounds for every workgroup where possible
        const long rounds = (T + fb - 1) / fb, per_wg = (rounds + n_cu - 1) / n_cu;
        long even = (T + per_wg * n_cu - 1) / (per_wg * n_cu);
        if (even >= 1 && even <= fb) fb = (int)even;
    }
    return fb;
}

const void* rows_kernel(bool il, int un) {
    if (un >= 4) return il ? (const void*)vet::k_spatial_rows<true, 4> : (const void*)vet::k_spatial_rows<false, 4>;
    return il ? (const void*)vet::k_spatial_rows<true, 2> : (const void*)vet::k_spatial_rows<false, 2>;
}

// one launch of k_spatial_rows (single video, or a batch whose descriptors carry block0 = first round)
int launch_rows(vet_plan* pl, const double* mu, const double* mv, int U, int T, const vet::VideoDesc* d_videos, int n_videos,
                long n_rounds, int FB, int HS, int ucap, double* d_entropy, int32_t* d_assign, double* d_weights,
                int32_t* d_present, int32_t* d_status, hipStream_t s) {
    vet_ctx* c = pl->ctx;
    const auto& F = pl->fused;
    vet::RowsParams q{};
    q.videos = d_videos; q.n_videos = n_videos;
    q.mu = mu; q.mv = mv; q.U = U; q.T = T; q.W = pl->W; q.H = pl->H;
    q.rec = F.d_rec; q.lens = F.d_lens; q.D = (int)pl->n_dirs; q.R = F.R;
    q.tab_w = F.d_w; q.tab_i = F.d_i; q.stride = F.stride; q.gs_log2 = F.gs_log2; q.interleaved = F.interleaved ? 1 : 0;
    q.lay = F.lay;
    q.entropy = d_entropy; q.assign = d_assign; q.weights = d_weights; q.present = d_present; q.status = d_status;
    q.FB = FB; q.HS = HS; q.UCAP = ucap; q.n_rounds = n_rounds;
    const size_t lds = vet::rows_lds_static(q.D, q.R) + vet::rows_lds_round(FB, HS, ucap, F.lay.N, F.lay.CF);
    long grid = c->n_cu;
    if (grid > n_rounds) grid = n_rounds;
    DevBuf dbg;
    if (getenv("VET_ROWS_DEBUG")) {          // development aid: cycles per phase (wave 0 of every workgroup), synchronous
        HIP_TRY(dbg.alloc(64));
        HIP_TRY(hipMemsetAsync(dbg.p, 0, 64, s));
        q.dbg = (unsigned long long*)dbg.p;
    }
    {
        ProfScope ps(c, s, KID_SPATIAL);
        void* args[] = {(void*)&q};
        const void* fn = rows_kernel(F.interleaved, env_int("VET_ROWS_UN", 1, 4, 4));
        HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(vet::ROWS_THREADS), args, lds, s));
        HIP_TRY(hipGetLastError());
    }
    if (dbg.p) {
        unsigned long long t[8] = {};
        HIP_TRY(hipMemcpyAsync(t, dbg.p, 64, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        fprintf(stderr, "[k_spatial_rows] grid %ld FB %d rounds %ld | cycles per workgroup (wave 0): loop %.0f  entropy %.0f  walk+own samples %.0f  late samples %.0f  barrier %.0f  - %.0f\n",
                grid, FB, n_rounds, (double)t[0] / grid, (double)t[1] / grid, (double)t[2] / grid, (double)t[3] / grid, (double)t[4] / grid, (double)t[5] / grid);
    }
    return VET_OK;
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

bool any_binned(const vet_plan* pl) {
    for (const auto& L : pl->lat) if (L.binned) return true;
    return false;
}

// The formulation of a weighted call is a function of the plan, the call's shape and (table does not fit the free
// device memory -> sweep) the memory left on the device — never of what the plan has processed before — and every
// formulation adds in a fixed order, so the same input gives the same floats:
//   table    policy +1, or policy 0 and the call holds at least 8 samples per direction of the table
//            (building a row costs about what the sweep spends on 30 samples; a gathered sample is ~6x
//            cheaper than a swept one), if the table fits and its error bound is inside the contract;
//   sweep    integer (2^-52) histogram, if its error bound is inside the contract;
//   precise  FP64 histogram and exact weights otherwise.
enum { F_TABLE = 0, F_SWEEP = 1, F_PRECISE = 2, F_FTABLE = 3 };

bool table_requested(const vet_plan* pl, long samples, int U) {
    if (!pl->weighted || pl->table_policy < 0 || any_binned(pl) || U >= 65536) return false;
    if ((int)pl->lat.size() > vet::MAX_LATTICES) return false;
    return pl->table_policy > 0 || samples >= 8 * (long)pl->n_dirs;
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
    const int dedup_users = env_int("VET_DEDUP_MIN_USERS", 1, 1 << 20, 128);
    const bool dedup = (uint64_t)pl->n_rows <= vet::DEDUP_MAX_DIRS && pl->d_dirrec && !getenv("VET_NO_DEDUP") &&
                       (d_videos ? batch_max_users : U) >= dedup_users;
    int blocks = blocks_batch, threads = 256;
    size_t lds = lds_batch;
    bool occ8 = true;
    // FP table: canonical row order through a bitmap over (row, mirrored) where that is small (<= 8 KB of LDS)
    q.sort_words = (fpt && dedup && 2 * pl->n_rows <= 65536) ? (int)((2 * pl->n_rows + 31) / 32) : 0;
    if (!d_videos) {
        q.UC = U < 2048 ? U : 2048;
        threads = fpt ? 256 : env_threads("VET_LUT_THREADS", threads);     // the FP table's row sort counts on 256 threads
        if (threads > 256) threads = 256;         // __launch_bounds__(256)
        int fpw = lut_frames_per_wg(U, T, c->n_cu, q.n_sum);
        fpw = env_int("VET_LUT_FPW", 1, 16, fpw);
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
    q.videos = d_videos; q.n_videos = n_videos;
    q.src = src; q.U = U; q.T = T;
    q.nearest = pl->lat[0].d_nearest; q.alias = nullptr; q.dirrec = F.d_dirrec; q.rec_meta = 1;
    q.K = 1; q.n_sum = F.lay.N;
    q.lat[0].tab_w = F.d_w; q.lat[0].tab_i = F.d_i; q.lat[0].tab_meta = F.d_meta;
    q.lat[0].stride = F.stride; q.lat[0].gs_log2 = F.gs_log2; q.lat[0].interleaved = F.interleaved ? 1 : 0;
    q.lat[0].n = F.lay.N; q.lat[0].hmax = 0.0; q.lat[0].zrow = (uint32_t)F.R;
    q.lay = F.lay;
    q.entropy = d_entropy; q.assign = d_assign; q.weights = d_weights; q.present = d_present; q.status = d_status;
    const int dedup_users = env_int("VET_DEDUP_MIN_USERS", 1, 1 << 20, 128);
    const bool dedup = !getenv("VET_NO_DEDUP") && (d_videos ? batch_max_users : U) >= dedup_users;
    int blocks = blocks_batch, threads = 256;
    size_t lds = lds_batch;
    if (!d_videos) {
        q.UC = U < 2048 ? U : 2048;
        int fpw = lut_frames_per_wg(U, T, c->n_cu, q.n_sum);
        fpw = env_int("VET_LUT_FPW", 1, 16, fpw);
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
    const bool occ8 = env_int("VET_LUT_OCC8", 0, 1, 0) != 0;
    DevBuf dbg;
    if (getenv("VET_LUT_DEBUG")) {           // development aid: cycles per stage (thread 0 of every workgroup), synchronous
        HIP_TRY(dbg.alloc(32));
        HIP_TRY(hipMemsetAsync(dbg.p, 0, 32, s));
        q.dbg = (unsigned long long*)dbg.p;
    }
    {
        ProfScope ps(c, s, KID_SPATIAL);
        void* args[] = {(void*)&q};
        HIP_TRY(hipLaunchKernel(lut_kernel_fused<FROM_IDS>(F.interleaved, occ8, dedup), dim3((unsigned)blocks), dim3(threads), args, lds, s));
        HIP_TRY(hipGetLastError());
    }
    if (dbg.p) {
        unsigned long long t[4] = {};
        HIP_TRY(hipMemcpyAsync(t, dbg.p, 32, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        fprintf(stderr, "[k_spatial_lut fused] blocks %d FPW %d lds %zu | cycles per workgroup (thread 0): samples->set %.0f  lists %.0f  walk %.0f  entropy %.0f\n",
                blocks, q.FPW, lds, (double)t[0] / blocks, (double)t[1] / blocks, (double)t[2] / blocks, (double)t[3] / blocks);
    }
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
        const size_t lds = g.lds + (size_t)g.UC * 4 + 16;        // + the users' direction ids (canonical order of the sums)
        if (lds > c->lds_max) return fail(VET_ERR_UNSUPPORTED, "resolver: %zu B of LDS", lds);
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

template <bool FROM_IDS>
int launch_spatial(vet_plan* pl, const vet::SampleSrc& src, int U, int T, double* d_entropy, int32_t* d_assign,
                   double* d_weights, int32_t* d_present, int32_t* d_status, hipStream_t s) {
    vet_ctx* c = pl->ctx;
    const int K = (int)pl->lat.size();
    double* ent_k = d_entropy;
    // ---- formulation per lattice (weighted Fibonacci lattices only)
    int form[64];
    if (K > 64) return fail(VET_ERR_UNSUPPORTED, "more than 64 lattices in one plan");
    const bool want_table = table_requested(pl, (long)U * T, U);
    // ---- weighted, integer table formulation on fused rows (k_spatial_rows): persistent workgroups, the direction
    // records in LDS, one row per distinct direction over all lattices — where the plan allows (ensure_fused)
    if (want_table && pl->weighted) {
        int rc = ensure_fused(pl, s);
        if (rc) return rc;
        int HS = 0;
        const int fb = FROM_IDS ? 0 : rows_frames_per_round(pl, U, T, c->n_cu, &HS);
        if (fb > 0) {
            rc = launch_rows(pl, src.mu, src.mv, U, T, nullptr, 0, ((long)T + fb - 1) / fb, fb, HS, U, d_entropy, d_assign,
                             d_weights, d_present, d_status, s);
            if (!rc) for (int k = 0; k < K; ++k) pl->lat[k].last_form = F_TABLE;
            return rc;
        }
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
        if (!hist_weighted && !FROM_IDS && U <= 4096 && !getenv("VET_U_NO_LDS")) {
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
                long grid = (long)c->n_cu * 2;
                grid = (long)c->n_cu * env_int("VET_U_WGS_PER_CU", 1, 8, 2);
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
    const int hs_pct = env_int("VET_T_HS_PCT", 100, 400, 200);      // bucket-hash slots per 100 users (100: no gain, 43.3 vs 43.7 us)
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
        if (U <= 4096 && lds_run <= lds_cap && !getenv("VET_T_GLOBAL")) {
            // persistent workgroups over contiguous runs of rows (k_transition_run); users per thread 1, 2, 4 or 8:
            // two waves per row up to 512 users (measured: 46 us vs 51 us with four, profiles/r02)
            int threads = U <= 512 ? 128 : (U <= 2048 ? 512 : 1024);
            threads = env_threads("VET_T_THREADS", threads);
            int upt = (U + threads - 1) / threads;
            upt = upt <= 1 ? 1 : (upt <= 2 ? 2 : (upt <= 4 ? 4 : 8));
            while ((long)upt * threads < U) threads *= 2;
            long per_cu = (long)(lds_cap / lds_run);
            const long by_waves = 32 / (threads / 64);
            if (per_cu > by_waves) per_cu = by_waves;
            per_cu = env_int("VET_T_WGS_PER_CU", 1, 16, (int)(per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu)));
            long grid = (long)c->n_cu * per_cu;
            if (grid > R) grid = R;
            p.run_q = (int)(R / grid); p.run_r = (int)(R % grid);
            const void* fn = transition_run_kernel<FROM_IDS>(upt, (long)upt * threads == U, threads);
            void* args[] = {(void*)&p};
            HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(threads), args, lds_run, s));
        } else {
            // more users than the LDS holds: bucket hash and per-user words in global scratch, one slice per
            // persistent workgroup (the reference accepts any number of users, entropy_utils.py:259-332)
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

int check_run_args(const vet_plan* pl, int U, int T, const void* out) {
    if (!pl) return fail(VET_ERR_INVALID, "plan is NULL");
    if (U <= 0 || T <= 0) return fail(VET_ERR_INVALID, "n_users and n_frames must be positive (got %d, %d)", U, T);
    if (!out) return fail(VET_ERR_INVALID, "entropy output pointer is NULL");
    return VET_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int vet_version(void) { return VET_VERSION; }
const char* vet_last_error(void) { return g_err.c_str(); }
const char* vet_kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? kKernelNames[kid] : ""; }

int vet_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vet_create(int device_id, vet_ctx** out) {
    if (!out) return fail(VET_ERR_INVALID, "out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(VET_ERR_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(VET_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    vet_ctx* c = new vet_ctx();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // keep two workgroups per CU resident: cap a workgroup at half of the 160 KiB LDS
    c->lds_max = prop.sharedMemPerBlock >= 160 * 1024 ? 80 * 1024 : (size_t)prop.sharedMemPerBlock;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_log2, 4097 * sizeof(double));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(vet::k_log2_table, dim3(17), dim3(256), 0, c->stream, c->d_log2, 4097);
        e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) {
        if (c->d_log2) (void)hipFree(c->d_log2);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(VET_ERR_DEVICE, "context set-up failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return VET_OK;
}

int vet_destroy(vet_ctx* c) {
    if (!c) return VET_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& ep : c->pending) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    if (c->ws) (void)hipFree(c->ws);
    if (c->d_log2) (void)hipFree(c->d_log2);
    for (void* q : c->pool) if (q) (void)hipFree(q);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return VET_OK;
}

int vet_synchronize(vet_ctx* c) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VET_OK;
}

int vet_profile_enable(vet_ctx* c, int on) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    c->profiling = on != 0;
    return VET_OK;
}
int vet_profile_reset(vet_ctx* c) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    int rc = collect_profile(c);
    if (rc) return rc;
    for (int i = 0; i < KID_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
    return VET_OK;
}
int vet_profile_get(vet_ctx* c, int kid, double* total_ms, int64_t* launches) {
    if (!c || kid < 0 || kid >= KID_COUNT) return fail(VET_ERR_INVALID, "bad ctx or kernel id");
    int rc = collect_profile(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->prof_ms[kid];
    if (launches) *launches = c->prof_n[kid];
    return VET_OK;
}

int vet_malloc(vet_ctx* c, size_t bytes, void** d_ptr) {
    if (!c || !d_ptr) return fail(VET_ERR_INVALID, "ctx or d_ptr is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 8));
    return VET_OK;
}
int vet_free(vet_ctx* c, void* d_ptr) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return VET_OK;
}
int vet_memcpy_h2d(vet_ctx* c, void* d, const void* h, size_t bytes) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VET_OK;
}
int vet_memcpy_d2h(vet_ctx* c, void* h, const void* d, size_t bytes) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VET_OK;
}

// ------------------------------------------------------------------------------------------------
int vet_plan_create(vet_ctx* c, const vet_plan_desc* d, vet_plan** out) {
    if (!c || !d || !out) return fail(VET_ERR_INVALID, "ctx, desc or out is NULL");
    const bool grid = d->h_lon_cos && d->h_lon_sin && d->h_lat_sin && d->h_lat_cos;
    if (!grid && !(d->h_dir_table && d->n_dirs > 0))
        return fail(VET_ERR_INVALID, "plan needs the four axis tables or an explicit direction table");
    if (grid && (d->video_width <= 0 || d->video_height <= 0))
        return fail(VET_ERR_INVALID, "Video dimensions must be positive");
    if (d->n_lattices <= 0 || !d->n_tiles || !d->h_tiles || !d->h_max_entropy)
        return fail(VET_ERR_INVALID, "Must specify at least one tile count");
    if (!(d->fov_angle > 0.0 && d->fov_angle <= 360.0))
        return fail(VET_ERR_INVALID, "FOV angle must be between 0 and 360 degrees");
    if (!(d->power_factor > 0.0)) return fail(VET_ERR_INVALID, "Power factor must be positive");
    for (int k = 0; k < d->n_lattices; ++k) {
        const bool binned = d->h_bin_lut && d->h_bin_lut[k];
        if (d->n_tiles[k] <= 0 || d->n_tiles[k] > 65535 || (!binned && !d->h_tiles[k]))
            return fail(VET_ERR_INVALID, "lattice %d: tile count %d outside [1, 65535]", k, d->n_tiles[k]);
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    vet_plan* pl = new vet_plan();
    pl->ctx = c;
    pl->grid = grid;
    pl->W = d->video_width; pl->H = d->video_height;
    pl->fov = d->fov_angle; pl->max_ang = d->max_angular_distance; pl->power = d->power_factor;
    pl->weighted = d->use_weight_distribution ? 1 : 0;
    // conservative cull on the cosine; the exact 'distance < max' test runs on the survivors
    pl->cos_cull = pl->max_ang >= 3.14159 ? -2.0 : std::cos(pl->max_ang) - 1e-9;
    pl->n_dirs = grid ? (int64_t)(pl->W + 1) * (pl->H + 1) : d->n_dirs;
    if (pl->n_dirs >= (1LL << 31)) { delete pl; return fail(VET_ERR_UNSUPPORTED, "direction table too large"); }

    auto cleanup = [&](int rc) { vet_plan_destroy(pl); return rc; };
#define PLAN_TRY(expr)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return cleanup(fail(VET_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_))); \
    } while (0)

    PLAN_TRY(hipMalloc((void**)&pl->d_dir_raw, (size_t)pl->n_dirs * 3 * sizeof(double)));
    PLAN_TRY(hipMalloc((void**)&pl->d_dir_unit, (size_t)pl->n_dirs * 3 * sizeof(double)));
    if (grid) {
        const size_t nw = (size_t)pl->W + 1, nh = (size_t)pl->H + 1;
        double* d_axes = nullptr;
        PLAN_TRY(hipMalloc((void**)&d_axes, (2 * nw + 2 * nh) * sizeof(double)));
        hipError_t e1 = hipMemcpyAsync(d_axes, d->h_lon_cos, nw * 8, hipMemcpyHostToDevice, s);
        hipError_t e2 = hipMemcpyAsync(d_axes + nw, d->h_lon_sin, nw * 8, hipMemcpyHostToDevice, s);
        hipError_t e3 = hipMemcpyAsync(d_axes + 2 * nw, d->h_lat_sin, nh * 8, hipMemcpyHostToDevice, s);
        hipError_t e4 = hipMemcpyAsync(d_axes + 2 * nw + nh, d->h_lat_cos, nh * 8, hipMemcpyHostToDevice, s);
        if (e1 || e2 || e3 || e4) { (void)hipFree(d_axes); return cleanup(fail(VET_ERR_DEVICE, "axis table upload failed")); }
        {
            ProfScope ps(c, s, KID_GRID);
            hipLaunchKernelGGL(vet::k_grid_dirs, dim3(grid_for(pl->n_dirs, 256, c->n_cu)), dim3(256), 0, s, d_axes,
                               d_axes + nw, d_axes + 2 * nw, d_axes + 2 * nw + nh, pl->W, pl->H, pl->d_dir_raw,
                               pl->d_dir_unit);
        }
        hipError_t e5 = hipStreamSynchronize(s);
        (void)hipFree(d_axes);
        if (e5 != hipSuccess) return cleanup(fail(VET_ERR_DEVICE, "k_grid_dirs failed: %s", hipGetErrorString(e5)));
    } else {
        PLAN_TRY(hipMemcpyAsync(pl->d_dir_raw, d->h_dir_table, (size_t)pl->n_dirs * 24, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(vet::k_unit_dirs, dim3(grid_for(pl->n_dirs, 256, c->n_cu)), dim3(256), 0, s, pl->d_dir_raw,
                           (long)pl->n_dirs, pl->d_dir_unit);
        PLAN_TRY(hipStreamSynchronize(s));
    }
    pl->lat.resize(d->n_lattices);
    for (int k = 0; k < d->n_lattices; ++k) {
        Lattice& L = pl->lat[k];
        L.n = d->n_tiles[k];
        L.hmax = d->h_max_entropy[k];
        L.norm_n = d->n_norm_tiles ? d->n_norm_tiles[k] : L.n;
        if (d->h_bin_lut && d->h_bin_lut[k]) {
            L.binned = true;
            for (int64_t i = 0; i < pl->n_dirs; ++i)
                if (d->h_bin_lut[k][i] >= L.n)
                    return cleanup(fail(VET_ERR_INVALID, "lattice %d: bin %u of direction %lld >= %d bins", k,
                                        (unsigned)d->h_bin_lut[k][i], (long long)i, L.n));
            // one spare entry: k_spatial_u_lds copies the LUT in 32-bit words
            PLAN_TRY(hipMalloc((void**)&L.d_nearest, ((size_t)pl->n_dirs + 1) * sizeof(uint16_t)));
            PLAN_TRY(hipMemsetAsync(L.d_nearest + pl->n_dirs, 0, sizeof(uint16_t), s));
            PLAN_TRY(hipMemcpyAsync(L.d_nearest, d->h_bin_lut[k], (size_t)pl->n_dirs * sizeof(uint16_t),
                                    hipMemcpyHostToDevice, s));
            PLAN_TRY(hipStreamSynchronize(s));
            continue;
        }
        std::vector<double> unit((size_t)L.n * 3);
        for (int t = 0; t < L.n; ++t) {
            const double x = d->h_tiles[k][3 * t], y = d->h_tiles[k][3 * t + 1], z = d->h_tiles[k][3 * t + 2];
            const double len = std::sqrt(x * x + y * y + z * z);
            if (!(len > 0.0)) return cleanup(fail(VET_ERR_INVALID, "Vector cannot have zero length (lattice %d tile %d)", k, t));
            unit[3 * t] = x / len; unit[3 * t + 1] = y / len; unit[3 * t + 2] = z / len;
        }
        PLAN_TRY(hipMalloc((void**)&L.d_tiles, unit.size() * sizeof(double)));
        PLAN_TRY(hipMalloc((void**)&L.d_nearest, ((size_t)pl->n_dirs + 1) * sizeof(uint16_t)));
        PLAN_TRY(hipMemsetAsync(L.d_nearest + pl->n_dirs, 0, sizeof(uint16_t), s));
        PLAN_TRY(hipMemcpyAsync(L.d_tiles, unit.data(), unit.size() * sizeof(double), hipMemcpyHostToDevice, s));
        PLAN_TRY(hipStreamSynchronize(s));   // 'unit' goes out of scope
        L.h_unit = unit;
        const size_t lds = (size_t)L.n * 3 * sizeof(double);
        if (lds > 160 * 1024 - 1024) return cleanup(fail(VET_ERR_UNSUPPORTED, "lattice of %d tiles exceeds the LDS tile cache", L.n));
        if (lds > 64 * 1024)
            PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_nearest_lut, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        {
            ProfScope ps(c, s, KID_NEAREST);
            hipLaunchKernelGGL(vet::k_nearest_lut, dim3(grid_for(pl->n_dirs, 64, c->n_cu * 4)), dim3(64), lds, s,
                               pl->d_dir_unit, (long)pl->n_dirs, L.d_tiles, L.n, L.d_nearest);
        }
        PLAN_TRY(hipGetLastError());
    }
    PLAN_TRY(hipStreamSynchronize(s));
    // the run kernels may need more than the default 64 KiB of dynamic LDS (set once per context)
    if (!c->attrs_set) {
    for (int R = 1; R <= 2; ++R) {
        for (int wm = 0; wm < 3; ++wm) {
            PLAN_TRY(hipFuncSetAttribute(spatial_w_kernel<false>(wm, R), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
            PLAN_TRY(hipFuncSetAttribute(spatial_w_kernel<true>(wm, R), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
        }
        PLAN_TRY(hipFuncSetAttribute(spatial_w_kernel<false>(0, R, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
        PLAN_TRY(hipFuncSetAttribute(spatial_w_kernel<true>(0, R, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    }
    for (int v = 0; v < 8; ++v) {
        PLAN_TRY(hipFuncSetAttribute(lut_kernel_fused<false>(v & 1, v & 2, v & 4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
        PLAN_TRY(hipFuncSetAttribute(lut_kernel_fused<true>(v & 1, v & 2, v & 4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
        PLAN_TRY(hipFuncSetAttribute(lut_kernel<false>(v & 1, v & 2, v & 4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
        PLAN_TRY(hipFuncSetAttribute(lut_kernel<true>(v & 1, v & 2, v & 4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
        if (!(v & 2)) {
            PLAN_TRY(hipFuncSetAttribute(lut_kernel<false>(v & 1, false, v & 4, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
            PLAN_TRY(hipFuncSetAttribute(lut_kernel<true>(v & 1, false, v & 4, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
        }
    }
    for (int un : {2, 4})
        for (int il = 0; il < 2; ++il)
            PLAN_TRY(hipFuncSetAttribute(rows_kernel(il != 0, un), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRowsLdsCap));
    PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_spatial_u_lds<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_spatial_u_lds<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_spatial_u_lds<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_spatial_u_lds<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_spatial_u_lds<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_spatial_u_lds<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_spatial_u<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    PLAN_TRY(hipFuncSetAttribute((const void*)vet::k_spatial_u<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_max));
    {
        std::vector<const void*> tk = {(const void*)vet::k_transition_any<false>, (const void*)vet::k_transition_any<true>};
        for (int upt : {1, 2, 4, 8})
            for (int ex = 0; ex < 2; ++ex)
                for (int threads : {128, 0}) {
                    tk.push_back(transition_run_kernel<false>(upt, ex != 0, threads));
                    tk.push_back(transition_run_kernel<true>(upt, ex != 0, threads));
                }
        for (int upt : {1, 2, 4, 8})
            for (int ex = 0; ex < 2; ++ex) tk.push_back(transition_batch_kernel(upt, ex != 0));
        for (const void* f : tk) PLAN_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    }
    c->attrs_set = true;
    }
#undef PLAN_TRY
    *out = pl;
    return VET_OK;
}

int vet_plan_destroy(vet_plan* pl) {
    if (!pl) return VET_OK;
    (void)hipSetDevice(pl->ctx->device);
    (void)hipStreamSynchronize(pl->ctx->stream);
    if (pl->d_dir_raw) (void)hipFree(pl->d_dir_raw);
    if (pl->d_dir_unit) (void)hipFree(pl->d_dir_unit);
    for (auto& L : pl->lat) {
        if (L.d_tiles) (void)hipFree(L.d_tiles);
        if (L.d_nearest) (void)hipFree(L.d_nearest);
        if (L.d_tab_w) (void)hipFree(L.d_tab_w);
        if (L.d_tab_i) (void)hipFree(L.d_tab_i);
        if (L.d_tab_meta) (void)hipFree(L.d_tab_meta);
        if (L.d_row_s) (void)hipFree(L.d_row_s);
        if (L.d_row_e) (void)hipFree(L.d_row_e);
    }
    if (pl->d_alias) (void)hipFree(pl->d_alias);
    if (pl->d_canon) (void)hipFree(pl->d_canon);
    if (pl->d_dirrec) (void)hipFree(pl->d_dirrec);
    {
        auto& F = pl->fused;
        for (void* q : {(void*)F.d_canon, (void*)F.d_rec, (void*)F.d_lens, (void*)F.d_row_s, (void*)F.d_w, (void*)F.d_i, (void*)F.d_meta, (void*)F.d_dirrec})
            if (q) (void)hipFree(q);
    }
    delete pl;
    return VET_OK;
}

int64_t vet_plan_n_dirs(const vet_plan* pl) { return pl ? pl->n_dirs : 0; }

int vet_plan_set_table_policy(vet_plan* pl, int policy) {
    if (!pl) return fail(VET_ERR_INVALID, "plan is NULL");
    pl->table_policy = policy > 0 ? 1 : (policy < 0 ? -1 : 0);
    return VET_OK;
}

int64_t vet_plan_table_rows(const vet_plan* pl) { return pl ? pl->n_rows : 0; }

int vet_plan_table_stride(const vet_plan* pl, int k) {
    if (!pl || k < 0 || k >= (int)pl->lat.size()) return 0;
    // plans on the fused table (one row per direction over all lattices) never build the per-lattice ones
    if (pl->lat[k].stride == 0 && pl->fused.state == 1) return pl->fused.stride;
    return pl->lat[k].stride;
}

// get_fb_tile_boundaries (utilities/data_utils.py:58-189) for one lattice; synchronous, host buffers
int vet_fb_tile_boundaries(vet_ctx* c, const double* h_tiles, int n, int max_edges, double* h_edges, int32_t* h_count) {
    if (!c || !h_tiles || !h_edges || !h_count) return fail(VET_ERR_INVALID, "ctx, tiles or an output is NULL");
    if (n <= 0 || max_edges <= 0) return fail(VET_ERR_INVALID, "need n > 0 tiles and max_edges > 0");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t eb = (size_t)n * max_edges * 6 * sizeof(double);
    DevBuf tiles, edges, count, err;
    HIP_TRY(tiles.alloc((size_t)n * 24));
    HIP_TRY(edges.alloc(eb));
    HIP_TRY(count.alloc((size_t)n * 4));
    HIP_TRY(err.alloc(4));
    HIP_TRY(hipMemcpyAsync(tiles.p, h_tiles, (size_t)n * 24, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(edges.p, 0xFF, eb, s));            // NaN padding
    HIP_TRY(hipMemsetAsync(err.p, 0, 4, s));
    hipLaunchKernelGGL(vet::k_fb_boundaries, dim3((n + 63) / 64), dim3(64), 0, s, (const double*)tiles.p, n, max_edges,
                       (double*)edges.p, (int32_t*)count.p, (int32_t*)err.p);
    HIP_TRY(hipGetLastError());
    int32_t bad = 0;
    HIP_TRY(hipMemcpyAsync(h_edges, edges.p, eb, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h_count, count.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&bad, err.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (bad) return fail(VET_ERR_UNSUPPORTED, "%d tile(s) with more than %d neighbours or more than %d edges", bad,
                         vet::FB_MAX_NEIGHBOURS, max_edges);
    return VET_OK;
}

int vet_plan_last_formulation(const vet_plan* pl, int k) {
    if (!pl || k < 0 || k >= (int)pl->lat.size()) return -1;
    return pl->lat[k].last_form;
}

int vet_plan_error_bounds(vet_plan* pl, int k, double* table_bound, double* sweep_bound) {
    if (!pl) return fail(VET_ERR_INVALID, "plan is NULL");
    if (k < 0 || k >= (int)pl->lat.size()) return fail(VET_ERR_INVALID, "lattice index %d out of range", k);
    if (pl->lat[k].binned) return fail(VET_ERR_INVALID, "lattice %d is binned (integer counts, exact)", k);
    HIP_TRY(hipSetDevice(pl->ctx->device));
    int rc = ensure_all_stats(pl, pl->ctx->stream);
    if (rc) return rc;
    // plans with weights that underflow (the reference's NaN frames) never use an integer formulation
    const double inf = std::numeric_limits<double>::infinity();
    if (table_bound) *table_bound = pl->ultra ? inf : pl->lat[k].crit_tab;
    if (sweep_bound) *sweep_bound = pl->ultra ? inf : pl->lat[k].crit_base * std::ldexp(1.0, -52);
    return VET_OK;
}

int vet_plan_read_dirs(vet_plan* pl, double* h_xyz) {
    if (!pl || !h_xyz) return fail(VET_ERR_INVALID, "plan or output is NULL");
    HIP_TRY(hipMemcpy(h_xyz, pl->d_dir_raw, (size_t)pl->n_dirs * 24, hipMemcpyDeviceToHost));
    return VET_OK;
}

int vet_plan_read_nearest(vet_plan* pl, int k, int32_t* h_nearest) {
    if (!pl || !h_nearest) return fail(VET_ERR_INVALID, "plan or output is NULL");
    if (k < 0 || k >= (int)pl->lat.size()) return fail(VET_ERR_INVALID, "lattice index %d out of range", k);
    std::vector<uint16_t> tmp((size_t)pl->n_dirs);
    HIP_TRY(hipMemcpy(tmp.data(), pl->lat[k].d_nearest, tmp.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) h_nearest[i] = tmp[i];
    return VET_OK;
}

// ------------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------------
// Batch of videos in ONE launch (weighted table formulation): short videos are launch-bound one at
// a time (config 2: 43 us of kernel per call), so their frame blocks share a grid.  Falls back to
// one call per video when the table formulation does not apply.
static int pooled(vet_ctx* c, int slot, size_t bytes, void** out);

// Unweighted (nearest-tile) batch: every video's frame blocks in ONE k_spatial_u_lds launch per lattice; with several
// lattices the per-lattice values go through the workspace and k_finalize_batch forms the means.  Returns launched =
// false when the batch does not fit the kernel (odd shapes, LUT too large for LDS): the caller loops over the videos.
static int batch_unweighted(vet_plan* pl, int n_videos, const vet_video* videos, int32_t* d_status, hipStream_t s, bool* launched) {
    *launched = false;
    vet_ctx* c = pl->ctx;
    const int K = (int)pl->lat.size();
    if (getenv("VET_U_NO_LDS")) return VET_OK;
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
    void* d_desc = nullptr;
    int rc = pooled(c, 7, desc.size() * sizeof(vet::VideoDesc) + (frame0.size() + outs.size()) * 8, &d_desc);
    if (rc) return rc;
    char* base = (char*)d_desc;
    long* d_frame0 = (long*)(base + desc.size() * sizeof(vet::VideoDesc));
    double** d_outs = (double**)(d_frame0 + frame0.size());
    c->batch_desc = desc;                          // host copies stay alive until the copies below have run
    HIP_TRY(hipMemcpyAsync(base, c->batch_desc.data(), desc.size() * sizeof(vet::VideoDesc), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_frame0, frame0.data(), frame0.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_outs, outs.data(), outs.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));              // frame0 / outs are locals (pageable copies are staged, this is belt and braces)
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
        long grid = (long)c->n_cu * env_int("VET_U_WGS_PER_CU", 1, 8, 2);
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
            const bool dedup = !getenv("VET_NO_DEDUP") && max_users >= env_int("VET_DEDUP_MIN_USERS", 1, 1 << 20, 128);
            std::vector<vet::VideoDesc>& desc = c->batch_desc;
            desc.resize(n_videos);
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
                void* d_desc = nullptr;
                rc = pooled(c, 7, desc.size() * sizeof(vet::VideoDesc), &d_desc);
                if (rc) return rc;
                HIP_TRY(hipMemcpyAsync(d_desc, desc.data(), desc.size() * sizeof(vet::VideoDesc), hipMemcpyHostToDevice, s));
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
    std::vector<vet::VideoDesc>& desc = c->batch_desc;
    size_t lds_max = 0;
    if (table) {
        const bool dedup = (uint64_t)pl->n_rows <= vet::DEDUP_MAX_DIRS && pl->d_dirrec && !getenv("VET_NO_DEDUP") &&
                           max_users >= env_int("VET_DEDUP_MIN_USERS", 1, 1 << 20, 128);
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
            void* d_desc = nullptr;
            int rc = pooled(c, 7, desc.size() * sizeof(vet::VideoDesc), &d_desc);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(d_desc, desc.data(), desc.size() * sizeof(vet::VideoDesc), hipMemcpyHostToDevice, s));
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
    bool one_launch = rows > 0 && max_users <= 4096 && lds_run <= lds_cap && !getenv("VET_T_GLOBAL");
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
    threads = env_threads("VET_T_THREADS", threads);
    int upt = (max_users + threads - 1) / threads;
    upt = upt <= 1 ? 1 : (upt <= 2 ? 2 : (upt <= 4 ? 4 : 8));
    while ((long)upt * threads < max_users) threads *= 2;
    long per_cu = (long)(lds_cap / lds_run);
    const long by_waves = 32 / (threads / 64);
    if (per_cu > by_waves) per_cu = by_waves;
    per_cu = env_int("VET_T_WGS_PER_CU", 1, 16, (int)(per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu)));
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
    void* d_buf = nullptr;
    int rc = pooled(c, 7, tv.size() * sizeof(vet::TransVideo) + (row0.size() + outs.size()) * 8, &d_buf);
    if (rc) return rc;
    char* base = (char*)d_buf;
    long* d_row0 = (long*)(base + tv.size() * sizeof(vet::TransVideo));
    double** d_outs = (double**)(d_row0 + row0.size());
    HIP_TRY(hipMemcpyAsync(base, tv.data(), tv.size() * sizeof(vet::TransVideo), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_row0, row0.data(), row0.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_outs, outs.data(), outs.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));              // the descriptors are locals
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

// ------------------------------------------------------------------------------------------------
// host-buffer variants: stage through the context's grow-only device buffers (synchronous)

// slot-indexed staging buffer of at least `bytes` bytes (kept by the context between calls)
static int pooled(vet_ctx* c, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 8;
    if (c->pool_cap[slot] < bytes) {
        if (c->pool[slot]) HIP_TRY(hipFree(c->pool[slot]));
        c->pool[slot] = nullptr; c->pool_cap[slot] = 0;
        const size_t want = bytes + bytes / 8;
        HIP_TRY(hipMalloc(&c->pool[slot], want));
        c->pool_cap[slot] = want;
    }
    *out = c->pool[slot];
    return VET_OK;
}
#define POOL(slot, bytes, var) do { int rc_ = pooled(c, slot, bytes, &var); if (rc_) return rc_; } while (0)

struct vet_result {
    int device = 0;                      // the result may outlive its context: only the device id is kept
    void* d[2] = {nullptr, nullptr};     // 0: assign / pairs, 1: weights / srccount
    size_t row_bytes[2] = {0, 0};
    int64_t rows = 0;
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
        if (hipMalloc(&res->d[0], a_bytes ? a_bytes : 8) != hipSuccess || hipMalloc(&res->d[1], b_bytes ? b_bytes : 8) != hipSuccess) {
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
    HIP_TRY(hipMemcpy(h_dst, (const char*)r->d[which] + (size_t)row0 * r->row_bytes[which], (size_t)n_rows * r->row_bytes[which],
                      hipMemcpyDeviceToHost));
    return VET_OK;
}

int vet_result_free(vet_result* r) {
    if (!r) return VET_OK;
    (void)hipSetDevice(r->device);
    for (void* q : r->d) if (q) (void)hipFree(q);
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
