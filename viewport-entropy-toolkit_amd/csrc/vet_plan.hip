// vet_plan.hip — the device tables of a plan (C-ABI: include/vet.h): direction table, unit lattices, one nearest-tile LUT
// per lattice, and — built on first use — the alias table (directions with the same Vector / mirror images share a
// row), the row statistics with the proven error bounds, the direction weight tables (per lattice, fused).  Also the
// parity read-back hooks, vet_angular_distances and the tile boundary geometry.  No CPU compute path.
#include "vet_host.hpp"
#include "vet_plan_kernels.hpp"
#include "vet_weight_table.hpp"
#include "vet_geometry.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>

namespace vh {

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
    vet_ctx* c = pl->ctx;
    const long D = (long)pl->n_dirs;
    if (D <= 0 || D >= (long)0x7FFFFFFF) return fail(VET_ERR_UNSUPPORTED, "direction table of %ld entries", D);
    // mirror symmetry of every lattice, bit for bit on the unit vectors the kernels use (host: a few thousand values)
    bool mirror = !c->tune.no_mirror && pl->weighted;
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
    // the set, the classes, the mirror partners and the dense row numbering on the device (vet_plan_kernels.hpp: k_alias_*,
    // k_canon_*), on the null stream: ensure_alias is synchronous, like the host map of rounds 2-5 it replaces
    unsigned long long slots = 64;
    while (slots < 2ull * (unsigned long long)D) slots <<= 1;
    const int n_tiles = (int)((D + vet::CANON_TILE - 1) / vet::CANON_TILE);
    DevBuf set, a0, a1, tcount, toff, rowid;
    HIP_TRY(set.alloc(slots * 4));
    HIP_TRY(a0.alloc((size_t)D * 4));
    HIP_TRY(a1.alloc((size_t)D * 4));
    HIP_TRY(tcount.alloc((size_t)n_tiles * 4));
    HIP_TRY(toff.alloc(((size_t)n_tiles + 1) * 4));
    HIP_TRY(rowid.alloc((size_t)D * 4));
    uint32_t* d_alias = nullptr;
    int* d_canon = nullptr;
    HIP_TRY(hipMemset(set.p, 0xFF, slots * 4));
    const int grid = grid_for(D, 256, c->n_cu);
    hipLaunchKernelGGL(vet::k_alias_insert, dim3(grid), dim3(256), 0, 0, (const double*)pl->d_dir_raw, D, (uint32_t*)set.p, slots - 1);
    hipLaunchKernelGGL(vet::k_alias_lookup, dim3(grid), dim3(256), 0, 0, (const double*)pl->d_dir_raw, D, (const uint32_t*)set.p,
                       slots - 1, (uint32_t*)a0.p);
    const uint32_t* classes = (const uint32_t*)a0.p;
    if (mirror) {
        hipLaunchKernelGGL(vet::k_alias_mirror, dim3(grid), dim3(256), 0, 0, (const double*)pl->d_dir_raw, D, (const uint32_t*)set.p,
                           slots - 1, (uint32_t*)a0.p);
        hipLaunchKernelGGL(vet::k_alias_resolve, dim3(grid), dim3(256), 0, 0, (const uint32_t*)a0.p, D, (uint32_t*)a1.p);
        classes = (const uint32_t*)a1.p;
    }
    hipLaunchKernelGGL(vet::k_canon_count, dim3(n_tiles), dim3(256), 0, 0, classes, D, (uint32_t*)tcount.p);
    hipLaunchKernelGGL(vet::k_canon_scan, dim3(1), dim3(1024), 0, 0, (const uint32_t*)tcount.p, n_tiles, (uint32_t*)toff.p);
    HIP_TRY(hipGetLastError());
    uint32_t n_rows = 0;
    HIP_TRY(hipMemcpy(&n_rows, (const uint32_t*)toff.p + n_tiles, 4, hipMemcpyDeviceToHost));
    if (n_rows == 0) return fail(VET_ERR_DEVICE, "alias table: no canonical direction");
    hipError_t e = hipMalloc((void**)&d_alias, (size_t)D * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_canon, (size_t)n_rows * sizeof(int));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(vet::k_canon_fill, dim3(n_tiles), dim3(256), 0, 0, classes, D, (const uint32_t*)toff.p, d_canon, (uint32_t*)rowid.p);
        hipLaunchKernelGGL(vet::k_alias_rows, dim3(grid), dim3(256), 0, 0, classes, D, (const uint32_t*)rowid.p, d_alias);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();           // complete before the temporaries go and before any stream reads the tables
    if (e != hipSuccess) {
        if (d_alias) (void)hipFree(d_alias);
        if (d_canon) (void)hipFree(d_canon);
        return fail(VET_ERR_DEVICE, "alias table build failed: %s", hipGetErrorString(e));
    }
    pl->d_alias = d_alias;
    pl->d_canon = d_canon;
    pl->n_rows = (int)n_rows;
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
    // rows start on 128-byte lines (u16 tile rows) / 256 bytes (u32 weight rows); whole 64-entry blocks
    const int align = c->tune.stride_align;
    int stride = ((longest > 0 ? longest : 1) + align - 1) / align * align;
    const size_t rows = (size_t)R + 1;            // one extra, all-zero row (index n_rows) for the gather's idle lanes
    const size_t bytes = rows * stride * 6 + rows * 4;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = kMaxTableBytes;
    // (the gather addresses entries with 32-bit offsets: fewer than 2^32 of them)
    if (stride > 65535 || bytes > kMaxTableBytes || rows * (size_t)stride >= ((size_t)1 << 32) || bytes + ((size_t)64 << 20) > free_b) { L.stride = -1; return VET_OK; }
    // whatever goes wrong below, no half-built table stays behind (a retry starts from nothing)
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
    struct DropOnError { decltype(drop)& d; bool armed = true; ~DropOnError() { if (armed) d(); } } guard{drop};
    // 4 entries per lane and 2 rows in flight per group; lanes per row (part of the row layout) = the
    // smallest power of two whose 4-entry chunks cover the longest row, at most 16 (measured best for
    // long rows, profiles/r01/v4_table_vs_xcd_partition_sweep.log), so the short rows of small
    // lattices do not idle most of a group
    L.gs_log2 = 1;
    while (L.gs_log2 < 4 && (4 << L.gs_log2) < longest) ++L.gs_log2;
    if (c->tune.gs_log2) L.gs_log2 = c->tune.gs_log2;
    // 16-lane rows with at least one block that is 3/4 full get the class-dealt layout (k_wtab)
    L.interleaved = L.gs_log2 == 4 && stride % 64 == 0 && 4 * longest >= 3 * 64 && c->tune.tab_interleave != 0;
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
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_dirrec, dim3(grid_for(pl->n_dirs, 256, c->n_cu)), dim3(256), 0, s, pl->d_alias,
                           L.d_nearest, L.d_tab_meta, (long)pl->n_dirs, pl->d_dirrec);
        HIP_TRY(hipGetLastError());
    }
    // the table is complete before this returns: a later call may run on another stream (first use only)
    int markers = 0;
    HIP_TRY(hipMemcpyAsync(&markers, d_max + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    guard.armed = false;
    L.markers = markers;
    L.stride = stride;
    return VET_OK;
}

// Exact FP64 weight rows of lattice 0 for the weights pass (vet_host.hpp: WeightsCore::Exact).  Rows = the plan's
// canonical directions (ensure_alias: a direction and its mirror image share a row; the dot products are identical bit
// for bit, so are the weights).  10 bytes per entry; capped at a quarter of the free device memory and 8 GB.
int ensure_exact_weights(vet_plan* pl, hipStream_t s) {
    WeightsCore::Exact& X = pl->wcore->ex;
    if (X.state != 0) return VET_OK;
    vet_ctx* c = pl->ctx;
    const Lattice& L = pl->lat[0];
    if (!pl->weighted || L.binned || !L.d_tiles || c->tune.no_exact_rows) { X.state = -1; return VET_OK; }
    int rc = ensure_alias(pl);
    if (rc) return rc;
    const long R = pl->n_rows;
    if (R <= 0) { X.state = -1; return VET_OK; }
    DevBuf d_max;
    HIP_TRY(d_max.alloc(sizeof(int)));
    HIP_TRY(hipMemsetAsync(d_max.p, 0, sizeof(int), s));
    vet::WtabParams p{};
    p.dir_unit = pl->d_dir_unit; p.D = R;
    p.canon = pl->d_canon; p.shift_by_dir = 1; p.nl = 0;
    p.tiles = L.d_tiles; p.n = L.n;
    p.cos_cull = pl->cos_cull;
    p.wc.max_ang = pl->max_ang; p.wc.inv_max = 1.0 / pl->max_ang; p.wc.power = pl->power; p.wc.shift = 0;
    p.maxcount = (int*)d_max.p; p.gs_log2 = -1;
    const int blocks = grid_for(R * vet::WAVE, 256, c->n_cu * 2);
    hipLaunchKernelGGL(vet::k_wtab<false>, dim3(blocks), dim3(256), 0, s, p);       // longest row (conservative cone test)
    HIP_TRY(hipGetLastError());
    int longest = 0;
    HIP_TRY(hipMemcpyAsync(&longest, d_max.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int stride = ((longest > 0 ? longest : 1) + 63) / 64 * 64;
    const size_t D = (size_t)pl->n_dirs, entries = (size_t)R * stride, bytes = entries * 10 + (size_t)R * 4 + D * 4;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    if (bytes > ((size_t)8 << 30) || bytes > free_b / 4 || stride > 65535) { X.state = -1; return VET_OK; }
    auto dev_free = [](void* q) { if (q) (void)hipFree(q); };
    auto dev_alloc = [&](size_t b) {
        void* q = nullptr;
        if (hipMalloc(&q, b ? b : 8) != hipSuccess) { (void)hipGetLastError(); q = nullptr; }
        return std::shared_ptr<void>(q, dev_free);
    };
    auto alias = dev_alloc(D * 4), idx = dev_alloc(entries * 2), w = dev_alloc(entries * 8), len = dev_alloc((size_t)R * 4);
    // out of memory at the first request: decided ONCE for the plan, like "does not fit" above — the precise sweep serves
    // every later weights request and every result of this plan (a retry that succeeded later would switch paths between an
    // eager weights call and the fetch of the same frames: same values, different last bits)
    if (!alias || !idx || !w || !len) { X.state = -1; return VET_OK; }
    HIP_TRY(hipMemcpyAsync(alias.get(), pl->d_alias, D * 4, hipMemcpyDeviceToDevice, s));
    vet::WexactParams q{};
    q.dir_unit = pl->d_dir_unit; q.canon = pl->d_canon; q.R = R; q.tiles = L.d_tiles; q.n = L.n;
    q.cos_cull = pl->cos_cull; q.wc = p.wc; q.stride = stride;
    q.idx = (uint16_t*)idx.get(); q.w = (double*)w.get(); q.len = (uint32_t*)len.get();
    {
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_wexact, dim3(blocks), dim3(256), 0, s, q);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));          // complete before this returns: a later fetch may run on another stream
    X.alias = alias; X.idx = idx; X.w = w; X.len = len;
    X.stride = stride; X.n_rows = (int)R;
    X.state = 1;
    return VET_OK;
}

bool any_binned(const vet_plan* pl) {
    for (const auto& L : pl->lat) if (L.binned) return true;
    return false;
}

// Builds the plan's fused table (first use; synchronises).  state = -1: not usable for this plan (a property of the
// plan, decided once); a device failure on the way leaves state = 0 and nothing allocated, so a later call may retry.
int ensure_fused(vet_plan* pl, hipStream_t s) {
    auto& F = pl->fused;
    if (F.state != 0) return VET_OK;
    vet_ctx* c = pl->ctx;
    const int K = (int)pl->lat.size();
    auto unusable = [&]() { F.state = -1; return VET_OK; };
    // one lattice: a fused row is the lattice's own row — nothing to share, and the per-lattice epilogue is a little
    // cheaper (clustered audience, config-3 shape: 0.48 vs 0.51 ms); VET_FUSED=1 fuses such plans too
    if (K == 1 && !c->tune.fused_single) return unusable();
    if (!pl->weighted || K > vet::MAX_LATTICES || any_binned(pl) || c->tune.no_fused) return unusable();
    if ((uint64_t)pl->n_dirs > vet::DEDUP_MAX_DIRS) return unusable();
    int rc = ensure_all_stats(pl, s);
    if (rc) return rc;
    if (pl->ultra) return unusable();
    for (const auto& L : pl->lat)
        if (!(L.crit_tab <= kContractMargin) || !L.d_row_s) return unusable();
    rc = ensure_alias(pl);
    if (rc) return rc;
    const size_t D = (size_t)pl->n_dirs;
    const int R = pl->n_rows;                  // canonical directions, densely numbered (ensure_alias)
    if (R == 0) return unusable();
    vet::FusedLayout& lay = F.lay;
    lay.K = K; lay.Hs = 0; lay.CF = 0;
    for (int k = 0; k < K; ++k) {
        lay.n[k] = pl->lat[k].n; lay.off[k] = 2 * K + lay.Hs; lay.Hs += pl->lat[k].n >> 1; lay.hmax[k] = pl->lat[k].hmax;
        lay.CF += (pl->lat[k].n + vet::WAVE - 1) / vet::WAVE;
    }
    lay.N = 2 * (lay.Hs + K) + 4 * K;
    if (lay.N > 65535 || lay.N < 32) return unusable();

    // every buffer of the fused table is released again unless the build completes
    auto drop = [&]() {
        for (void** q : {(void**)&F.d_row_s, (void**)&F.d_w, (void**)&F.d_i, (void**)&F.d_meta, (void**)&F.d_dirrec})
            if (*q) { (void)hipFree(*q); *q = nullptr; }
    };
    struct DropOnExit { decltype(drop)& d; bool armed = true; ~DropOnExit() { if (armed) d(); } } guard{drop};
    DevBuf ptrs_d, delta_d, max_d;
    HIP_TRY(ptrs_d.alloc(sizeof(void*) * vet::MAX_LATTICES));
    HIP_TRY(delta_d.alloc(sizeof(int) * vet::MAX_LATTICES));
    HIP_TRY(max_d.alloc(sizeof(int)));
    HIP_TRY(hipMalloc((void**)&F.d_row_s, (size_t)R + 1));
    const uint8_t* ptrs[vet::MAX_LATTICES] = {};
    for (int k = 0; k < K; ++k) ptrs[k] = pl->lat[k].d_row_s;
    HIP_TRY(hipMemcpyAsync(ptrs_d.p, ptrs, sizeof(ptrs), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(delta_d.p, 0, sizeof(int) * vet::MAX_LATTICES, s));
    HIP_TRY(hipMemsetAsync(max_d.p, 0, sizeof(int), s));
    HIP_TRY(hipMemsetAsync(F.d_row_s + R, vet::TAB_X, 1, s));
    hipLaunchKernelGGL(vet::k_fuse_shifts, dim3(grid_for(R, 256, c->n_cu)), dim3(256), 0, s, (const int*)pl->d_canon, R, K,
                       (const uint8_t* const*)ptrs_d.p, F.d_row_s, (int*)delta_d.p);
    vet::WtabParams p{};
    p.dir_unit = pl->d_dir_unit; p.D = R;
    p.tiles = nullptr; p.n = 0;
    p.cos_cull = pl->cos_cull;
    p.wc.max_ang = pl->max_ang; p.wc.inv_max = 1.0 / pl->max_ang; p.wc.power = pl->power; p.wc.shift = 0;
    p.stride = 0; p.w = nullptr; p.idx = nullptr; p.meta = nullptr; p.row_s = F.d_row_s; p.row_e = nullptr; p.fp = 0;
    p.markers = nullptr; p.maxcount = (int*)max_d.p; p.gs_log2 = -1;
    p.canon = pl->d_canon; p.shift_by_dir = 0; p.nl = K; p.Hs = lay.Hs; p.N = lay.N;
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
        if (!(bound <= kContractMargin)) return unusable();
    }
    if (longest >= 65536) return unusable();
    const int stride = ((longest > 0 ? longest : 1) + 63) / 64 * 64;
    const size_t rows = (size_t)R + 1;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = kMaxTableBytes;
    const size_t bytes = rows * stride * 6;
    if (bytes > kMaxTableBytes || rows * (size_t)stride >= ((size_t)1 << 32) || bytes + ((size_t)64 << 20) > free_b) return unusable();
    if (hipMalloc((void**)&F.d_w, rows * stride * 4) != hipSuccess || hipMalloc((void**)&F.d_i, rows * stride * 2) != hipSuccess ||
        hipMalloc((void**)&F.d_meta, rows * 4) != hipSuccess || hipMalloc((void**)&F.d_dirrec, D * sizeof(uint2)) != hipSuccess) {
        (void)hipGetLastError();               // out of memory today: the per-lattice tables / the sweep still work, a later call retries
        return VET_OK;
    }
    F.gs_log2 = 1;
    while (F.gs_log2 < 4 && (4 << F.gs_log2) < longest) ++F.gs_log2;
    // rows of 65..96 entries: three 32-entry blocks of an 8-lane group instead of two 64-entry blocks, the second mostly
    // empty (config 4: 88-94 entries: 128 -> 96 slots per row walk)
    if (longest > 64 && longest <= 96 && c->tune.fused_narrow) F.gs_log2 = 3;
    if (c->tune.gs_log2) F.gs_log2 = c->tune.gs_log2;
    // class-dealt blocks (k_wtab): 16-lane rows with a block of 64 at least 3/4 full, 8-lane rows with one of 32
    F.interleaved = ((F.gs_log2 == 4 && 4 * longest >= 3 * 64) || (F.gs_log2 == 3 && 4 * longest >= 3 * 32 && c->tune.narrow_deal)) &&
                    c->tune.tab_interleave != 0;
    p.stride = stride; p.w = F.d_w; p.idx = F.d_i; p.meta = F.d_meta; p.maxcount = nullptr;
    p.gs_log2 = F.interleaved ? F.gs_log2 : -1;
    {
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_wtab<true>, dim3(blocks), dim3(256), 0, s, p);
    }
    HIP_TRY(hipGetLastError());
    {
        ProfScope ps(c, s, KID_WTAB);
        hipLaunchKernelGGL(vet::k_dirrec, dim3(grid_for((long)D, 256, c->n_cu)), dim3(256), 0, s, (const uint32_t*)pl->d_alias,
                           (const uint16_t*)pl->lat[0].d_nearest, (const uint32_t*)F.d_meta, (long)D, F.d_dirrec);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    guard.armed = false;
    F.R = R; F.stride = stride;
    F.state = 1;
    return VET_OK;
}

}  // namespace vh

using namespace vh;

extern "C" {

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
        int rc = spatial_set_attrs(c);
        if (!rc) rc = transition_set_attrs(c);
        if (rc) return cleanup(rc);
        c->attrs_set = true;
    }
#undef PLAN_TRY
    {   // the weights-only pass (vet_host.hpp: WeightsCore) shares the direction table and lattice 0's tiles
        auto dev_free = [](void* q) { if (q) (void)hipFree(q); };
        pl->wcore = std::make_shared<vh::WeightsCore>();
        vh::WeightsCore& w = *pl->wcore;
        w.device = c->device; w.lds_max = c->lds_max; w.n_cu = c->n_cu;
        w.dir_unit = std::shared_ptr<void>((void*)pl->d_dir_unit, dev_free);
        if (pl->lat[0].d_tiles) w.tiles0 = std::shared_ptr<void>((void*)pl->lat[0].d_tiles, dev_free);
        w.n0 = pl->lat[0].n; w.n_dirs = pl->n_dirs;
        w.cos_cull = pl->cos_cull; w.max_ang = pl->max_ang; w.power = pl->power;
    }
    *out = pl;
    return VET_OK;
}

int vet_plan_destroy(vet_plan* pl) {
    if (!pl) return VET_OK;
    (void)hipSetDevice(pl->ctx->device);
    (void)hipStreamSynchronize(pl->ctx->stream);
    if (pl->d_dir_raw) (void)hipFree(pl->d_dir_raw);
    // d_dir_unit and lat[0].d_tiles belong to pl->wcore once the plan is complete (shared with device-resident results)
    if (pl->d_dir_unit && !(pl->wcore && pl->wcore->dir_unit)) (void)hipFree(pl->d_dir_unit);
    for (auto& L : pl->lat) {
        if (L.d_tiles && !(&L == &pl->lat[0] && pl->wcore && pl->wcore->tiles0)) (void)hipFree(L.d_tiles);
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
        for (void* q : {(void*)F.d_row_s, (void*)F.d_w, (void*)F.d_i, (void*)F.d_meta, (void*)F.d_dirrec})
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

int vet_plan_set_raw_weights(vet_plan* pl, int on) {
    if (!pl) return fail(VET_ERR_INVALID, "plan is NULL");
    pl->raw_weights = on != 0;
    return VET_OK;
}

int64_t vet_plan_table_rows(const vet_plan* pl) { return pl ? pl->n_rows : 0; }

int vet_plan_table_stride(const vet_plan* pl, int k) {
    if (!pl || k < 0 || k >= (int)pl->lat.size()) return 0;
    // plans on the fused table (one row per direction over all lattices) never build the per-lattice ones
    if (pl->lat[k].stride == 0 && pl->fused.state == 1) return pl->fused.stride;
    return pl->lat[k].stride;
}

// vector_angle_distance / find_angular_distances (utilities/entropy_utils.py:41-87): synchronous, host buffers
int vet_angular_distances(vet_ctx* c, const double* h_vectors, int64_t m, const double* h_tiles, int n, double* h_out) {
    if (!c || !h_vectors || !h_tiles || !h_out) return fail(VET_ERR_INVALID, "ctx, vectors, tiles or output is NULL");
    if (m <= 0 || n <= 0) return fail(VET_ERR_INVALID, "need m > 0 vectors and n > 0 tile centres");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    DevBuf vecs, tiles, out;
    HIP_TRY(vecs.alloc((size_t)m * 24));
    HIP_TRY(tiles.alloc((size_t)n * 24));
    HIP_TRY(out.alloc((size_t)m * n * 8));
    HIP_TRY(hipMemcpyAsync(vecs.p, h_vectors, (size_t)m * 24, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(tiles.p, h_tiles, (size_t)n * 24, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(vet::k_angular_distances, dim3(grid_for((long)m * n, 256, c->n_cu)), dim3(256), 0, s, (const double*)vecs.p,
                       (long)m, (const double*)tiles.p, n, (double*)out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_out, out.p, (size_t)m * n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return VET_OK;
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

}  // extern "C"
