// vet_spatial_u.hpp — k_spatial_u / k_spatial_u_lds: nearest-tile (unweighted) and naive-grid spatial entropy, an HBM stream
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_spatial_sweep.hpp"

namespace vet {

// ------------------------------------------------------------------------------------------
// k_spatial_u — the same entropy with use_weight_distribution = False: every present user adds
// weight 1.0 to its nearest tile (entropy_utils.py:139-142), so the frame histogram is an integer
// count per tile and the path is a pure stream: 16 B in, LUT gather, 4 B out per sample.
// LDS: cnt u32 [FPW][n].  Wave w owns frames w, w+NW, ... of the workgroup's FPW frames.
// ------------------------------------------------------------------------------------------
template <bool FROM_IDS>
__global__ void k_spatial_u(const SpatialParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* cnt = (unsigned*)smem;                                             // [FPW][n]
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const long f0 = (long)blockIdx.x * p.FPW;
    const int nf = (int)min((long)p.FPW, (long)p.T - f0);
    for (int i = tid; i < p.FPW * p.n; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    bool bad = false;
    const long base = f0 * (long)p.U, total = (long)nf * p.U;
    if (!FROM_IDS && (p.U & 1) == 0) {
        // 16-byte loads: a lane takes two neighbouring users; four such pairs are in flight.
        // (frame, pair-in-frame) of the flat pair index is tracked incrementally: no division
        // in the loop.
        constexpr int UN = 4;
        const double2* mu2 = (const double2*)(p.src.mu + base);
        const double2* mv2 = (const double2*)(p.src.mv + base);
        int2* out2 = (int2*)(p.assign ? p.assign + base : nullptr);
        const int ppf = p.U >> 1;                                   // pairs per frame
        const int pairs = nf * ppf;
        const int dq = (int)blockDim.x / ppf, dr = (int)blockDim.x % ppf;
        int fl0 = tid / ppf, j0 = tid % ppf;
        for (int i0 = tid; i0 < pairs; i0 += UN * (int)blockDim.x) {
            double2 a[UN], b[UN];
            int fl[UN];
            int fk = fl0, jk = j0;
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int i = i0 + k * (int)blockDim.x;
                fl[k] = fk;
                if (i < pairs) { a[k] = mu2[i]; b[k] = mv2[i]; }
                fk += dq; jk += dr;
                if (jk >= ppf) { jk -= ppf; ++fk; }
            }
            fl0 = fk; j0 = jk;
            int near[UN][2];
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                near[k][0] = near[k][1] = -1;
                if (i0 + k * (int)blockDim.x < pairs) {
                    const int id0 = grid_dir(a[k].x, b[k].x, p.src.W, p.src.H, bad);
                    const int id1 = grid_dir(a[k].y, b[k].y, p.src.W, p.src.H, bad);
                    if (id0 >= 0) near[k][0] = p.nearest[id0];
                    if (id1 >= 0) near[k][1] = p.nearest[id1];
                }
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int i = i0 + k * (int)blockDim.x;
                if (i < pairs) {
                    unsigned* row = cnt + (size_t)fl[k] * p.n;
                    if (near[k][0] >= 0) atomicAdd(&row[near[k][0]], 1u);
                    if (near[k][1] >= 0) atomicAdd(&row[near[k][1]], 1u);
                    if (out2) out2[i] = make_int2(near[k][0], near[k][1]);
                }
            }
        }
    } else {
        for (long i = tid; i < total; i += blockDim.x) {
            const int fl = (int)(i / p.U);
            const long idx = base + i;
            const int id = sample_dir<FROM_IDS>(p.src, idx, bad);
            int near = -1;
            if (id >= 0) {
                near = p.nearest[id];
                atomicAdd(&cnt[(size_t)fl * p.n + near], 1u);
            }
            if (p.assign) __builtin_nontemporal_store(near, p.assign + idx);
        }
    }
    __syncthreads();
    for (int fl = wv; fl < nf; fl += NW) {
        const unsigned* row = cnt + (size_t)fl * p.n;
        int np = 0;
        for (int t = lane; t < p.n; t += WAVE) np += (int)row[t];
        np = wave_sum(np);
        const double tw = (double)np;             // total_weight == number of present users
        double h = 0.0;
        for (int t = lane; t < p.n; t += WAVE) {
            const unsigned v = row[t];
            if (v) {
                const double q = (double)v / tw;
                h -= q * log2(q);
            }
            if (p.weights) __builtin_nontemporal_store((double)v, p.weights + (f0 + fl) * (long)p.n + t);
        }
        h = wave_sum(h);
        if (lane == 0) {
            double hmax = p.hmax;                  // entropy_utils.py:201-206
            if (!(tw > (double)p.norm_n) && !p.full_norm) {
                const double mp = 1.0 / tw;
                hmax = -tw * mp * log2(mp);
            }
            double e = h / hmax;
            if (np == 0) {
                e = __builtin_nan("");
                if (p.status) atomicAdd(&p.status[1], 1);
            }
            p.ent_k[f0 + fl] = e;
            if (p.present) p.present[f0 + fl] = np;
        }
    }
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

// ------------------------------------------------------------------------------------------
// k_spatial_u_lds — k_spatial_u for plans whose nearest-tile LUT fits the LDS (40 KB at the
// default 100x200 grid).  Measured on MI355X (tools/stream_probe.hip): the 2-byte LUT gather
// from global memory runs at about one lane per cycle per CU and costs 70 us of a 177 us
// kernel, while the same stream with the LUT in LDS reaches 5.1 TB/s.  So: persistent
// workgroups (1024 threads, 2 per CU) load the LUT into LDS once and walk the frame axis in
// blocks of FB frames (FB * U/2 <= 2048 sample pairs, two pairs per thread, 16-byte loads).
// Per sample: 16 B in, one ds_read_u16, one ds_add_u32, 4 B out.  Frame f of a round is reduced
// to its entropy by wave f.  Requires an even U <= 4096 and grid samples.
// Counts are integers <= U, so log2(v/N) is taken as lg[v] - lg[N] from an LDS table of log2(k),
// k = 1..U, copied from a per-context table (keeps ocml's log2 out of this kernel: 64 VGPRs, no spills).
// LDS: lut u16 [n_dirs] | lg f64 [U+1] | cnt u32 [FB][n]
// ------------------------------------------------------------------------------------------
// WEIGHTS: also write the per-frame tile counts (the analyzers' tile_weights).  PAIRS: 16-byte loads,
// two users per lane (even U); otherwise one user per lane with 8-byte loads, any U.
// BATCH: the launch covers a list of videos (p.videos); the single-video instantiation has none of that code
template <bool WEIGHTS, bool PAIRS, bool BATCH = false>
__global__ __launch_bounds__(1024, 8) void k_spatial_u_lds(const SpatialParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PPT = 2;
    const int NW = blockDim.x >> 6;
    const int FB = p.FPW;                                                        // frames per round
    const long D = p.src.n_dirs;
    uint16_t* lut = (uint16_t*)smem;
    double* lg = (double*)(smem + ((D * 2 + 15) & ~15L));                        // [U+1]
    unsigned* cnt = (unsigned*)(lg + p.U + 1);                                   // [FB][n]
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    for (long i = tid; i < (D + 1) / 2; i += blockDim.x) ((unsigned*)lut)[i] = ((const unsigned*)p.nearest)[i];
    for (int i = tid; i <= p.U; i += blockDim.x) lg[i] = p.log2_tab[i];
    for (int i = tid; i < FB * p.n; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    bool bad = false;
    // the video and frames of a block: the launch's only video, or one of a batch (vet_spatial_entropy_batch: block0 = the
    // video's first block, FPW = its frames per block; every video's U fits the launch's LDS tables)
    struct Blk { const double* mu; const double* mv; int U, ipf, nf; long f0; double* ent; int32_t* assign; int32_t* present; double* weights; float inv_ipf; };
    const long nblocks = BATCH ? (long)p.n_blocks : ((long)p.T + FB - 1) / FB;
    auto locate = [&](long blk) {
        Blk x;
        x.mu = p.src.mu; x.mv = p.src.mv; x.U = p.U; x.ent = p.ent_k; x.assign = p.assign; x.present = p.present; x.weights = p.weights;
        int T = p.T, fb = FB;
        if (BATCH) {
            int lo = 0, hi = p.n_videos - 1;                   // last video with block0 <= blk
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((long)p.videos[mid].block0 <= blk) lo = mid; else hi = mid - 1;
            }
            const VideoDesc& d = p.videos[lo];
            x.mu = d.mu; x.mv = d.mv; x.U = d.U; T = d.T; fb = d.FPW;
            x.ent = d.entropy; x.assign = d.assign; x.present = d.present; x.weights = nullptr;
            blk -= d.block0;
        }
        x.f0 = blk * fb;
        x.nf = (int)min((long)fb, (long)T - x.f0);
        x.ipf = PAIRS ? x.U >> 1 : x.U;                        // items (pairs or users) per frame
        x.inv_ipf = 1.0f / (float)x.ipf;
        return x;
    };
    // PAIRS: the next round's samples are requested before the barriers of this round (the barriers
    // wait for LDS traffic only, see lds_barrier), so HBM loads stay in flight while the waves
    // reduce the round's histograms.
    double2 a[PPT], b[PPT];
    if (PAIRS && (long)blockIdx.x < nblocks) {
        const Blk x = locate(blockIdx.x);
        const int nitems = x.nf * x.ipf;
        const double2* mu2 = (const double2*)(x.mu + x.f0 * (long)x.U);
        const double2* mv2 = (const double2*)(x.mv + x.f0 * (long)x.U);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * (int)blockDim.x;
            if (i < nitems) { a[k] = nt_load(mu2 + i); b[k] = nt_load(mv2 + i); }
        }
    }
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const Blk cur = locate(blk);
        const long f0 = cur.f0;
        const int nf = cur.nf, ipf = cur.ipf;
        const float inv_ipf = cur.inv_ipf;
        const int nitems = nf * ipf;
        if (PAIRS) {
            int2* out2 = (int2*)(cur.assign ? cur.assign + f0 * (long)cur.U : nullptr);
            int near[PPT][2];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int i = tid + k * (int)blockDim.x;
                near[k][0] = near[k][1] = -1;
                if (i < nitems) {
                    const int id0 = grid_dir(a[k].x, b[k].x, p.src.W, p.src.H, bad);
                    const int id1 = grid_dir(a[k].y, b[k].y, p.src.W, p.src.H, bad);
                    if (id0 >= 0) near[k][0] = (int)lut[id0];
                    if (id1 >= 0) near[k][1] = (int)lut[id1];
                }
            }
            // next round's loads go out ahead of this round's stores
            const long nb = blk + gridDim.x;
            if (nb < nblocks) {
                const Blk nx = locate(nb);
                const int nnext = nx.nf * nx.ipf;
                const double2* mu2 = (const double2*)(nx.mu + nx.f0 * (long)nx.U);
                const double2* mv2 = (const double2*)(nx.mv + nx.f0 * (long)nx.U);
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int i = tid + k * (int)blockDim.x;
                    if (i < nnext) { a[k] = nt_load(mu2 + i); b[k] = nt_load(mv2 + i); }
                }
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int i = tid + k * (int)blockDim.x;
                if (i < nitems) {
                    const int fl = (int)(((float)i + 0.5f) * inv_ipf);           // exact: i < 2^12
                    unsigned* row = cnt + (size_t)fl * p.n;
                    if (near[k][0] >= 0) atomicAdd(&row[near[k][0]], 1u);
                    if (near[k][1] >= 0) atomicAdd(&row[near[k][1]], 1u);
                    if (out2) nt_store(out2 + i, make_int2(near[k][0], near[k][1]));
                }
            }
        } else {
            const double* mu1 = cur.mu + f0 * (long)cur.U;
            const double* mv1 = cur.mv + f0 * (long)cur.U;
            int* out1 = cur.assign ? cur.assign + f0 * (long)cur.U : nullptr;
            double a[2 * PPT], b[2 * PPT];
#pragma unroll
            for (int k = 0; k < 2 * PPT; ++k) {
                const int i = tid + k * (int)blockDim.x;
                if (i < nitems) { a[k] = __builtin_nontemporal_load(mu1 + i); b[k] = __builtin_nontemporal_load(mv1 + i); }
            }
#pragma unroll
            for (int k = 0; k < 2 * PPT; ++k) {
                const int i = tid + k * (int)blockDim.x;
                if (i < nitems) {
                    const int fl = (int)(((float)i + 0.5f) * inv_ipf);           // exact: i < 2^13
                    const int id0 = grid_dir(a[k], b[k], p.src.W, p.src.H, bad);
                    const int n0 = id0 >= 0 ? (int)lut[id0] : -1;
                    if (n0 >= 0) atomicAdd(&cnt[(size_t)fl * p.n + n0], 1u);
                    if (out1) __builtin_nontemporal_store(n0, out1 + i);
                }
            }
        }
        lds_barrier();
        // entropy (entropy_utils.py:194-211): wave f reduces frame f and clears its histogram
        for (int f = wv; f < nf; f += NW) {
            unsigned* row = cnt + (size_t)f * p.n;
            int np = 0;                            // users present = histogram total (exact)
            for (int t = lane; t < p.n; t += WAVE) np += (int)row[t];
            np = wave_sum(np);
            const double tw = (double)np, lgn = lg[np], inv_tw = 1.0 / tw;
            double h = 0.0;
            double* wout = WEIGHTS ? cur.weights + (f0 + f) * (long)p.n : nullptr;
            for (int t = lane; t < p.n; t += WAVE) {
                const unsigned v = row[t];
                if (v) h -= ((double)v * inv_tw) * (lg[v] - lgn);
                if (WEIGHTS) __builtin_nontemporal_store((double)v, wout + t);
                row[t] = 0u;
            }
            h = wave_sum(h);
            if (lane == 0) {
                double hmax = p.hmax;              // entropy_utils.py:201-206
                if (!(tw > (double)p.norm_n) && !p.full_norm) hmax = -tw * (1.0 / tw) * -lgn;   // log2(1/N) = -log2 N
                double e = h / hmax;
                if (np == 0) {
                    e = __builtin_nan("");
                    if (p.status) atomicAdd(&p.status[1], 1);
                }
                cur.ent[f0 + f] = e;
                if (cur.present) cur.present[f0 + f] = np;
            }
        }
        lds_barrier();
    }
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

}  // namespace vet
