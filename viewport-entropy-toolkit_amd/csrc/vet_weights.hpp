// vet_weights.hpp — the FoV weight of a (direction, tile) pair and the per-frame entropy of a weighted histogram
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_common.hpp"

namespace vet {

// ------------------------------------------------------------------------------------------
// FoV weight of one (direction, tile) pair from their cosine
// calculate_tile_weights, entropy_utils.py:124-137:  d = arccos(clip(c)); if d < max:
//   w = ((max - d) / max) ** power.  Returned in 64-bit fixed point: w * 2^(52 - shift).
// ------------------------------------------------------------------------------------------
struct WeightCfg {
    double max_ang;     // np.radians(fov/2)
    double inv_max;     // 1 / max_ang
    double power;
    int shift;          // fixed point = 2^(52-shift); shift = max(0, ceil(log2 U) - 10)
};

// WMODE: 0 generic (ocml acos, pow)   1 fast acos, power == 2   2 fast acos, power == 1
// The fast acos needs max_ang <= 60 deg (fov <= 120): then c >= 0.5 - 1e-9 and
//   theta = 2 asin(s), s = sqrt(z), z = (1 - c)/2 <= 0.2502,
//   asin(s) = s + s z P(z), P of degree 9 fitted on [0, 0.2502]: |d theta| / theta < 2e-14.
__device__ __forceinline__ double fast_theta(double c) {
    const double z = fmax((1.0 - c) * 0.5, 1e-300);
    // sqrt(z): hardware rsq seed, one Goldschmidt step and one residual correction
    const double y = __builtin_amdgcn_rsq(z);
    double s = z * y, h = 0.5 * y;
    const double e = fma(-h, s, 0.5);
    s = fma(s, e, s);
    h = fma(h, e, h);
    s = fma(fma(-s, s, z), h, s);
    double P = 2.80476016723745745e-02;
    P = fma(P, z, -3.09562448984870928e-03);
    P = fma(P, z, 1.57475990547630423e-02);
    P = fma(P, z, 1.31700206864407612e-02);
    P = fma(P, z, 1.74440881411108591e-02);
    P = fma(P, z, 2.23658455433679397e-02);
    P = fma(P, z, 3.03821932887589595e-02);
    P = fma(P, z, 4.46428521871264916e-02);
    P = fma(P, z, 7.50000000381451232e-02);
    P = fma(P, z, 1.66666666666618335e-01);
    const double a = fma(s * z, P, s);
    return a + a;
}

// w in [0, 1] -> round(w * 2^52) through the mantissa of 1 + w, then >> shift
__device__ __forceinline__ unsigned long long unit_to_fx(double w, int shift) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(1.0 + w);
    return (bits - 0x3FF0000000000000ull) >> shift;
}

// the reference's FoV test and weight, evaluated as the reference does (entropy_utils.py:124-137).  True when
// distance < max: the tile is then a key of the reference's dict WHATEVER the weight — a power factor large
// enough makes (..) ** power underflow to exactly 0.0, the key stays (and 0 * log2 0 makes the frame NaN, :195-198).
__device__ __forceinline__ bool fov_weight_cone(double c, const WeightCfg& w, double& wt) {
    c = fmin(fmax(c, -1.0), 1.0);
    const double d = acos(c);
    wt = 0.0;
    if (!(d < w.max_ang)) return false;
    wt = pow((w.max_ang - d) / w.max_ang, w.power);
    return true;
}
__device__ __forceinline__ double fov_weight_exact(double c, const WeightCfg& w) {
    double wt;
    (void)fov_weight_cone(c, w, wt);
    return wt;
}

// A tile with distance < max is a key of the reference's dict however small its weight (entropy_utils.py:131-136):
// the fixed-point weight of an in-FoV tile is at least one unit (the truncation error stays below one unit, as the
// error bound of the sweep assumes), so "histogram slot != 0" is exactly "key".  0 = outside the FoV.
template <int WMODE>
__device__ __forceinline__ unsigned long long fov_weight_fx(double c, const WeightCfg& w) {
    double r;
    if (WMODE == 0) {
        c = fmin(fmax(c, -1.0), 1.0);
        const double d = acos(c);
        if (!(d < w.max_ang)) return 0ull;
        r = pow((w.max_ang - d) / w.max_ang, w.power);
    } else {
        r = (w.max_ang - fast_theta(c)) * w.inv_max;
        if (r < 1e-9) {            // at the rim of the cone the reference's own test decides (rare: |d - max| < 1e-9 rad)
            const double d = acos(fmin(fmax(c, -1.0), 1.0));
            if (!(d < w.max_ang)) return 0ull;
            r = (w.max_ang - d) / w.max_ang;
        }
        if (WMODE == 1) r = r * r;
    }
    const unsigned long long fx = unit_to_fx(r, w.shift);
    return fx ? fx : 1ull;
}

// ------------------------------------------------------------------------------------------
// Entropy of the workgroup's frames from their fixed-point tile histograms
// (entropy_utils.py:194-211, weighted mode: normaliser log2(n)).  Wave w takes frames w, w+NW, ...
// ------------------------------------------------------------------------------------------
template <typename HT>
__device__ __forceinline__ void weighted_frame_entropy(const HT* hist, const int* cnt_frame, int nf,
                                                       long f0, int n, double inv_unit, double hmax,
                                                       double* ent_k, double* weights, int32_t* present,
                                                       int32_t* status) {
    const int NW = blockDim.x >> 6, lane = lane_id(), wv = wave_id();
    for (int fl = wv; fl < nf; fl += NW) {
        const HT* hrow = hist + (size_t)fl * n;
        // total weight: up to U*n/4 fixed-point units, which can exceed 64 bits, so it is summed
        // in FP64 (fixed lane order + butterfly => still a pure function of the histogram)
        double totd = 0.0;
        for (int t = lane; t < n; t += WAVE) totd += (double)hrow[t];
        totd = wave_sum(totd);
        double h = 0.0;
        for (int t = lane; t < n; t += WAVE) {
            const HT v = hrow[t];
            if (v != (HT)0) {
                const double q = (double)v / totd;
                h -= q * log2(q);
            }
            if (weights) __builtin_nontemporal_store((double)v * inv_unit, weights + (f0 + fl) * (long)n + t);
        }
        h = wave_sum(h);
        if (lane == 0) {
            const int np = cnt_frame[fl];
            double e = h / hmax;
            if (np == 0) {
                e = __builtin_nan("");
                if (status) atomicAdd(&status[1], 1);
            }
            ent_k[f0 + fl] = e;
            if (present) present[f0 + fl] = np;
        }
    }
}

}  // namespace vet
