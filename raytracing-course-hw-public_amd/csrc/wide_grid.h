// wide_grid.h — the scene-wide grids of the packed wide node (WideGrid, rt_device_types.h), shared by the host collapse (wide_build.cpp), the
// device collapse (rt_bvh_device.hip: host code computes the grid, the emission kernel snaps with it) and the pack pass (rt_wide_pack.hip).
//
// Node origins. The 80-byte WideNode stores a node's lower corner as three floats; the packed node has 20 bits per axis for it. So the
// builders SNAP every origin down to base + m * g (m < 2^20, g a power of two) before they quantise the children against it: the child boxes
// are then conservative with respect to the snapped origin (floor / ceil on the node's own cell grid, as before), and packing is a lossless
// re-encoding. g is chosen so that (a) 2^20 - 2 steps span the scene's largest extent and (b) every multiple of g inside the scene is an
// exactly representable float (g >= one ulp of the largest coordinate magnitude, times two): origin = fma(float(m), g, base) is then exact
// on the device. With 20 bits an origin moves by less than 1e-6 of the scene's extent, far below a leaf node's own cell size unless the
// scene spans more than ~2^20 leaf cells per axis.
// Cell exponents. A node's cells are 2^e wide, e = ceil(log2(extent / 255)). The packed node keeps e - e_base in 4 bits, e_base = (largest
// e any node can have) - 15: a node whose extent would ask for a finer cell gets 2^e_base (coarser cells are always valid: fewer of the 255
// steps are used). 2^-15 of the root's cell is below float resolution of the scene's own coordinates, so nothing representable is lost.
#pragma once
#include <cmath>
#include <cstdint>

#include "rt_device_types.h"

namespace rt {

inline WideGrid make_wide_grid(const float lo[3], const float hi[3]) {
    WideGrid G{};
    double ext = 0.0, mag = 0.0;
    for (int c = 0; c < 3; ++c) {
        ext = std::fmax(ext, (double)hi[c] - (double)lo[c]);
        mag = std::fmax(mag, std::fmax(std::fabs((double)lo[c]), std::fabs((double)hi[c])));
    }
    // (non-finite bounds cannot reach the device: rt_create refuses such scenes; the raw host entry point rt_bvh_wide_build_host must merely
    // stay memory safe and terminate on them, so every loop below is bounded and every conversion guarded)
    if (!std::isfinite(ext))
        ext = 3.0e38;
    if (!std::isfinite(mag))
        mag = 3.0e38;
    int k = -120;
    if (ext > 0.0) {
        k = (int)std::ceil(std::log2(ext / (double)((1u << RT_WIDE_ORIGIN_BITS) - 4u)));
        while (k < 120 && std::ldexp((double)((1u << RT_WIDE_ORIGIN_BITS) - 4u), k) < ext)
            ++k;
    }
    if (mag > 0.0) { // every multiple of g up to `mag` must be a float: g >= 2^(exponent(mag) - 22)
        int em;
        (void)std::frexp(mag, &em); // mag = f * 2^em, f in [0.5, 1)
        k = std::max(k, em - 23);
    }
    k = std::min(std::max(k, -120), 120);
    G.g = (float)std::ldexp(1.0, k);
    for (int c = 0; c < 3; ++c) {
        const double l = std::isfinite(lo[c]) ? (double)lo[c] : -3.0e38;
        G.base[c] = (float)(std::floor(l / (double)G.g) * (double)G.g); // exact: |lo| / g < 2^23
    }
    // largest cell exponent any node can need: the scene box measured from `base` (one g more than the true extent at most)
    int e_top = -126;
    for (int c = 0; c < 3; ++c) {
        double e = (double)hi[c] - (double)G.base[c];
        if (!std::isfinite(e))
            e = 6.0e38;
        if (e > 0.0) {
            int ee = (int)std::ceil(std::log2(e / 255.0));
            while (ee < 126 && std::ldexp(255.0, ee) < e)
                ++ee;
            e_top = std::max(e_top, ee);
        }
    }
    e_top = std::min(std::max(e_top, -111), 126);
    G.e_base = e_top - 15;
    return G;
}

} // namespace rt

// Snap a lower bound to the origin grid: the largest base + m * g that is <= lo (m clamped to the 20-bit range). Host and device.
#ifdef __HIPCC__
__host__ __device__
#endif
inline float wide_snap_origin(const WideGrid &G, int axis, float lo, uint32_t *m_out) {
    double m = floor(((double)lo - (double)G.base[axis]) / (double)G.g);
    const double m_max = (double)((1u << RT_WIDE_ORIGIN_BITS) - 1u);
    m = m > 0.0 ? (m > m_max ? m_max : m) : 0.0; // (NaN -> 0)
    if (m_out)
        *m_out = (uint32_t)m;
    return (float)((double)G.base[axis] + m * (double)G.g); // exact by construction of g
}
// The cell exponent (unbiased) of a node axis of extent `ext` (measured from the snapped origin), clamped to what 4 bits above e_base hold.
#ifdef __HIPCC__
__host__ __device__
#endif
inline int wide_cell_exponent(const WideGrid &G, double ext) {
    int e = G.e_base;
    if (ext > 0.0 && ext < 1.0e300) { // (a non-finite extent keeps e_base: see make_wide_grid)
        e = (int)ceil(log2(ext / 255.0));
        while (e < 127 && ldexp(255.0, e) < ext) // rounding of log2: the grid must span the box
            ++e;
    }
    e = e < G.e_base ? G.e_base : e;
    return e > G.e_base + 15 ? G.e_base + 15 : e; // (> e_base + 15 cannot happen for a node inside the scene box)
}
