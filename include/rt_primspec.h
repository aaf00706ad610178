/*
 * rt_primspec.h — ray intersection of the analytic primitives of the scene-txt front end (rt_primitive_desc, rt_abi.h),
 * written once so that the HIP kernels and the CPU oracle evaluate the same IEEE-754 operation sequence (like
 * rt_devspec.h: +,-,*,/ and sqrt on float only; -ffp-contract=off on both compilers).
 *
 * Status: the reference at HEAD has no analytic primitive (using shape = triangle, geometry.h:505). The ellipsoid test
 * follows its unused intersect_ray_sphere (raytracer.h:61-77: ray scaled by the radii, half-b quadratic, both roots),
 * the plane test is the textbook one. "Parity unpinned" against the reference; oracle == GPU bit for bit.
 *
 * A hit is accepted for the nearest root t >= min_dst (as intersect(ray, triangle, min_dst), bvh.h:52-65); the returned
 * normal faces the ray (flipped when the ray starts inside / behind), as to_intersection_info flips the triangle's
 * (bvh.h:86-87, 118).
 */
#ifndef RT_PRIMSPEC_H
#define RT_PRIMSPEC_H

#include "rt_abi.h"
#include "rt_devspec.h"

/* v rotated by the quaternion q = (x, y, z, w):  v + 2 w (u x v) + 2 u x (u x v),  u = (x, y, z) */
RT_HD void rt_quat_rotate(float qx, float qy, float qz, float qw, const float v[3], float out[3]) {
    const float cx = qy * v[2] - qz * v[1], cy = qz * v[0] - qx * v[2], cz = qx * v[1] - qy * v[0];
    const float dx = qy * cz - qz * cy, dy = qz * cx - qx * cz, dz = qx * cy - qy * cx;
    out[0] = v[0] + 2.0f * (qw * cx) + 2.0f * dx;
    out[1] = v[1] + 2.0f * (qw * cy) + 2.0f * dy;
    out[2] = v[2] + 2.0f * (qw * cz) + 2.0f * dz;
}

RT_HD float rt_sqrtf(float x) { return __builtin_sqrtf(x); }

/* 1 = hit: *t_out = distance along d, n_out = unit normal facing the ray. NaNs make every comparison false -> miss. */
RT_HD int rt_prim_intersect(const rt_primitive_desc *p, const float o[3], const float d[3], float min_dst, float *t_out, float n_out[3]) {
    if (p->kind == RT_PRIM_ELLIPSOID) {
        const float qx = p->rotation[0], qy = p->rotation[1], qz = p->rotation[2], qw = p->rotation[3];
        const float rel[3] = {o[0] - p->position[0], o[1] - p->position[1], o[2] - p->position[2]};
        float lo[3], ld[3];
        rt_quat_rotate(-qx, -qy, -qz, qw, rel, lo); /* into the primitive's frame: conjugate rotation */
        rt_quat_rotate(-qx, -qy, -qz, qw, d, ld);
        const float sx = lo[0] / p->param[0], sy = lo[1] / p->param[1], sz = lo[2] / p->param[2]; /* ray.start / r */
        const float ex = ld[0] / p->param[0], ey = ld[1] / p->param[1], ez = ld[2] / p->param[2]; /* ray.dir / r */
        const float a = ex * ex + ey * ey + ez * ez;
        const float hb = sx * ex + sy * ey + sz * ez;
        const float c = (sx * sx + sy * sy + sz * sz) - 1.0f;
        const float hd2 = hb * hb - a * c;
        if (!(hd2 >= 0.0f))
            return 0;
        const float hd = rt_sqrtf(hd2);
        const float t1 = (-hb - hd) / a, t2 = (-hb + hd) / a;
        float t;
        int inside;
        if (t1 >= min_dst && t1 <= 3.4028234663852886e38f) {
            t = t1;
            inside = 0;
        } else if (t2 >= min_dst && t2 <= 3.4028234663852886e38f) {
            t = t2;
            inside = 1;
        } else {
            return 0;
        }
        /* gradient of (x/rx)^2 + (y/ry)^2 + (z/rz)^2 at the hit point, in the primitive's frame */
        float g[3] = {(lo[0] + ld[0] * t) / p->param[0] / p->param[0], (lo[1] + ld[1] * t) / p->param[1] / p->param[1], (lo[2] + ld[2] * t) / p->param[2] / p->param[2]};
        const float gl = rt_sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        g[0] = g[0] / gl;
        g[1] = g[1] / gl;
        g[2] = g[2] / gl;
        float n[3];
        rt_quat_rotate(qx, qy, qz, qw, g, n);
        n_out[0] = inside ? -n[0] : n[0];
        n_out[1] = inside ? -n[1] : n[1];
        n_out[2] = inside ? -n[2] : n[2];
        *t_out = t;
        return 1;
    }
    if (p->kind == RT_PRIM_PLANE) {
        const float nl = rt_sqrtf(p->param[0] * p->param[0] + p->param[1] * p->param[1] + p->param[2] * p->param[2]);
        const float n[3] = {p->param[0] / nl, p->param[1] / nl, p->param[2] / nl};
        const float dn = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];
        const float h = (p->position[0] - o[0]) * n[0] + (p->position[1] - o[1]) * n[1] + (p->position[2] - o[2]) * n[2];
        const float t = h / dn;
        if (!(t >= min_dst && t <= 3.4028234663852886e38f))
            return 0;
        const int behind = dn > 0.0f;
        n_out[0] = behind ? -n[0] : n[0];
        n_out[1] = behind ? -n[1] : n[1];
        n_out[2] = behind ? -n[2] : n[2];
        *t_out = t;
        return 1;
    }
    return 0;
}

#endif /* RT_PRIMSPEC_H */
