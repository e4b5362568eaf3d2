// Container shapes 7..14 of OBBConstraints.comp:144-296 (star prism, superellipsoid, trefoil
// tube, Moebius band, DNA double helix, heart tube, gyroid, coil).
//
// These run as their own pass (k_obb_ext) after the SPH pass instead of being fused into its
// epilogue: they need transcendental functions and a 48..128-sample search per particle, which
// would only add registers and code to the hot kernel that the box / round shapes never use.
//
// cos, atan(y,x) and pow are fully specified fp32 routines (same constants and operation order
// as the parity oracle, semantic 10), so results are identical on any device.  Curves the shader
// samples at fixed parameter values are tabulated once per dispatch on the host (sph_host.h
// shape_table) and read here.
#pragma once
#include "sph_device.h"

namespace sph {

struct ShapeTab {
    const float4* pts;     // sampled curve (xyz), shapes 9 / 11 / 12 / 14
    int count;
    float b0x, b0y, b0z;   // the shader's initial "best" point
};

__device__ __forceinline__ float sph_cosf(float x) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.549789948768648e-8f;
    float q = rintf(x * TWO_OVER_PI);
    float r = fmaf(q, -P1, x);
    r = fmaf(q, -P2, r);
    r = fmaf(q, -P3, r);
    int n = ((int)(q - 4.0f * floorf(q * 0.25f)) + 1) & 3;   // cos x = sin(x + pi/2)
    float r2 = r * r, res;
    if (n & 1) {
        float c = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
        c = fmaf(c, r2, 4.166664568298827e-2f);
        c = fmaf(c, r2, -0.5f);
        res = fmaf(c, r2, 1.0f);
    } else {
        float s = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
        s = fmaf(s, r2, -1.6666654611e-1f);
        s = s * r2;
        res = fmaf(s, r, r);
    }
    return (n & 2) ? -res : res;
}

__device__ __forceinline__ float sph_atanf(float x) {
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return sign * y;
}
__device__ __forceinline__ float sph_atan2f(float y, float x) {
    const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966f;
    if (x > 0.0f) return sph_atanf(y / x);
    if (x < 0.0f) return (y >= 0.0f) ? (sph_atanf(y / x) + PI) : (sph_atanf(y / x) - PI);
    return (y > 0.0f) ? PIO2 : ((y < 0.0f) ? -PIO2 : 0.0f);
}

__device__ __forceinline__ float sph_log2f(float x) {
    uint32_t ix = fbits(x);
    int e = (int)(ix - 0x3f3504f3u) >> 23;
    ix = ix - ((uint32_t)e << 23);
    float m = bitsf(ix);
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s, w = z * z;
    float t1 = w * fmaf(w, 0.24279078841f, 0.40000972152f);
    float t2 = z * fmaf(w, 0.28498786688f, 0.66666662693f);
    float R = t2 + t1;
    float hfsq = 0.5f * (f * f);
    float ln1pf = f - (hfsq - s * (hfsq + R));
    return fmaf(ln1pf, 1.44269504088896341f, (float)e);
}
__device__ __forceinline__ float sph_exp2f(float y) {
    y = fminf(fmaxf(y, -126.0f), 127.0f);
    float n = rintf(y);
    float r = y - n;
    float p = fmaf(r, 1.535336188319500e-4f, 1.339887440266574e-3f);
    p = fmaf(p, r, 9.618437357674640e-3f);
    p = fmaf(p, r, 5.550357186158072e-2f);
    p = fmaf(p, r, 2.402264791363012e-1f);
    p = fmaf(p, r, 6.931472028550421e-1f);
    p = fmaf(p, r, 1.0f);
    return p * bitsf((uint32_t)((int)n + 127) << 23);
}
__device__ __forceinline__ float sph_powf(float x, float p) {
    if (!(x > 0.0f)) return (p > 0.0f) ? 0.0f : ((p == 0.0f) ? 1.0f : __builtin_inff());
    return sph_exp2f(p * sph_log2f(x));
}

// Nearest tabulated sample, then radial projection onto the tube of radius r (:192-201 etc.).
__device__ __forceinline__ bool tube_project(const ShapeTab& T, float r, float px, float py, float pz,
                                             float& qx, float& qy, float& qz, float& nx, float& ny, float& nz) {
    float bx = T.b0x, by = T.b0y, bz = T.b0z, bestD2 = 1e30f;
    for (int k = 0; k < T.count; ++k) {
        const float4 c = T.pts[k];
        const float d0 = px - c.x, d1 = py - c.y, d2v = pz - c.z;
        const float d2 = dot3(d0, d1, d2v, d0, d1, d2v);
        if (d2 < bestD2) { bestD2 = d2; bx = c.x; by = c.y; bz = c.z; }
    }
    const float dx = px - bx, dy = py - by, dz = pz - bz;
    const float dl = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
    if (dl > r) {
        const float m = fmaxf(dl, 1e-6f);
        nx = dx / m; ny = dy / m; nz = dz / m;
        qx = bx + nx * r; qy = by + ny * r; qz = bz + nz * r;
        return true;
    }
    return false;
}

__device__ __forceinline__ bool shape_project_ext(const SimK& k, const ShapeTab& T, float px, float py, float pz,
                                                  float& qx, float& qy, float& qz, float& nx, float& ny, float& nz) {
    qx = px; qy = py; qz = pz; nx = ny = nz = 0.0f;
    switch (k.shape) {
    case 7: {                                               // star prism :144-163
        const float R = k.bhx, H = k.bhy;
        const float pts = fmaxf(3.0f, k.auxx), depth = clampf(k.auxy, 0.0f, 0.9f);
        const float yC = clampf(py, -H, H);
        const float ang = sph_atan2f(pz, px);
        const float rMax = R * (1.0f - depth * (0.5f + 0.5f * sph_cosf(pts * ang)));
        const float lxz = sqrtf(fmaf(pz, pz, px * px));
        float cx = px, cz = pz;
        if (lxz > rMax) { const float s = rMax / fmaxf(lxz, 1e-6f); cx = px * s; cz = pz * s; }
        qx = cx; qy = yC; qz = cz;
        const float ex = px - qx, ey = py - qy, ez = pz - qz;
        const float dl = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
        if (dl > 1e-6f) { nx = ex / dl; ny = ey / dl; nz = ez / dl; return true; }
        return false;
    }
    case 8: {                                               // superellipsoid :164-179
        const float a = fmaxf(k.bhx, 1e-6f), b = fmaxf(k.bhy, 1e-6f);
        const float n = clampf(k.auxz, 0.6f, 8.0f);
        const float ux = fabsf(px) / a, uy = fabsf(py) / b, uz = fabsf(pz) / a;
        const float F = (sph_powf(ux, n) + sph_powf(uy, n)) + sph_powf(uz, n);
        if (F > 1.0f) {
            const float s = sph_powf(F, -1.0f / n);
            qx = px * s; qy = py * s; qz = pz * s;
            const float n1 = n - 1.0f;
            const float gx = (signf(px) * sph_powf(fmaxf(fabsf(qx) / a, 1e-6f), n1)) / a;
            const float gy = (signf(py) * sph_powf(fmaxf(fabsf(qy) / b, 1e-6f), n1)) / b;
            const float gz = (signf(pz) * sph_powf(fmaxf(fabsf(qz) / a, 1e-6f), n1)) / a;
            const float gl = sqrtf(dot3(gx, gy, gz, gx, gy, gz));
            nx = gx / gl; ny = gy / gl; nz = gz / gl;
            return true;
        }
        return false;
    }
    case 9: case 11: case 12: case 14:                      // trefoil :180-202, DNA :224-241, heart :242-257, coil :282-296
        return tube_project(T, k.bhy, px, py, pz, qx, qy, qz, nx, ny, nz);
    case 10: {                                              // Moebius band :203-223
        const float R = k.bhx, wHalf = k.bhy, tHalf = fmaxf(k.auxx, 0.05f);
        const float phi = sph_atan2f(pz, px);
        const float cp = sph_cosf(phi), sp = sph_sinf(phi);
        const float c0 = R * cp, c2 = R * sp;
        const float psi = 0.5f * phi;
        const float cps = sph_cosf(psi), sps = sph_sinf(psi);
        const float w0 = cps * cp, w1 = sps, w2 = cps * sp;
        const float t0 = (-sps) * cp, t1 = cps, t2 = (-sps) * sp;
        const float o0 = px - c0, o1 = py - 0.0f, o2 = pz - c2;
        const float cu = clampf(dot3(o0, o1, o2, w0, w1, w2), -wHalf, wHalf);
        const float cv = clampf(dot3(o0, o1, o2, t0, t1, t2), -tHalf, tHalf);
        qx = (c0 + cu * w0) + cv * t0; qy = (0.0f + cu * w1) + cv * t1; qz = (c2 + cu * w2) + cv * t2;
        const float ex = px - qx, ey = py - qy, ez = pz - qz;
        const float dl = sqrtf(dot3(ex, ey, ez, ex, ey, ez));
        if (dl > 1e-5f) { nx = ex / dl; ny = ey / dl; nz = ez / dl; return true; }
        return false;
    }
    case 13: {                                              // gyroid :258-281
        const float R = k.bhx, sc = fmaxf(k.auxx, 0.1f), th = clampf(k.auxy, 0.2f, 2.5f);
        const float lp = sqrtf(dot3(px, py, pz, px, py, pz));
        if (lp > R) {
            const float m = fmaxf(lp, 1e-6f);
            nx = px / m; ny = py / m; nz = pz / m;
            qx = nx * R; qy = ny * R; qz = nz * R;
            return true;
        }
        const float ax = px * sc, ay = py * sc, az = pz * sc;
        const float sx = sph_sinf(ax), cx = sph_cosf(ax), sy = sph_sinf(ay), cy = sph_cosf(ay), sz = sph_sinf(az), cz = sph_cosf(az);
        const float g = (sx * cy + sy * cz) + sz * cx;
        if (fabsf(g) > th) {
            const float g0 = sc * (cx * cy - sz * sx), g1 = sc * ((-sx) * sy + cy * cz), g2 = sc * ((-sy) * sz + cz * cx);
            const float gl = fmaxf(sqrtf(dot3(g0, g1, g2, g0, g1, g2)), 1e-5f);
            const float sg = signf(g), step = (fabsf(g) - th) / gl;
            nx = (sg * g0) / gl; ny = (sg * g1) / gl; nz = (sg * g2) / gl;
            qx = px - nx * step; qy = py - ny * step; qz = pz - nz * step;
            return true;
        }
        return false;
    }
    default: return false;
    }
}

// OBBConstraints.comp main() for one non-ghost particle and a shape of this file.
__device__ __forceinline__ void obb_apply_ext(const SimK& k, const ShapeTab& T, float& px, float& py, float& pz,
                                              float& vx, float& vy, float& vz) {
    float lx, ly, lz, qx, qy, qz, nx, ny, nz;
    obb_to_local(k, px, py, pz, lx, ly, lz);
    if (shape_project_ext(k, T, lx, ly, lz, qx, qy, qz, nx, ny, nz)) obb_respond(k, qx, qy, qz, nx, ny, nz, px, py, pz, vx, vy, vz);
}

}  // namespace sph
