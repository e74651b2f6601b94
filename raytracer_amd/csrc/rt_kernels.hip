// HIP kernels for gfx950 (MI355X): primary-ray generation, BVH closest-hit, hard-shadow rays,
// Blinn-Phong shading, mirror bounces, gamma.  Compiled with -ffp-contract=off.
//
// Numerics.  Box tests run in FP32 against boxes padded by 2^-18 * max(|origin|, scene radius); the
// padding exceeds every FP32 rounding error of the slab test by > 3x, so a box that could contain the
// closest hit is never skipped.  Everything that decides WHICH primitive is hit and HOW FAR - the leaf
// intersectors and the re-projected distance - is FP64 and follows the reference's operation order
// (s2899221/Code/mesh.cpp:159-280 cube, 371-514 cylinder, 597-681 sphere, 795-920 plane), so hit ids and
// distances equal the reference's brute-force one_pass_intersection (raytracer.cpp:143-157) bit for bit;
// equal distances resolve to the lowest id, as its strict '<' over ascending ids does.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

#define DEV __device__ __forceinline__

namespace {

struct RayD { double ox, oy, oz, dx, dy, dz, dlen; };

struct HitD {            // full surface record of one hit (shading inputs)
    double dist;
    double px, py, pz;
    double nx, ny, nz;
    double rx, ry, rz;   // mirror direction
};

DEV double dsqrt(double x) { return __builtin_sqrt(x); }

// Re-projected distance shared by all intersectors (mesh.cpp:254-260 and siblings).
DEV double ray_distance(const RayD& r, double px, double py, double pz) {
    const double dx = px - r.ox, dy = py - r.oy, dz = pz - r.oz;
    return (r.dlen > 1e-15) ? (dx*r.dx + dy*r.dy + dz*r.dz) / r.dlen : dsqrt(dx*dx + dy*dy + dz*dz);
}

// Mirror direction r = d - 2 (d.n) n, normalised (mesh.cpp:236-243).
template <bool FULL> DEV void finish(const RayD& r, double px, double py, double pz, double nx, double ny, double nz, HitD& h) {
    h.dist = ray_distance(r, px, py, pz);
    if (FULL) {
        const double ddn = r.dx*nx + r.dy*ny + r.dz*nz;
        const double tx = r.dx - 2.0 * ddn * nx, ty = r.dy - 2.0 * ddn * ny, tz = r.dz - 2.0 * ddn * nz;
        double len = dsqrt(tx*tx + ty*ty + tz*tz);
        if (len < 1e-15) len = 1.0;
        h.rx = tx/len; h.ry = ty/len; h.rz = tz/len;
        h.px = px; h.py = py; h.pz = pz; h.nx = nx; h.ny = ny; h.nz = nz;
    }
}

// World ray into the unit frame of a translated / rotated / scaled primitive (mesh.cpp:171-180).
// c: T[0..2], R[3..11] row-major, S[12..14], 1/S[15..17].
DEV void to_local(const double* c, const RayD& r, double ro[3], double rd[3]) {
    const double tx = r.ox - c[0], ty = r.oy - c[1], tz = r.oz - c[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = c[3 + i]*tx   + c[6 + i]*ty   + c[9 + i]*tz;      // row i of R^T
        const double b = c[3 + i]*r.dx + c[6 + i]*r.dy + c[9 + i]*r.dz;
        ro[i] = a * c[15 + i];
        rd[i] = b * c[15 + i];
    }
}

// Local hit back to the world (mesh.cpp:223-233): n ~ R S^-1 n_local, p = T + R S p_local.
template <bool FULL> DEV void from_local(const double* c, const RayD& r, const double nl[3], const double pl[3], HitD& h) {
    const double s0 = pl[0] * c[12], s1 = pl[1] * c[13], s2 = pl[2] * c[14];
    const double px = (c[3]*s0 + c[4]*s1 + c[5]*s2) + c[0];
    const double py = (c[6]*s0 + c[7]*s1 + c[8]*s2) + c[1];
    const double pz = (c[9]*s0 + c[10]*s1 + c[11]*s2) + c[2];
    double nx = 0, ny = 0, nz = 0;
    if (FULL) {
        const double m0 = nl[0] * c[15], m1 = nl[1] * c[16], m2 = nl[2] * c[17];
        const double wx = c[3]*m0 + c[4]*m1 + c[5]*m2;
        const double wy = c[6]*m0 + c[7]*m1 + c[8]*m2;
        const double wz = c[9]*m0 + c[10]*m1 + c[11]*m2;
        const double len = dsqrt(wx*wx + wy*wy + wz*wz);
        nx = wx/len; ny = wy/len; nz = wz/len;
    }
    finish<FULL>(r, px, py, pz, nx, ny, nz, h);
}

// Cube::intersect, mesh.cpp:159-280 (slabs on [-1,1]^3, exit face when the origin is inside).
template <bool FULL> DEV bool hit_cube(const double* c, const RayD& r, HitD& h) {
    double ro[3], rd[3];
    to_local(c, r, ro, rd);
    double tmin = -__builtin_inf(), tmax = __builtin_inf();
    double ne[3] = { 0, 0, 0 }, nx_[3] = { 0, 0, 0 };
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double o = ro[a], d = rd[a];
        if (__builtin_fabs(d) < 1e-12) {
            if (o < -1.0 || o > 1.0) return false;
            continue;
        }
        double t1 = (-1.0 - o) / d;
        double t2 = (1.0 - o) / d;
        double s1 = -1.0, s2 = 1.0;
        if (t1 > t2) { const double t = t1; t1 = t2; t2 = t; s1 = 1.0; s2 = -1.0; }
        if (t1 > tmin) { tmin = t1; if (FULL) { ne[0] = ne[1] = ne[2] = 0.0; ne[a] = s1; } }
        if (t2 < tmax) { tmax = t2; if (FULL) { nx_[0] = nx_[1] = nx_[2] = 0.0; nx_[a] = s2; } }
        if (tmin > tmax) return false;
    }
    if (tmax < 0.0) return false;
    const bool entering = (tmin >= 0.0);
    const double t = entering ? tmin : tmax;
    const double pl[3] = { ro[0] + t * rd[0], ro[1] + t * rd[1], ro[2] + t * rd[2] };
    double nl[3] = { 0, 0, 0 };
    if (FULL) { nl[0] = entering ? ne[0] : nx_[0]; nl[1] = entering ? ne[1] : nx_[1]; nl[2] = entering ? ne[2] : nx_[2]; }
    from_local<FULL>(c, r, nl, pl, h);
    return true;
}

// Cylinder::intersect, mesh.cpp:371-514 (unit cylinder x^2 + y^2 = 1, z in [-1,1], both caps).
template <bool FULL> DEV bool hit_cylinder(const double* c, const RayD& r, HitD& h) {
    double ro[3], rd[3];
    to_local(c, r, ro, rd);
    const double eps = 1e-12;
    double best_t = __builtin_inf();
    double bp[3] = { 0, 0, 0 }, bn[3] = { 0, 0, 0 };
    bool found = false;

    const double A = rd[0]*rd[0] + rd[1]*rd[1];
    const double B = 2.0 * (ro[0]*rd[0] + ro[1]*rd[1]);
    const double C = ro[0]*ro[0] + ro[1]*ro[1] - 1.0;
    if (A > eps) {
        const double disc = B*B - 4.0*A*C;
        if (disc >= 0.0) {
            const double sq = dsqrt(disc);
            const double ts[2] = { (-B - sq) / (2.0*A), (-B + sq) / (2.0*A) };
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double t = ts[k];
                if (t < 0.0) continue;
                const double z = ro[2] + t * rd[2];
                if (z < -1.0 - eps || z > 1.0 + eps) continue;
                if (t < best_t) {
                    best_t = t;
                    bp[0] = ro[0] + t*rd[0]; bp[1] = ro[1] + t*rd[1]; bp[2] = z;
                    if (FULL) {
                        const double len_xy = dsqrt(bp[0]*bp[0] + bp[1]*bp[1]);
                        if (len_xy > eps) { bn[0] = bp[0]/len_xy; bn[1] = bp[1]/len_xy; bn[2] = 0.0; }
                        else              { bn[0] = 0.0; bn[1] = 0.0; bn[2] = 1.0; }
                    }
                    found = true;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double zc = k ? -1.0 : 1.0;
        if (__builtin_fabs(rd[2]) < eps) continue;
        const double t = (zc - ro[2]) / rd[2];
        if (t < 0.0) continue;
        const double x = ro[0] + t * rd[0];
        const double y = ro[1] + t * rd[1];
        if (x*x + y*y <= 1.0 + 1e-9) {
            if (t < best_t) {
                best_t = t;
                bp[0] = x; bp[1] = y; bp[2] = zc;
                bn[0] = 0.0; bn[1] = 0.0; bn[2] = zc;
                found = true;
            }
        }
    }
    if (!found) return false;
    from_local<FULL>(c, r, bn, bp, h);
    return true;
}

// Sphere::intersect (axis-aligned ellipsoid), mesh.cpp:597-681.  c: C[0..2], S[3..5], 1/S[6..8].
template <bool FULL> DEV bool hit_sphere(const double* c, const RayD& r, HitD& h) {
    const double ro0 = (r.ox - c[0]) * c[6], ro1 = (r.oy - c[1]) * c[7], ro2 = (r.oz - c[2]) * c[8];
    const double rd0 = r.dx * c[6], rd1 = r.dy * c[7], rd2 = r.dz * c[8];
    const double A = rd0*rd0 + rd1*rd1 + rd2*rd2;
    const double B = 2.0 * (ro0*rd0 + ro1*rd1 + ro2*rd2);
    const double C = ro0*ro0 + ro1*ro1 + ro2*ro2 - 1.0;
    if (A < 1e-12) return false;
    const double disc = B*B - 4.0*A*C;
    if (disc < 0.0) return false;
    const double sq = dsqrt(disc);
    double t0 = (-B - sq) / (2.0*A);
    double t1 = (-B + sq) / (2.0*A);
    if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
    const double t = (t0 >= 0.0) ? t0 : t1;
    if (t < 0.0) return false;
    const double l0 = ro0 + t*rd0, l1 = ro1 + t*rd1, l2 = ro2 + t*rd2;
    const double px = c[0] + l0 * c[3], py = c[1] + l1 * c[4], pz = c[2] + l2 * c[5];
    double nx = 0, ny = 0, nz = 0;
    if (FULL) {
        const double wx = l0 * c[6], wy = l1 * c[7], wz = l2 * c[8];
        const double len = dsqrt(wx*wx + wy*wy + wz*wz);
        nx = wx/len; ny = wy/len; nz = wz/len;
    }
    finish<FULL>(r, px, py, pz, nx, ny, nz, h);
    return true;
}

// Plane::intersect (parallelogram P0 + u e1 + v e2), mesh.cpp:795-920, and the triangle extension
// (same frame, inside test u >= -eps, v >= -eps, u + v <= 1 + eps).
// c: W = [e1 e2 n]^-1 row-major [0..8], P0[9..11], e1[12..14], e2[15..17].
template <bool FULL, bool TRI> DEV bool hit_planar(const double* c, const RayD& r, HitD& h) {
    const double qx = r.ox - c[9], qy = r.oy - c[10], qz = r.oz - c[11];
    const double ro2 = c[6]*qx + c[7]*qy + c[8]*qz;
    const double rd2 = c[6]*r.dx + c[7]*r.dy + c[8]*r.dz;
    const double eps = 1e-12;
    if (__builtin_fabs(rd2) < eps) return false;
    const double t = -ro2 / rd2;
    if (t < 0.0) return false;
    const double ro0 = c[0]*qx + c[1]*qy + c[2]*qz, rd0 = c[0]*r.dx + c[1]*r.dy + c[2]*r.dz;
    const double ro1 = c[3]*qx + c[4]*qy + c[5]*qz, rd1 = c[3]*r.dx + c[4]*r.dy + c[5]*r.dz;
    const double u = ro0 + t * rd0;
    const double v = ro1 + t * rd1;
    if (TRI) { if (u < -eps || v < -eps || u + v > 1.0 + eps) return false; }
    else     { if (u < -eps || u > 1.0 + eps || v < -eps || v > 1.0 + eps) return false; }
    // p = P0 + M (u, v, 0) with M = [e1 e2 n] (mesh.cpp:849-853, 885).  The reference also adds n * 0.0,
    // which can only turn a -0.0 component into +0.0: no effect on any comparison, so the distance-only
    // path leaves the normal out.
    double nx = 0, ny = 0, nz = 0, zx = 0, zy = 0, zz = 0;
    if (FULL) {
        // n = (e1 x e2) / |e1 x e2| exactly as the host derived it for W
        const double cx = c[13]*c[17] - c[14]*c[16], cy = c[14]*c[15] - c[12]*c[17], cz = c[12]*c[16] - c[13]*c[15];
        const double nl = dsqrt(cx*cx + cy*cy + cz*cz);
        nx = cx/nl; ny = cy/nl; nz = cz/nl;
        zx = nx*0.0; zy = ny*0.0; zz = nz*0.0;
    }
    const double px = c[9]  + (c[12]*u + c[15]*v + zx);
    const double py = c[10] + (c[13]*u + c[16]*v + zy);
    const double pz = c[11] + (c[14]*u + c[17]*v + zz);
    finish<FULL>(r, px, py, pz, nx, ny, nz, h);
    return true;
}

// 16-byte loads of a record's doubles into registers.
template <int N> DEV void load_rec(const rt_prim_rec* rec, double (&c)[18]) {
    const double2* v = reinterpret_cast<const double2*>(rec);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { const double2 t = v[i]; c[2*i] = t.x; c[2*i + 1] = t.y; }
    if (N & 1) c[N - 1] = rec->d[N - 1];
}

template <bool FULL> DEV bool hit_record(const rt_prim_rec* rec, uint32_t kind, const RayD& r, HitD& h) {
    double c[18];
    switch (kind) {
        case RT_CUBE:     load_rec<18>(rec, c); return hit_cube<FULL>(c, r, h);
        case RT_SPHERE:   load_rec<9>(rec, c);  return hit_sphere<FULL>(c, r, h);
        case RT_PLANE:    load_rec<18>(rec, c); return hit_planar<FULL, false>(c, r, h);
        case RT_TRIANGLE: load_rec<18>(rec, c); return hit_planar<FULL, true>(c, r, h);
        case RT_CYLINDER: load_rec<18>(rec, c); return hit_cylinder<FULL>(c, r, h);
        default: return false;
    }
}

// ---- BVH traversal ---------------------------------------------------------------------------

struct BoxRay { float opx, opy, opz, omx, omy, omz, ix, iy, iz; };

DEV float f32_up(double x) {                       // smallest float >= max(x, 0); NaN -> +inf
    if (!(x == x)) return __builtin_inff();
    float f = static_cast<float>(x);
    if (static_cast<double>(f) < x) {               // step to the next float up
        const uint32_t u = __float_as_uint(f);
        f = __uint_as_float(f > 0.0f ? u + 1u : (f == 0.0f ? 1u : u - 1u));
    }
    return (f > 0.0f) ? f : 0.0f;
}

DEV void box_ray_init(BoxRay& b, const RayD& r, float radius) {
    const float ox = static_cast<float>(r.ox), oy = static_cast<float>(r.oy), oz = static_cast<float>(r.oz);
    const float m = fmaxf(fmaxf(radius, fabsf(ox)), fmaxf(fabsf(oy), fabsf(oz)));
    const float pad = ldexpf(m, -RT_PAD_SHIFT);
    b.opx = ox + pad; b.opy = oy + pad; b.opz = oz + pad;
    b.omx = ox - pad; b.omy = oy - pad; b.omz = oz - pad;
    b.ix = 1.0f / static_cast<float>(r.dx / r.dlen);
    b.iy = 1.0f / static_cast<float>(r.dy / r.dlen);
    b.iz = 1.0f / static_cast<float>(r.dz / r.dlen);
}

// Entry distance (>= 0) into the padded box, +inf on a miss or when the box starts beyond `limit`.
DEV float box_entry(const float4 lo, const float4 hi, const BoxRay& b, float limit) {
    const float ax = (lo.x - b.opx) * b.ix, bx = (hi.x - b.omx) * b.ix;
    const float ay = (lo.y - b.opy) * b.iy, by = (hi.y - b.omy) * b.iy;
    const float az = (lo.z - b.opz) * b.iz, bz = (hi.z - b.omz) * b.iz;
    float tn = fmaxf(0.0f, fminf(ax, bx));
    float tf = fminf(limit, fmaxf(ax, bx));
    tn = fmaxf(tn, fminf(ay, by)); tf = fminf(tf, fmaxf(ay, by));
    tn = fmaxf(tn, fminf(az, bz)); tf = fminf(tf, fmaxf(az, bz));
    return (tn <= tf) ? tn : __builtin_inff();
}

struct Best { double dist; uint32_t slot; int32_t id; };

// MODE 0: closest hit.  MODE 1: shadow verdict for a light at light_dist with the reference's rule
// "blocked iff the CLOSEST hit lies in (1e-4, light_dist - 1e-4)" (raytracer.cpp:230-235): hits at or
// beyond light_dist - 1e-4 are irrelevant; once one hit inside the interval is known only hits at
// <= 1e-4 can still change the verdict, so the search limit collapses to 1e-4; any such hit ends it.
// stack: this lane's column of the block's LDS stack (stride = blockDim.x words).
template <int MODE, bool COUNT>
DEV bool traverse(const DevScene& sc, const RayD& r, uint32_t* stack, uint32_t stride, double light_dist,
                  Best& best, uint32_t& vn, uint32_t& vp) {
    BoxRay br; box_ray_init(br, r, sc.radius);
    best.dist = __builtin_inf(); best.slot = 0xFFFFFFFFu; best.id = -1;
    double limit = (MODE == 0) ? __builtin_inf() : (light_dist - 1e-4);
    float limit_f = f32_up(limit);
    bool found_mid = false;

    auto test_slot = [&](uint32_t slot) -> bool {       // returns true when the walk can stop (MODE 1)
        const rt_prim_rec* rec = sc.recs + slot;
        const uint2 tag = *reinterpret_cast<const uint2*>(&rec->kind);
        if (COUNT) ++vp;
        HitD h;
        if (!hit_record<false>(rec, tag.x, r, h)) return false;
        if (MODE == 0) {
            const int32_t id = static_cast<int32_t>(tag.y);
            if (h.dist < best.dist || (h.dist == best.dist && id < best.id)) {
                best.dist = h.dist; best.slot = slot; best.id = id;
                limit = h.dist; limit_f = f32_up(limit);
            }
        } else {
            if (h.dist <= 1e-4) return true;
            if (h.dist < light_dist - 1e-4 && !found_mid) { found_mid = true; limit = 1e-4; limit_f = f32_up(limit); }
        }
        return false;
    };

    bool closed = false;                                 // MODE 1: a hit at <= 1e-4 was found
    if (sc.n_nodes) {
        const float4* nodes = reinterpret_cast<const float4*>(sc.nodes);
        uint32_t sp = 0;
        float4 lo = nodes[0], hi = nodes[1];
        if (COUNT) vn += 1;
        bool have = box_entry(lo, hi, br, limit_f) != __builtin_inff();
        for (;;) {
            if (have) {
                const uint32_t first = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
                if (count) {
                    for (uint32_t s = 0; s < count; ++s)
                        if (test_slot(first + s)) { closed = true; break; }
                    if (closed) break;
                    have = false;
                } else {
                    const float4 llo = nodes[2*first], lhi = nodes[2*first + 1];
                    const float4 rlo = nodes[2*first + 2], rhi = nodes[2*first + 3];
                    if (COUNT) vn += 2;
                    const float tl = box_entry(llo, lhi, br, limit_f), tr = box_entry(rlo, rhi, br, limit_f);
                    const bool hl = tl != __builtin_inff(), hr = tr != __builtin_inff();
                    if (hl && hr) {
                        if (tr < tl) { stack[sp * stride] = first;     ++sp; lo = rlo; hi = rhi; }
                        else         { stack[sp * stride] = first + 1; ++sp; lo = llo; hi = lhi; }
                    } else if (hl) { lo = llo; hi = lhi; }
                    else if (hr)   { lo = rlo; hi = rhi; }
                    else have = false;
                }
            }
            if (!have) {
                if (sp == 0) break;
                --sp;
                const uint32_t idx = stack[sp * stride];
                lo = nodes[2*idx]; hi = nodes[2*idx + 1];
                if (COUNT) vn += 1;
                have = box_entry(lo, hi, br, limit_f) != __builtin_inff();
            }
        }
    }
    // unbounded primitives (a zero or non-finite scale): tested against every ray
    if (!closed)
        for (uint32_t s = 0; s < sc.n_loose; ++s)
            if (test_slot(sc.n_slots + s)) { closed = true; break; }
    if (MODE == 0) return best.id >= 0;
    return found_mid && !closed;
}

// ---- shading -----------------------------------------------------------------------------------

DEV void normalize3(double x, double y, double z, double& ox, double& oy, double& oz) {
    const double len = dsqrt(x*x + y*y + z*z);
    if (len <= 1e-15) { ox = 0.0; oy = 0.0; oz = 1.0; return; }
    ox = x/len; oy = y/len; oz = z/len;
}
DEV double clampd(double v, double lo, double hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
DEV uint32_t round_u8(double c) {
    int ci = static_cast<int>(__builtin_round(c));
    ci = ci < 0 ? 0 : (ci > 255 ? 255 : ci);
    return static_cast<uint32_t>(ci);
}

struct ShadeCounters { uint32_t shadow_rays, vn, vp; };

// RayTracer::shade, raytracer.cpp:159-279; returns linear 8-bit colour packed r | g << 8 | b << 16.
template <bool COUNT>
DEV uint32_t shade(const DevScene& sc, const HitD& h, int32_t id, double view_ox, double view_oy, double view_oz,
                   uint32_t* stack, uint32_t stride, ShadeCounters& ctr) {
    if (sc.n_lights == 0) return 0;
    double nx, ny, nz; normalize3(h.nx, h.ny, h.nz, nx, ny, nz);
    double vx, vy, vz; normalize3(view_ox - h.px, view_oy - h.py, view_oz - h.pz, vx, vy, vz);
    const rt_mat_rec m = sc.mats[id];
    const double metallic = clampd(m.metallic, 0.0, 1.0);
    const double roughness = clampd(m.roughness, 0.001, 1.0);
    const double e_raw = (2.0 / (roughness * roughness)) - 2.0;
    const double phong_exp = (2.0 < e_raw) ? e_raw : 2.0;

    double lighting = 0.0;
    for (uint32_t li = 0; li < sc.n_lights; ++li) {
        const rt_light L = sc.lights[li];
        const double lx = L.pos[0] - h.px, ly = L.pos[1] - h.py, lz = L.pos[2] - h.pz;
        const double len2 = lx*lx + ly*ly + lz*lz;
        const double dist2 = (1e-12 < len2) ? len2 : 1e-12;
        double dx, dy, dz; normalize3(lx, ly, lz, dx, dy, dz);
        const double light_dist = dsqrt(dist2);
        const double eps = 1e-4;
        const double ndl = nx*dx + ny*dy + nz*dz;
        const double s = (ndl >= 0.0) ? 1.0 : -1.0;
        RayD sr;
        sr.ox = h.px + s * nx * eps; sr.oy = h.py + s * ny * eps; sr.oz = h.pz + s * nz * eps;
        sr.dx = dx; sr.dy = dy; sr.dz = dz;
        sr.dlen = dsqrt(dx*dx + dy*dy + dz*dz);
        Best b;
        ++ctr.shadow_rays;
        if (traverse<1, COUNT>(sc, sr, stack, stride, light_dist, b, ctr.vn, ctr.vp)) continue;

        const double n_dot_l = (0.0 < ndl) ? ndl : 0.0;
        const double diffuse = 1.0 * (1.0 - metallic) * n_dot_l;
        double hx, hy, hz; normalize3(dx + vx, dy + vy, dz + vz, hx, hy, hz);
        const double ndh = nx*hx + ny*hy + nz*hz;
        const double n_dot_h = (0.0 < ndh) ? ndh : 0.0;
        const double specular = (0.5 * 1.25) * pow(n_dot_h, phong_exp);
        const double attenuation = L.intensity / dist2;
        lighting += attenuation * (diffuse + specular);
    }
    double intensity = (1.0 - metallic) * 0.05 + lighting;
    if (intensity < 0.0) intensity = 0.0;
    if (intensity > 1.0) intensity = 1.0;
    return round_u8(intensity * static_cast<double>(m.albedo[0]))
         | round_u8(intensity * static_cast<double>(m.albedo[1])) << 8
         | round_u8(intensity * static_cast<double>(m.albedo[2])) << 16;
}

// Camera::pixelToRay, persp_camera.cpp:94-118 (effective sensor size precomputed on the host).
DEV void primary_ray(const DevCamera& cam, double px, double py, RayD& r) {
    const double nx = px / cam.width, ny = py / cam.height;
    const double xs = cam.eff_w * (nx - 0.5), ys = cam.eff_h * (0.5 - ny), zs = cam.focal;
    const double wx = cam.rot[0]*xs + cam.rot[1]*ys + cam.rot[2]*zs;
    const double wy = cam.rot[3]*xs + cam.rot[4]*ys + cam.rot[5]*zs;
    const double wz = cam.rot[6]*xs + cam.rot[7]*ys + cam.rot[8]*zs;
    const double len = dsqrt(wx*wx + wy*wy + wz*wz);
    r.ox = cam.pos[0]; r.oy = cam.pos[1]; r.oz = cam.pos[2];
    if (len == 0) { r.dx = r.dy = r.dz = 0.0; } else { r.dx = wx/len; r.dy = wy/len; r.dz = wz/len; }
    r.dlen = dsqrt(r.dx*r.dx + r.dy*r.dy + r.dz*r.dz);
}

// ---- pixel enumeration -------------------------------------------------------------------------

DEV uint32_t compact3(uint32_t v) {        // bits 0,2,4 -> 0,1,2
    return (v & 1u) | ((v >> 1) & 2u) | ((v >> 2) & 4u);
}

// Blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of logical blocks so that
// the tiles sharing an L2 are neighbours in the image.
DEV uint32_t xcd_contiguous(uint32_t b, uint32_t nb) {
    const uint32_t q = nb >> 3, rem = nb & 7u, x = b & 7u;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + (b >> 3);
}

struct Pixel { int x, y; uint32_t out; bool valid; };

// One 256-thread block covers 16x16 pixels: wave w owns the 8x8 quadrant (w & 1, w >> 1), lanes in
// Z-order inside it.  Logical blocks run in Z-order inside 8x8-block super tiles (128x128 pixels).
DEV Pixel map_pixel(const PixelMap& pm) {
    const uint32_t L = xcd_contiguous(blockIdx.x, gridDim.x);
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    const int lx = static_cast<int>((wave & 1u) * 8u + compact3(lane));
    const int ly = static_cast<int>((wave >> 1) * 8u + compact3(lane >> 1));
    Pixel p; p.valid = false; p.x = 0; p.y = 0; p.out = 0;
    if (pm.mode == 0) {
        const uint32_t st = L >> 6, in = L & 63u;
        const uint32_t bx = (st % pm.super_x) * 8u + compact3(in), by = (st / pm.super_x) * 8u + compact3(in >> 1);
        const int rx = static_cast<int>(bx) * 16 + lx, ry = static_cast<int>(by) * 16 + ly;
        if (rx < pm.w && ry < pm.h) {
            p.valid = true; p.x = pm.x0 + rx; p.y = pm.y0 + ry;
            p.out = static_cast<uint32_t>(ry) * static_cast<uint32_t>(pm.w) + static_cast<uint32_t>(rx);
        }
    } else {
        const uint32_t bpt_x = static_cast<uint32_t>(pm.tile) / 16u, bpt = bpt_x * bpt_x;
        const uint32_t lt = L / bpt, in = L % bpt;
        const uint32_t gt = lt * static_cast<uint32_t>(pm.n_ranks) + static_cast<uint32_t>(pm.rank);
        if (gt < static_cast<uint32_t>(pm.n_tiles)) {
            const int tx = static_cast<int>(gt % static_cast<uint32_t>(pm.tiles_x)) * pm.tile;
            const int ty = static_cast<int>(gt / static_cast<uint32_t>(pm.tiles_x)) * pm.tile;
            const int ix = static_cast<int>(in % bpt_x) * 16 + lx, iy = static_cast<int>(in / bpt_x) * 16 + ly;
            if (tx + ix < pm.w && ty + iy < pm.h) {
                p.valid = true; p.x = pm.x0 + tx + ix; p.y = pm.y0 + ty + iy;
                p.out = lt * static_cast<uint32_t>(pm.tile * pm.tile) + static_cast<uint32_t>(iy * pm.tile + ix);
            }
        }
    }
    return p;
}

struct BlockCounters { unsigned long long v[8]; };

DEV void flush_counters(unsigned long long* lds_ctr, rt_counters* out) {
    __syncthreads();
    if (threadIdx.x < 8 && lds_ctr[threadIdx.x])
        atomicAdd(reinterpret_cast<unsigned long long*>(out) + threadIdx.x, lds_ctr[threadIdx.x]);
}

} // namespace

// ---- kernels -------------------------------------------------------------------------------------

// Primary rays over a pixel map: closest hit ids and distances.
template <bool COUNT>
__global__ void __launch_bounds__(RT_BLOCK) k_trace_primary(DevScene sc, DevCamera cam, PixelMap pm,
                                                            int32_t* __restrict__ ids, double* __restrict__ dist,
                                                            rt_counters* ctr) {
    __shared__ uint32_t stack[RT_STACK_DEPTH * RT_BLOCK];
    __shared__ unsigned long long lctr[8];
    if (threadIdx.x < 8) lctr[threadIdx.x] = 0;
    __syncthreads();
    const Pixel p = map_pixel(pm);
    if (p.valid) {
        RayD r; primary_ray(cam, static_cast<double>(p.x) + 0.5, static_cast<double>(p.y) + 0.5, r);
        Best b; uint32_t vn = 0, vp = 0;
        const bool hit = traverse<0, COUNT>(sc, r, stack + threadIdx.x, RT_BLOCK, 0.0, b, vn, vp);
        if (ids) ids[p.out] = b.id;
        if (dist) dist[p.out] = b.dist;
        atomicAdd(&lctr[0], 1ull);
        if (hit) atomicAdd(&lctr[1], 1ull);
        if (COUNT) { atomicAdd(&lctr[4], static_cast<unsigned long long>(vn)); atomicAdd(&lctr[5], static_cast<unsigned long long>(vp)); }
    }
    flush_counters(lctr, ctr);
}

// Arbitrary rays (origin, direction as 6 doubles): closest hit, optionally shaded.
template <bool COUNT, bool SHADE>
__global__ void __launch_bounds__(RT_BLOCK) k_trace_rays(DevScene sc, const double* __restrict__ rays, uint64_t n,
                                                         int32_t* __restrict__ ids, double* __restrict__ dist,
                                                         uint8_t* __restrict__ rgb, rt_counters* ctr) {
    __shared__ uint32_t stack[RT_STACK_DEPTH * RT_BLOCK];
    __shared__ unsigned long long lctr[8];
    if (threadIdx.x < 8) lctr[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * RT_BLOCK + threadIdx.x;
    if (i < n) {
        RayD r;
        r.ox = rays[i*6]; r.oy = rays[i*6 + 1]; r.oz = rays[i*6 + 2];
        r.dx = rays[i*6 + 3]; r.dy = rays[i*6 + 4]; r.dz = rays[i*6 + 5];
        r.dlen = dsqrt(r.dx*r.dx + r.dy*r.dy + r.dz*r.dz);
        Best b; uint32_t vn = 0, vp = 0;
        const bool hit = traverse<0, COUNT>(sc, r, stack + threadIdx.x, RT_BLOCK, 0.0, b, vn, vp);
        if (ids) ids[i] = b.id;
        if (dist) dist[i] = b.dist;
        atomicAdd(&lctr[0], 1ull);
        if (hit) atomicAdd(&lctr[1], 1ull);
        if (COUNT) { atomicAdd(&lctr[4], static_cast<unsigned long long>(vn)); atomicAdd(&lctr[5], static_cast<unsigned long long>(vp)); }
        if (SHADE) {
            uint32_t c = 0;
            if (hit) {
                HitD h;
                const rt_prim_rec* rec = sc.recs + b.slot;
                hit_record<true>(rec, rec->kind, r, h);
                ShadeCounters sctr = { 0, 0, 0 };
                c = shade<COUNT>(sc, h, b.id, r.ox, r.oy, r.oz, stack + threadIdx.x, RT_BLOCK, sctr);
                atomicAdd(&lctr[2], static_cast<unsigned long long>(sctr.shadow_rays));
                if (COUNT) { atomicAdd(&lctr[6], static_cast<unsigned long long>(sctr.vn)); atomicAdd(&lctr[7], static_cast<unsigned long long>(sctr.vp)); }
            }
            rgb[i*3] = c & 255u; rgb[i*3 + 1] = (c >> 8) & 255u; rgb[i*3 + 2] = (c >> 16) & 255u;
        }
    }
    flush_counters(lctr, ctr);
}

// The pixel pipeline of render_unaccelerated_ppm at one sample per pixel (raytracer.cpp:401-438) with
// trace_ray_recursive (raytracer.cpp:281-371) unrolled into a loop: walk down the mirror chain keeping
// each level's local colour, Fresnel weight and tint, then fold the 8-bit blends back up.
template <bool COUNT, bool BOUNCE>
__global__ void __launch_bounds__(RT_BLOCK) k_render(DevScene sc, DevCamera cam, PixelMap pm, int max_scatters,
                                                     double min_scatter_intensity, uint8_t* __restrict__ rgb,
                                                     int32_t* __restrict__ ids, rt_counters* ctr) {
    __shared__ uint32_t stack[RT_STACK_DEPTH * RT_BLOCK];
    __shared__ unsigned long long lctr[8];
    if (threadIdx.x < 8) lctr[threadIdx.x] = 0;
    __syncthreads();
    const Pixel p = map_pixel(pm);
    if (p.valid) {
        uint32_t* my_stack = stack + threadIdx.x;
        RayD r; primary_ray(cam, static_cast<double>(p.x) + 0.5, static_cast<double>(p.y) + 0.5, r);
        ShadeCounters sctr = { 0, 0, 0 };
        uint32_t vn = 0, vp = 0, bounce_rays = 0;
        uint32_t colour = 0;

        // per-level state of the mirror chain (only touched when BOUNCE)
        double   lvl_fr[RT_MAX_SCATTERS];
        uint32_t lvl_local[RT_MAX_SCATTERS + 1];
        uint32_t lvl_tint[RT_MAX_SCATTERS];      // albedo | metallic flag << 24
        int last = 0;

        for (int depth = 0; ; ++depth) {
            Best b;
            const bool hit = traverse<0, COUNT>(sc, r, my_stack, RT_BLOCK, 0.0, b, vn, vp);
            if (depth == 0) {
                if (ids) ids[p.out] = b.id;
                atomicAdd(&lctr[0], 1ull);
                if (hit) atomicAdd(&lctr[1], 1ull);
            }
            uint32_t local = 0;
            HitD h;
            if (hit) {
                const rt_prim_rec* rec = sc.recs + b.slot;
                hit_record<true>(rec, rec->kind, r, h);
                local = shade<COUNT>(sc, h, b.id, r.ox, r.oy, r.oz, my_stack, RT_BLOCK, sctr);
            }
            if (!BOUNCE) { colour = local; break; }
            lvl_local[depth] = local; last = depth;
            if (depth >= max_scatters || !hit) break;

            const rt_mat_rec m = sc.mats[b.id];
            const double inv_len = (r.dlen > 1e-15) ? (1.0 / r.dlen) : 1.0;
            const double vx = -r.dx * inv_len, vy = -r.dy * inv_len, vz = -r.dz * inv_len;
            const double ct = h.nx*vx + h.ny*vy + h.nz*vz;
            const double cos_theta = (0.0 < ct) ? ct : 0.0;
            double f0;
            if (m.metallic <= 0.0) { const double r0 = (m.ior - 1.0) / (m.ior + 1.0); f0 = clampd(r0 * r0, 0.0, 1.0); }
            else {
                const double avg = (static_cast<double>(m.albedo[0]) + static_cast<double>(m.albedo[1]) + static_cast<double>(m.albedo[2])) / (3.0 * 255.0);
                f0 = clampd(avg, 0.0, 1.0);
            }
            const double fr = f0 + (1.0 - f0) * pow(1.0 - cos_theta, 5.0);
            if (fr <= 0.0 || fr < min_scatter_intensity) break;
            lvl_fr[depth] = fr;
            lvl_tint[depth] = static_cast<uint32_t>(m.albedo[0]) | static_cast<uint32_t>(m.albedo[1]) << 8
                            | static_cast<uint32_t>(m.albedo[2]) << 16 | (m.metallic > 0.0 ? 1u << 24 : 0u);
            const double eps = 1e-6;
            RayD nr;
            nr.ox = h.px + h.rx * eps; nr.oy = h.py + h.ry * eps; nr.oz = h.pz + h.rz * eps;
            nr.dx = h.rx; nr.dy = h.ry; nr.dz = h.rz;
            nr.dlen = dsqrt(nr.dx*nr.dx + nr.dy*nr.dy + nr.dz*nr.dz);
            r = nr; ++bounce_rays;
            last = depth + 1;
        }
        if (BOUNCE) {
            colour = lvl_local[last];
            // lvl_local[last] is only valid if the loop reached level `last`; when the chain stopped on the
            // Fresnel cut at level k, last == k and the colour is that level's local one.
            for (int k = last - 1; k >= 0; --k) {
                const uint32_t tint = lvl_tint[k];
                uint32_t out = 0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const double sec = static_cast<double>((colour >> (8*ch)) & 255u);
                    double tinted = sec;
                    if (tint >> 24) tinted = static_cast<double>(round_u8((static_cast<double>((tint >> (8*ch)) & 255u) / 255.0) * sec));
                    const double loc = static_cast<double>((lvl_local[k] >> (8*ch)) & 255u);
                    out |= round_u8((1.0 - lvl_fr[k]) * loc + lvl_fr[k] * tinted) << (8*ch);
                }
                colour = out;
            }
        }
        uint8_t* px = rgb + static_cast<size_t>(p.out) * 3;
        px[0] = sc.gamma[colour & 255u]; px[1] = sc.gamma[(colour >> 8) & 255u]; px[2] = sc.gamma[(colour >> 16) & 255u];
        atomicAdd(&lctr[2], static_cast<unsigned long long>(sctr.shadow_rays));
        if (BOUNCE && bounce_rays) atomicAdd(&lctr[3], static_cast<unsigned long long>(bounce_rays));
        if (COUNT) {
            atomicAdd(&lctr[4], static_cast<unsigned long long>(vn)); atomicAdd(&lctr[5], static_cast<unsigned long long>(vp));
            atomicAdd(&lctr[6], static_cast<unsigned long long>(sctr.vn)); atomicAdd(&lctr[7], static_cast<unsigned long long>(sctr.vp));
        }
    }
    flush_counters(lctr, ctr);
}

// Scatter gathered per-rank tile buffers into a row-major frame (one thread per pixel).
__global__ void __launch_bounds__(256) k_assemble_tiles(int width, int height, int tile, int tiles_x, int n_ranks, int tiles_per_rank,
                                                        const uint8_t* __restrict__ gathered, uint8_t* __restrict__ frame) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= width || y >= height) return;
    const int gt = (y / tile) * tiles_x + (x / tile);
    const int rank = gt % n_ranks, lt = gt / n_ranks;
    const size_t src = ((static_cast<size_t>(rank) * tiles_per_rank + lt) * tile * tile + static_cast<size_t>(y % tile) * tile + (x % tile)) * 3;
    const size_t dst = (static_cast<size_t>(y) * width + x) * 3;
    frame[dst] = gathered[src]; frame[dst + 1] = gathered[src + 1]; frame[dst + 2] = gathered[src + 2];
}

// ---- host-callable launchers (called from rt_capi.cpp) ---------------------------------------------

extern "C" hipError_t rt_launch_trace_primary(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm,
                                              uint32_t n_blocks, int counting, int32_t* ids, double* dist, rt_counters* ctr) {
    if (counting) hipLaunchKernelGGL(k_trace_primary<true>, dim3(n_blocks), dim3(RT_BLOCK), 0, st, *sc, *cam, *pm, ids, dist, ctr);
    else          hipLaunchKernelGGL(k_trace_primary<false>, dim3(n_blocks), dim3(RT_BLOCK), 0, st, *sc, *cam, *pm, ids, dist, ctr);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_trace_rays(hipStream_t st, const DevScene* sc, const double* rays, uint64_t n, int counting, int shade_it,
                                           int32_t* ids, double* dist, uint8_t* rgb, rt_counters* ctr) {
    const uint32_t nb = static_cast<uint32_t>((n + RT_BLOCK - 1) / RT_BLOCK);
    if (nb == 0) return hipSuccess;
    if (shade_it) {
        if (counting) hipLaunchKernelGGL((k_trace_rays<true, true>), dim3(nb), dim3(RT_BLOCK), 0, st, *sc, rays, n, ids, dist, rgb, ctr);
        else          hipLaunchKernelGGL((k_trace_rays<false, true>), dim3(nb), dim3(RT_BLOCK), 0, st, *sc, rays, n, ids, dist, rgb, ctr);
    } else {
        if (counting) hipLaunchKernelGGL((k_trace_rays<true, false>), dim3(nb), dim3(RT_BLOCK), 0, st, *sc, rays, n, ids, dist, rgb, ctr);
        else          hipLaunchKernelGGL((k_trace_rays<false, false>), dim3(nb), dim3(RT_BLOCK), 0, st, *sc, rays, n, ids, dist, rgb, ctr);
    }
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_render(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, uint32_t n_blocks,
                                       int counting, int max_scatters, double min_si, uint8_t* rgb, int32_t* ids, rt_counters* ctr) {
    if (max_scatters > 0) {
        if (counting) hipLaunchKernelGGL((k_render<true, true>), dim3(n_blocks), dim3(RT_BLOCK), 0, st, *sc, *cam, *pm, max_scatters, min_si, rgb, ids, ctr);
        else          hipLaunchKernelGGL((k_render<false, true>), dim3(n_blocks), dim3(RT_BLOCK), 0, st, *sc, *cam, *pm, max_scatters, min_si, rgb, ids, ctr);
    } else {
        if (counting) hipLaunchKernelGGL((k_render<true, false>), dim3(n_blocks), dim3(RT_BLOCK), 0, st, *sc, *cam, *pm, max_scatters, min_si, rgb, ids, ctr);
        else          hipLaunchKernelGGL((k_render<false, false>), dim3(n_blocks), dim3(RT_BLOCK), 0, st, *sc, *cam, *pm, max_scatters, min_si, rgb, ids, ctr);
    }
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_assemble(hipStream_t st, int width, int height, int tile, int tiles_x, int n_ranks, int tiles_per_rank,
                                         const uint8_t* gathered, uint8_t* frame) {
    hipLaunchKernelGGL(k_assemble_tiles, dim3((width + 15) / 16, (height + 15) / 16), dim3(256), 0, st,
                       width, height, tile, tiles_x, n_ranks, tiles_per_rank, gathered, frame);
    return hipGetLastError();
}
