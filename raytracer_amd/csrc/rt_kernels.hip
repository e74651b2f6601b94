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
//
// Execution shape.  One wave (64 lanes) per workgroup; a wave owns an 8x8 pixel tile with lanes in Z-order,
// so neighbouring lanes walk neighbouring BVH paths.  Each lane keeps its traversal stack in LDS
// ([entry][lane], conflict-free), with a private overflow for pathological depth.  The walk is
// "while-while": lanes step through internal nodes until every lane of the wave holds a leaf (or is done),
// then all leaf tests run together - FP64 leaf code is 10-40x a box test, so it must not run lane by lane.
// The frame pipeline for max_scatters = 0: k_entry (per 8x8 tile of camera rays: the subtrees its beam can touch, as the
// wide node its walks start from), k_rays<0> (closest hit: the winning record per pixel), k_surface (dense: the full
// surface record of each hit), k_entry again (per tile and light, for the shadow beams), k_rays<1> (shadow rays, the 64
// rays of a (tile, light) group side by side: all aim at the same light; a verdict per ray), then the resolve (dense:
// the terms of the lights that reach each point in light order, ambient, albedo, gamma) - k_resolve_decided and
// k_resolve_listed when the lights fit one batch, k_resolve per batch otherwise.  The ray kernels only walk: what can
// run densely does.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "rt_device.h"

#define DEV __device__ __forceinline__
#ifndef RT_ENTRY_ITERS_PRIMARY       // refinement steps of a tile's entry search at most: later steps cost the search more than
#define RT_ENTRY_ITERS_PRIMARY 24    // they save the walks (frame time at 8 / 10 / 16 / 24 / 96 steps for both kinds of beam:
#endif                               // 6.36 / 5.97 / 5.81 / 5.82 / 5.93 ms on the 1.3 M-primitive 4K frame)
#ifndef RT_ENTRY_ITERS_SHADOW
#define RT_ENTRY_ITERS_SHADOW 16
#endif
#ifndef RT_ENTRY_QUAD_MAX_BEAMS
#define RT_ENTRY_QUAD_MAX_BEAMS 600000u   // entry searches of up to this many beams run with four lanes per beam (k_entry_quad)
#endif
#ifndef RT_INNER_EXIT_REL
#define RT_INNER_EXIT_REL 4          // 0: lanes wait until every ray of the wave has left the inner nodes
#endif

namespace {

struct RayD { double ox, oy, oz, dx, dy, dz, dlen; };

struct HitD {            // full surface record of one hit (shading inputs)
    double dist;
    double px, py, pz;
    double nx, ny, nz;
    double rx, ry, rz;   // mirror direction
    double u, v;         // surface coordinates (only filled on request)
};

DEV double dsqrt(double x) { return __builtin_sqrt(x); }
DEV double clamp_unit(double v) { return (v < -1.0) ? -1.0 : ((1.0 < v) ? 1.0 : v); }

// ---- leaf intersectors -----------------------------------------------------------------------------
// Each "local" solver works in the primitive's own frame and returns the local hit point (and, when FULL,
// the local normal); the shared head and tail move the ray in and the hit out.

// Cube::intersect core, mesh.cpp:182-221: slabs on [-1,1]^3, exit face when the origin is inside.
template <bool FULL> DEV bool solve_cube(const double ro[3], const double rd[3], double pl[3], double nl[3]) {
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
    pl[0] = ro[0] + t * rd[0]; pl[1] = ro[1] + t * rd[1]; pl[2] = ro[2] + t * rd[2];
    if (FULL) { nl[0] = entering ? ne[0] : nx_[0]; nl[1] = entering ? ne[1] : nx_[1]; nl[2] = entering ? ne[2] : nx_[2]; }
    return true;
}

// The same decision and the same local hit point without the normals and without branches, for the walks (which only
// need the distance): every lane of a wave runs all three axes anyway, so the reference's early exits save nothing there.
// Equivalent to solve_cube<false>: an axis the ray runs along constrains nothing (or rejects when the origin lies outside
// its slab); the interval only ever shrinks, so "empty after some axis" is "empty at the end"; the comparisons that pick
// the larger entry and the smaller exit are the reference's own (strict, first value kept on equality).
DEV bool solve_cube_point(const double ro[3], const double rd[3], double pl[3]) {
    double tmin = -__builtin_inf(), tmax = __builtin_inf();
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double o = ro[a], d = rd[a];
        const bool along = __builtin_fabs(d) < 1e-12;
        miss = miss || (along && (o < -1.0 || o > 1.0));
        const double ta = (-1.0 - o) / d;
        const double tb = (1.0 - o) / d;
        const bool swap = ta > tb;
        double t1 = swap ? tb : ta;
        double t2 = swap ? ta : tb;
        t1 = along ? -__builtin_inf() : t1;
        t2 = along ? __builtin_inf() : t2;
        tmin = (t1 > tmin) ? t1 : tmin;
        tmax = (t2 < tmax) ? t2 : tmax;
    }
    miss = miss || tmin > tmax || tmax < 0.0;
    const double t = (tmin >= 0.0) ? tmin : tmax;
    pl[0] = ro[0] + t * rd[0]; pl[1] = ro[1] + t * rd[1]; pl[2] = ro[2] + t * rd[2];
    return !miss;
}

// Cylinder::intersect core, mesh.cpp:394-459: unit cylinder x^2 + y^2 = 1, z in [-1,1], both caps.
template <bool FULL> DEV bool solve_cylinder(const double ro[3], const double rd[3], double pl[3], double nl[3]) {
    const double eps = 1e-12;
    double best_t = __builtin_inf();
    bool found = false;
    pl[0] = pl[1] = pl[2] = 0.0; nl[0] = nl[1] = nl[2] = 0.0;
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
                    pl[0] = ro[0] + t*rd[0]; pl[1] = ro[1] + t*rd[1]; pl[2] = z;
                    if (FULL) {
                        const double len_xy = dsqrt(pl[0]*pl[0] + pl[1]*pl[1]);
                        if (len_xy > eps) { nl[0] = pl[0]/len_xy; nl[1] = pl[1]/len_xy; nl[2] = 0.0; }
                        else              { nl[0] = 0.0; nl[1] = 0.0; nl[2] = 1.0; }
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
                pl[0] = x; pl[1] = y; pl[2] = zc;
                nl[0] = 0.0; nl[1] = 0.0; nl[2] = zc;
                found = true;
            }
        }
    }
    return found;
}

// 16-byte loads of a record's doubles into registers.
typedef double rt_d2 __attribute__((ext_vector_type(2)));
template <int N> DEV void load_rec(const rt_prim_rec* rec, double (&c)[18]) {
    const rt_d2* v = reinterpret_cast<const rt_d2*>(rec);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const rt_d2 t = v[i];
        c[2*i] = t.x; c[2*i + 1] = t.y;
    }
    if (N & 1) c[N - 1] = rec->d[N - 1];
}

// One primitive record against one ray.  The three families (ellipsoid / planar / rotated-scaled box frame)
// are laid out so that a wave holding a mix of kinds serialises only the family-specific middle part.
// c: the record's 18 doubles, already in registers (an ellipsoid uses the first 9).
template <bool FULL, bool UV = false> DEV bool hit_loaded(const double (&c)[18], uint32_t kind, const RayD& r, HitD& h) {
    const double pi = 3.14159265358979323846;
    double px = 0, py = 0, pz = 0, nx = 0, ny = 0, nz = 0;
    if (kind == RT_SPHERE) {
        // Sphere::intersect (axis-aligned ellipsoid), mesh.cpp:597-642.  c: C[0..2], S[3..5], 1/S[6..8].
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
        px = c[0] + l0 * c[3]; py = c[1] + l1 * c[4]; pz = c[2] + l2 * c[5];
        if (FULL) {
            const double wx = l0 * c[6], wy = l1 * c[7], wz = l2 * c[8];
            const double len = dsqrt(wx*wx + wy*wy + wz*wz);
            nx = wx/len; ny = wy/len; nz = wz/len;
        }
        if (UV) {       // spherical mapping of the local point, mesh.cpp:669-679
            double ux = l0, uy = l1, uz = l2;
            const double plen = dsqrt(ux*ux + uy*uy + uz*uz);
            if (plen > 1e-15) { ux /= plen; uy /= plen; uz /= plen; }
            h.u = 0.5 + atan2(uz, ux) / (2.0 * pi);
            h.v = 0.5 - asin(clamp_unit(uy)) / pi;
        }
    } else if (kind == RT_PLANE || kind == RT_TRIANGLE) {
        // Plane::intersect (parallelogram P0 + u e1 + v e2), mesh.cpp:855-886, and the triangle extension
        // (same frame, inside test u >= -eps, v >= -eps, u + v <= 1 + eps).
        // c: W = [e1 e2 n]^-1 row-major [0..8], P0[9..11], e1[12..14], e2[15..17].
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
        const bool outside = (kind == RT_TRIANGLE) ? (u < -eps || v < -eps || u + v > 1.0 + eps)
                                                   : (u < -eps || u > 1.0 + eps || v < -eps || v > 1.0 + eps);
        if (outside) return false;
        // p = P0 + M (u, v, 0) with M = [e1 e2 n].  The reference also adds n * 0.0, which can only turn a -0.0
        // component into +0.0: no effect on any comparison, so the distance-only path leaves the normal out.
        double zx = 0, zy = 0, zz = 0;
        if (FULL) {
            // n = (e1 x e2) / |e1 x e2| exactly as the host derived it for W
            const double cx = c[13]*c[17] - c[14]*c[16], cy = c[14]*c[15] - c[12]*c[17], cz = c[12]*c[16] - c[13]*c[15];
            const double nl = dsqrt(cx*cx + cy*cy + cz*cz);
            nx = cx/nl; ny = cy/nl; nz = cz/nl;
            zx = nx*0.0; zy = ny*0.0; zz = nz*0.0;
        }
        px = c[9]  + (c[12]*u + c[15]*v + zx);
        py = c[10] + (c[13]*u + c[16]*v + zy);
        pz = c[11] + (c[14]*u + c[17]*v + zz);
        if (UV) { h.u = u; h.v = v; }
    } else if (kind == RT_CUBE || kind == RT_CYLINDER) {
        // c: T[0..2], R[3..11] row-major, S[12..14], 1/S[15..17].  World ray into the unit frame
        // (mesh.cpp:171-180), local solve, local hit back to the world (mesh.cpp:223-233).
        double ro[3], rd[3], pl[3], nl[3] = { 0, 0, 0 };
        const double tx = r.ox - c[0], ty = r.oy - c[1], tz = r.oz - c[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double a = c[3 + i]*tx   + c[6 + i]*ty   + c[9 + i]*tz;      // row i of R^T
            const double b = c[3 + i]*r.dx + c[6 + i]*r.dy + c[9 + i]*r.dz;
            ro[i] = a * c[15 + i];
            rd[i] = b * c[15 + i];
        }
        const bool ok = (kind == RT_CUBE) ? (FULL ? solve_cube<FULL>(ro, rd, pl, nl) : solve_cube_point(ro, rd, pl)) : solve_cylinder<FULL>(ro, rd, pl, nl);
        if (!ok) return false;
        const double s0 = pl[0] * c[12], s1 = pl[1] * c[13], s2 = pl[2] * c[14];
        px = (c[3]*s0 + c[4]*s1 + c[5]*s2) + c[0];
        py = (c[6]*s0 + c[7]*s1 + c[8]*s2) + c[1];
        pz = (c[9]*s0 + c[10]*s1 + c[11]*s2) + c[2];
        if (FULL) {
            const double m0 = nl[0] * c[15], m1 = nl[1] * c[16], m2 = nl[2] * c[17];
            const double wx = c[3]*m0 + c[4]*m1 + c[5]*m2;
            const double wy = c[6]*m0 + c[7]*m1 + c[8]*m2;
            const double wz = c[9]*m0 + c[10]*m1 + c[11]*m2;
            const double len = dsqrt(wx*wx + wy*wy + wz*wz);
            nx = wx/len; ny = wy/len; nz = wz/len;
        }
        if (UV) {
            if (kind == RT_CUBE) {          // per-face planar mapping, mesh.cpp:262-278
                if (__builtin_fabs(nl[0]) > 0.5)      { h.u = 0.5 * (pl[2] + 1.0); h.v = 0.5 * (pl[1] + 1.0); }
                else if (__builtin_fabs(nl[1]) > 0.5) { h.u = 0.5 * (pl[0] + 1.0); h.v = 0.5 * (pl[2] + 1.0); }
                else                                  { h.u = 0.5 * (pl[0] + 1.0); h.v = 0.5 * (pl[1] + 1.0); }
            } else if (__builtin_fabs(nl[2]) < 0.5) {   // cylinder side: unwrap, mesh.cpp:497-512
                h.u = (atan2(pl[1], pl[0]) + pi) / (2.0 * pi);
                h.v = 0.5 * (pl[2] + 1.0);
            } else {
                h.u = 0.5 * (pl[0] + 1.0); h.v = 0.5 * (pl[1] + 1.0);
            }
        }
    } else {
        return false;
    }
    // Shared tail (mesh.cpp:236-260 and siblings): re-projected distance, mirror direction.
    {
        const double dx = px - r.ox, dy = py - r.oy, dz = pz - r.oz;
        h.dist = (r.dlen > 1e-15) ? (dx*r.dx + dy*r.dy + dz*r.dz) / r.dlen : dsqrt(dx*dx + dy*dy + dz*dz);
    }
    if (FULL) {
        const double ddn = r.dx*nx + r.dy*ny + r.dz*nz;
        const double tx = r.dx - 2.0 * ddn * nx, ty = r.dy - 2.0 * ddn * ny, tz = r.dz - 2.0 * ddn * nz;
        double len = dsqrt(tx*tx + ty*ty + tz*tz);
        if (len < 1e-15) len = 1.0;
        h.rx = tx/len; h.ry = ty/len; h.rz = tz/len;
        h.px = px; h.py = py; h.pz = pz; h.nx = nx; h.ny = ny; h.nz = nz;
    }
    return true;
}

// The same with the record still in memory: fetches what the kind needs.
template <bool FULL, bool UV = false> DEV bool hit_record(const rt_prim_rec* rec, uint32_t kind, const RayD& r, HitD& h) {
    double c[18];
    if (kind == RT_SPHERE) load_rec<9>(rec, c); else load_rec<18>(rec, c);
    return hit_loaded<FULL, UV>(c, kind, r, h);
}

// ---- BVH traversal ---------------------------------------------------------------------------

// What a box test needs of a ray.  A plane x = c of the padded box is crossed at t = (c - (o +- pad)) / d, evaluated as
// one fused multiply-add c * (1/d) + (-(o +- pad) * (1/d)): ix.. are the reciprocals (clamped to +-2^100, so that a ray
// running along an axis gives huge finite distances of the right sign instead of inf - inf).  Which of a box's two planes
// on an axis the ray meets first follows from the sign of the reciprocal alone, so it is decided once per ray: cnx.. is
// the addend of the plane met first (-(o + pad) / d for the lower plane when d > 0, -(o - pad) / d for the upper one when
// d < 0: the padding always moves a plane outwards), cfx.. that of the other, and kx.. the byte offset inside a wide node
// of the vector that holds the four children's first-met planes on that axis (the other plane's vector is at a fixed
// distance).  The entry distance is then max3 of the near crossings and the exit min3 of the far ones, with no min / max
// per axis; the values are the ones min / max would pick (a padded box is never thinner than -2 pad).
// Rounding: the products carry 2^-24 of |o| / |d|, i.e. |o| 2^-24 along the ray's path - a sixty-fourth of the padding
// (2^-18 max(|o|, scene radius)), like the other two roundings of the form.
struct BoxRay { float ix, iy, iz, cnx, cny, cnz, cfx, cfy, cfz; uint32_t kx, ky, kz; };
constexpr uint32_t WN_LOX = 0u, WN_LOY = 16u, WN_LOZ = 32u, WN_HIX = 48u, WN_HIY = 64u, WN_HIZ = 80u, WN_REF = 96u;   // wide node: byte offsets

DEV float f32_up(double x) {                       // smallest float >= max(x, 0); NaN -> +inf
    if (!(x == x)) return __builtin_inff();
    float f = static_cast<float>(x);
    if (static_cast<double>(f) < x) {               // step to the next float up
        const uint32_t u = __float_as_uint(f);
        f = __uint_as_float(f > 0.0f ? u + 1u : (f == 0.0f ? 1u : u - 1u));
    }
    return (f > 0.0f) ? f : 0.0f;
}

DEV float f32_below(double x) {                    // a float <= x (x finite)
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) {              // one float down
        const uint32_t u = __float_as_uint(f);
        f = __uint_as_float(f > 0.0f ? u - 1u : (f == 0.0f ? 0x80000001u : u + 1u));
    }
    return f;
}
DEV float f32_above(double x) {                    // a float >= x (x finite)
    float f = static_cast<float>(x);
    if (static_cast<double>(f) < x) {              // one float up
        const uint32_t u = __float_as_uint(f);
        f = __uint_as_float(f > 0.0f ? u + 1u : (f == 0.0f ? 1u : u - 1u));
    }
    return f;
}

// QUICK: the reciprocals as len * rcp(d) (v_rcp_f32, one unit in the last place) instead of a division.  That moves a
// crossing by up to (|plane| + |o|) 2^-23 along the path, a sixteenth of the padding - with the three roundings of a
// sixty-fourth each still under an eighth of it - and takes twenty-odd instructions out of a set-up that walks through
// instances repeat at every instance they enter and leave
// (the 100 M-triangle frame 19.9 -> 19.6 ms).  One-level scenes keep the division: their counting kernels are compared
// box for box with a CPU walk that divides.
template <bool QUICK = false>
DEV void box_ray_init(BoxRay& b, const RayD& r, float radius) {
    const float ox = static_cast<float>(r.ox), oy = static_cast<float>(r.oy), oz = static_cast<float>(r.oz);
    const float m = fmaxf(fmaxf(radius, fabsf(ox)), fmaxf(fabsf(oy), fabsf(oz)));
    const float pad = ldexpf(m, -RT_PAD_SHIFT);
    const float big = 1.2676506e30f;                                  // 2^100
    const float len = static_cast<float>(r.dlen);                     // 1 / (d / |d|) in one FP32 division per axis
    const float fdx = static_cast<float>(r.dx), fdy = static_cast<float>(r.dy), fdz = static_cast<float>(r.dz);
    b.ix = fminf(fmaxf(QUICK ? len * __builtin_amdgcn_rcpf(fdx) : len / fdx, -big), big);
    b.iy = fminf(fmaxf(QUICK ? len * __builtin_amdgcn_rcpf(fdy) : len / fdy, -big), big);
    b.iz = fminf(fmaxf(QUICK ? len * __builtin_amdgcn_rcpf(fdz) : len / fdz, -big), big);
    const float npx = -(ox + pad) * b.ix, npy = -(oy + pad) * b.iy, npz = -(oz + pad) * b.iz;     // goes with a lower plane
    const float nmx = -(ox - pad) * b.ix, nmy = -(oy - pad) * b.iy, nmz = -(oz - pad) * b.iz;     // goes with an upper plane
    const bool fx = b.ix > 0.0f, fy = b.iy > 0.0f, fz = b.iz > 0.0f;                              // lower plane first
    b.cnx = fx ? npx : nmx; b.cfx = fx ? nmx : npx; b.kx = fx ? WN_LOX : WN_HIX;
    b.cny = fy ? npy : nmy; b.cfy = fy ? nmy : npy; b.ky = fy ? WN_LOY : WN_HIY;
    b.cnz = fz ? npz : nmz; b.cfz = fz ? nmz : npz; b.kz = fz ? WN_LOZ : WN_HIZ;
}

// Entry distance (>= 0) into the padded box from its planes in the order the ray meets them, +inf on a miss or when the
// box starts beyond `limit`.
DEV float box_entry_ordered(float nx, float ny, float nz, float fx, float fy, float fz, const BoxRay& b, float limit) {
    const float ax = __builtin_fmaf(nx, b.ix, b.cnx), bx = __builtin_fmaf(fx, b.ix, b.cfx);
    const float ay = __builtin_fmaf(ny, b.iy, b.cny), by = __builtin_fmaf(fy, b.iy, b.cfy);
    const float az = __builtin_fmaf(nz, b.iz, b.cnz), bz = __builtin_fmaf(fz, b.iz, b.cfz);
    const float tn = fmaxf(0.0f, fmaxf(fmaxf(ax, ay), az));
    const float tf = fminf(limit, fminf(fminf(bx, by), bz));
    return (tn <= tf) ? tn : __builtin_inff();
}
// The same for a box given as its corners.
DEV float box_entry(const float4 lo, const float4 hi, const BoxRay& b, float limit) {
    const bool fx = b.kx == WN_LOX, fy = b.ky == WN_LOY, fz = b.kz == WN_LOZ;
    return box_entry_ordered(fx ? lo.x : hi.x, fy ? lo.y : hi.y, fz ? lo.z : hi.z, fx ? hi.x : lo.x, fy ? hi.y : lo.y, fz ? hi.z : lo.z, b, limit);
}

// A node reference: bits 31..28 = primitive count (0 = internal), bits 27..0 = first slot (leaf) or
// index of the child pair / 2 (internal).
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;
DEV uint32_t make_ref(uint32_t first, uint32_t count) { return count ? (count << 28) | first : (first >> 1); }
DEV bool ref_internal(uint32_t r) { return (r >> 28) == 0u; }

// Per-lane stack: entries 0..RT_STACK_LDS-1 in LDS ([entry][lane]), deeper ones in a private array that lives
// outside this struct (a dynamically indexed member would drag every field of the walker into scratch memory).
// BIG: two-level scenes (both trees and the marker between them) and the wide walk (up to three postponed subtrees per step)
template <bool BIG> struct DeepStack { uint2 e[(BIG ? RT_STACK_DEPTH_INST : RT_STACK_DEPTH) - RT_STACK_LDS]; };
// The LDS part is one file-scope array, indexed directly, so that the compiler emits ds_read / ds_write (a pointer
// kept in a struct degrades to flat addressing).  Every ray kernel runs one wave per workgroup.
__shared__ uint2 g_lds_stack[RT_STACK_LDS * 64];
struct Stack {
    uint2* deep;                      // DeepStack::e of the owning kernel
    // Where the next entry goes, as the byte offset it would have in the LDS part: entries * 512 + lane * 8 (a row of 64
    // entries is 512 bytes and a lane's column starts below that, so entries = top >> 9 and "none" is top < 512).
    // Kept in this form a push or pop costs one addition, not a shift and an addition.
    uint32_t top;
    DEV void reset(uint32_t lane) { top = lane * 8u; }
    DEV void push(uint32_t ref, float t) {
        const uint2 e = make_uint2(ref, __float_as_uint(t));
        if (top < RT_STACK_LDS * 512u) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(g_lds_stack) + top) = e;
        else deep[(top >> 9) - RT_STACK_LDS] = e;
        top += 512u;
    }
    // next entry that still starts before `limit`, or REF_NONE
    DEV uint32_t pop(float limit) {
        while (top >= 512u) {
            top -= 512u;
            // an LDS read in any case (of a clamped slot), replaced from the private part when the entry lives there:
            // left to itself the compiler picks between the two addresses and reads through one flat load, which is
            // much slower than ds_read
            const uint32_t at = top < RT_STACK_LDS * 512u ? top : (RT_STACK_LDS - 1u) * 512u + (top & 511u);
            uint2 e = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(g_lds_stack) + at);
            asm volatile("" : "+v"(e.x), "+v"(e.y));          // keeps the read where it is, as a ds_read
            if (top >= RT_STACK_LDS * 512u) e = deep[(top >> 9) - RT_STACK_LDS];
            if (__uint_as_float(e.y) <= limit) return e.x;
        }
        return REF_NONE;
    }
};

// Two-level scenes: while a walk is inside an instance its ray in the world waits here ([component][lane]); it is only
// needed again when the walk leaves the instance, and fourteen registers are worth more than seven LDS reads then.
__shared__ double g_lds_world[7 * 64];
DEV void world_ray_park(const RayD& r) {
    const uint32_t l = threadIdx.x & 63u;
    g_lds_world[l] = r.ox; g_lds_world[64 + l] = r.oy; g_lds_world[128 + l] = r.oz;
    g_lds_world[192 + l] = r.dx; g_lds_world[256 + l] = r.dy; g_lds_world[320 + l] = r.dz; g_lds_world[384 + l] = r.dlen;
}
DEV void world_ray_fetch(RayD& r) {
    const uint32_t l = threadIdx.x & 63u;
    r.ox = g_lds_world[l]; r.oy = g_lds_world[64 + l]; r.oz = g_lds_world[128 + l];
    r.dx = g_lds_world[192 + l]; r.dy = g_lds_world[256 + l]; r.dz = g_lds_world[320 + l]; r.dlen = g_lds_world[384 + l];
}

struct Best { double dist; uint32_t slot; int32_t id; uint32_t inst; };

// Two-level scenes.  A ray enters an instance as o' = rot^T (o - trans), d' = rot^T d; what it finds there comes back
// as p = rot p' + trans and n = rot n', r = rot r'.  The distance is the one found in the instance's frame.
constexpr uint32_t REF_LEAVE = 0xFFFFFFFEu;        // stack marker: the walk is back in the top-level tree
DEV void ray_into_instance(const rt_inst_rec* in, const RayD& w, RayD& o) {
    double c[18];
    load_rec<12>(reinterpret_cast<const rt_prim_rec*>(in), c);          // rot[0..8], trans[9..11]
    const double ex = w.ox - c[9], ey = w.oy - c[10], ez = w.oz - c[11];
    o.ox = c[0]*ex + c[3]*ey + c[6]*ez;
    o.oy = c[1]*ex + c[4]*ey + c[7]*ez;
    o.oz = c[2]*ex + c[5]*ey + c[8]*ez;
    o.dx = c[0]*w.dx + c[3]*w.dy + c[6]*w.dz;
    o.dy = c[1]*w.dx + c[4]*w.dy + c[7]*w.dz;
    o.dz = c[2]*w.dx + c[5]*w.dy + c[8]*w.dz;
    o.dlen = dsqrt(o.dx*o.dx + o.dy*o.dy + o.dz*o.dz);
}
DEV void hit_out_of_instance(const rt_inst_rec* in, HitD& h) {
    double c[18];
    load_rec<12>(reinterpret_cast<const rt_prim_rec*>(in), c);
    const double px = c[0]*h.px + c[1]*h.py + c[2]*h.pz + c[9];
    const double py = c[3]*h.px + c[4]*h.py + c[5]*h.pz + c[10];
    const double pz = c[6]*h.px + c[7]*h.py + c[8]*h.pz + c[11];
    const double nx = c[0]*h.nx + c[1]*h.ny + c[2]*h.nz, ny = c[3]*h.nx + c[4]*h.ny + c[5]*h.nz, nz = c[6]*h.nx + c[7]*h.ny + c[8]*h.nz;
    const double rx = c[0]*h.rx + c[1]*h.ry + c[2]*h.rz, ry = c[3]*h.rx + c[4]*h.ry + c[5]*h.rz, rz = c[6]*h.rx + c[7]*h.ry + c[8]*h.rz;
    h.px = px; h.py = py; h.pz = pz; h.nx = nx; h.ny = ny; h.nz = nz; h.rx = rx; h.ry = ry; h.rz = rz;
}

// One ray's walk through the BVH as a resumable state machine, so that a wave can interleave the walks of its
// lanes ("while-while") and swap finished rays for fresh ones.
// MODE 0: closest hit.  MODE 1: shadow verdict for a light at light_dist with the reference's rule
// "blocked iff the CLOSEST hit lies in (1e-4, light_dist - 1e-4)" (raytracer.cpp:230-235): hits at or
// beyond light_dist - 1e-4 are irrelevant; once one hit inside the interval is known only hits at
// <= 1e-4 can still change the verdict, so the search limit collapses to 1e-4; any such hit ends it.
// vn counts node records fetched (1 for the root, 2 per internal node visited, 4 per wide node), vp primitive records.
// WIDE: the scene carries wide nodes (four subtrees per node) and the walk goes through them.
template <int MODE, bool COUNT, bool INST = false, bool WIDE = false>
struct Walker {
    RayD r;                             // in the frame being walked: the world, or the current instance's
    uint32_t inst; int32_t id_off;      // INST: instance slot being walked (REF_NONE in the top level), its id offset
    BoxRay br;
    Stack st;
    uint32_t cur;
    uint32_t self_ref;                  // MODE 1: leaf reference (one primitive) of the record the ray starts on when its test is known to miss
    uint32_t self_inst;                 // ... and, in a two-level scene, the instance it belongs to
    double limit; float limit_f;
    Best best;                          // MODE 0
    double light_dist;                  // MODE 1
    bool found_mid, closed;             // MODE 1
    uint32_t vn, vp;

    template <bool DROP = false> DEV void test_slot(const DevScene& sc, uint32_t slot) {
        if (MODE == 1 && ((1u << 28) | slot) == self_ref && (!INST || inst == self_inst)) return;
        // kind and all 18 doubles in one go: waiting for the kind before asking for the numbers would put two trips
        // to memory one behind the other, and a record is rarely in a cache
        const rt_prim_rec* rec = sc.recs + slot;
        const uint2 tag = *reinterpret_cast<const uint2*>(&rec->kind);
        double c[18];
        load_rec<18>(rec, c);
        if (DROP) __builtin_amdgcn_s_setprio(0);           // the fetch is on its way: step down for the arithmetic (see k_rays)
        if (COUNT) ++vp;
        HitD h;
        if (!hit_loaded<false>(c, tag.x, r, h)) return;
        if (MODE == 0) {
            const int32_t id = static_cast<int32_t>(tag.y) + (INST ? id_off : 0);
            if (h.dist < best.dist || (h.dist == best.dist && id < best.id)) {
                best.dist = h.dist; best.slot = slot; best.id = id;
                if (INST) best.inst = inst;
                limit = h.dist; limit_f = f32_up(limit);
            }
        } else {
            if (h.dist <= 1e-4) { closed = true; return; }
            if (h.dist < light_dist - 1e-4 && !found_mid) { found_mid = true; limit = 1e-4; limit_f = f32_up(limit); }
        }
    }

    // Root box test; afterwards cur is the first node reference or REF_NONE.
    // entry: null, or the entry node of the ray's tile (k_entry): a wide node made for the tile's beam of rays, whose up to
    // four children are the subtrees the beam can touch at all.  The walk then starts with a step on that node.
    // (given as entry_base + entry_off bytes; entry_base null: none)
    DEV void start(const DevScene& sc, const RayD& ray, uint2* deep_stack, double ldist, const char* entry_base = nullptr, uint32_t entry_off = 0) {
        r = ray;
        if (INST) { world_ray_park(ray); inst = REF_NONE; id_off = 0; }
        box_ray_init<INST>(br, r, sc.radius);
        st.deep = deep_stack; st.reset(threadIdx.x & 63u);
        best.dist = __builtin_inf(); best.slot = 0xFFFFFFFFu; best.id = -1; best.inst = REF_NONE;
        light_dist = ldist; found_mid = false; closed = false;
        self_ref = 0xF0000000u;             // a leaf of fifteen: no tree has one, so neither cur nor a slot ever equals it
        self_inst = REF_NONE;
        limit = (MODE == 0) ? __builtin_inf() : (ldist - 1e-4);
        limit_f = f32_up(limit);
        cur = REF_NONE;
        if (WIDE && entry_base) { step_wide<true>(entry_base, entry_off); return; }
        if (WIDE) {
            // the root's box travels with the kernel arguments, with the wide node that holds the root's children
            if (COUNT) vn += 1;
            if (box_entry(make_float4(sc.root_lo[0], sc.root_lo[1], sc.root_lo[2], 0.0f), make_float4(sc.root_hi[0], sc.root_hi[1], sc.root_hi[2], 0.0f),
                          br, limit_f) != __builtin_inff()) cur = sc.root_ref;
        } else if (sc.n_nodes) {
            const float4* nodes = reinterpret_cast<const float4*>(sc.nodes);
            const float4 lo = nodes[0], hi = nodes[1];
            if (COUNT) vn += 1;
            if (box_entry(lo, hi, br, limit_f) != __builtin_inff()) cur = make_ref(__float_as_uint(lo.w), __float_as_uint(hi.w));
        }
    }
    // A shadow ray leaves the primitive it starts on: when that primitive's own test is known to miss (set_self), its
    // leaf is passed over as soon as the walk arrives there, instead of waiting for a leaf round to find nothing.
    DEV void set_self(uint32_t slot, uint32_t instance) {
        self_ref = (1u << 28) | slot; self_inst = instance;
        skip_self();
    }
    DEV void skip_self() { if (MODE == 1 && cur == self_ref && (!INST || inst == self_inst)) cur = st.pop(limit_f); }
    DEV bool at_internal() const { return ref_internal(cur); }
    DEV bool done() const { return cur == REF_NONE; }

    // One wide node: 128 contiguous bytes, four boxes.  The children that the ray enters are put in order of their
    // entry distance (a five-exchange sorting network); the walk goes on in the nearest, the others wait on the
    // stack, farthest pushed first.  An empty place is told by its reference; only a tile's entry node (ENTRY) can have
    // its first two places empty.  The node is `off` bytes into `base` (a wave-uniform pointer: scalar base + 32-bit offset).
    template <bool ENTRY> DEV void step_wide(const char* base, uint32_t off) {
        // off is a multiple of 128, so the low bits of off + k are k itself and the other plane's vector is one XOR away
        const uint32_t ax = off + br.kx, ay = off + br.ky, az = off + br.kz;
        const float4 nx = *reinterpret_cast<const float4*>(base + ax), fx = *reinterpret_cast<const float4*>(base + (ax ^ (WN_LOX ^ WN_HIX)));
        const float4 ny = *reinterpret_cast<const float4*>(base + ay), fy = *reinterpret_cast<const float4*>(base + (ay ^ (WN_LOY ^ WN_HIY)));
        const float4 nz = *reinterpret_cast<const float4*>(base + az), fz = *reinterpret_cast<const float4*>(base + (az ^ (WN_LOZ ^ WN_HIZ)));
        const float4 rf = *reinterpret_cast<const float4*>(base + (off + WN_REF));
        wide_decide<ENTRY>(nx, fx, ny, fy, nz, fz, rf);
    }
    // n*: the four children's planes the ray meets first on each axis, f*: the others; rf: the four references.
    template <bool ENTRY> DEV void wide_decide(const float4 nx, const float4 fx, const float4 ny, const float4 fy, const float4 nz, const float4 fz, const float4 rf) {
        if (COUNT) vn += 4;
        float t0 = box_entry_ordered(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, br, limit_f);
        float t1 = box_entry_ordered(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, br, limit_f);
        float t2 = box_entry_ordered(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, br, limit_f);
        float t3 = box_entry_ordered(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, br, limit_f);
        // an empty place holds an inverted box (lower planes at +inf, upper at -inf: every builder and the entry search write
        // it so), which the ordered test cannot enter: its near crossings are +inf and its far ones -inf whatever the ray
        uint32_t r0 = __float_as_uint(rf.x), r1 = __float_as_uint(rf.y), r2 = __float_as_uint(rf.z), r3 = __float_as_uint(rf.w);
#define RT_EXCHANGE(ta, ra, tb, rb) { const bool sw = tb < ta; const float tt = sw ? tb : ta; tb = sw ? ta : tb; ta = tt; \
                                      const uint32_t rr = sw ? rb : ra; rb = sw ? ra : rb; ra = rr; }
        RT_EXCHANGE(t0, r0, t1, r1) RT_EXCHANGE(t2, r2, t3, r3) RT_EXCHANGE(t0, r0, t2, r2) RT_EXCHANGE(t1, r1, t3, r3) RT_EXCHANGE(t1, r1, t2, r2)
#undef RT_EXCHANGE
        if (t3 != __builtin_inff()) st.push(r3, t3);
        if (t2 != __builtin_inff()) st.push(r2, t2);
        if (t1 != __builtin_inff()) st.push(r1, t1);
        cur = (t0 != __builtin_inff()) ? r0 : st.pop(limit_f);
    }

    // One internal node: fetch both children (64 contiguous bytes), test, descend into the nearer, postpone the other.
    DEV void step_internal(const DevScene& sc) {
        if (WIDE) { step_wide<false>(reinterpret_cast<const char*>(sc.nodes4), cur << 7); return; }
        const float4* pair = reinterpret_cast<const float4*>(sc.nodes) + 4u * static_cast<size_t>(cur);
        const float4 llo = pair[0], lhi = pair[1], rlo = pair[2], rhi = pair[3];
        if (COUNT) vn += 2;
        const float tl = box_entry(llo, lhi, br, limit_f), tr = box_entry(rlo, rhi, br, limit_f);
        const bool hl = tl != __builtin_inff(), hr = tr != __builtin_inff();
        const uint32_t lref = make_ref(__float_as_uint(llo.w), __float_as_uint(lhi.w));
        const uint32_t rref = make_ref(__float_as_uint(rlo.w), __float_as_uint(rhi.w));
        if (hl && hr) {
            if (tr < tl) { st.push(lref, tl); cur = rref; }
            else         { st.push(rref, tr); cur = lref; }
        } else if (hl) cur = lref;
        else if (hr)   cur = rref;
        else           cur = st.pop(limit_f);
    }

    // cur is a leaf: test its primitives, then take the next postponed node.
    DEV void step_leaf(const DevScene& sc) {
        if (INST) {
            if (cur == REF_LEAVE) {                 // the instance is exhausted: back to the world
                world_ray_fetch(r); inst = REF_NONE; id_off = 0;
                box_ray_init<INST>(br, r, sc.radius);
                cur = st.pop(limit_f);
                return;
            }
            if (inst == REF_NONE) {                 // a leaf of the top-level tree: one instance
                const uint32_t slot = cur & 0x0FFFFFFFu;
                const rt_inst_rec* in = sc.inst + slot;
                const uint4 tail = *reinterpret_cast<const uint4*>(&in->root_ref);
                st.push(REF_LEAVE, 0.0f);
                { const RayD world = r; ray_into_instance(in, world, r); }      // in the top-level tree r is the world ray
                box_ray_init<INST>(br, r, __uint_as_float(tail.z));
                inst = slot; id_off = static_cast<int32_t>(tail.y);
                cur = tail.x;
                if (COUNT) ++vn;
                return;
            }
        }
        const uint32_t first = cur & 0x0FFFFFFFu, count = cur >> 28;
        for (uint32_t s = 0; s < count && !(MODE == 1 && closed); ++s) test_slot<true>(sc, first + s);
        cur = (MODE == 1 && closed) ? REF_NONE : st.pop(limit_f);
    }

    // Unbounded primitives (a zero or non-finite scale) are tested against every ray, after the walk.
    // A ray that misses a loose primitive's padded box (or meets it beyond the search limit) is spared the FP64 test, as
    // with a leaf of the tree; the box ray is the world's again by then.
    DEV void finish(const DevScene& sc) {
        for (uint32_t s = 0; s < sc.n_loose && !closed; ++s) {
            if (!INST && box_entry(sc.loose_boxes[2u * s], sc.loose_boxes[2u * s + 1u], br, limit_f) == __builtin_inff()) continue;
            test_slot(sc, sc.n_slots + s);
        }
    }
    DEV bool hit() const { return best.id >= 0; }                  // MODE 0
    DEV bool blocked() const { return found_mid && !closed; }      // MODE 1
};

// A whole walk in one go (arbitrary-ray and bounce kernels).
template <int MODE, bool COUNT, bool INST = false, bool WIDE = false>
DEV bool traverse(const DevScene& sc, const RayD& r, double light_dist,
                  Best& best, uint32_t& vn, uint32_t& vp) {
    Walker<MODE, COUNT, INST, WIDE> w; w.vn = 0; w.vp = 0;
    DeepStack<INST || WIDE> deep;
    w.start(sc, r, deep.e, light_dist);
    while (!w.done()) {
        // inner nodes: the wave stays here until every lane holds a leaf or has run out of nodes
        while (w.at_internal()) w.step_internal(sc);
        if (!w.done()) w.step_leaf(sc);
    }
    w.finish(sc);
    best = w.best; vn += w.vn; vp += w.vp;
    return MODE == 0 ? w.hit() : w.blocked();
}

// The full surface record of the hit a walk found, in the world; mat_id = the primitive whose material applies.
template <bool UV, bool INST>
DEV void best_record(const DevScene& sc, const Best& b, const RayD& world_ray, HitD& h, int32_t& mat_id) {
    const rt_prim_rec* rec = sc.recs + b.slot;
    const uint2 tag = *reinterpret_cast<const uint2*>(&rec->kind);
    if (INST) {
        const rt_inst_rec* in = sc.inst + b.inst;
        RayD ro; ray_into_instance(in, world_ray, ro);
        hit_record<true, UV>(rec, tag.x, ro, h);
        hit_out_of_instance(in, h);
        mat_id = static_cast<int32_t>(tag.y);
    } else {
        hit_record<true, UV>(rec, tag.x, world_ray, h);
        mat_id = static_cast<int32_t>(tag.y);
    }
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

// Shadow ray of RayTracer::shade (raytracer.cpp:210-228) from a surface point with unit normal n towards light L:
// origin nudged 1e-4 along the normal on the light's side, unit direction.  Returns the light distance.
DEV double shadow_ray(const rt_light& L, double px, double py, double pz, double nx, double ny, double nz, RayD& sr, bool* outward = nullptr) {
    const double lx = L.pos[0] - px, ly = L.pos[1] - py, lz = L.pos[2] - pz;
    const double len2 = lx*lx + ly*ly + lz*lz;
    const double len = dsqrt(len2);                 // one root serves the direction and the light's distance
    double dx = 0.0, dy = 0.0, dz = 1.0;            // normalize3(lx, ly, lz)
    if (!(len <= 1e-15)) { dx = lx/len; dy = ly/len; dz = lz/len; }
    const double eps = 1e-4;
    const double ndl = nx*dx + ny*dy + nz*dz;
    const double s = (ndl >= 0.0) ? 1.0 : -1.0;
    if (outward) *outward = ndl >= 0.0;             // the ray starts on the normal's side of the surface and leaves it
    sr.ox = px + s * nx * eps; sr.oy = py + s * ny * eps; sr.oz = pz + s * nz * eps;
    sr.dx = dx; sr.dy = dy; sr.dz = dz;
    sr.dlen = dsqrt(dx*dx + dy*dy + dz*dz);
    return (1e-12 < len2) ? len : 1e-6;             // sqrt(max(1e-12, |L - p|^2)), raytracer.cpp:215
}

// An unblocked light's term, attenuation * (diffuse + specular) (raytracer.cpp:242-259); (dx,dy,dz) is the unit
// direction to the light exactly as shadow_ray produced it.
DEV double light_value(const rt_light& L, const rt_mat_rec& m, double px, double py, double pz,
                       double nx, double ny, double nz, double dx, double dy, double dz,
                       double view_ox, double view_oy, double view_oz) {
    const double lx = L.pos[0] - px, ly = L.pos[1] - py, lz = L.pos[2] - pz;
    const double len2 = lx*lx + ly*ly + lz*lz;
    const double dist2 = (1e-12 < len2) ? len2 : 1e-12;
    const double ndl = nx*dx + ny*dy + nz*dz;
    const double metallic = clampd(m.metallic, 0.0, 1.0);
    const double roughness = clampd(m.roughness, 0.001, 1.0);
    const double e_raw = (2.0 / (roughness * roughness)) - 2.0;
    const double phong_exp = (2.0 < e_raw) ? e_raw : 2.0;
    double vx, vy, vz; normalize3(view_ox - px, view_oy - py, view_oz - pz, vx, vy, vz);
    const double n_dot_l = (0.0 < ndl) ? ndl : 0.0;
    const double diffuse = 1.0 * (1.0 - metallic) * n_dot_l;
    double hx, hy, hz; normalize3(dx + vx, dy + vy, dz + vz, hx, hy, hz);
    const double ndh = nx*hx + ny*hy + nz*hz;
    const double n_dot_h = (0.0 < ndh) ? ndh : 0.0;
    // x^e for a sharp highlight is zero nearly everywhere: ln x <= x - 1, so e (x - 1) log2(e) < -1100 puts x^e below
    // 2^-1100, which is 0 in FP64 however it is rounded - the value pow returns, without its three hundred instructions
    // (a mirror's exponent is two million: only normals within a fiftieth of a degree of the half vector get past this)
    const double specular = (phong_exp * (n_dot_h - 1.0) * 1.4426950408889634 < -1100.0) ? 0.0 : (0.5 * 1.25) * pow(n_dot_h, phong_exp);
    const double attenuation = L.intensity / dist2;
    return attenuation * (diffuse + specular);
}

// The same term for the dense resolve kernels, in mixed precision.  Positions are of size 10^2..10^3 and what the term
// depends on are their differences, so L - p and view - p are taken in FP64 and everything after them - normalisations,
// dot products, the power, the attenuation - runs in FP32 (the result is quantised to 8 bits a few operations later).
// The power is 2^(e log2 x); its relative error grows with e times the error of x, so exponents above 64 (roughness below
// 0.174) take the FP64 form above.  Degenerate vectors normalise to (0, 0, 1) as in normalize3.
#ifndef RT_MIXED_RESOLVE
#define RT_MIXED_RESOLVE 1
#endif
DEV void normalize3f(float x, float y, float z, float& ox, float& oy, float& oz) {
    const float l2 = x*x + y*y + z*z;
    if (!(l2 > 1e-30f)) { ox = 0.0f; oy = 0.0f; oz = 1.0f; return; }
    const float r = __builtin_amdgcn_rsqf(l2);
    ox = x*r; oy = y*r; oz = z*r;
}
// bound: grows by an upper bound of this term's absolute error (attenuation times the relative error of diffuse + specular:
// a few roundings of 2^-24 each, and e times as much again through the power).
// mixed_phong_exp: the exponent as this form computes it; above RT_MIXED_MAX_EXP the FP64 form is taken (FALLBACK), or
// the caller has made sure beforehand that the material does not get here (the first pass of the two-pass resolve).
constexpr float RT_MIXED_MAX_EXP = 64.0f;
DEV float mixed_phong_exp(const rt_mat_rec& m) {
    const float roughness = fminf(fmaxf(static_cast<float>(m.roughness), 0.001f), 1.0f);
    return fmaxf(2.0f / (roughness * roughness) - 2.0f, 2.0f);
}
template <bool FALLBACK = true>
DEV double light_value_mixed(const rt_light& L, const rt_mat_rec& m, double px, double py, double pz,
                             double nx, double ny, double nz, double view_ox, double view_oy, double view_oz, float& bound) {
    const float phong_exp = mixed_phong_exp(m);
    if (FALLBACK && phong_exp > RT_MIXED_MAX_EXP) {
        double dx, dy, dz; normalize3(L.pos[0] - px, L.pos[1] - py, L.pos[2] - pz, dx, dy, dz);
        return light_value(L, m, px, py, pz, nx, ny, nz, dx, dy, dz, view_ox, view_oy, view_oz);        // exact: adds nothing to the bound
    }
    const float lx = static_cast<float>(L.pos[0] - px), ly = static_cast<float>(L.pos[1] - py), lz = static_cast<float>(L.pos[2] - pz);
    const float wx = static_cast<float>(view_ox - px), wy = static_cast<float>(view_oy - py), wz = static_cast<float>(view_oz - pz);
    const float fnx = static_cast<float>(nx), fny = static_cast<float>(ny), fnz = static_cast<float>(nz);
    const float len2 = lx*lx + ly*ly + lz*lz;
    const float dist2 = fmaxf(len2, 1e-12f);
    float dx, dy, dz; normalize3f(lx, ly, lz, dx, dy, dz);
    float vx, vy, vz; normalize3f(wx, wy, wz, vx, vy, vz);
    float hx, hy, hz; normalize3f(dx + vx, dy + vy, dz + vz, hx, hy, hz);
    const float metallic = fminf(fmaxf(static_cast<float>(m.metallic), 0.0f), 1.0f);
    const float n_dot_l = fmaxf(fnx*dx + fny*dy + fnz*dz, 0.0f);
    const float n_dot_h = fmaxf(fnx*hx + fny*hy + fnz*hz, 0.0f);
    const float specular = 0.625f * __builtin_exp2f(phong_exp * __builtin_log2f(n_dot_h));
    const float attenuation = static_cast<float>(L.intensity) / dist2;
    bound += __builtin_fabsf(attenuation) * (2e-6f + 4e-7f * phong_exp);
    return static_cast<double>(attenuation * ((1.0f - metallic) * n_dot_l + specular));
}

// One light's term of RayTracer::shade (raytracer.cpp:210-260) in one go; 0 when the light is blocked.
template <bool COUNT, bool INST = false, bool WIDE = false>
DEV double light_term(const DevScene& sc, const rt_light& L, const rt_mat_rec& m, double px, double py, double pz,
                      double nx, double ny, double nz, double view_ox, double view_oy, double view_oz,
                      ShadeCounters& ctr) {
    RayD sr;
    const double light_dist = shadow_ray(L, px, py, pz, nx, ny, nz, sr);
    Best b;
    ++ctr.shadow_rays;
    if (traverse<1, COUNT, INST, WIDE>(sc, sr, light_dist, b, ctr.vn, ctr.vp)) return 0.0;
    return light_value(L, m, px, py, pz, nx, ny, nz, sr.dx, sr.dy, sr.dz, view_ox, view_oy, view_oz);
}

// Mesh::evaluate_albedo (mesh.cpp:21-45) as r | g << 8 | b << 16: the texel under (u, v) - wrap, nearest,
// floor(u * (w - 1)) - when the primitive carries a texture, its solid colour otherwise.
DEV uint32_t solid_albedo(const rt_mat_rec& m) {
    return static_cast<uint32_t>(m.albedo[0]) | static_cast<uint32_t>(m.albedo[1]) << 8 | static_cast<uint32_t>(m.albedo[2]) << 16;
}
DEV uint32_t albedo_at(const DevScene& sc, const rt_mat_rec& m, double u, double v) {
    if (m.tex < 0) return solid_albedo(m);
    const rt_tex_rec t = sc.tex_table[m.tex];
    u = u - __builtin_floor(u); v = v - __builtin_floor(v);
    int x = static_cast<int>(__builtin_floor(u * static_cast<double>(t.width - 1)));
    int y = static_cast<int>(__builtin_floor(v * static_cast<double>(t.height - 1)));
    if (x < 0) x = 0;
    if (x >= t.width) x = t.width - 1;
    if (y < 0) y = 0;
    if (y >= t.height) y = t.height - 1;
    const uint8_t* px = sc.texels + t.offset + (static_cast<size_t>(y) * t.width + x) * 3u;
    return static_cast<uint32_t>(px[0]) | static_cast<uint32_t>(px[1]) << 8 | static_cast<uint32_t>(px[2]) << 16;
}

// Ambient, clamp, albedo, 8-bit quantisation (raytracer.cpp:262-278); packed r | g << 8 | b << 16.
DEV uint32_t finish_shade(const rt_mat_rec& m, uint32_t albedo, double lighting) {
    const double metallic = clampd(m.metallic, 0.0, 1.0);
    double intensity = (1.0 - metallic) * 0.05 + lighting;
    if (intensity < 0.0) intensity = 0.0;
    if (intensity > 1.0) intensity = 1.0;
    return round_u8(intensity * static_cast<double>(albedo & 255u))
         | round_u8(intensity * static_cast<double>((albedo >> 8) & 255u)) << 8
         | round_u8(intensity * static_cast<double>((albedo >> 16) & 255u)) << 16;
}

// RayTracer::shade, raytracer.cpp:159-279, all lights in order inside one lane (megakernel path).
template <bool COUNT, bool INST = false, bool WIDE = false>
DEV uint32_t shade_material(const DevScene& sc, const HitD& h, const rt_mat_rec& m, uint32_t albedo, double view_ox, double view_oy,
                            double view_oz, ShadeCounters& ctr) {
    if (sc.n_lights == 0) return 0;
    double nx, ny, nz; normalize3(h.nx, h.ny, h.nz, nx, ny, nz);
    double lighting = 0.0;
    for (uint32_t li = 0; li < sc.n_lights; ++li) {
        const rt_light L = sc.lights[li];
        lighting += light_term<COUNT, INST, WIDE>(sc, L, m, h.px, h.py, h.pz, nx, ny, nz, view_ox, view_oy, view_oz, ctr);
    }
    return finish_shade(m, albedo, lighting);
}

// h must carry u, v when the scene has textures.
template <bool COUNT, bool INST = false, bool WIDE = false>
DEV uint32_t shade(const DevScene& sc, const HitD& h, int32_t id, double view_ox, double view_oy, double view_oz,
                   ShadeCounters& ctr) {
    const rt_mat_rec m = sc.mats[id];
    return shade_material<COUNT, INST, WIDE>(sc, h, m, albedo_at(sc, m, h.u, h.v), view_ox, view_oy, view_oz, ctr);
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

// Pixel (x, y): centre (x + 0.5, y + 0.5) (raytracer.cpp:131-134), plus - for samples after the first - a sub-pixel
// offset in [-0.5, 0.5)^2.  The reference draws that offset from a random_device-seeded mt19937
// (raytracer.cpp:398-409); here it is a counter-based hash of (seed, pixel), so a frame is reproducible.
DEV void primary_ray(const DevCamera& cam, int x, int y, RayD& r) {
    double jx = 0.0, jy = 0.0;
    if (cam.jitter_seed) {
        uint32_t h = cam.jitter_seed * 0x9E3779B1u ^ (static_cast<uint32_t>(x) * 0x85EBCA77u) ^ (static_cast<uint32_t>(y) * 0xC2B2AE3Du);
        h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
        uint32_t g = h * 0x9E3779B1u + 0x7F4A7C15u;
        g ^= g >> 16; g *= 0x7FEB352Du; g ^= g >> 15; g *= 0x846CA68Bu; g ^= g >> 16;
        jx = static_cast<double>(h) * (1.0 / 4294967296.0) - 0.5;
        jy = static_cast<double>(g) * (1.0 / 4294967296.0) - 0.5;
    }
    primary_ray(cam, static_cast<double>(x) + 0.5 + jx, static_cast<double>(y) + 0.5 + jy, r);
}

// ---- pixel enumeration -------------------------------------------------------------------------

DEV uint32_t compact1by1(uint32_t v) {        // even bits of v packed together (Z-order decode)
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0F0F0F0Fu;
    v = (v | (v >> 4)) & 0x00FF00FFu;
    return v;
}

// Workgroups are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of logical ids so that
// the tiles sharing an L2 are neighbours in the image.
DEV uint32_t xcd_contiguous(uint32_t b, uint32_t nb) {
    const uint32_t q = nb >> 3, rem = nb & 7u, x = b & 7u;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + (b >> 3);
}

struct Pixel { int x, y; uint32_t out; bool valid; };

// n / k.d, remainder in rem (DivU32 in rt_device.h).
DEV uint32_t div_by(uint32_t n, const DivU32 k, uint32_t& rem) {
    uint32_t q = k.m ? __umulhi(n, k.m) : n;
    int32_t r = static_cast<int32_t>(n - q * k.d);
    if (r < 0) { --q; r += static_cast<int32_t>(k.d); }
    rem = static_cast<uint32_t>(r);
    return q;
}
DEV DivU32 make_div(uint32_t d) { DivU32 k; k.d = d; k.m = d > 1u ? (0xFFFFFFFFu / d) + 1u : 0u; return k; }     // ceil(2^32 / d) = floor((2^32 - 1) / d) + 1

// Wave `w` of the pixel map owns one 8x8 tile, lanes in Z-order.  Rectangle mode: waves run in Z-order inside
// 128x128-pixel super tiles.  Tile mode: a rank's tiles one after the other, 8x8 blocks row-major inside a tile.
DEV Pixel map_pixel(const PixelMap& pm, uint32_t w, uint32_t lane) {
    const int lx = static_cast<int>(compact1by1(lane)), ly = static_cast<int>(compact1by1(lane >> 1));
    Pixel p; p.valid = false; p.x = 0; p.y = 0; p.out = 0;
    if (pm.mode == 0) {
        const uint32_t st = w >> 8, in = w & 255u;
        uint32_t sx;
        const uint32_t sy = div_by(st, pm.d_super_x, sx);
        const int rx = static_cast<int>(sx * 128u + compact1by1(in) * 8u) + lx;
        const int ry = static_cast<int>(sy * 128u + compact1by1(in >> 1) * 8u) + ly;
        if (rx < pm.w && ry < pm.h) {
            p.valid = true; p.x = pm.x0 + rx; p.y = pm.y0 + ry;
            p.out = static_cast<uint32_t>(ry) * static_cast<uint32_t>(pm.w) + static_cast<uint32_t>(rx);
        }
    } else {
        uint32_t in;
        const uint32_t lt = div_by(w, pm.d_wpt, in);
        const uint32_t gt = lt * static_cast<uint32_t>(pm.n_ranks) + static_cast<uint32_t>(pm.rank);
        if (gt < static_cast<uint32_t>(pm.n_tiles)) {
            uint32_t gx, bx;
            const uint32_t gy = div_by(gt, pm.d_tiles_x, gx), by = div_by(in, pm.d_wpt_x, bx);
            const int tx = static_cast<int>(gx) * pm.tile, ty = static_cast<int>(gy) * pm.tile;
            const int ix = static_cast<int>(bx) * 8 + lx, iy = static_cast<int>(by) * 8 + ly;
            if (tx + ix < pm.w && ty + iy < pm.h) {
                p.valid = true; p.x = pm.x0 + tx + ix; p.y = pm.y0 + ty + iy;
                p.out = lt * static_cast<uint32_t>(pm.tile * pm.tile) + static_cast<uint32_t>(iy * pm.tile + ix);
            }
        }
    }
    return p;
}

// The other way round: which tile of the pixel map (wave index of map_pixel) and which lane own output slot i.
DEV uint32_t part1by1(uint32_t v) {           // bits of v spread to the even positions (Z-order encode)
    v &= 0x0000FFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
DEV void slot_to_ray(const PixelMap& pm, size_t i, uint32_t& tile_w, uint32_t& lane) {
    uint32_t ix, iy;
    // output slots are 32-bit throughout (a launch holds fewer than 2^32 / 64 / RT_LIGHT_BATCH tiles)
    if (pm.mode == 0) {
        iy = div_by(static_cast<uint32_t>(i), pm.d_w, ix);
        tile_w = ((iy >> 7) * pm.super_x + (ix >> 7)) * 256u + (part1by1((ix >> 3) & 15u) | part1by1((iy >> 3) & 15u) << 1);
    } else {
        const uint32_t wpt_x = pm.d_wpt_x.d;
        uint32_t rem;
        const uint32_t lt = div_by(static_cast<uint32_t>(i), pm.d_t2, rem);
        iy = div_by(rem, pm.d_tile, ix);
        tile_w = lt * pm.d_wpt.d + (iy >> 3) * wpt_x + (ix >> 3);
    }
    lane = part1by1(ix & 7u) | part1by1(iy & 7u) << 1;
}

// ---- counters ------------------------------------------------------------------------------------
// Wave-level sums, then one atomic per wave and counter into one of RT_CTR_SHARDS cache lines.

DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

DEV void add_counter(rt_counters* ctr, uint32_t shard, int field, unsigned long long lane_value) {
    const unsigned long long total = wave_sum(lane_value);
    if ((threadIdx.x & 63u) == 0 && total)
        atomicAdd(reinterpret_cast<unsigned long long*>(ctr) + static_cast<size_t>(shard) * RT_CTR_STRIDE + field, total);
}

// Issue priority (s_setprio, 0..3) of a ray kernel's wave outside its leaf rounds, which run at 0.
#ifndef RT_WALK_PRIO
#define RT_WALK_PRIO 1
#endif
enum { F_PRIMARY = 0, F_HITS = 1, F_SHADOW = 2, F_BOUNCE = 3, F_VN_P = 4, F_VP_P = 5, F_VN_S = 6, F_VP_S = 7 };

} // namespace

// ---- kernels -------------------------------------------------------------------------------------

// Persistent ray kernel.  Every wave loops: lanes whose ray has finished write their result, then all free lanes
// take the next rays of the wave's current chunk (chunks of RT_CHUNK consecutive rays are handed out by one atomic
// counter), then the wave advances all its walks by one while-while round.  Finished lanes are only swapped once
// RT_REFILL_MIN of them have piled up, so that the result / set-up code also runs on a well-filled wave.
// MODE 0 - closest hits: ray i = lane (i % 64) of tile (i / 64) of the pixel map; writes ids / dist (optional) and,
//          with RECORD != 0, the record slot that was hit (and the instance), from which k_surface makes the
//          surface record the shadow pass starts from.
//          RECORD & 2: a level of a mirror-bounce frame - deeper levels take their rays from the slots (left there by
//          k_bounce_resolve).
// MODE 1 - shadow rays: ray i = lane (i % 64) of tile ((i / 64) / n_batch) towards light light0 + (i / 64) % n_batch;
//          the 64 rays of a group aim at one light from one tile, and a tile's lights follow one another, so its
//          surface records are still in a cache when the next light asks for them.  Writes the verdict.
template <int MODE, bool COUNT, int RECORD, bool INST = false, bool WIDE = false>
__global__ void __launch_bounds__(64, RT_RAYS_MIN_WAVES) k_rays(DevScene sc, DevCamera cam, PixelMap pm, uint32_t n_tiles, uint32_t light0,
                                             uint32_t n_rays, int32_t* __restrict__ ids, double* __restrict__ dist,
                                             FrameBufs fb, rt_counters* ctr, unsigned int* queue, int refill_min) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_batch = MODE == 1 ? (n_rays >> 6) / n_tiles : 1u;      // lights of this shadow launch
    const DivU32 d_batch = make_div(n_batch);
    enum { IDLE = 0, TRAV = 1, DONE = 2 };
    uint32_t state = IDLE, out = 0, ridx = 0;
    Walker<MODE, COUNT, INST, WIDE> w; w.vn = 0; w.vp = 0; w.cur = REF_NONE;
    DeepStack<INST || WIDE> deep;
    uint32_t n_cast = 0, n_hit = 0;
    uint32_t q_next = 0, q_end = 0;          // wave-uniform: the unread part of the current chunk
    bool q_empty = false;
    // The chunk sequence is cut into RT_QUEUE_PARTS contiguous runs, one per XCD (workgroups go round-robin over the
    // XCDs): the waves behind one L2 work through neighbouring rays.  A wave whose run is used up helps with the others.
    // camera rays are cheap to set up and coherent: their launch is faster with twice the chunk (1.22 -> 1.18 ms on the 4K frame),
    // the shadow launch slower (3.60 -> 3.65 ms)
    constexpr uint32_t CHUNK = (MODE == 0 && !(RECORD & 2)) ? RT_CHUNK_PRIMARY : RT_CHUNK;
    const uint32_t n_chunks = static_cast<uint32_t>((static_cast<unsigned long long>(n_rays) + CHUNK - 1) / CHUNK);
    uint32_t q_part = blockIdx.x % RT_QUEUE_PARTS, q_tried = 0;
#ifdef RT_SWAP_TIMING
    unsigned long long sw_res = 0, sw_fetch = 0, sw_load = 0, sw_comp = 0, sw_n = 0, sw_rays = 0, sw_t = 0;
    const unsigned long long sw_begin = __builtin_amdgcn_s_memtime();
#endif
#ifdef RT_PHASE_TIMING
    unsigned long long pt_swap = 0, pt_inner = 0, pt_leaf = 0, pt_inner_it = 0, pt_inner_lanes = 0, pt_leaf_it = 0, pt_leaf_lanes = 0;
    const unsigned long long pt_begin = __builtin_amdgcn_s_memtime();
#endif

    __builtin_amdgcn_s_setprio(RT_WALK_PRIO);
    for (;;) {
#ifdef RT_PHASE_TIMING
        const unsigned long long pt_t0 = __builtin_amdgcn_s_memtime();
#endif
        const int n_trav = __popcll(__ballot(state == TRAV));
        const int n_done = __popcll(__ballot(state == DONE));
        const bool swap = n_trav == 0 || (q_empty ? n_done >= refill_min : 64 - n_trav >= refill_min);
        if (swap) {
#ifdef RT_SWAP_TIMING
            sw_t = __builtin_amdgcn_s_memtime(); sw_n += 1;
#endif
            // ---- results of finished rays -----------------------------------------------------------------
            if (state == DONE) {
                if (sc.n_loose) w.finish(sc);
                if (MODE == 0) {
                    const bool hit = w.hit();
                    n_hit += hit ? 1u : 0u;
                    if (ids) ids[out] = w.best.id;
                    if (dist) dist[out] = w.best.dist;
                    if (RECORD) {
                        // which record was hit (and in which instance); k_surface turns that into the surface record
                        fb.slot[out] = w.best.slot;
                        if (INST) fb.inst[out] = w.best.inst;
                    }
                } else {
                    // only the verdict leaves this kernel: the light's term is evaluated by the resolve kernels, densely.
                    // Verdicts lie in ray order - the 64 rays of a tile towards one light share one 64-byte line - so the
                    // bytes a swap writes fall into a few lines instead of one line each (per-pixel order cost 16 bytes of
                    // memory traffic per verdict byte; one bit per ray set with atomics cost 11: atomics go to memory).
                    fb.lit[ridx] = w.blocked() ? 0 : 1;
                }
                state = IDLE;
            }
#ifdef RT_SWAP_TIMING
            { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_amdgcn_s_memtime(); sw_res += t - sw_t; sw_t = t; }
#endif
            // ---- fresh rays for the free lanes ---------------------------------------------------------------
            // next chunk of the wave's run of the chunk sequence (or of the other runs, once that one is used up)
            auto fetch_chunk = [&](uint32_t& c_next, uint32_t& c_end) -> bool {
                while (q_tried < RT_QUEUE_PARTS) {
                    const uint32_t p0 = static_cast<uint32_t>(static_cast<unsigned long long>(n_chunks) * q_part / RT_QUEUE_PARTS);
                    const uint32_t p1 = static_cast<uint32_t>(static_cast<unsigned long long>(n_chunks) * (q_part + 1) / RT_QUEUE_PARTS);
                    unsigned int c = 0;
                    if (lane == 0) c = atomicAdd(queue + q_part, 1u);
                    c = __builtin_amdgcn_readfirstlane(c);
                    if (c < p1 - p0) {
                        c_next = (p0 + c) * CHUNK;
                        c_end = (n_rays - c_next < CHUNK) ? n_rays : c_next + CHUNK;
                        return true;
                    }
                    q_part = (q_part + 1) % RT_QUEUE_PARTS; ++q_tried;
                }
                return false;
            };
            // ---- set-up of one fresh ray (index ridx of the launch) in this lane ---------------------------------------
            auto setup_ray = [&]() {
                const uint32_t g = ridx >> 6;
                uint32_t g_light = 0;                            // MODE 1: the light of the ray's group within the batch
                const uint32_t g_tile = MODE == 0 ? g : div_by(g, d_batch, g_light);
                const Pixel p = map_pixel(pm, g_tile, ridx & 63u);
                if (MODE == 1) {
                    // Everything the new ray needs is asked for at once and unconditionally (slot 0 stands in where the
                    // lane has no pixel): asking for the surface record only after the id has arrived would put two
                    // trips to memory one behind the other in every swap.
                    const uint32_t o = p.valid ? p.out : 0u;
                    int32_t hid = fb.id[o];
                    const uint32_t own = fb.slot[o];                                   // with k_surface's marks
                    const uint32_t own_inst = INST ? fb.inst[o] : REF_NONE;
                    const double hx = fb.px[o], hy = fb.py[o], hz = fb.pz[o], hnx = fb.nx[o], hny = fb.ny[o], hnz = fb.nz[o];
                    const char* entry = WIDE ? reinterpret_cast<const char*>(fb.entry_s) : nullptr;    // node g = tile * n_batch + light
                    const rt_light L = sc.lights[light0 + g_light];
                    asm volatile("" : "+v"(hid));                           // the loads stay up here
                    __builtin_amdgcn_s_setprio(0);                         // ... and the FP64 set-up below gives way to other waves' chains
#ifdef RT_SWAP_TIMING
                    { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_amdgcn_s_memtime(); sw_load += t - sw_t; sw_t = t; }
#endif
                    RayD r;
                    bool outward;
                    const double ld = shadow_ray(L, hx, hy, hz, hnx, hny, hnz, r, &outward);
                    __builtin_amdgcn_s_setprio(RT_WALK_PRIO);
                    if (p.valid && hid >= 0) {
                        out = o;
                        w.start(sc, r, deep.e, ld, entry, g << 7);
                        // the ray leaves the convex primitive it starts on (k_surface marked it): that one test is a miss
                        if ((own >> 31) || ((own >> 30) & (outward ? 1u : 0u))) w.set_self(own & 0x0FFFFFFFu, own_inst);
                        ++n_cast;
                        state = w.done() ? DONE : TRAV;
                    }
                } else if (p.valid) {
                    out = p.out;
                    if (MODE == 0) {
                        RayD r;
                        bool have = true;
                        if ((RECORD & 2) && fb.level > 0) {
                            // a bounce level: the ray was left in the slot by the level above
                            if (fb.alive[out]) {
                                r.ox = fb.ray[out]; r.oy = fb.ray[fb.n + out]; r.oz = fb.ray[2*fb.n + out];
                                r.dx = fb.ray[3*fb.n + out]; r.dy = fb.ray[4*fb.n + out]; r.dz = fb.ray[5*fb.n + out];
                                r.dlen = dsqrt(r.dx*r.dx + r.dy*r.dy + r.dz*r.dz);
                            } else { have = false; fb.slot[out] = REF_NONE; }
                        } else {
                            primary_ray(cam, p.x, p.y, r);
                        }
                        if (have) {
                            const char* entry = (WIDE && !((RECORD & 2) && fb.level > 0)) ? reinterpret_cast<const char*>(fb.entry_p) : nullptr;
                            w.start(sc, r, deep.e, 0.0, entry, g << 7);
                            ++n_cast;
                            state = w.done() ? DONE : TRAV;
                        }
                    }
                }
            };
            while (!q_empty) {
                const unsigned long long idle = __ballot(state == IDLE);
                if (!idle) break;
                if (q_next == q_end && !fetch_chunk(q_next, q_end)) { q_empty = true; break; }
                // when the rest of the chunk cannot fill the free lanes the next chunk is opened in the same pass: the
                // set-up below is long, and running it twice for one swap (once per chunk) is what this avoids
                const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle));
                const uint32_t take_a = n_idle < q_end - q_next ? n_idle : q_end - q_next;
                uint32_t b_next = 0, b_end = 0;
                if (take_a < n_idle && !fetch_chunk(b_next, b_end)) q_empty = true;
                const uint32_t take_b = n_idle - take_a < b_end - b_next ? n_idle - take_a : b_end - b_next;
                const uint32_t take = take_a + take_b;
#ifdef RT_SWAP_TIMING
                { const unsigned long long t = __builtin_amdgcn_s_memtime(); sw_fetch += t - sw_t; sw_t = t; sw_rays += take; }
#endif
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32),
                                                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
                if (state == IDLE && rank < take) {
                    ridx = rank < take_a ? q_next + rank : b_next + (rank - take_a);
                    setup_ray();
                }
                if (b_end != b_next) { q_next = b_next + take_b; q_end = b_end; } else q_next += take_a;
#ifdef RT_SWAP_TIMING
                { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_amdgcn_s_memtime(); sw_comp += t - sw_t; sw_t = t; }
#endif
            }
            if (__ballot(state != IDLE) == 0) { if (q_empty) break; else continue; }
            if (__ballot(state == TRAV) == 0) continue;          // only finished rays on board: write them out first
        }
        // ---- one while-while round for the rays in flight ----------------------------------------------------------
#ifdef RT_PHASE_TIMING
        const unsigned long long pt_t1 = __builtin_amdgcn_s_memtime();
        pt_swap += pt_t1 - pt_t0;
        unsigned long long pt_t2 = pt_t1;
#endif
        if (state == TRAV) {
            // Lanes step through inner nodes together; once fewer than a quarter of the rays in flight are still among
            // inner nodes the others test their leaves, and the stragglers resume in the next round (they would
            // otherwise hold three quarters of the wave idle; measured 10.96 -> 10.03 ms per frame on the 1.3 M scene).
#if RT_INNER_EXIT_REL > 0
            const int in_flight = __popcll(__ballot(true));
            while (w.at_internal()) {
#ifdef RT_PHASE_TIMING
                pt_inner_lanes += 1; if (lane == static_cast<uint32_t>(__ffsll(static_cast<long long>(__ballot(true))) - 1)) pt_inner_it += 1;
#endif
                w.step_internal(sc);
                w.skip_self();
                if (__popcll(__ballot(w.at_internal())) * RT_INNER_EXIT_REL < in_flight) break;
            }
#ifdef RT_PHASE_TIMING
            pt_t2 = __builtin_amdgcn_s_memtime();
#endif
            if (!w.done() && !w.at_internal()) {
#ifdef RT_PHASE_TIMING
                pt_leaf_lanes += 1; if (lane == static_cast<uint32_t>(__ffsll(static_cast<long long>(__ballot(true))) - 1)) pt_leaf_it += 1;
#endif
                // The leaf tests are long runs of FP64 arithmetic behind one fetch; a node step or a swap is a short piece of
                // a chain of round trips.  When both want the SIMD the chain goes first: this wave steps down for the
                // arithmetic of its leaf round, once the record is asked for (test_slot), and up again after it
                // (measured on the 1 M-cell frame: 5.16 -> 5.09 ms; the other way round, leaf rounds first, 5.19).
                w.step_leaf(sc);
                w.skip_self();
                __builtin_amdgcn_s_setprio(RT_WALK_PRIO);
            }
#else
            while (w.at_internal()) w.step_internal(sc);
            if (!w.done()) w.step_leaf(sc);
#endif
            if (w.done()) state = DONE;
        }
#ifdef RT_PHASE_TIMING
        {
            // the lanes reconverge here; lane-dependent sums were taken inside divergent code by whichever lanes were there,
            // so take the maximum over the wave afterwards (below)
            const unsigned long long pt_t3 = __builtin_amdgcn_s_memtime();
            const unsigned long long trav = __ballot(pt_t2 != pt_t1);
            if (trav) pt_t2 = __shfl(pt_t2, __ffsll(static_cast<long long>(trav)) - 1);
            pt_inner += pt_t2 - pt_t1; pt_leaf += pt_t3 - pt_t2;
        }
#endif
    }
    const uint32_t shard = blockIdx.x % RT_CTR_SHARDS;
#ifdef RT_SWAP_TIMING
    if (MODE == 1) {
        const bool l0 = lane == 0;
        const unsigned long long total = __builtin_amdgcn_s_memtime() - sw_begin;
        add_counter(ctr, shard, F_VN_P, l0 ? sw_res : 0); add_counter(ctr, shard, F_VP_P, l0 ? sw_fetch : 0);
        add_counter(ctr, shard, F_VN_S, l0 ? sw_load : 0); add_counter(ctr, shard, F_VP_S, l0 ? sw_comp : 0);
        add_counter(ctr, shard, F_HITS, l0 ? sw_n : 0); add_counter(ctr, shard, F_BOUNCE, l0 ? total : 0);
        add_counter(ctr, shard, F_PRIMARY, l0 ? sw_rays : 0);
        return;
    }
    return;
#endif
#ifdef RT_PHASE_TIMING
    if (MODE == 1) {
        // wave-level figures: every lane holds its own partial view of the divergent parts; reduce with max
        const unsigned long long total = __builtin_amdgcn_s_memtime() - pt_begin;
        const unsigned long long a = pt_swap, b = pt_inner, c = pt_leaf, d = wave_sum(pt_inner_it), e = wave_sum(pt_inner_lanes), f = wave_sum(pt_leaf_it), g = wave_sum(pt_leaf_lanes);
        const bool l0 = lane == 0;
        add_counter(ctr, shard, F_VN_P, l0 ? a : 0); add_counter(ctr, shard, F_VP_P, l0 ? b : 0); add_counter(ctr, shard, F_VN_S, l0 ? c : 0);
        add_counter(ctr, shard, F_VP_S, l0 ? total : 0);
        add_counter(ctr, shard, F_HITS, l0 ? d : 0); add_counter(ctr, shard, F_BOUNCE, l0 ? e : 0);
        add_counter(ctr, shard, F_PRIMARY, l0 ? f : 0); add_counter(ctr, shard, F_SHADOW, l0 ? g : 0);
        return;
    }
    return;
#endif
    if (MODE == 0) {
        const bool bounce_level = (RECORD & 2) && fb.level > 0;
        add_counter(ctr, shard, bounce_level ? F_BOUNCE : F_PRIMARY, n_cast);
        add_counter(ctr, shard, F_HITS, bounce_level ? 0u : n_hit);
        if (COUNT) { add_counter(ctr, shard, F_VN_P, w.vn); add_counter(ctr, shard, F_VP_P, w.vp); }
    } else {
        add_counter(ctr, shard, F_SHADOW, n_cast);
        if (COUNT) { add_counter(ctr, shard, F_VN_S, w.vn); add_counter(ctr, shard, F_VP_S, w.vp); }
    }
}

// Surface records for the hits of a closest-hit launch, densely: one lane per pixel of the pixel map re-creates its ray
// (camera ray, or the ray k_bounce_resolve left for this level), evaluates the winning primitive in full (point,
// normal; surface coordinates and albedo when the scene has textures) and stores what the shadow and resolve passes read.
// FLAGS & 2: a level of a mirror-bounce frame (keeps the raw normal, and at level 0 the camera ray, for the Fresnel
// term and the next level); FLAGS & 4: textures.
template <int FLAGS, bool INST>
__global__ void __launch_bounds__(256) k_surface(DevScene sc, DevCamera cam, PixelMap pm, uint32_t n_tiles, FrameBufs fb) {
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= n_tiles) return;                                       // whole waves leave together
    const Pixel p = map_pixel(pm, g, threadIdx.x & 63u);
    float blo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() };
    float bhi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
    if (p.valid) {
        const uint32_t out = p.out;
        const uint32_t slot = fb.slot[out];
        RayD r;
        bool have = slot != REF_NONE;                               // REF_NONE: no ray here, or a miss
        if ((FLAGS & 2) && fb.level > 0) {
            if (have) {
                r.ox = fb.ray[out]; r.oy = fb.ray[fb.n + out]; r.oz = fb.ray[2*fb.n + out];
                r.dx = fb.ray[3*fb.n + out]; r.dy = fb.ray[4*fb.n + out]; r.dz = fb.ray[5*fb.n + out];
                r.dlen = dsqrt(r.dx*r.dx + r.dy*r.dy + r.dz*r.dz);
            }
        } else {
            primary_ray(cam, p.x, p.y, r);
            if (FLAGS & 2) {
                fb.ray[out] = r.ox; fb.ray[fb.n + out] = r.oy; fb.ray[2*fb.n + out] = r.oz;
                fb.ray[3*fb.n + out] = r.dx; fb.ray[4*fb.n + out] = r.dy; fb.ray[5*fb.n + out] = r.dz;
            }
        }
        if (!have) fb.id[out] = -1;
        else {
            Best b; b.dist = 0.0; b.slot = slot; b.id = 0; b.inst = INST ? fb.inst[out] : REF_NONE;
            HitD h; h.u = 0.0; h.v = 0.0;
            int32_t mat_id = -1;
            best_record<(FLAGS & 4) != 0, INST>(sc, b, r, h, mat_id);
            // the slot keeps the primitive whose material applies: in a two-level scene the mesh primitive
            fb.id[out] = mat_id;
            {
                // Marks for the shadow pass.  A shadow ray starts 1e-4 off this surface, on the side its light is on.
                // The primitive is convex and lies behind the tangent plane at the hit; a ray that starts in front of that
                // plane and does not point back at it cannot meet the primitive again.  Bit 31: flat (parallelogram,
                // triangle) - true for either side.  Bit 30: solid with a proper scale - true when the ray leaves along
                // the outward normal; towards the other side it starts inside and must be tested.
                const rt_prim_rec* rec = sc.recs + slot;
                const uint32_t kind = rec->kind;
                uint32_t mark = 0;
                if (kind == RT_PLANE || kind == RT_TRIANGLE) mark = 1u << 31;
                else if (kind == RT_CUBE || kind == RT_CYLINDER || kind == RT_SPHERE) {
                    const int k0 = kind == RT_SPHERE ? 3 : 12;
                    const double s0 = rec->d[k0], s1 = rec->d[k0 + 1], s2 = rec->d[k0 + 2];
                    const bool proper = s0 > 0.0 && s1 > 0.0 && s2 > 0.0 && s0 < 1e300 && s1 < 1e300 && s2 < 1e300;
                    mark = proper ? 1u << 30 : 0u;
                }
                fb.slot[out] = slot | mark;
            }
            if (FLAGS & 4) fb.albedo[out] = albedo_at(sc, sc.mats[mat_id], h.u, h.v);
            double nx, ny, nz; normalize3(h.nx, h.ny, h.nz, nx, ny, nz);   // shade() re-normalises (raytracer.cpp:182)
            fb.px[out] = h.px; fb.py[out] = h.py; fb.pz[out] = h.pz;
            fb.nx[out] = nx;   fb.ny[out] = ny;   fb.nz[out] = nz;
            if (FLAGS & 2) { fb.rawn[out] = h.nx; fb.rawn[fb.n + out] = h.ny; fb.rawn[2*fb.n + out] = h.nz; }
            if (fb.tile_box) {
                blo[0] = f32_below(h.px); blo[1] = f32_below(h.py); blo[2] = f32_below(h.pz);
                bhi[0] = f32_above(h.px); bhi[1] = f32_above(h.py); bhi[2] = f32_above(h.pz);
            }
        }
    }
    // the box of the tile's hit points, for the entry search of its shadow beams (every lane of the wave is here)
    if (fb.tile_box) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                blo[k] = fminf(blo[k], __shfl_xor(blo[k], off, 64));
                bhi[k] = fmaxf(bhi[k], __shfl_xor(bhi[k], off, 64));
            }
        if ((threadIdx.x & 63u) == 0) {
            float* t = fb.tile_box + static_cast<size_t>(g) * 8u;
            t[0] = blo[0]; t[1] = blo[1]; t[2] = blo[2]; t[3] = bhi[0]; t[4] = bhi[1]; t[5] = bhi[2];
        }
    }
}

// ---- entry search ------------------------------------------------------------------------------------
// The 64 rays of a tile form a narrow beam, and near the root they all take the same turns.  One lane per beam refines a
// set of up to four subtrees, starting from the root: it replaces a subtree by those of its children that the beam can
// touch, for as long as the set still fits in four places, and leaves the set as one wide node - the tile's entry node -
// from which the tile's rays start their walks.  A child is passed over only when NO ray of the beam can touch its box,
// so nothing a ray could hit is lost and results do not change; the tests are FP64 and conservative.

// Camera rays of a tile: common origin o, directions within [wlo, whi] per axis (not normalised: a box test does not
// depend on the direction's length when the ray has no far limit).  Interval arithmetic on the slab test; the reciprocals
// of the direction bounds belong to the beam and are made once.
struct PrimaryBeam { double o[3], i0[3], i1[3]; int sign[3]; };      // sign: +1 / -1 all directions point that way, 0 mixed
DEV void primary_beam_init(PrimaryBeam& p, const double o[3], const double wlo[3], const double whi[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p.o[k] = o[k];
        p.sign[k] = (wlo[k] <= 0.0 && whi[k] >= 0.0) ? 0 : (wlo[k] > 0.0 ? 1 : -1);
        p.i0[k] = p.sign[k] ? 1.0 / wlo[k] : 0.0; p.i1[k] = p.sign[k] ? 1.0 / whi[k] : 0.0;
    }
}
DEV bool beam_touches_primary(const PrimaryBeam& p, const double lo[3], const double hi[3]) {
    double tn = 0.0, tf = __builtin_inf();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (p.sign[k] == 0) continue;                       // some ray may run along this slab: no constraint from it
        // entry through the face the directions point away from, exit through the other one
        const double a = (p.sign[k] > 0 ? lo[k] : hi[k]) - p.o[k], b = (p.sign[k] > 0 ? hi[k] : lo[k]) - p.o[k];
        const double near_lb = fmin(a * p.i0[k], a * p.i1[k]), far_ub = fmax(b * p.i0[k], b * p.i1[k]);
        tn = fmax(tn, near_lb); tf = fmin(tf, far_ub);
    }
    return tn <= tf + 1e-9 * fmax(1.0, tf);
}

// Shadow rays of a tile towards one light: origins within [olo, ohi], all aiming at L.  At parameter t in [0, 1] the rays
// are inside (1 - t) [olo, ohi] + t L; the box can be touched only if that slice overlaps it for some t, which is six
// linear conditions on t.  (The rays' 1e-4 nudge along the normal is in the caller's padding of the box.)
// The conditions' slopes belong to the beam, not to the box: b1 = L - olo, b2 = ohi - L and their reciprocals are made
// once per beam (a product with the reciprocal is within one rounding of the quotient, far inside the slack below).
struct ShadowBeam { double olo[3], ohi[3], b1[3], i1[3], b2[3], i2[3]; };
DEV void shadow_beam_init(ShadowBeam& s, const double olo[3], const double ohi[3], const double L[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.olo[k] = olo[k]; s.ohi[k] = ohi[k];
        s.b1[k] = L[k] - olo[k]; s.i1[k] = s.b1[k] != 0.0 ? 1.0 / s.b1[k] : 0.0;
        s.b2[k] = ohi[k] - L[k]; s.i2[k] = s.b2[k] != 0.0 ? 1.0 / s.b2[k] : 0.0;
    }
}
DEV bool beam_touches_shadow(const ShadowBeam& s, const double lo[3], const double hi[3]) {
    double tn = 0.0, tf = 1.0;
    bool out = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // olo + t (L - olo) <= hi
        { const double r = hi[k] - s.olo[k], q = r * s.i1[k];
          if (s.b1[k] > 0.0) tf = fmin(tf, q); else if (s.b1[k] < 0.0) tn = fmax(tn, q); else out = out || r < 0.0; }
        // ohi + t (L - ohi) >= lo
        { const double r = s.ohi[k] - lo[k], q = r * s.i2[k];
          if (s.b2[k] > 0.0) tf = fmin(tf, q); else if (s.b2[k] < 0.0) tn = fmax(tn, q); else out = out || r < 0.0; }
    }
    return !out && tn <= tf + 1e-9;
}

// n_batch = 0: beams of camera rays, one per tile -> fb.entry_p.  Otherwise beams of shadow rays, one per tile and light
// of the batch -> fb.entry_s (needs fb.tile_box from k_surface).
template <bool SHADOW>
__global__ void __launch_bounds__(256) k_entry(DevScene sc, DevCamera cam, PixelMap pm, uint32_t n_tiles, uint32_t light0, uint32_t n_batch,
                                               int counting, FrameBufs fb, rt_counters* ctr) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    constexpr bool shadow = SHADOW;
    const uint32_t n_beams = shadow ? n_tiles * n_batch : n_tiles;
    uint32_t vn = 0;
    if (b < n_beams) {
        const uint32_t tile = shadow ? b / n_batch : b;
        double o_lo[3], o_hi[3], d_lo[3], d_hi[3], L[3];
        bool any = true;
        // boxes grow by what the FP64 arithmetic here and the rays' own rounding can lose, and for shadow rays by the
        // 1e-4 their origins are nudged off the surface (raytracer.cpp:221-225)
        double pad = 1e-7 * static_cast<double>(sc.radius);
        if (shadow) {
            const float* t = fb.tile_box + static_cast<size_t>(tile) * 8u;
            for (int k = 0; k < 3; ++k) { o_lo[k] = t[k]; o_hi[k] = t[3 + k]; }
            any = o_lo[0] <= o_hi[0];
            const rt_light Lt = sc.lights[light0 + b % n_batch];
            for (int k = 0; k < 3; ++k) { L[k] = Lt.pos[k]; d_lo[k] = 0.0; d_hi[k] = 0.0; }
            pad += 1.001e-4;
        } else {
            const Pixel p = map_pixel(pm, tile, 0u);             // lane 0 is the tile's top left pixel
            any = p.valid;
            for (int k = 0; k < 3; ++k) { o_lo[k] = cam.pos[k]; o_hi[k] = cam.pos[k]; L[k] = 0.0; d_lo[k] = __builtin_inf(); d_hi[k] = -__builtin_inf(); }
            // directions are linear in the pixel coordinates before normalisation (persp_camera.cpp:110-116): their range
            // over the tile's 8 x 8 pixel squares (sub-pixel offsets included) is spanned by the four corners
            for (int c = 0; c < 4 && any; ++c) {
                const double px = static_cast<double>(p.x + ((c & 1) ? 8 : 0)), py = static_cast<double>(p.y + ((c & 2) ? 8 : 0));
                const double xs = cam.eff_w * (px / cam.width - 0.5), ys = cam.eff_h * (0.5 - py / cam.height), zs = cam.focal;
                const double w[3] = { cam.rot[0]*xs + cam.rot[1]*ys + cam.rot[2]*zs, cam.rot[3]*xs + cam.rot[4]*ys + cam.rot[5]*zs,
                                      cam.rot[6]*xs + cam.rot[7]*ys + cam.rot[8]*zs };
                for (int k = 0; k < 3; ++k) { d_lo[k] = fmin(d_lo[k], w[k]); d_hi[k] = fmax(d_hi[k], w[k]); }
            }
            double wmax = 0.0;
            for (int k = 0; k < 3; ++k) wmax = fmax(wmax, fmax(fabs(d_lo[k]), fabs(d_hi[k])));
            for (int k = 0; k < 3; ++k) { d_lo[k] -= 1e-9 * wmax; d_hi[k] += 1e-9 * wmax; }
        }
        ShadowBeam sb;
        PrimaryBeam pb;
        if (shadow) shadow_beam_init(sb, o_lo, o_hi, L); else primary_beam_init(pb, o_lo, d_lo, d_hi);
        auto touches = [&](float lx, float ly, float lz, float hx, float hy, float hz) -> bool {
            const double lo[3] = { lx - pad, ly - pad, lz - pad }, hi[3] = { hx + pad, hy + pad, hz + pad };
            return shadow ? beam_touches_shadow(sb, lo, hi) : beam_touches_primary(pb, lo, hi);
        };
        // the set: four places, each a reference and its box (REF_NONE = empty place); `closed` marks places whose
        // subtree cannot be refined at present because its touched children would not fit
        uint32_t ref[4] = { REF_NONE, REF_NONE, REF_NONE, REF_NONE };
        float blo[4][3], bhi[4][3];                          // an empty place keeps an inverted box: the walks rely on it
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int a = 0; a < 3; ++a) { blo[k][a] = __builtin_inff(); bhi[k][a] = -__builtin_inff(); }
        uint32_t closed = 0;
        if (any) {
            vn += 1;
            if (touches(sc.root_lo[0], sc.root_lo[1], sc.root_lo[2], sc.root_hi[0], sc.root_hi[1], sc.root_hi[2])) {
                ref[0] = sc.root_ref;
                for (int k = 0; k < 3; ++k) { blo[0][k] = sc.root_lo[k]; bhi[0][k] = sc.root_hi[k]; }
            }
        }
        for (int iter = 0; iter < (SHADOW ? RT_ENTRY_ITERS_SHADOW : RT_ENTRY_ITERS_PRIMARY); ++iter) {
            // the widest subtree that can still be refined goes first: the set then shrinks evenly around the beam
            int pick = -1; float widest = -1.0f; uint32_t count = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (ref[k] == REF_NONE) continue;
                ++count;
                if (!ref_internal(ref[k]) || ((closed >> k) & 1u)) continue;
                const float ex = bhi[k][0] - blo[k][0], ey = bhi[k][1] - blo[k][1], ez = bhi[k][2] - blo[k][2];
                const float size = ex * ey + ey * ez + ez * ex;
                if (size > widest) { widest = size; pick = k; }
            }
            if (pick < 0) break;
            uint32_t node = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k == pick) node = ref[k];
            const float4* n4 = sc.nodes4 + 8u * static_cast<size_t>(node);
            const float4 lox = n4[0], loy = n4[1], loz = n4[2], hix = n4[3], hiy = n4[4], hiz = n4[5], rf = n4[6];
            vn += 4;
            const uint32_t cr[4] = { __float_as_uint(rf.x), __float_as_uint(rf.y), __float_as_uint(rf.z), __float_as_uint(rf.w) };
            const float cl[4][3] = { { lox.x, loy.x, loz.x }, { lox.y, loy.y, loz.y }, { lox.z, loy.z, loz.z }, { lox.w, loy.w, loz.w } };
            const float ch[4][3] = { { hix.x, hiy.x, hiz.x }, { hix.y, hiy.y, hiz.y }, { hix.z, hiy.z, hiz.z }, { hix.w, hiy.w, hiz.w } };
            bool h[4]; uint32_t nh = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                h[c] = cr[c] != REF_NONE && touches(cl[c][0], cl[c][1], cl[c][2], ch[c][0], ch[c][1], ch[c][2]);
                nh += h[c] ? 1u : 0u;
            }
            if (count - 1u + nh > 4u) { closed |= 1u << pick; continue; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k == pick) {
                    ref[k] = REF_NONE;
#pragma unroll
                    for (int a = 0; a < 3; ++a) { blo[k][a] = __builtin_inff(); bhi[k][a] = -__builtin_inff(); }
                }
            if (nh == 0) closed = 0;                              // a place came free: the others get another chance
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!h[c]) continue;
                int place = -1;
#pragma unroll
                for (int k = 3; k >= 0; --k) if (ref[k] == REF_NONE) place = k;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k == place) {
                        ref[k] = cr[c]; closed &= ~(1u << k);
                        for (int a = 0; a < 3; ++a) { blo[k][a] = cl[c][a]; bhi[k][a] = ch[c][a]; }
                    }
            }
        }
        float4* out = (shadow ? fb.entry_s : fb.entry_p) + 8u * static_cast<size_t>(b);      // b = tile, or tile * n_batch + light
        out[0] = make_float4(blo[0][0], blo[1][0], blo[2][0], blo[3][0]);
        out[1] = make_float4(blo[0][1], blo[1][1], blo[2][1], blo[3][1]);
        out[2] = make_float4(blo[0][2], blo[1][2], blo[2][2], blo[3][2]);
        out[3] = make_float4(bhi[0][0], bhi[1][0], bhi[2][0], bhi[3][0]);
        out[4] = make_float4(bhi[0][1], bhi[1][1], bhi[2][1], bhi[3][1]);
        out[5] = make_float4(bhi[0][2], bhi[1][2], bhi[2][2], bhi[3][2]);
        out[6] = make_float4(__uint_as_float(ref[0]), __uint_as_float(ref[1]), __uint_as_float(ref[2]), __uint_as_float(ref[3]));
    }
    if (counting) add_counter(ctr, blockIdx.x % RT_CTR_SHARDS, shadow ? F_VN_S : F_VN_P, vn);
}

// The same search with four lanes per beam, one per place of the set and per child of the node being opened: the four
// box tests of a step run side by side, and there are four times as many waves to overlap the steps' fetches.  The lanes
// of a beam are an aligned quad, so what one needs of another comes by quad broadcasts (DPP) or one lane-indexed read;
// every decision - which subtree to open, whether its children fit, which place a child takes - is the one the
// one-lane search takes, so the entry nodes are the same bit for bit.
template <int K> DEV uint32_t quad_lane(uint32_t v) {           // the value lane K of this lane's quad holds
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), K * 0x55, 0xF, 0xF, true));
}
template <int K> DEV float quad_lane(float v) { return __uint_as_float(quad_lane<K>(__float_as_uint(v))); }
DEV uint32_t quad_bits(bool p) {                                // bit k = p in lane k of this lane's quad
    return static_cast<uint32_t>(__ballot(p) >> (threadIdx.x & 60u)) & 15u;
}
template <bool SHADOW>
__global__ void __launch_bounds__(256) k_entry_quad(DevScene sc, DevCamera cam, PixelMap pm, uint32_t n_tiles, uint32_t light0, uint32_t n_batch,
                                                    int counting, FrameBufs fb, rt_counters* ctr) {
    const uint32_t b = (blockIdx.x * 256u + threadIdx.x) >> 2, q = threadIdx.x & 3u;
    constexpr bool shadow = SHADOW;
    const uint32_t n_beams = shadow ? n_tiles * n_batch : n_tiles;
    uint32_t vn = 0;
    if (b < n_beams) {
        const uint32_t tile = shadow ? b / n_batch : b;
        double o_lo[3], o_hi[3], d_lo[3], d_hi[3], L[3];
        bool any = true;
        double pad = 1e-7 * static_cast<double>(sc.radius);
        if (shadow) {
            const float* t = fb.tile_box + static_cast<size_t>(tile) * 8u;
            for (int k = 0; k < 3; ++k) { o_lo[k] = t[k]; o_hi[k] = t[3 + k]; }
            any = o_lo[0] <= o_hi[0];
            const rt_light Lt = sc.lights[light0 + b % n_batch];
            for (int k = 0; k < 3; ++k) { L[k] = Lt.pos[k]; d_lo[k] = 0.0; d_hi[k] = 0.0; }
            pad += 1.001e-4;
        } else {
            const Pixel p = map_pixel(pm, tile, 0u);
            any = p.valid;
            for (int k = 0; k < 3; ++k) { o_lo[k] = cam.pos[k]; o_hi[k] = cam.pos[k]; L[k] = 0.0; d_lo[k] = __builtin_inf(); d_hi[k] = -__builtin_inf(); }
            for (int c = 0; c < 4 && any; ++c) {
                const double px = static_cast<double>(p.x + ((c & 1) ? 8 : 0)), py = static_cast<double>(p.y + ((c & 2) ? 8 : 0));
                const double xs = cam.eff_w * (px / cam.width - 0.5), ys = cam.eff_h * (0.5 - py / cam.height), zs = cam.focal;
                const double w[3] = { cam.rot[0]*xs + cam.rot[1]*ys + cam.rot[2]*zs, cam.rot[3]*xs + cam.rot[4]*ys + cam.rot[5]*zs,
                                      cam.rot[6]*xs + cam.rot[7]*ys + cam.rot[8]*zs };
                for (int k = 0; k < 3; ++k) { d_lo[k] = fmin(d_lo[k], w[k]); d_hi[k] = fmax(d_hi[k], w[k]); }
            }
            double wmax = 0.0;
            for (int k = 0; k < 3; ++k) wmax = fmax(wmax, fmax(fabs(d_lo[k]), fabs(d_hi[k])));
            for (int k = 0; k < 3; ++k) { d_lo[k] -= 1e-9 * wmax; d_hi[k] += 1e-9 * wmax; }
        }
        ShadowBeam sb;
        PrimaryBeam pb;
        if (shadow) shadow_beam_init(sb, o_lo, o_hi, L); else primary_beam_init(pb, o_lo, d_lo, d_hi);
        auto touches = [&](float lx, float ly, float lz, float hx, float hy, float hz) -> bool {
            const double lo[3] = { lx - pad, ly - pad, lz - pad }, hi[3] = { hx + pad, hy + pad, hz + pad };
            return shadow ? beam_touches_shadow(sb, lo, hi) : beam_touches_primary(pb, lo, hi);
        };
        // this lane's place of the set (an empty place keeps an inverted box: the walks rely on it)
        uint32_t ref = REF_NONE;
        float blo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, bhi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
        bool closed = false;
        if (any) {
            if (q == 0) vn += 1;
            if (q == 0 && touches(sc.root_lo[0], sc.root_lo[1], sc.root_lo[2], sc.root_hi[0], sc.root_hi[1], sc.root_hi[2])) {
                ref = sc.root_ref;
                for (int k = 0; k < 3; ++k) { blo[k] = sc.root_lo[k]; bhi[k] = sc.root_hi[k]; }
            }
        }
        for (int iter = 0; iter < (SHADOW ? RT_ENTRY_ITERS_SHADOW : RT_ENTRY_ITERS_PRIMARY); ++iter) {
            // the widest subtree that can still be refined goes first (the lowest place among equals)
            const uint32_t count = static_cast<uint32_t>(__popc(quad_bits(ref != REF_NONE)));
            float size = -1.0f;
            if (ref != REF_NONE && ref_internal(ref) && !closed) {
                const float ex = bhi[0] - blo[0], ey = bhi[1] - blo[1], ez = bhi[2] - blo[2];
                size = ex * ey + ey * ez + ez * ex;
            }
            const float s0 = quad_lane<0>(size), s1 = quad_lane<1>(size), s2 = quad_lane<2>(size), s3 = quad_lane<3>(size);
            int pick = -1; float widest = -1.0f;
            if (s0 > widest) { widest = s0; pick = 0; }
            if (s1 > widest) { widest = s1; pick = 1; }
            if (s2 > widest) { widest = s2; pick = 2; }
            if (s3 > widest) { widest = s3; pick = 3; }
            if (pick < 0) break;
            const uint32_t r0 = quad_lane<0>(ref), r1 = quad_lane<1>(ref), r2 = quad_lane<2>(ref), r3 = quad_lane<3>(ref);
            const uint32_t node = pick == 0 ? r0 : (pick == 1 ? r1 : (pick == 2 ? r2 : r3));
            // child q of that node
            const float* n4 = reinterpret_cast<const float*>(sc.nodes4 + 8u * static_cast<size_t>(node)) + q;
            const float cl[3] = { n4[0], n4[4], n4[8] }, ch[3] = { n4[12], n4[16], n4[20] };
            const uint32_t cr = __float_as_uint(n4[24]);
            if (q == 0) vn += 4;
            const bool h = cr != REF_NONE && touches(cl[0], cl[1], cl[2], ch[0], ch[1], ch[2]);
            const uint32_t hits = quad_bits(h), nh = static_cast<uint32_t>(__popc(hits));
            if (count - 1u + nh > 4u) { if (static_cast<int>(q) == pick) closed = true; continue; }
            if (static_cast<int>(q) == pick) {
                ref = REF_NONE;
                for (int a = 0; a < 3; ++a) { blo[a] = __builtin_inff(); bhi[a] = -__builtin_inff(); }
            }
            if (nh == 0) closed = false;                          // a place came free: the others get another chance
            // the touched children, in order, take the free places, in order
            const uint32_t free_places = quad_bits(ref == REF_NONE);
            const uint32_t j = static_cast<uint32_t>(__popc(free_places & ((1u << q) - 1u)));       // this place is the j-th free one
            const bool take = ref == REF_NONE && j < nh;
            // the j-th touched child: drop the j lowest set bits of `hits`, take the lowest that is left
            uint32_t rest = hits;
            for (uint32_t d = 0; d < 3u; ++d) if (d < j) rest &= rest - 1u;
            const uint32_t src = (threadIdx.x & 60u) + (rest ? static_cast<uint32_t>(__ffs(static_cast<int>(rest)) - 1) : 0u);
            const uint32_t t_ref = __shfl(cr, static_cast<int>(src), 64);
            const float t_l0 = __shfl(cl[0], static_cast<int>(src), 64), t_l1 = __shfl(cl[1], static_cast<int>(src), 64), t_l2 = __shfl(cl[2], static_cast<int>(src), 64);
            const float t_h0 = __shfl(ch[0], static_cast<int>(src), 64), t_h1 = __shfl(ch[1], static_cast<int>(src), 64), t_h2 = __shfl(ch[2], static_cast<int>(src), 64);
            if (take) {
                ref = t_ref; closed = false;
                blo[0] = t_l0; blo[1] = t_l1; blo[2] = t_l2; bhi[0] = t_h0; bhi[1] = t_h1; bhi[2] = t_h2;
            }
        }
        float* out = reinterpret_cast<float*>((shadow ? fb.entry_s : fb.entry_p) + 8u * static_cast<size_t>(b)) + q;   // b = tile, or tile * n_batch + light
        out[0] = blo[0]; out[4] = blo[1]; out[8] = blo[2]; out[12] = bhi[0]; out[16] = bhi[1]; out[20] = bhi[2];
        out[24] = __uint_as_float(ref);
    }
    if (counting) add_counter(ctr, blockIdx.x % RT_CTR_SHARDS, shadow ? F_VN_S : F_VN_P, vn);
}

// The terms of the lights of one batch that reach surface point i, added in light order (the reference's accumulation
// order, raytracer.cpp:210-260) to `lighting`.  view = origin of the ray that found the surface (raytracer.cpp:185-186).
// MIXED: the terms in mixed precision (light_value_mixed), *bound = an upper bound of the sum's absolute error.
template <int MIXED = 0>          // 0: FP64; 1: mixed precision, FP64 for sharp highlights; 2: mixed precision only
DEV double add_lit_lights(const DevScene& sc, const FrameBufs& fb, const PixelMap& pm, size_t i, uint32_t light0, uint32_t n_batch, const rt_mat_rec& m,
                          double vox, double voy, double voz, double lighting, float* bound = nullptr) {
    const double px = fb.px[i], py = fb.py[i], pz = fb.pz[i], nx = fb.nx[i], ny = fb.ny[i], nz = fb.nz[i];
    // the slot's verdicts: ray (tile, light j, lane) of the shadow launch wrote byte (tile * n_batch + j) * 64 + lane
    uint32_t tile_w, lane;
    slot_to_ray(pm, i, tile_w, lane);
    const uint8_t* lit = fb.lit + (static_cast<size_t>(tile_w) * n_batch) * 64u + lane;
    for (uint32_t j = 0; j < n_batch; ++j) {
        if (!lit[64u * j]) continue;
        const rt_light L = sc.lights[light0 + j];
        if (MIXED) { lighting += light_value_mixed<MIXED == 1>(L, m, px, py, pz, nx, ny, nz, vox, voy, voz, *bound); continue; }
        double dx, dy, dz; normalize3(L.pos[0] - px, L.pos[1] - py, L.pos[2] - pz, dx, dy, dz);   // the shadow ray's direction
        lighting += light_value(L, m, px, py, pz, nx, ny, nz, dx, dy, dz, vox, voy, voz);
    }
    return lighting;
}

// Can a light sum that is off by up to `bound` round a channel of the colour differently?
DEV bool mixed_sum_undecided(const rt_mat_rec& m, uint32_t albedo, double lighting, float bound) {
    const double metallic = clampd(m.metallic, 0.0, 1.0);
    double intensity = (1.0 - metallic) * 0.05 + lighting;
    intensity = intensity < 0.0 ? 0.0 : (intensity > 1.0 ? 1.0 : intensity);
    const double tol = 1e-4 + 255.0 * static_cast<double>(bound);
    bool undecided = !(tol < 0.25);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double c = intensity * static_cast<double>((albedo >> (8 * k)) & 255u);
        const double f = c - __builtin_floor(c);
        undecided = undecided || __builtin_fabs(f - 0.5) < tol;
    }
    return undecided;
}
// The light sum of slot i after this batch, for the resolve kernels.  When the batch holds all the lights the sum is made
// in mixed precision first, with a bound on its error: the colour is round(intensity * albedo) per channel, so it can
// only differ from the FP64 result where a channel's product lies within that error of a half; such slots (with the
// bench scene's lights about one in four hundred) are summed again in FP64, every other one has the FP64 colour already.
// Several batches: FP64 throughout (the running sum waits in fb.lighting).
DEV double resolve_lighting(const DevScene& sc, const FrameBufs& fb, const PixelMap& pm, size_t i, uint32_t light0, uint32_t n_batch, const rt_mat_rec& m,
                            uint32_t albedo, double vox, double voy, double voz, int first, int last) {
    bool exact = true;
    double lighting = 0.0;
#if RT_MIXED_RESOLVE
    if (first && last) {
        float bound = 0.0f;
        lighting = add_lit_lights<1>(sc, fb, pm, i, light0, n_batch, m, vox, voy, voz, 0.0, &bound);
        exact = mixed_sum_undecided(m, albedo, lighting, bound);
    }
#endif
    if (exact) lighting = add_lit_lights<0>(sc, fb, pm, i, light0, n_batch, m, vox, voy, voz, first ? 0.0 : fb.lighting[i]);
    return lighting;
}

// Light terms, then ambient, albedo, gamma.  Lights beyond one batch: the running sum is parked in fb.lighting between
// batches.
__global__ void __launch_bounds__(256) k_resolve(DevScene sc, DevCamera cam, PixelMap pm, FrameBufs fb, uint32_t light0, uint32_t n_batch, int first, int last,
                                                 uint8_t* __restrict__ rgb) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= fb.n) return;
    const int32_t id = fb.id[i];
    if (id < 0) {
        if (last) { rgb[i*3] = sc.gamma[0]; rgb[i*3 + 1] = sc.gamma[0]; rgb[i*3 + 2] = sc.gamma[0]; }
        return;
    }
    const rt_mat_rec m = sc.mats[id];
    const uint32_t albedo = fb.albedo ? fb.albedo[i] : solid_albedo(m);
    const double lighting = resolve_lighting(sc, fb, pm, i, light0, n_batch, m, albedo, cam.pos[0], cam.pos[1], cam.pos[2], first, last);
    if (!last) { fb.lighting[i] = lighting; return; }
    const uint32_t c = finish_shade(m, albedo, lighting);
    rgb[i*3] = sc.gamma[c & 255u]; rgb[i*3 + 1] = sc.gamma[(c >> 8) & 255u]; rgb[i*3 + 2] = sc.gamma[(c >> 16) & 255u];
}

// The same for a frame whose lights fit one batch, in two passes.  The FP64 form of the light terms (its `pow` above all)
// costs the kernel that holds it twice the registers of the mixed-precision form, and only one slot in a few hundred
// needs it.  So the first pass knows the mixed form only: it colours every slot whose sum is decided (resolve_lighting's
// test) and appends the others - and every slot of a material with a highlight too sharp for the mixed form - to a list;
// the second pass colours the listed slots in FP64.  The colours are the ones k_resolve writes.
// list: fb.lighting's memory (unused when there is one batch), as 32-bit slot numbers; *head: their count, zero before.
// lights: sc.lights + light0 once more, as a parameter of its own that promises not to alias anything written here - the
// batch's lights are then scalar operands wherever the loop stands (read through `sc` they are fetched per lane, one
// more trip to memory per light, because the sky colour has been stored by then).
__global__ void __launch_bounds__(256) k_resolve_decided(DevScene sc, DevCamera cam, PixelMap pm, FrameBufs fb, const rt_light* __restrict__ lights,
                                                         uint32_t n_batch, uint8_t* __restrict__ rgb, unsigned int* __restrict__ heads, uint32_t part_cap) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    const bool valid = i < fb.n;
    const size_t ii = valid ? i : 0;
    // Whatever depends on the slot number alone is asked for in one go - the id, the surface point and normal, the
    // verdict bytes of all the batch's lights (ray (tile, light j, lane) of the shadow launch wrote byte
    // (tile * n_batch + j) * 64 + lane) - so that a slot costs two trips to memory (these, then its material), not one
    // per verdict byte.
    int32_t id = fb.id[ii];
    const double px = fb.px[ii], py = fb.py[ii], pz = fb.pz[ii], nx = fb.nx[ii], ny = fb.ny[ii], nz = fb.nz[ii];
    uint32_t tile_w, ray_lane;
    slot_to_ray(pm, ii, tile_w, ray_lane);
    const uint8_t* lit = fb.lit + (static_cast<size_t>(tile_w) * n_batch) * 64u + ray_lane;
    uint32_t verdict[RT_LIGHT_BATCH];                       // places beyond the batch repeat its last byte and are not looked at
#pragma unroll
    for (uint32_t j = 0; j < RT_LIGHT_BATCH; ++j) verdict[j] = lit[64u * (j < n_batch ? j : n_batch - 1u)];
    uint32_t tex_albedo = fb.albedo ? fb.albedo[ii] : 0u;
    asm volatile("" : "+v"(id), "+v"(tex_albedo));                  // the loads stay up here
    bool defer = false;
    if (valid) {
        if (id < 0) { rgb[i*3] = sc.gamma[0]; rgb[i*3 + 1] = sc.gamma[0]; rgb[i*3 + 2] = sc.gamma[0]; }
        else {
            const rt_mat_rec m = sc.mats[id];
            const uint32_t albedo = fb.albedo ? tex_albedo : solid_albedo(m);
            defer = mixed_phong_exp(m) > RT_MIXED_MAX_EXP;
            if (!defer) {
                float bound = 0.0f;
                double lighting = 0.0;
                // every lane goes through every light of the batch and keeps the terms of those that reach it (in light
                // order, as the reference adds them): a branch per lane would not save the wave anything, and the light
                // stays a scalar operand
#pragma unroll
                for (uint32_t j = 0; j < RT_LIGHT_BATCH; ++j) {
                    if (j >= n_batch) break;
                    const rt_light L = lights[j];
                    float b = 0.0f;
                    const double term = light_value_mixed<false>(L, m, px, py, pz, nx, ny, nz, cam.pos[0], cam.pos[1], cam.pos[2], b);
                    lighting = verdict[j] ? lighting + term : lighting;
                    bound = verdict[j] ? bound + b : bound;
                }
                defer = mixed_sum_undecided(m, albedo, lighting, bound);
                if (!defer) {
                    const uint32_t c = finish_shade(m, albedo, lighting);
                    rgb[i*3] = sc.gamma[c & 255u]; rgb[i*3 + 1] = sc.gamma[(c >> 8) & 255u]; rgb[i*3 + 2] = sc.gamma[(c >> 16) & 255u];
                }
            }
        }
    }
    // one addition to a count per wave.  The list is kept in RT_RESOLVE_PARTS parts, each with a count of its own (blocks
    // take them in turn, so part_cap = 256 * the blocks of a part bounds what a part can hold): additions to one word from
    // all over the chip are carried out one after the other on the memory side, and twenty thousand of them on one
    // word - or on eight - cost three times what the second pass saves.
    const unsigned long long dm = __ballot(defer);
    if (dm) {
        const uint32_t lane = threadIdx.x & 63u, part = blockIdx.x % RT_RESOLVE_PARTS;
        unsigned int base = 0;
        if (lane == static_cast<uint32_t>(__ffsll(static_cast<long long>(dm)) - 1)) base = atomicAdd(heads + part, static_cast<unsigned int>(__popcll(dm)));
        base = __shfl(base, __ffsll(static_cast<long long>(dm)) - 1, 64);
        if (defer) reinterpret_cast<uint32_t*>(fb.lighting)[static_cast<size_t>(part) * part_cap + base + static_cast<uint32_t>(__popcll(dm & ((1ull << lane) - 1ull)))] = static_cast<uint32_t>(i);
    }
}
// blockIdx.y: the part of the list; one wave per block and a few blocks per part, all resident at once: a part holds
// a hundred slots or so, unless the scene's highlights are sharp - then it holds all its slots, and the waves loop.
__global__ void __launch_bounds__(64) k_resolve_listed(DevScene sc, DevCamera cam, PixelMap pm, FrameBufs fb, const rt_light* __restrict__ lights,
                                                       uint32_t n_batch, uint8_t* __restrict__ rgb, const unsigned int* __restrict__ heads, uint32_t part_cap) {
    const uint32_t n_listed = heads[blockIdx.y];
    const uint32_t* list = reinterpret_cast<const uint32_t*>(fb.lighting) + static_cast<size_t>(blockIdx.y) * part_cap;
    for (uint32_t j = blockIdx.x * 64u + threadIdx.x; j < n_listed; j += gridDim.x * 64u) {
        const size_t i = list[j];
        // as in the first pass: what depends on the slot number alone in one go, then the material
        int32_t id = fb.id[i];
        const double px = fb.px[i], py = fb.py[i], pz = fb.pz[i], nx = fb.nx[i], ny = fb.ny[i], nz = fb.nz[i];
        uint32_t tile_w, ray_lane;
        slot_to_ray(pm, i, tile_w, ray_lane);
        const uint8_t* lit = fb.lit + (static_cast<size_t>(tile_w) * n_batch) * 64u + ray_lane;
        uint32_t verdict[RT_LIGHT_BATCH];
#pragma unroll
        for (uint32_t k = 0; k < RT_LIGHT_BATCH; ++k) verdict[k] = lit[64u * (k < n_batch ? k : n_batch - 1u)];
        uint32_t tex_albedo = fb.albedo ? fb.albedo[i] : 0u;
        asm volatile("" : "+v"(id), "+v"(tex_albedo));
        const rt_mat_rec m = sc.mats[id];
        const uint32_t albedo = fb.albedo ? tex_albedo : solid_albedo(m);
        double lighting = 0.0;
        for (uint32_t k = 0; k < n_batch; ++k) {
            if (!verdict[k < RT_LIGHT_BATCH ? k : 0]) continue;
            const rt_light L = lights[k];
            double dx, dy, dz; normalize3(L.pos[0] - px, L.pos[1] - py, L.pos[2] - pz, dx, dy, dz);   // the shadow ray's direction
            lighting += light_value(L, m, px, py, pz, nx, ny, nz, dx, dy, dz, cam.pos[0], cam.pos[1], cam.pos[2]);
        }
        const uint32_t c = finish_shade(m, albedo, lighting);
        rgb[i*3] = sc.gamma[c & 255u]; rgb[i*3 + 1] = sc.gamma[(c >> 8) & 255u]; rgb[i*3 + 2] = sc.gamma[(c >> 16) & 255u];
    }
}

// One byte per output slot: bit j set = light j of the batch reaches the slot's surface point; 0 where nothing was hit.
__global__ void __launch_bounds__(256) k_verdict_mask(PixelMap pm, FrameBufs fb, uint32_t n_batch, uint8_t* __restrict__ mask) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= fb.n) return;
    uint32_t m = 0;
    if (fb.id[i] >= 0) {
        uint32_t tile_w, lane;
        slot_to_ray(pm, i, tile_w, lane);
        const uint8_t* lit = fb.lit + (static_cast<size_t>(tile_w) * n_batch) * 64u + lane;
        for (uint32_t j = 0; j < n_batch; ++j) m |= lit[64u * j] ? (1u << j) : 0u;
    }
    mask[i] = static_cast<uint8_t>(m);
}

// Arbitrary rays (origin, direction as 6 doubles): closest hit, optionally shaded (linear colour, no gamma).
template <bool COUNT, bool SHADE, bool INST = false, bool WIDE = false>
__global__ void __launch_bounds__(64) k_trace_rays(DevScene sc, const double* __restrict__ rays, uint64_t n,
                                                   int32_t* __restrict__ ids, double* __restrict__ dist,
                                                   uint8_t* __restrict__ rgb, double* __restrict__ records, rt_counters* ctr) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 64 + threadIdx.x;
    bool hit = false;
    uint32_t vn = 0, vp = 0;
    ShadeCounters sctr = { 0, 0, 0 };
    if (i < n) {
        RayD r;
        r.ox = rays[i*6]; r.oy = rays[i*6 + 1]; r.oz = rays[i*6 + 2];
        r.dx = rays[i*6 + 3]; r.dy = rays[i*6 + 4]; r.dz = rays[i*6 + 5];
        r.dlen = dsqrt(r.dx*r.dx + r.dy*r.dy + r.dz*r.dz);
        Best b;
        hit = traverse<0, COUNT, INST, WIDE>(sc, r, 0.0, b, vn, vp);
        if (ids) ids[i] = b.id;
        if (dist) dist[i] = b.dist;
        if (records) {      // the Hit of the reference (utils.h:68-112): 16 doubles per ray
            double* o = records + i * 16;
            HitD h; h.u = 0.0; h.v = 0.0;
            int32_t mat_id;
            if (hit) best_record<true, INST>(sc, b, r, h, mat_id);
            o[0] = hit ? 1.0 : 0.0; o[1] = static_cast<double>(b.id); o[2] = b.dist;
            o[3] = hit ? h.px : 0.0; o[4] = hit ? h.py : 0.0; o[5] = hit ? h.pz : 0.0;
            o[6] = hit ? h.nx : 0.0; o[7] = hit ? h.ny : 0.0; o[8] = hit ? h.nz : 0.0;
            o[9] = hit ? h.rx : 0.0; o[10] = hit ? h.ry : 0.0; o[11] = hit ? h.rz : 0.0;
            o[12] = hit ? h.u : 0.0; o[13] = hit ? h.v : 0.0; o[14] = 0.0; o[15] = 0.0;
        }
        if (SHADE) {
            uint32_t c = 0;
            if (hit) {
                HitD h; h.u = 0.0; h.v = 0.0;
                int32_t mat_id;
                best_record<true, INST>(sc, b, r, h, mat_id);
                c = shade<COUNT, INST, WIDE>(sc, h, mat_id, r.ox, r.oy, r.oz, sctr);
            }
            rgb[i*3] = c & 255u; rgb[i*3 + 1] = (c >> 8) & 255u; rgb[i*3 + 2] = (c >> 16) & 255u;
        }
    }
    const uint32_t shard = blockIdx.x % RT_CTR_SHARDS;
    add_counter(ctr, shard, F_PRIMARY, i < n ? 1 : 0);
    add_counter(ctr, shard, F_HITS, hit ? 1 : 0);
    if (SHADE) add_counter(ctr, shard, F_SHADOW, sctr.shadow_rays);
    if (COUNT) {
        add_counter(ctr, shard, F_VN_P, vn); add_counter(ctr, shard, F_VP_P, vp);
        if (SHADE) { add_counter(ctr, shard, F_VN_S, sctr.vn); add_counter(ctr, shard, F_VP_S, sctr.vp); }
    }
}

// RayTracer::one_pass_intersection as the reference runs it (raytracer.cpp:143-157): every primitive against every ray,
// no hierarchy.  The loop index is the same for all lanes, so a record is fetched once per wave and every lane runs the
// same intersector; records sit in leaf order, not id order, so equal distances are resolved to the lowest id explicitly
// (the reference's strict '<' over ascending ids).  rays = null: ray i is the camera ray of pixel i of the rectangle in pm
// (row-major).  Same outputs as k_trace_rays.
__global__ void __launch_bounds__(256) k_brute_rays(DevScene sc, DevCamera cam, PixelMap pm, const double* __restrict__ rays, uint64_t n,
                                                    int32_t* __restrict__ ids, double* __restrict__ dist, double* __restrict__ records,
                                                    int counting, rt_counters* ctr) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    bool hit = false;
    uint32_t vp = 0;
    if (i < n) {
        RayD r;
        if (rays) {
            r.ox = rays[i*6]; r.oy = rays[i*6 + 1]; r.oz = rays[i*6 + 2];
            r.dx = rays[i*6 + 3]; r.dy = rays[i*6 + 4]; r.dz = rays[i*6 + 5];
            r.dlen = dsqrt(r.dx*r.dx + r.dy*r.dy + r.dz*r.dz);
        } else {
            primary_ray(cam, pm.x0 + static_cast<int>(i % static_cast<uint64_t>(pm.w)), pm.y0 + static_cast<int>(i / static_cast<uint64_t>(pm.w)), r);
        }
        Best b; b.dist = __builtin_inf(); b.slot = 0xFFFFFFFFu; b.id = -1; b.inst = REF_NONE;
        const uint32_t n_recs = sc.n_slots + sc.n_loose;
        for (uint32_t s = 0; s < n_recs; ++s) {
            const rt_prim_rec* rec = sc.recs + s;
            const uint32_t kind = rec->kind;
            const int32_t id = static_cast<int32_t>(rec->id);
            HitD h;
            ++vp;
            if (!hit_record<false>(rec, kind, r, h)) continue;
            if (h.dist < b.dist || (h.dist == b.dist && id < b.id)) { b.dist = h.dist; b.slot = s; b.id = id; }
        }
        hit = b.id >= 0;
        if (ids) ids[i] = b.id;
        if (dist) dist[i] = b.dist;
        if (records) {
            double* o = records + i * 16;
            HitD h; h.u = 0.0; h.v = 0.0;
            int32_t mat_id;
            if (hit) best_record<true, false>(sc, b, r, h, mat_id);
            o[0] = hit ? 1.0 : 0.0; o[1] = static_cast<double>(b.id); o[2] = b.dist;
            o[3] = hit ? h.px : 0.0; o[4] = hit ? h.py : 0.0; o[5] = hit ? h.pz : 0.0;
            o[6] = hit ? h.nx : 0.0; o[7] = hit ? h.ny : 0.0; o[8] = hit ? h.nz : 0.0;
            o[9] = hit ? h.rx : 0.0; o[10] = hit ? h.ry : 0.0; o[11] = hit ? h.rz : 0.0;
            o[12] = hit ? h.u : 0.0; o[13] = hit ? h.v : 0.0; o[14] = 0.0; o[15] = 0.0;
        }
    }
    const uint32_t shard = blockIdx.x % RT_CTR_SHARDS;
    add_counter(ctr, shard, F_PRIMARY, i < n ? 1 : 0);
    add_counter(ctr, shard, F_HITS, hit ? 1 : 0);
    if (counting) add_counter(ctr, shard, F_VP_P, vp);
}

// Mesh::intersect of one primitive for n rays (mesh.cpp:159, 371, 597, 795): the record layout of k_trace_rays with the
// primitive id slot holding 0.
__global__ void __launch_bounds__(64) k_intersect_one(const rt_prim_rec* __restrict__ rec, const double* __restrict__ rays, uint64_t n,
                                                      double* __restrict__ records) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 64 + threadIdx.x;
    if (i >= n) return;
    RayD r;
    r.ox = rays[i*6]; r.oy = rays[i*6 + 1]; r.oz = rays[i*6 + 2];
    r.dx = rays[i*6 + 3]; r.dy = rays[i*6 + 4]; r.dz = rays[i*6 + 5];
    r.dlen = dsqrt(r.dx*r.dx + r.dy*r.dy + r.dz*r.dz);
    HitD h; h.u = 0.0; h.v = 0.0;
    const bool hit = hit_record<true, true>(rec, rec->kind, r, h);
    double* o = records + i * 16;
    o[0] = hit ? 1.0 : 0.0; o[1] = hit ? 0.0 : -1.0; o[2] = hit ? h.dist : __builtin_inf();
    o[3] = hit ? h.px : 0.0; o[4] = hit ? h.py : 0.0; o[5] = hit ? h.pz : 0.0;
    o[6] = hit ? h.nx : 0.0; o[7] = hit ? h.ny : 0.0; o[8] = hit ? h.nz : 0.0;
    o[9] = hit ? h.rx : 0.0; o[10] = hit ? h.ry : 0.0; o[11] = hit ? h.rz : 0.0;
    o[12] = hit ? h.u : 0.0; o[13] = hit ? h.v : 0.0; o[14] = 0.0; o[15] = 0.0;
}

// RayTracer::shade for caller-supplied hits: 10 doubles each (point, normal, primitive id, u, v, has-uv flag) and a
// view origin (3 doubles) per hit; linear 8-bit colour out.
template <bool INST, bool WIDE>
__global__ void __launch_bounds__(64) k_shade_hits(DevScene sc, const double* __restrict__ hits, const double* __restrict__ view,
                                                   uint64_t n, uint8_t* __restrict__ rgb, rt_counters* ctr) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 64 + threadIdx.x;
    ShadeCounters sctr = { 0, 0, 0 };
    if (i < n) {
        HitD h;
        h.px = hits[i*10]; h.py = hits[i*10 + 1]; h.pz = hits[i*10 + 2];
        h.nx = hits[i*10 + 3]; h.ny = hits[i*10 + 4]; h.nz = hits[i*10 + 5];
        h.u = hits[i*10 + 7]; h.v = hits[i*10 + 8];
        const double idf = hits[i*10 + 6];
        uint32_t c = 0;
        if (idf >= 0.0 && idf < static_cast<double>(sc.n_prims)) {
            const rt_mat_rec m = sc.mats[static_cast<int32_t>(idf)];
            // a texture is only consulted when the Hit carries surface coordinates (mesh.cpp:23)
            const uint32_t alb = (hits[i*10 + 9] != 0.0) ? albedo_at(sc, m, h.u, h.v) : solid_albedo(m);
            c = shade_material<false, INST, WIDE>(sc, h, m, alb, view[i*3], view[i*3 + 1], view[i*3 + 2], sctr);
        } else if (idf == -2.0) {
            // a hit that names no primitive: white, non-metallic, roughness 0.5 (raytracer.cpp:204-208, 250-254, 268-272)
            rt_mat_rec m;
            m.metallic = 0.0; m.roughness = 0.5; m.ior = 1.5; m.albedo[0] = 255; m.albedo[1] = 255; m.albedo[2] = 255;
            m.tex = -1;
            c = shade_material<false, INST, WIDE>(sc, h, m, 0x00FFFFFFu, view[i*3], view[i*3 + 1], view[i*3 + 2], sctr);
        }
        rgb[i*3] = c & 255u; rgb[i*3 + 1] = (c >> 8) & 255u; rgb[i*3 + 2] = (c >> 16) & 255u;
    }
    add_counter(ctr, blockIdx.x % RT_CTR_SHARDS, F_SHADOW, sctr.shadow_rays);
}

#ifndef RT_BOUNCE_RESOLVE_WAVES
#define RT_BOUNCE_RESOLVE_WAVES 4      // waves per SIMD the kernel below is compiled for: it streams 250 bytes per slot, and three waves are too few for that
#endif
// Mirror bounces (trace_ray_recursive, raytracer.cpp:281-371) as a level-by-level pipeline: each level is traced and
// shadow-tested by the same ray kernels as a plain frame; this kernel closes a level - local colour, Fresnel weight,
// tint, and the ray of the next level - and k_bounce_fold blends the levels back up once the chain has ended.
__global__ void __launch_bounds__(256, RT_BOUNCE_RESOLVE_WAVES) k_bounce_resolve(DevScene sc, DevCamera cam, PixelMap pm, FrameBufs fb, LevelBufs lv, uint32_t light0, uint32_t n_batch, int first, int last,
                                                        int max_scatters, double min_scatter_intensity) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= fb.n) return;
    const int level = fb.level;
    if (level > 0 && !fb.alive[i]) return;
    const int32_t id = fb.id[i];
    uint32_t local = 0;
    if (id >= 0) {
        const rt_mat_rec m0 = sc.mats[id];
        double vox = cam.pos[0], voy = cam.pos[1], voz = cam.pos[2];
        if (level > 0) { vox = fb.ray[i]; voy = fb.ray[fb.n + i]; voz = fb.ray[2*fb.n + i]; }
        const uint32_t albedo0 = fb.albedo ? fb.albedo[i] : solid_albedo(m0);
        const double lighting = resolve_lighting(sc, fb, pm, i, light0, n_batch, m0, albedo0, vox, voy, voz, first, last);
        if (!last) { fb.lighting[i] = lighting; return; }
        local = finish_shade(m0, albedo0, lighting);
    } else if (!last) return;
    lv.local[static_cast<size_t>(level) * fb.n + i] = local;
    lv.last[i] = static_cast<uint8_t>(level);
    bool go = false;
    if (id >= 0 && level < max_scatters) {
        const rt_mat_rec m = sc.mats[id];
        const double dx = fb.ray[3*fb.n + i], dy = fb.ray[4*fb.n + i], dz = fb.ray[5*fb.n + i];
        const double nx = fb.rawn[i], ny = fb.rawn[fb.n + i], nz = fb.rawn[2*fb.n + i];
        const double dlen = dsqrt(dx*dx + dy*dy + dz*dz);
        const double inv_len = (dlen > 1e-15) ? (1.0 / dlen) : 1.0;
        const double vx = -dx * inv_len, vy = -dy * inv_len, vz = -dz * inv_len;
        const double ct = nx*vx + ny*vy + nz*vz;
        const double cos_theta = (0.0 < ct) ? ct : 0.0;
        const uint32_t alb = fb.albedo ? fb.albedo[i] : solid_albedo(m);
        double f0;
        if (m.metallic <= 0.0) { const double r0 = (m.ior - 1.0) / (m.ior + 1.0); f0 = clampd(r0 * r0, 0.0, 1.0); }
        else {
            const double avg = (static_cast<double>(alb & 255u) + static_cast<double>((alb >> 8) & 255u) + static_cast<double>((alb >> 16) & 255u)) / (3.0 * 255.0);
            f0 = clampd(avg, 0.0, 1.0);
        }
        const double fr = f0 + (1.0 - f0) * pow(1.0 - cos_theta, 5.0);
        if (!(fr <= 0.0 || fr < min_scatter_intensity)) {
            go = true;
            lv.fr[static_cast<size_t>(level) * fb.n + i] = fr;
            lv.tint[static_cast<size_t>(level) * fb.n + i] = alb | (m.metallic > 0.0 ? 1u << 24 : 0u);
            // mirror direction exactly as the intersectors leave it in the Hit (mesh.cpp:236-247 and siblings)
            const double ddn = dx*nx + dy*ny + dz*nz;
            const double tx = dx - 2.0 * ddn * nx, ty = dy - 2.0 * ddn * ny, tz = dz - 2.0 * ddn * nz;
            double len = dsqrt(tx*tx + ty*ty + tz*tz);
            if (len < 1e-15) len = 1.0;
            const double rx = tx/len, ry = ty/len, rz = tz/len;
            const double eps = 1e-6;                                        // raytracer.cpp:332-336
            fb.ray[i] = fb.px[i] + rx * eps; fb.ray[fb.n + i] = fb.py[i] + ry * eps; fb.ray[2*fb.n + i] = fb.pz[i] + rz * eps;
            fb.ray[3*fb.n + i] = rx; fb.ray[4*fb.n + i] = ry; fb.ray[5*fb.n + i] = rz;
        }
    }
    fb.alive[i] = go ? 1 : 0;
}

// Fold the 8-bit blends back up the chain (raytracer.cpp:342-370), then the display transfer.
__global__ void __launch_bounds__(256) k_bounce_fold(DevScene sc, LevelBufs lv, size_t n, uint8_t* __restrict__ rgb) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const int last = lv.last[i];
    uint32_t colour = lv.local[static_cast<size_t>(last) * n + i];
    for (int k = last - 1; k >= 0; --k) {
        const uint32_t tint = lv.tint[static_cast<size_t>(k) * n + i];
        const uint32_t loc_k = lv.local[static_cast<size_t>(k) * n + i];
        const double fr = lv.fr[static_cast<size_t>(k) * n + i];
        uint32_t out = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const double sec = static_cast<double>((colour >> (8*ch)) & 255u);
            double tinted = sec;
            if (tint >> 24) tinted = static_cast<double>(round_u8((static_cast<double>((tint >> (8*ch)) & 255u) / 255.0) * sec));
            const double loc = static_cast<double>((loc_k >> (8*ch)) & 255u);
            out |= round_u8((1.0 - fr) * loc + fr * tinted) << (8*ch);
        }
        colour = out;
    }
    rgb[i*3] = sc.gamma[colour & 255u]; rgb[i*3 + 1] = sc.gamma[(colour >> 8) & 255u]; rgb[i*3 + 2] = sc.gamma[(colour >> 16) & 255u];
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

// Fixed workspace of rt_render_rect_static: the frame working set and the queue heads of one call, for callers that
// must not allocate.  The array's name is a host-side handle, not a device address: host code asks the runtime for
// this GPU's address once per device and uses that pointer everywhere (kernel arguments, hipMemsetAsync).
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

extern "C" hipError_t rt_static_scratch(int device, void** ptr, size_t* bytes) {
    constexpr int kMaxDevices = 64;
    static void* per_device[kMaxDevices] = {};
    static std::mutex guard;
    if (!ptr || !bytes || device < 0 || device >= kMaxDevices) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(guard);
    if (!per_device[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        per_device[device] = p;
    }
    *ptr = per_device[device];
    *bytes = sizeof(scratch);
    return hipSuccess;
}

// Multi-sample frames: per-sample linear 8-bit colours are summed per channel, and the mean is rounded and gamma
// encoded once (raytracer.cpp:419-436).
__global__ void __launch_bounds__(256) k_accumulate(const uint8_t* __restrict__ rgb, uint32_t* __restrict__ acc, size_t n, int first) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) acc[i] = (first ? 0u : acc[i]) + rgb[i];
}

__global__ void __launch_bounds__(256) k_finish_samples(const uint32_t* __restrict__ acc, size_t n, int spp, const uint8_t* __restrict__ gamma,
                                                        uint8_t* __restrict__ rgb) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    int m = static_cast<int>(__builtin_round(static_cast<double>(acc[i]) / static_cast<double>(spp)));
    m = m < 0 ? 0 : (m > 255 ? 255 : m);
    rgb[i] = gamma[m];
}

extern "C" hipError_t rt_launch_accumulate(hipStream_t st, const uint8_t* rgb, uint32_t* acc, size_t n, int first) {
    if (n) hipLaunchKernelGGL(k_accumulate, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, st, rgb, acc, n, first);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_finish_samples(hipStream_t st, const uint32_t* acc, size_t n, int spp, const uint8_t* gamma, uint8_t* rgb) {
    if (n) hipLaunchKernelGGL(k_finish_samples, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, st, acc, n, spp, gamma, rgb);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_primary(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm,
                                        uint32_t n_tiles, uint32_t n_waves, int counting, int32_t* ids, double* dist,
                                        const FrameBufs* fb, rt_counters* ctr, unsigned int* queue, int refill_min) {
    const FrameBufs none{};
    const uint32_t n_rays = n_tiles * 64u;
    const dim3 grid(n_waves), block(64);
    const int variant = !fb ? 0 : (1 | (fb->ray ? 2 : 0));
    const int flags = !fb ? 0 : (variant | (fb->albedo ? 4 : 0));
#define RT_LAUNCH_PRIMARY_IW(R, FB, I, W)                                                                                                \
    do {                                                                                                                                 \
        if (counting) hipLaunchKernelGGL((k_rays<0, true, R, I, W>), grid, block, 0, st, *sc, *cam, *pm, n_tiles, 0u, n_rays, ids, dist, FB, ctr, queue, refill_min);  \
        else          hipLaunchKernelGGL((k_rays<0, false, R, I, W>), grid, block, 0, st, *sc, *cam, *pm, n_tiles, 0u, n_rays, ids, dist, FB, ctr, queue, refill_min); \
    } while (0)
#define RT_LAUNCH_PRIMARY_I(R, FB, I)                                                                                                    \
    do { if (sc->nodes4) RT_LAUNCH_PRIMARY_IW(R, FB, I, true); else RT_LAUNCH_PRIMARY_IW(R, FB, I, false); } while (0)
#define RT_LAUNCH_SURFACE(F, I) hipLaunchKernelGGL((k_surface<F, I>), dim3((n_tiles + 3u) / 4u), dim3(256), 0, st, *sc, *cam, *pm, n_tiles, *fb)
    if (sc->inst) {                     // two-level scene (no textures there)
        switch (variant) {
            case 0: RT_LAUNCH_PRIMARY_I(0, none, true); break;
            case 1: RT_LAUNCH_PRIMARY_I(1, *fb, true); RT_LAUNCH_SURFACE(1, true); break;
            default: RT_LAUNCH_PRIMARY_I(3, *fb, true); RT_LAUNCH_SURFACE(3, true); break;
        }
        return hipGetLastError();
    }
    switch (variant) {
        case 0: RT_LAUNCH_PRIMARY_I(0, none, false); break;
        case 1: RT_LAUNCH_PRIMARY_I(1, *fb, false); break;
        default: RT_LAUNCH_PRIMARY_I(3, *fb, false); break;
    }
    switch (flags) {
        case 0: break;
        case 1: RT_LAUNCH_SURFACE(1, false); break;
        case 3: RT_LAUNCH_SURFACE(3, false); break;
        case 5: RT_LAUNCH_SURFACE(5, false); break;
        default: RT_LAUNCH_SURFACE(7, false); break;
    }
#undef RT_LAUNCH_SURFACE
#undef RT_LAUNCH_PRIMARY_I
#undef RT_LAUNCH_PRIMARY_IW
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_entry_search(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, uint32_t n_tiles,
                                             uint32_t light0, uint32_t n_batch, int counting, const FrameBufs* fb, rt_counters* ctr) {
    const uint64_t n_beams = static_cast<uint64_t>(n_tiles) * (n_batch ? n_batch : 1u);
    if (!sc->nodes4 || !fb || !fb->entry_p || n_beams == 0) return hipSuccess;
    // One lane per beam is paced by its own instruction stream once there are eight or more waves per SIMD; four lanes
    // per beam run more instructions in total but a quarter as many per wave, which is what counts while the launch is
    // too small to fill the SIMDs: a rank's share of a tiled frame, or any frame up to 4K (searches of the 4K frame 0.217
    // against 0.212 ms, of an eighth of it 0.110 against 0.060; of the 8K frame 0.64 against 0.72).
    if (n_beams > RT_ENTRY_QUAD_MAX_BEAMS) {
        const dim3 grid(static_cast<uint32_t>((n_beams + 255u) / 256u));
        if (n_batch) hipLaunchKernelGGL(k_entry<true>, grid, dim3(256), 0, st, *sc, *cam, *pm, n_tiles, light0, n_batch, counting, *fb, ctr);
        else         hipLaunchKernelGGL(k_entry<false>, grid, dim3(256), 0, st, *sc, *cam, *pm, n_tiles, light0, n_batch, counting, *fb, ctr);
        return hipGetLastError();
    }
    const dim3 grid(static_cast<uint32_t>((n_beams + 63u) / 64u));          // four lanes per beam
    if (n_batch) hipLaunchKernelGGL(k_entry_quad<true>, grid, dim3(256), 0, st, *sc, *cam, *pm, n_tiles, light0, n_batch, counting, *fb, ctr);
    else         hipLaunchKernelGGL(k_entry_quad<false>, grid, dim3(256), 0, st, *sc, *cam, *pm, n_tiles, light0, n_batch, counting, *fb, ctr);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_shadow(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm,
                                       uint32_t n_tiles, uint32_t n_waves, uint32_t light0, uint32_t n_batch, int counting,
                                       const FrameBufs* fb, rt_counters* ctr, unsigned int* queue, int refill_min) {
    const uint32_t n_rays = n_tiles * n_batch * 64u;
#define RT_LAUNCH_SHADOW(I, W)                                                                                                           \
    do {                                                                                                                                 \
        if (counting) hipLaunchKernelGGL((k_rays<1, true, 0, I, W>), dim3(n_waves), dim3(64), 0, st, *sc, *cam, *pm, n_tiles, light0, n_rays, nullptr, nullptr, *fb, ctr, queue, refill_min);  \
        else          hipLaunchKernelGGL((k_rays<1, false, 0, I, W>), dim3(n_waves), dim3(64), 0, st, *sc, *cam, *pm, n_tiles, light0, n_rays, nullptr, nullptr, *fb, ctr, queue, refill_min); \
    } while (0)
    if (sc->inst) { if (sc->nodes4) RT_LAUNCH_SHADOW(true, true); else RT_LAUNCH_SHADOW(true, false); }
    else          { if (sc->nodes4) RT_LAUNCH_SHADOW(false, true); else RT_LAUNCH_SHADOW(false, false); }
#undef RT_LAUNCH_SHADOW
    return hipGetLastError();
}

// list_heads: RT_RESOLVE_PARTS zeroed 32-bit words for the two-pass form (taken when the batch holds all the lights), or null.
extern "C" hipError_t rt_launch_resolve(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, const FrameBufs* fb, uint32_t light0,
                                        uint32_t n_batch, int first, int last, uint8_t* rgb, unsigned int* list_heads) {
    const uint32_t nb = static_cast<uint32_t>((fb->n + 255) / 256);
    if (nb == 0) return hipSuccess;
#if RT_MIXED_RESOLVE
    // the list lives in fb->lighting (8 n bytes, unused when there is one batch): RT_RESOLVE_PARTS parts of part_cap slot numbers
    const uint32_t per_part = (nb + RT_RESOLVE_PARTS - 1u) / RT_RESOLVE_PARTS;
    const uint64_t part_cap = 256ull * per_part;
    if (first && last && list_heads && 4ull * RT_RESOLVE_PARTS * part_cap <= 8ull * fb->n && fb->n < (1ull << 32)) {
        hipLaunchKernelGGL(k_resolve_decided, dim3(nb), dim3(256), 0, st, *sc, *cam, *pm, *fb, sc->lights + light0, n_batch, rgb, list_heads,
                           static_cast<uint32_t>(part_cap));
        hipLaunchKernelGGL(k_resolve_listed, dim3(per_part < 2u ? 4u * per_part : 8u, RT_RESOLVE_PARTS), dim3(64), 0, st, *sc, *cam, *pm, *fb, sc->lights + light0, n_batch, rgb,
                           list_heads, static_cast<uint32_t>(part_cap));
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(k_resolve, dim3(nb), dim3(256), 0, st, *sc, *cam, *pm, *fb, light0, n_batch, first, last, rgb);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_verdict_mask(hipStream_t st, const PixelMap* pm, const FrameBufs* fb, uint32_t n_batch, uint8_t* mask) {
    const uint32_t nb = static_cast<uint32_t>((fb->n + 255) / 256);
    if (nb) hipLaunchKernelGGL(k_verdict_mask, dim3(nb), dim3(256), 0, st, *pm, *fb, n_batch, mask);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_trace_rays(hipStream_t st, const DevScene* sc, const double* rays, uint64_t n, int counting, int shade_it,
                                           int32_t* ids, double* dist, uint8_t* rgb, double* records, rt_counters* ctr) {
    const uint32_t nb = static_cast<uint32_t>((n + 63) / 64);
    if (nb == 0) return hipSuccess;
#define RT_LAUNCH_TRACE(I, W)                                                                                                                      \
    do {                                                                                                                                           \
        if (shade_it) {                                                                                                                            \
            if (counting) hipLaunchKernelGGL((k_trace_rays<true, true, I, W>), dim3(nb), dim3(64), 0, st, *sc, rays, n, ids, dist, rgb, records, ctr);  \
            else          hipLaunchKernelGGL((k_trace_rays<false, true, I, W>), dim3(nb), dim3(64), 0, st, *sc, rays, n, ids, dist, rgb, records, ctr); \
        } else {                                                                                                                                   \
            if (counting) hipLaunchKernelGGL((k_trace_rays<true, false, I, W>), dim3(nb), dim3(64), 0, st, *sc, rays, n, ids, dist, rgb, records, ctr); \
            else          hipLaunchKernelGGL((k_trace_rays<false, false, I, W>), dim3(nb), dim3(64), 0, st, *sc, rays, n, ids, dist, rgb, records, ctr);\
        }                                                                                                                                          \
    } while (0)
    if (sc->inst) { if (sc->nodes4) RT_LAUNCH_TRACE(true, true); else RT_LAUNCH_TRACE(true, false); }
    else          { if (sc->nodes4) RT_LAUNCH_TRACE(false, true); else RT_LAUNCH_TRACE(false, false); }
#undef RT_LAUNCH_TRACE
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_brute(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, const double* rays, uint64_t n,
                                      int counting, int32_t* ids, double* dist, double* records, rt_counters* ctr) {
    const uint32_t nb = static_cast<uint32_t>((n + 255) / 256);
    if (nb) hipLaunchKernelGGL(k_brute_rays, dim3(nb), dim3(256), 0, st, *sc, *cam, *pm, rays, n, ids, dist, records, counting, ctr);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_intersect_one(hipStream_t st, const rt_prim_rec* rec, const double* rays, uint64_t n, double* records) {
    const uint32_t nb = static_cast<uint32_t>((n + 63) / 64);
    if (nb) hipLaunchKernelGGL(k_intersect_one, dim3(nb), dim3(64), 0, st, rec, rays, n, records);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_shade_hits(hipStream_t st, const DevScene* sc, const double* hits, const double* view, uint64_t n,
                                           uint8_t* rgb, rt_counters* ctr) {
    const uint32_t nb = static_cast<uint32_t>((n + 63) / 64);
    if (nb && sc->inst && sc->nodes4) hipLaunchKernelGGL((k_shade_hits<true, true>), dim3(nb), dim3(64), 0, st, *sc, hits, view, n, rgb, ctr);
    else if (nb && sc->inst)   hipLaunchKernelGGL((k_shade_hits<true, false>), dim3(nb), dim3(64), 0, st, *sc, hits, view, n, rgb, ctr);
    else if (nb && sc->nodes4) hipLaunchKernelGGL((k_shade_hits<false, true>), dim3(nb), dim3(64), 0, st, *sc, hits, view, n, rgb, ctr);
    else if (nb)               hipLaunchKernelGGL((k_shade_hits<false, false>), dim3(nb), dim3(64), 0, st, *sc, hits, view, n, rgb, ctr);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_bounce_resolve(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, const FrameBufs* fb, const LevelBufs* lv,
                                               uint32_t light0, uint32_t n_batch, int first, int last, int max_scatters, double min_si) {
    const uint32_t nb = static_cast<uint32_t>((fb->n + 255) / 256);
    if (nb) hipLaunchKernelGGL(k_bounce_resolve, dim3(nb), dim3(256), 0, st, *sc, *cam, *pm, *fb, *lv, light0, n_batch, first, last, max_scatters, min_si);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_bounce_fold(hipStream_t st, const DevScene* sc, const LevelBufs* lv, size_t n, uint8_t* rgb) {
    const uint32_t nb = static_cast<uint32_t>((n + 255) / 256);
    if (nb) hipLaunchKernelGGL(k_bounce_fold, dim3(nb), dim3(256), 0, st, *sc, *lv, n, rgb);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_assemble(hipStream_t st, int width, int height, int tile, int tiles_x, int n_ranks, int tiles_per_rank,
                                         const uint8_t* gathered, uint8_t* frame) {
    hipLaunchKernelGGL(k_assemble_tiles, dim3((width + 15) / 16, (height + 15) / 16), dim3(256), 0, st,
                       width, height, tile, tiles_x, n_ranks, tiles_per_rank, gathered, frame);
    return hipGetLastError();
}
