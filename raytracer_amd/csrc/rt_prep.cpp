// Host-side, per-primitive precomputation.  Everything that depends only on the primitive is derived
// here once, with exactly the reference's sequence of IEEE double operations, so that the device-side
// intersectors reproduce the reference's distances bit for bit without ever calling sin/cos.
// Build with -ffp-contract=off (no FMA contraction).
#include <cmath>
#include <cstring>
#include <limits>

#include "rt_scene.h"

namespace {

inline double inv_or_zero(double s) { return (s != 0.0) ? (1.0 / s) : 0.0; }

inline bool all_finite(const double* v, int n) {
    for (int i = 0; i < n; ++i) if (!std::isfinite(v[i])) return false;
    return true;
}

inline void box_reset(rt_box64& b) {
    for (int k = 0; k < 3; ++k) { b.lo[k] = std::numeric_limits<double>::infinity(); b.hi[k] = -b.lo[k]; }
}
inline void box_add(rt_box64& b, const double p[3]) {
    for (int k = 0; k < 3; ++k) { if (p[k] < b.lo[k]) b.lo[k] = p[k]; if (p[k] > b.hi[k]) b.hi[k] = p[k]; }
}

// 8 corners of the scaled, rotated, translated [-1,1]^3 box: the hit points of cube and cylinder
// intersections all lie inside it (compute_aabb, s2899221/Code/mesh.cpp:288-309, 517-538).
void corners_box(const double T[3], const double R[9], const double S[3], rt_box64& b) {
    box_reset(b);
    for (int dx = -1; dx <= 1; dx += 2)
        for (int dy = -1; dy <= 1; dy += 2)
            for (int dz = -1; dz <= 1; dz += 2) {
                const double ps[3] = { dx * S[0], dy * S[1], dz * S[2] };
                const double pw[3] = { R[0]*ps[0] + R[1]*ps[1] + R[2]*ps[2] + T[0],
                                       R[3]*ps[0] + R[4]*ps[1] + R[5]*ps[2] + T[1],
                                       R[6]*ps[0] + R[7]*ps[1] + R[8]*ps[2] + T[2] };
                box_add(b, pw);
            }
}

} // namespace

// R = Rz(yaw) * (Ry(pitch) * Rx(roll)) through two full 3x3 products, structural zeros and ones
// included, so each entry carries the reference's rounding (build_rotation_matrix_rpy, mesh.cpp:122-145).
void rt_rotation_rpy(double roll, double pitch, double yaw, double R[9]) {
    // The reference takes cos and sin of each angle side by side (mesh.cpp:123-125); g++ -O2, the build its results
    // are pinned on, turns each pair into one sincos() call, and glibc's sincos() differs from its sin() in the last
    // bit for about one angle in a hundred.  Call sincos() outright so that the matrix does not depend on which
    // compiler built this file.
    double cr, sr, cp, sp, cy, sy;
    ::sincos(roll, &sr, &cr);
    ::sincos(pitch, &sp, &cp);
    ::sincos(yaw, &sy, &cy);
    const double Z[9] = { cy, -sy, 0, sy, cy, 0, 0, 0, 1 };
    const double Y[9] = { cp, 0, sp, 0, 1, 0, -sp, 0, cp };
    const double X[9] = { 1, 0, 0, 0, cr, -sr, 0, sr, cr };
    double YX[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            YX[i*3+j] = Y[i*3+0]*X[0*3+j] + Y[i*3+1]*X[1*3+j] + Y[i*3+2]*X[2*3+j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i*3+j] = Z[i*3+0]*YX[0*3+j] + Z[i*3+1]*YX[1*3+j] + Z[i*3+2]*YX[2*3+j];
}

rt_prim_class rt_make_record(const rt_prim_desc& p, uint32_t id, rt_prim_rec& rec, rt_box64& box) {
    std::memset(&rec, 0, sizeof(rec));
    rec.kind = static_cast<uint32_t>(p.kind);
    rec.id = id;
    box_reset(box);
    const double* q = p.q;
    switch (p.kind) {
    case RT_CUBE:
    case RT_CYLINDER: {
        double* d = rec.d;
        d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
        rt_rotation_rpy(q[3], q[4], q[5], d + 3);
        for (int k = 0; k < 3; ++k) { d[12 + k] = q[6 + k]; d[15 + k] = inv_or_zero(q[6 + k]); }
        // A zero scale makes the local ray coordinate on that axis identically zero, so the primitive
        // is hit by rays anywhere along that axis: not bounded by its corner box.
        if (!all_finite(q, 9) || q[6] == 0.0 || q[7] == 0.0 || q[8] == 0.0) return RT_CLASS_LOOSE;
        corners_box(d, d + 3, d + 12, box);
        return RT_CLASS_BOUNDED;
    }
    case RT_SPHERE: {
        double* d = rec.d;
        for (int k = 0; k < 3; ++k) { d[k] = q[k]; d[3 + k] = q[3 + k]; d[6 + k] = inv_or_zero(q[3 + k]); }
        if (!all_finite(q, 6) || q[3] == 0.0 || q[4] == 0.0 || q[5] == 0.0) return RT_CLASS_LOOSE;
        for (int k = 0; k < 3; ++k) { const double a = std::fabs(q[3 + k]); box.lo[k] = q[k] - a; box.hi[k] = q[k] + a; }
        return RT_CLASS_BOUNDED;
    }
    case RT_PLANE:
    case RT_TRIANGLE: {
        // Frame of Plane::intersect (mesh.cpp:799-847): e1 = P1 - P0, e2 = P3 - P0 (a plane's corner 2 is
        // never read), n = e1 x e2 normalised, W = inverse of [e1 e2 n] by the explicit adjugate.
        if (p.kind == RT_PLANE && p.reserved != 0) return RT_CLASS_NEVER;
        const double* P0 = q;
        const double* P1 = q + 3;
        const double* P3 = (p.kind == RT_PLANE) ? q + 9 : q + 6;
        if (!all_finite(q, p.kind == RT_PLANE ? 12 : 9)) return RT_CLASS_NEVER;   // every distance would be NaN
        const double e1[3] = { P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2] };
        const double e2[3] = { P3[0] - P0[0], P3[1] - P0[1], P3[2] - P0[2] };
        const double nc[3] = { e1[1]*e2[2] - e1[2]*e2[1], e1[2]*e2[0] - e1[0]*e2[2], e1[0]*e2[1] - e1[1]*e2[0] };
        const double n_len = std::sqrt(nc[0]*nc[0] + nc[1]*nc[1] + nc[2]*nc[2]);
        if (n_len < 1e-15) return RT_CLASS_NEVER;
        const double n[3] = { nc[0]/n_len, nc[1]/n_len, nc[2]/n_len };
        const double M[3][3] = { { e1[0], e2[0], n[0] }, { e1[1], e2[1], n[1] }, { e1[2], e2[2], n[2] } };
        const double det = M[0][0]*(M[1][1]*M[2][2] - M[1][2]*M[2][1])
                         - M[0][1]*(M[1][0]*M[2][2] - M[1][2]*M[2][0])
                         + M[0][2]*(M[1][0]*M[2][1] - M[1][1]*M[2][0]);
        if (std::fabs(det) < 1e-15) return RT_CLASS_NEVER;
        const double id_ = 1.0 / det;
        double* W = rec.d;
        W[0] =  (M[1][1]*M[2][2] - M[1][2]*M[2][1]) * id_;
        W[1] = -(M[0][1]*M[2][2] - M[0][2]*M[2][1]) * id_;
        W[2] =  (M[0][1]*M[1][2] - M[0][2]*M[1][1]) * id_;
        W[3] = -(M[1][0]*M[2][2] - M[1][2]*M[2][0]) * id_;
        W[4] =  (M[0][0]*M[2][2] - M[0][2]*M[2][0]) * id_;
        W[5] = -(M[0][0]*M[1][2] - M[0][2]*M[1][0]) * id_;
        W[6] =  (M[1][0]*M[2][1] - M[1][1]*M[2][0]) * id_;
        W[7] = -(M[0][0]*M[2][1] - M[0][1]*M[2][0]) * id_;
        W[8] =  (M[0][0]*M[1][1] - M[0][1]*M[1][0]) * id_;
        for (int k = 0; k < 3; ++k) { rec.d[9 + k] = P0[k]; rec.d[12 + k] = e1[k]; rec.d[15 + k] = e2[k]; }
        // Reachable hit points: P0 + u e1 + v e2 with u, v in [0, 1] (parallelogram) - NOT the four given
        // corners, whose third entry the intersector ignores.
        box_add(box, P0);
        const double a[3] = { P0[0] + e1[0], P0[1] + e1[1], P0[2] + e1[2] };
        const double b[3] = { P0[0] + e2[0], P0[1] + e2[1], P0[2] + e2[2] };
        box_add(box, a); box_add(box, b);
        if (p.kind == RT_PLANE) {
            const double c[3] = { a[0] + e2[0], a[1] + e2[1], a[2] + e2[2] };
            box_add(box, c);
        }
        return RT_CLASS_BOUNDED;
    }
    default:
        return RT_CLASS_NEVER;
    }
}

// clamp_u8_from_unit and the solid branch of Mesh::evaluate_albedo (mesh.cpp:15-45).
static uint8_t unit_to_u8(double x) {
    if (x <= 0.0) return 0;
    if (x >= 1.0) return 255;
    return static_cast<uint8_t>(std::round(x * 255.0));
}

void rt_make_material(const rt_prim_desc& p, rt_mat_rec& m) {
    std::memset(&m, 0, sizeof(m));
    m.metallic = p.metallic; m.roughness = p.roughness; m.ior = p.ior;
    for (int k = 0; k < 3; ++k) m.albedo[k] = unit_to_u8(p.albedo[k]);
    m.tex = -1;
}

// The reference's own compute_aabb, type by type (mesh.cpp:288-309, 517-538, 683-701, 922-930).  A plane
// uses all four given corners here (unlike the reachable-point box the BVH is built from).
void rt_reference_aabb(const rt_prim_desc& p, double out6[6]) {
    rt_box64 b; box_reset(b);
    const double* q = p.q;
    if (p.kind == RT_PLANE || p.kind == RT_TRIANGLE) {
        // a plane given fewer than four corners spans only those (mesh.cpp:922-930); none at all gives a zero box
        int nc = 3;
        if (p.kind == RT_PLANE) nc = (p.reserved == 0) ? 4 : ((p.reserved > 0 && p.reserved < 4) ? p.reserved : 0);
        for (int c = 0; c < nc; ++c) box_add(b, q + 3*c);
        if (nc == 0) { for (int k = 0; k < 6; ++k) out6[k] = 0.0; return; }
    } else if (p.kind == RT_SPHERE) {
        for (int dx = -1; dx <= 1; dx += 2) for (int dy = -1; dy <= 1; dy += 2) for (int dz = -1; dz <= 1; dz += 2) {
            const double pw[3] = { q[0] + dx * q[3], q[1] + dy * q[4], q[2] + dz * q[5] };
            box_add(b, pw);
        }
    } else {
        double R[9]; rt_rotation_rpy(q[3], q[4], q[5], R);
        for (int dx = -1; dx <= 1; dx += 2) for (int dy = -1; dy <= 1; dy += 2) for (int dz = -1; dz <= 1; dz += 2) {
            const double ps[3] = { dx * q[6], dy * q[7], dz * q[8] };
            const double pr[3] = { R[0]*ps[0] + R[1]*ps[1] + R[2]*ps[2], R[3]*ps[0] + R[4]*ps[1] + R[5]*ps[2],
                                   R[6]*ps[0] + R[7]*ps[1] + R[8]*ps[2] };
            const double pw[3] = { pr[0] + q[0], pr[1] + q[1], pr[2] + q[2] };
            box_add(b, pw);
        }
    }
    for (int k = 0; k < 3; ++k) { out6[k] = b.lo[k]; out6[3 + k] = b.hi[k]; }
}

// ---- camera (persp_camera.cpp:17, 35-66, 94-118) --------------------------------------------

namespace {
inline void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1]*b[2] - a[2]*b[1]; o[1] = a[2]*b[0] - a[0]*b[2]; o[2] = a[0]*b[1] - a[1]*b[0];
}
inline double length3(const double a[3]) { return std::sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }
inline void unit3(const double a[3], double o[3]) {          // exact zero length -> zero vector
    const double len = length3(a);
    if (len == 0) { o[0] = o[1] = o[2] = 0.0; return; }
    o[0] = a[0]/len; o[1] = a[1]/len; o[2] = a[2]/len;
}
}

void rt_camera_basis(const double pos[3], const double dir[3], const double* up, rt_camera* cam) {
    for (int k = 0; k < 3; ++k) cam->pos[k] = pos[k];
    double f[3], r[3], u[3], t[3];
    unit3(dir, f);
    bool have_right = false;
    if (up) {
        double ug[3]; unit3(up, ug);
        const double fu = f[0]*ug[0] + f[1]*ug[1] + f[2]*ug[2];
        if (!(length3(ug) < 1e-6 || std::fabs(fu) > 0.999)) { cross3(f, ug, t); unit3(t, r); have_right = true; }
    }
    if (!have_right) {
        const double z_up[3] = { 0, 0, 1 }, y_up[3] = { 0, 1, 0 };
        cross3(f, z_up, t); unit3(t, r);
        if (length3(r) < 1e-6) { cross3(f, y_up, t); unit3(t, r); }
    }
    cross3(r, f, u);
    for (int i = 0; i < 3; ++i) { cam->rot[i*3+0] = r[i]; cam->rot[i*3+1] = u[i]; cam->rot[i*3+2] = f[i]; }
}

void rt_pixel_to_ray(const rt_camera* cam, double px, double py, double o[3], double d[3]) {
    const double nx = px / static_cast<double>(cam->width);
    const double ny = py / static_cast<double>(cam->height);
    const double aspect_img = static_cast<double>(cam->width) / static_cast<double>(cam->height);
    const double aspect_sensor = (cam->sensor_h > 0.0) ? (cam->sensor_w / cam->sensor_h) : aspect_img;
    double ew, eh;
    if (aspect_img >= aspect_sensor) { ew = cam->sensor_w; eh = ew / aspect_img; }
    else                             { eh = cam->sensor_h; ew = eh * aspect_img; }
    const double xs = ew * (nx - 0.5), ys = eh * (0.5 - ny), zs = cam->focal;
    const double* R = cam->rot;
    const double w[3] = { R[0]*xs + R[1]*ys + R[2]*zs, R[3]*xs + R[4]*ys + R[5]*zs, R[6]*xs + R[7]*ys + R[8]*zs };
    o[0] = cam->pos[0]; o[1] = cam->pos[1]; o[2] = cam->pos[2];
    unit3(w, d);
}

// gamma_encode for every 8-bit input (raytracer.cpp:428-432); evaluated with the host libm so the table
// matches the reference exactly, the device only indexes it.
void rt_gamma_table(uint8_t lut[256]) {
    for (int c = 0; c < 256; ++c) {
        const double v = static_cast<double>(c) / 255.0;
        lut[c] = static_cast<uint8_t>(static_cast<int>(std::round(std::pow(v, 1.0 / 2.2) * 255.0)));
    }
}
