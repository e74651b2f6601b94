#define _GNU_SOURCE
/* TEST INFRASTRUCTURE ONLY - see rt_oracle.h.  Plain-C FP64 restatement of the reference hot path.
 * Every function cites the reference file:line it follows.  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off, no fast-math). */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RTO_INF (__builtin_inf())

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                               */

/* M v with the summation order of mul_mat_vec (s2899221/Code/mesh.cpp:147-151). */
static void mat_vec(const double M[3][3], const double v[3], double out[3]) {
    out[0] = M[0][0]*v[0] + M[0][1]*v[1] + M[0][2]*v[2];
    out[1] = M[1][0]*v[0] + M[1][1]*v[1] + M[1][2]*v[2];
    out[2] = M[2][0]*v[0] + M[2][1]*v[1] + M[2][2]*v[2];
}

/* R = Rz(yaw) * (Ry(pitch) * Rx(roll)) formed by two full 3x3 products, zeros and ones included
 * (build_rotation_matrix_rpy, s2899221/Code/mesh.cpp:122-145). */
static void rotation_rpy(double roll, double pitch, double yaw, double R[3][3]) {
    /* sincos(), not sin() and cos(): that is what g++ -O2 makes of the reference's side-by-side calls, and glibc's
     * sincos() is not bit-identical to its sin() for every angle. */
    double cr, sr, cp, sp, cy, sy;
    sincos(roll, &sr, &cr);
    sincos(pitch, &sp, &cp);
    sincos(yaw, &sy, &cy);
    const double Z[3][3] = { { cy, -sy, 0 }, { sy, cy, 0 }, { 0, 0, 1 } };
    const double Y[3][3] = { { cp, 0, sp }, { 0, 1, 0 }, { -sp, 0, cp } };
    const double X[3][3] = { { 1, 0, 0 }, { 0, cr, -sr }, { 0, sr, cr } };
    double YX[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            YX[i][j] = Y[i][0]*X[0][j] + Y[i][1]*X[1][j] + Y[i][2]*X[2][j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i][j] = Z[i][0]*YX[0][j] + Z[i][1]*YX[1][j] + Z[i][2]*YX[2][j];
}

static double inv_or_zero(double s) { return (s != 0.0) ? (1.0 / s) : 0.0; }

static void hit_clear(rto_hit* h) {
    memset(h, 0, sizeof(*h));
    h->id = -1; h->prim = -1; h->inst = -1;
    h->dist = RTO_INF;
}

/* Shared tail of all four intersectors: mirror direction, re-projected distance
 * (s2899221/Code/mesh.cpp:236-260, 472-495, 645-668, 892-916). */
static void finish_hit(const double o[3], const double d[3], const double pw[3], const double n[3], rto_hit* out) {
    const double d_dot_n = d[0]*n[0] + d[1]*n[1] + d[2]*n[2];
    const double rt[3] = { d[0] - 2.0 * d_dot_n * n[0],
                           d[1] - 2.0 * d_dot_n * n[1],
                           d[2] - 2.0 * d_dot_n * n[2] };
    double r_len = sqrt(rt[0]*rt[0] + rt[1]*rt[1] + rt[2]*rt[2]);
    if (r_len < 1e-15) r_len = 1.0;
    out->r[0] = rt[0]/r_len; out->r[1] = rt[1]/r_len; out->r[2] = rt[2]/r_len;
    out->p[0] = pw[0]; out->p[1] = pw[1]; out->p[2] = pw[2];
    out->n[0] = n[0];  out->n[1] = n[1];  out->n[2] = n[2];
    const double dx = pw[0] - o[0];
    const double dy = pw[1] - o[1];
    const double dz = pw[2] - o[2];
    const double d_len = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]);
    out->dist = (d_len > 1e-15) ? (dx*d[0] + dy*d[1] + dz*d[2]) / d_len
                                : sqrt(dx*dx + dy*dy + dz*dz);
    out->hit = 1;
}

/* World ray -> unit-primitive frame for cube and cylinder
 * (s2899221/Code/mesh.cpp:161-180 and 373-392). */
static void to_local_trs(const double* q, const double o[3], const double d[3],
                         double R[3][3], double invS[3], double ro[3], double rd[3]) {
    rotation_rpy(q[3], q[4], q[5], R);
    double RT[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RT[i][j] = R[j][i];
    const double rel[3] = { o[0] - q[0], o[1] - q[1], o[2] - q[2] };
    double ro_r[3], rd_r[3];
    mat_vec(RT, rel, ro_r);
    mat_vec(RT, d, rd_r);
    invS[0] = inv_or_zero(q[6]); invS[1] = inv_or_zero(q[7]); invS[2] = inv_or_zero(q[8]);
    for (int k = 0; k < 3; ++k) { ro[k] = ro_r[k] * invS[k]; rd[k] = rd_r[k] * invS[k]; }
}

/* Local normal and point back to the world for cube and cylinder
 * (s2899221/Code/mesh.cpp:223-233 and 461-469). */
static void to_world_trs(const double* q, const double R[3][3], const double invS[3],
                         const double n_local[3], const double p_local[3], double n_unit[3], double pw[3]) {
    const double ns[3] = { n_local[0] * invS[0], n_local[1] * invS[1], n_local[2] * invS[2] };
    double nw[3]; mat_vec(R, ns, nw);
    const double nlen = sqrt(nw[0]*nw[0] + nw[1]*nw[1] + nw[2]*nw[2]);
    n_unit[0] = nw[0]/nlen; n_unit[1] = nw[1]/nlen; n_unit[2] = nw[2]/nlen;
    const double ps[3] = { p_local[0] * q[6], p_local[1] * q[7], p_local[2] * q[8] };
    double pr[3]; mat_vec(R, ps, pr);
    pw[0] = pr[0] + q[0]; pw[1] = pr[1] + q[1]; pw[2] = pr[2] + q[2];
}

/* ------------------------------------------------------------------------------------------ */
/* intersectors                                                                                */

/* Cube::intersect, s2899221/Code/mesh.cpp:159-280. */
static int isect_cube(const double* q, const double o[3], const double d[3], rto_hit* out) {
    double R[3][3], invS[3], ro[3], rd[3];
    to_local_trs(q, o, d, R, invS, ro, rd);

    double tmin = -RTO_INF, tmax = RTO_INF;
    double n_enter[3] = { 0, 0, 0 }, n_exit[3] = { 0, 0, 0 };
    for (int axis = 0; axis < 3; ++axis) {
        const double oo = ro[axis], dd = rd[axis];
        if (fabs(dd) < 1e-12) {
            if (oo < -1.0 || oo > 1.0) return 0;
            continue;
        }
        double t1 = (-1.0 - oo) / dd;
        double t2 = (1.0 - oo) / dd;
        double s1 = -1.0, s2 = 1.0;                 /* face normal component on this axis */
        if (t1 > t2) { double t = t1; t1 = t2; t2 = t; s1 = 1.0; s2 = -1.0; }
        if (t1 > tmin) { tmin = t1; n_enter[0] = n_enter[1] = n_enter[2] = 0.0; n_enter[axis] = s1; }
        if (t2 < tmax) { tmax = t2; n_exit[0]  = n_exit[1]  = n_exit[2]  = 0.0; n_exit[axis]  = s2; }
        if (tmin > tmax) return 0;
    }
    if (tmax < 0.0) return 0;
    const int entering = (tmin >= 0.0);
    const double t_hit = entering ? tmin : tmax;
    const double* n_local = entering ? n_enter : n_exit;
    const double p_local[3] = { ro[0] + t_hit * rd[0], ro[1] + t_hit * rd[1], ro[2] + t_hit * rd[2] };

    double n_unit[3], pw[3];
    to_world_trs(q, R, invS, n_local, p_local, n_unit, pw);
    finish_hit(o, d, pw, n_unit, out);

    if (fabs(n_local[0]) > 0.5)      { out->u = 0.5 * (p_local[2] + 1.0); out->v = 0.5 * (p_local[1] + 1.0); }
    else if (fabs(n_local[1]) > 0.5) { out->u = 0.5 * (p_local[0] + 1.0); out->v = 0.5 * (p_local[2] + 1.0); }
    else                             { out->u = 0.5 * (p_local[0] + 1.0); out->v = 0.5 * (p_local[1] + 1.0); }
    return 1;
}

/* Cylinder::intersect, s2899221/Code/mesh.cpp:371-514. */
typedef struct { double t; double p[3]; double n[3]; int found; } cyl_best;

static void cyl_try_side(cyl_best* b, const double ro[3], const double rd[3], double t) {
    const double eps = 1e-12;
    if (t < 0.0) return;
    const double z = ro[2] + t * rd[2];
    if (z < -1.0 - eps || z > 1.0 + eps) return;
    if (t < b->t) {
        b->t = t;
        b->p[0] = ro[0] + t*rd[0];
        b->p[1] = ro[1] + t*rd[1];
        b->p[2] = z;
        const double len_xy = sqrt(b->p[0]*b->p[0] + b->p[1]*b->p[1]);
        if (len_xy > eps) { b->n[0] = b->p[0]/len_xy; b->n[1] = b->p[1]/len_xy; b->n[2] = 0.0; }
        else              { b->n[0] = 0.0; b->n[1] = 0.0; b->n[2] = 1.0; }
        b->found = 1;
    }
}

static void cyl_try_cap(cyl_best* b, const double ro[3], const double rd[3], double z_cap, double nz) {
    if (fabs(rd[2]) < 1e-12) return;
    const double t = (z_cap - ro[2]) / rd[2];
    if (t < 0.0) return;
    const double x = ro[0] + t * rd[0];
    const double y = ro[1] + t * rd[1];
    if (x*x + y*y <= 1.0 + 1e-9) {
        if (t < b->t) {
            b->t = t;
            b->p[0] = x; b->p[1] = y; b->p[2] = z_cap;
            b->n[0] = 0.0; b->n[1] = 0.0; b->n[2] = nz;
            b->found = 1;
        }
    }
}

static int isect_cylinder(const double* q, const double o[3], const double d[3], rto_hit* out) {
    double R[3][3], invS[3], ro[3], rd[3];
    to_local_trs(q, o, d, R, invS, ro, rd);

    cyl_best b; b.t = RTO_INF; b.found = 0;
    b.p[0] = b.p[1] = b.p[2] = 0.0; b.n[0] = b.n[1] = b.n[2] = 0.0;

    const double A = rd[0]*rd[0] + rd[1]*rd[1];
    const double B = 2.0 * (ro[0]*rd[0] + ro[1]*rd[1]);
    const double C = ro[0]*ro[0] + ro[1]*ro[1] - 1.0;
    if (A > 1e-12) {
        const double disc = B*B - 4.0*A*C;
        if (disc >= 0.0) {
            const double sq = sqrt(disc);
            const double t1 = (-B - sq) / (2.0*A);
            const double t2 = (-B + sq) / (2.0*A);
            cyl_try_side(&b, ro, rd, t1);
            cyl_try_side(&b, ro, rd, t2);
        }
    }
    cyl_try_cap(&b, ro, rd, 1.0, 1.0);
    cyl_try_cap(&b, ro, rd, -1.0, -1.0);
    if (!b.found) return 0;

    double n_unit[3], pw[3];
    to_world_trs(q, R, invS, b.n, b.p, n_unit, pw);
    finish_hit(o, d, pw, n_unit, out);

    const double pi = 3.14159265358979323846;
    if (fabs(b.n[2]) < 0.5) {
        const double theta = atan2(b.p[1], b.p[0]);
        out->u = (theta + pi) / (2.0 * pi);
        out->v = 0.5 * (b.p[2] + 1.0);
    } else {
        out->u = 0.5 * (b.p[0] + 1.0);
        out->v = 0.5 * (b.p[1] + 1.0);
    }
    return 1;
}

/* Sphere::intersect (axis-aligned ellipsoid), s2899221/Code/mesh.cpp:597-681. */
static int isect_sphere(const double* q, const double o[3], const double d[3], rto_hit* out) {
    const double invS[3] = { inv_or_zero(q[3]), inv_or_zero(q[4]), inv_or_zero(q[5]) };
    const double ro[3] = { (o[0] - q[0]) * invS[0], (o[1] - q[1]) * invS[1], (o[2] - q[2]) * invS[2] };
    const double rd[3] = { d[0] * invS[0], d[1] * invS[1], d[2] * invS[2] };

    const double A = rd[0]*rd[0] + rd[1]*rd[1] + rd[2]*rd[2];
    const double B = 2.0 * (ro[0]*rd[0] + ro[1]*rd[1] + ro[2]*rd[2]);
    const double C = ro[0]*ro[0] + ro[1]*ro[1] + ro[2]*ro[2] - 1.0;
    if (A < 1e-12) return 0;
    const double disc = B*B - 4.0*A*C;
    if (disc < 0.0) return 0;
    const double sq = sqrt(disc);
    double t0 = (-B - sq) / (2.0*A);
    double t1 = (-B + sq) / (2.0*A);
    if (t0 > t1) { double t = t0; t0 = t1; t1 = t; }
    const double tl = (t0 >= 0.0) ? t0 : t1;
    if (tl < 0.0) return 0;

    const double pl[3] = { ro[0] + tl*rd[0], ro[1] + tl*rd[1], ro[2] + tl*rd[2] };
    const double pw[3] = { q[0] + pl[0] * q[3], q[1] + pl[1] * q[4], q[2] + pl[2] * q[5] };
    const double nl[3] = { pl[0] / 1.0, pl[1] / 1.0, pl[2] / 1.0 };
    const double nw[3] = { nl[0] * invS[0], nl[1] * invS[1], nl[2] * invS[2] };
    const double nlen = sqrt(nw[0]*nw[0] + nw[1]*nw[1] + nw[2]*nw[2]);
    const double n_unit[3] = { nw[0]/nlen, nw[1]/nlen, nw[2]/nlen };
    finish_hit(o, d, pw, n_unit, out);

    double px = pl[0], py = pl[1], pz = pl[2];
    const double plen = sqrt(px*px + py*py + pz*pz);
    if (plen > 1e-15) { px /= plen; py /= plen; pz /= plen; }
    const double pi = 3.14159265358979323846;
    const double pyc = (py < -1.0) ? -1.0 : ((1.0 < py) ? 1.0 : py);
    out->u = 0.5 + atan2(pz, px) / (2.0 * pi);
    out->v = 0.5 - asin(pyc) / pi;
    return 1;
}

/* Plane::intersect (finite parallelogram P0 + u e1 + v e2), s2899221/Code/mesh.cpp:795-920.
 * The same frame solve serves the triangle extension; only the inside test differs:
 * triangle != 0 keeps u >= -eps, v >= -eps, u + v <= 1 + eps. */
static int isect_planar(const double P0[3], const double P1[3], const double P3[3], int triangle,
                        const double o[3], const double d[3], rto_hit* out) {
    const double e1[3] = { P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2] };
    const double e2[3] = { P3[0] - P0[0], P3[1] - P0[1], P3[2] - P0[2] };
    const double nc[3] = { e1[1]*e2[2] - e1[2]*e2[1],
                           e1[2]*e2[0] - e1[0]*e2[2],
                           e1[0]*e2[1] - e1[1]*e2[0] };
    const double n_len = sqrt(nc[0]*nc[0] + nc[1]*nc[1] + nc[2]*nc[2]);
    if (n_len < 1e-15) return 0;
    const double n[3] = { nc[0]/n_len, nc[1]/n_len, nc[2]/n_len };

    const double M[3][3] = { { e1[0], e2[0], n[0] }, { e1[1], e2[1], n[1] }, { e1[2], e2[2], n[2] } };
    const double det = M[0][0]*(M[1][1]*M[2][2] - M[1][2]*M[2][1])
                     - M[0][1]*(M[1][0]*M[2][2] - M[1][2]*M[2][0])
                     + M[0][2]*(M[1][0]*M[2][1] - M[1][1]*M[2][0]);
    if (fabs(det) < 1e-15) return 0;
    const double id = 1.0 / det;
    double W[3][3];                                   /* adjugate / det */
    W[0][0] =  (M[1][1]*M[2][2] - M[1][2]*M[2][1]) * id;
    W[0][1] = -(M[0][1]*M[2][2] - M[0][2]*M[2][1]) * id;
    W[0][2] =  (M[0][1]*M[1][2] - M[0][2]*M[1][1]) * id;
    W[1][0] = -(M[1][0]*M[2][2] - M[1][2]*M[2][0]) * id;
    W[1][1] =  (M[0][0]*M[2][2] - M[0][2]*M[2][0]) * id;
    W[1][2] = -(M[0][0]*M[1][2] - M[0][2]*M[1][0]) * id;
    W[2][0] =  (M[1][0]*M[2][1] - M[1][1]*M[2][0]) * id;
    W[2][1] = -(M[0][0]*M[2][1] - M[0][1]*M[2][0]) * id;
    W[2][2] =  (M[0][0]*M[1][1] - M[0][1]*M[1][0]) * id;

    const double rel[3] = { o[0] - P0[0], o[1] - P0[1], o[2] - P0[2] };
    double ro[3], rd[3];
    mat_vec(W, rel, ro);
    mat_vec(W, d, rd);

    const double eps = 1e-12;
    if (fabs(rd[2]) < eps) return 0;
    const double t = -ro[2] / rd[2];
    if (t < 0.0) return 0;
    const double u = ro[0] + t * rd[0];
    const double v = ro[1] + t * rd[1];
    if (triangle) {
        if (u < -eps || v < -eps || u + v > 1.0 + eps) return 0;
    } else {
        if (u < -eps || u > 1.0 + eps || v < -eps || v > 1.0 + eps) return 0;
    }
    const double pl[3] = { u, v, 0.0 };
    double rel_w[3]; mat_vec(M, pl, rel_w);
    const double pw[3] = { P0[0] + rel_w[0], P0[1] + rel_w[1], P0[2] + rel_w[2] };
    finish_hit(o, d, pw, n, out);
    out->u = u; out->v = v;
    return 1;
}

int rto_intersect(const rto_prim* prim, const double o[3], const double d[3], rto_hit* out) {
    hit_clear(out);
    int ok = 0;
    switch (prim->kind) {
        case RTO_CUBE:     ok = isect_cube(prim->q, o, d, out); break;
        case RTO_CYLINDER: ok = isect_cylinder(prim->q, o, d, out); break;
        case RTO_SPHERE:   ok = isect_sphere(prim->q, o, d, out); break;
        /* reserved != 0 marks a plane that was given fewer than four corners: never hit (mesh.cpp:796) */
        case RTO_PLANE:    ok = prim->reserved ? 0 : isect_planar(prim->q, prim->q + 3, prim->q + 9, 0, o, d, out); break;
        case RTO_TRIANGLE: ok = isect_planar(prim->q, prim->q + 3, prim->q + 6, 1, o, d, out); break;
        default: break;
    }
    if (!ok) hit_clear(out);
    return ok;
}

/* compute_aabb of each type: s2899221/Code/mesh.cpp:288-309 (cube), 517-538 (cylinder),
 * 683-701 (sphere), 922-930 (plane); corner visiting order dx, dy, dz in {-1, +1}. */
void rto_aabb(const rto_prim* prim, double mn[3], double mx[3]) {
    mn[0] = mn[1] = mn[2] = RTO_INF;
    mx[0] = mx[1] = mx[2] = -RTO_INF;
    const double* q = prim->q;
    if (prim->kind == RTO_PLANE || prim->kind == RTO_TRIANGLE) {
        /* reserved on a plane: 0 = four corners, 1..3 = only that many were given, anything else = none */
        int nc = 3;
        if (prim->kind == RTO_PLANE) nc = (prim->reserved == 0) ? 4 : ((prim->reserved > 0 && prim->reserved < 4) ? prim->reserved : 0);
        if (nc == 0) { mn[0] = mn[1] = mn[2] = 0.0; mx[0] = mx[1] = mx[2] = 0.0; return; }
        for (int c = 0; c < nc; ++c)
            for (int k = 0; k < 3; ++k) {
                const double x = q[c*3 + k];
                if (x < mn[k]) mn[k] = x;
                if (x > mx[k]) mx[k] = x;
            }
        return;
    }
    double R[3][3];
    if (prim->kind != RTO_SPHERE) rotation_rpy(q[3], q[4], q[5], R);
    for (int dx = -1; dx <= 1; dx += 2)
        for (int dy = -1; dy <= 1; dy += 2)
            for (int dz = -1; dz <= 1; dz += 2) {
                double pw[3];
                if (prim->kind == RTO_SPHERE) {
                    pw[0] = q[0] + (double)dx * q[3];
                    pw[1] = q[1] + (double)dy * q[4];
                    pw[2] = q[2] + (double)dz * q[5];
                } else {
                    const double ps[3] = { (double)dx * q[6], (double)dy * q[7], (double)dz * q[8] };
                    double pr[3]; mat_vec(R, ps, pr);
                    pw[0] = pr[0] + q[0]; pw[1] = pr[1] + q[1]; pw[2] = pr[2] + q[2];
                }
                for (int k = 0; k < 3; ++k) {
                    if (pw[k] < mn[k]) mn[k] = pw[k];
                    if (pw[k] > mx[k]) mx[k] = pw[k];
                }
            }
}

/* RayTracer::one_pass_intersection, s2899221/Code/raytracer.cpp:143-157: strict '<', so the
 * lowest index wins exact ties. */
/* Conservative "can this ray touch the box": FP64 slabs against bounds padded far beyond any rounding error. */
static int may_touch(const double o[3], const double d[3], const double b[6]) {
    double m = 1.0;
    for (int k = 0; k < 6; ++k) if (fabs(b[k]) > m) m = fabs(b[k]);
    for (int k = 0; k < 3; ++k) if (fabs(o[k]) > m) m = fabs(o[k]);
    const double pad = 1e-9 * m;
    double tn = 0.0, tf = RTO_INF;
    for (int k = 0; k < 3; ++k) {
        const double lo = b[k] - pad, hi = b[3 + k] + pad;
        if (fabs(d[k]) < 1e-300) { if (o[k] < lo || o[k] > hi) return 0; continue; }
        double t1 = (lo - o[k]) / d[k], t2 = (hi - o[k]) / d[k];
        if (t1 > t2) { const double t = t1; t1 = t2; t2 = t; }
        if (t1 > tn) tn = t1;
        if (t2 < tf) tf = t2;
    }
    return tn <= tf * (1.0 + 1e-9) + pad;
}

/* Closest hit over the instances of a two-level scene (see rto_instance). */
static int inst_closest_hit(const rto_scene* sc, const double o[3], const double d[3], rto_hit* out) {
    rto_hit best; hit_clear(&best);
    int any = 0;
    int64_t base = 0;
    for (int64_t k = 0; k < sc->n_inst; ++k) {
        const rto_instance* in = &sc->inst[k];
        const double* R = in->rot;
        const double e[3] = { o[0] - in->trans[0], o[1] - in->trans[1], o[2] - in->trans[2] };
        double oo[3], dd[3];
        for (int i = 0; i < 3; ++i) {
            oo[i] = R[i]*e[0] + R[3 + i]*e[1] + R[6 + i]*e[2];
            dd[i] = R[i]*d[0] + R[3 + i]*d[1] + R[6 + i]*d[2];
        }
        if (!sc->inst_bounds || may_touch(oo, dd, sc->inst_bounds + 6*k)) {
            for (uint32_t j = 0; j < in->n_prims; ++j) {
                rto_hit h;
                if (rto_intersect(&sc->prims[in->first_prim + j], oo, dd, &h) && h.dist < best.dist) {
                    best = h; best.id = (int32_t)(base + j); best.prim = (int32_t)(in->first_prim + j); best.inst = (int32_t)k;
                    any = 1;
                }
            }
        }
        base += in->n_prims;
    }
    if (!any) { hit_clear(out); return 0; }
    const rto_instance* in = &sc->inst[best.inst];
    const double* R = in->rot;
    *out = best;
    for (int i = 0; i < 3; ++i) {
        out->p[i] = R[3*i]*best.p[0] + R[3*i + 1]*best.p[1] + R[3*i + 2]*best.p[2] + in->trans[i];
        out->n[i] = R[3*i]*best.n[0] + R[3*i + 1]*best.n[1] + R[3*i + 2]*best.n[2];
        out->r[i] = R[3*i]*best.r[0] + R[3*i + 1]*best.r[1] + R[3*i + 2]*best.r[2];
    }
    return 1;
}

int rto_closest_hit(const rto_scene* sc, const double o[3], const double d[3], rto_hit* out) {
    if (sc->instanced) return inst_closest_hit(sc, o, d, out);
    rto_hit best; hit_clear(&best);
    int any = 0;
    for (int64_t i = 0; i < sc->n_prims; ++i) {
        rto_hit h;
        if (sc->prim_bounds && !may_touch(o, d, sc->prim_bounds + 6*i)) continue;
        if (rto_intersect(&sc->prims[i], o, d, &h)) {
            if (h.dist < best.dist) { best = h; best.id = (int32_t)i; best.prim = (int32_t)i; any = 1; }
        }
    }
    if (any) *out = best; else hit_clear(out);
    return any;
}

/* ------------------------------------------------------------------------------------------ */
/* shading                                                                                     */

static void normalize3(double x, double y, double z, double out[3]) {
    const double len = sqrt(x*x + y*y + z*z);
    if (len <= 1e-15) { out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; return; }
    out[0] = x/len; out[1] = y/len; out[2] = z/len;
}

static double clampd(double v, double lo, double hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

/* clamp_u8_from_unit + solid branch of Mesh::evaluate_albedo, s2899221/Code/mesh.cpp:15-45. */
static uint8_t unit_to_u8(double x) {
    if (x <= 0.0) return 0;
    if (x >= 1.0) return 255;
    return (uint8_t)round(x * 255.0);
}
static void albedo_u8(const rto_prim* m, uint8_t c[3]) {
    c[0] = unit_to_u8(m->albedo[0]); c[1] = unit_to_u8(m->albedo[1]); c[2] = unit_to_u8(m->albedo[2]);
}
/* Mesh::evaluate_albedo, s2899221/Code/mesh.cpp:21-45: texture (wrap, nearest) when there is one, else solid. */
static void albedo_at(const rto_scene* sc, const rto_hit* hit, uint8_t c[3]) {
    const rto_prim* m = &sc->prims[hit->prim];
    if (sc->prim_tex && sc->textures && sc->prim_tex[hit->prim] >= 0 && sc->prim_tex[hit->prim] < sc->n_textures) {
        const rto_texture* t = &sc->textures[sc->prim_tex[hit->prim]];
        const int w = t->width, h = t->height;
        if (w > 0 && h > 0) {
            double u = hit->u, v = hit->v;
            u = u - floor(u);
            v = v - floor(v);
            int x = (int)floor(u * (double)(w - 1));
            int y = (int)floor(v * (double)(h - 1));
            if (x < 0) x = 0;
            if (x >= w) x = w - 1;
            if (y < 0) y = 0;
            if (y >= h) y = h - 1;
            const uint8_t* px = t->rgb + ((size_t)y * (size_t)w + (size_t)x) * 3u;
            c[0] = px[0]; c[1] = px[1]; c[2] = px[2];
            return;
        }
    }
    albedo_u8(m, c);
}

static uint8_t round_u8(double cc) {
    int ci = (int)round(cc);
    if (ci < 0) ci = 0;
    if (ci > 255) ci = 255;
    return (uint8_t)ci;
}

/* RayTracer::shade, s2899221/Code/raytracer.cpp:159-279.  The shadow query is injected so that the
 * same arithmetic serves the brute-force path and the BVH counting path. */
typedef int (*occluded_fn)(const void* ctx, const double o[3], const double d[3], double light_dist);

static int brute_occluded(const void* ctx, const double o[3], const double d[3], double light_dist) {
    const rto_scene* sc = (const rto_scene*)ctx;
    rto_hit sh;
    if (rto_closest_hit(sc, o, d, &sh)) {
        const double t = sh.dist;
        if (t > 1e-4 && t < light_dist - 1e-4) return 1;
    }
    return 0;
}

/* Origin of the shadow ray from p (unit normal n) towards a light in unit direction ldir, nudged 1e-4 along the normal
 * on the light's side; returns the light's distance (s2899221/Code/raytracer.cpp:217-228). */
static double shadow_origin(const double p[3], const double n[3], const double ldir[3], double dist2, double so[3]) {
    const double light_dist = sqrt(dist2);
    const double eps = 1e-4;
    const double ndl = n[0]*ldir[0] + n[1]*ldir[1] + n[2]*ldir[2];
    const double s = (ndl >= 0.0) ? 1.0 : -1.0;
    so[0] = p[0] + s * n[0] * eps; so[1] = p[1] + s * n[1] * eps; so[2] = p[2] + s * n[2] * eps;
    return light_dist;
}

static void shade_with(const rto_scene* sc, const rto_hit* hit, const double ray_o[3], uint8_t rgb[3],
                       occluded_fn occ, const void* occ_ctx, int64_t* shadow_rays) {
    rgb[0] = rgb[1] = rgb[2] = 0;
    if (!hit->hit) return;
    if (sc->n_lights <= 0) return;

    double n[3]; normalize3(hit->n[0], hit->n[1], hit->n[2], n);
    const double* p = hit->p;
    double v[3]; normalize3(ray_o[0] - p[0], ray_o[1] - p[1], ray_o[2] - p[2], v);

    const rto_prim* m = (hit->prim >= 0) ? &sc->prims[hit->prim] : 0;
    const double metallic_factor = m ? clampd(m->metallic, 0.0, 1.0) : 0.0;

    double lighting = 0.0;
    for (int64_t li = 0; li < sc->n_lights; ++li) {
        const rto_light* L = &sc->lights[li];
        const double lx = L->pos[0] - p[0];
        const double ly = L->pos[1] - p[1];
        const double lz = L->pos[2] - p[2];
        const double len2 = lx*lx + ly*ly + lz*lz;
        const double dist2 = (1e-12 < len2) ? len2 : 1e-12;
        double ldir[3]; normalize3(lx, ly, lz, ldir);
        {
            double so[3];
            const double light_dist = shadow_origin(p, n, ldir, dist2, so);
            if (shadow_rays) ++*shadow_rays;
            if (occ(occ_ctx, so, ldir, light_dist)) continue;
        }
        const double dot_nl = n[0]*ldir[0] + n[1]*ldir[1] + n[2]*ldir[2];
        const double n_dot_l = (0.0 < dot_nl) ? dot_nl : 0.0;
        const double diffuse = 1.0 * (1.0 - metallic_factor) * n_dot_l;
        double h[3]; normalize3(ldir[0] + v[0], ldir[1] + v[1], ldir[2] + v[2], h);
        const double dot_nh = n[0]*h[0] + n[1]*h[1] + n[2]*h[2];
        const double n_dot_h = (0.0 < dot_nh) ? dot_nh : 0.0;
        const double roughness = m ? clampd(m->roughness, 0.001, 1.0) : 0.5;
        const double e_raw = (2.0 / (roughness * roughness)) - 2.0;
        const double phong_exp = (2.0 < e_raw) ? e_raw : 2.0;
        const double specular = (0.5 * 1.25) * pow(n_dot_h, phong_exp);
        const double attenuation = L->intensity / dist2;
        lighting += attenuation * (diffuse + specular);
    }
    double intensity = (1.0 - metallic_factor) * 0.05 + lighting;
    if (intensity < 0.0) intensity = 0.0;
    if (intensity > 1.0) intensity = 1.0;

    uint8_t base[3] = { 255, 255, 255 };
    if (m) albedo_at(sc, hit, base);
    for (int c = 0; c < 3; ++c) rgb[c] = round_u8(intensity * (double)base[c]);
}

void rto_shade(const rto_scene* sc, const rto_hit* hit, const double ray_o[3], uint8_t rgb[3]) {
    shade_with(sc, hit, ray_o, rgb, brute_occluded, sc, 0);
}

/* RayTracer::trace_ray_recursive, s2899221/Code/raytracer.cpp:281-371. */
static void trace_rec(const rto_scene* sc, const double o[3], const double d[3], int depth, int max_scatters,
                      double min_si, uint8_t rgb[3], int64_t* shadow_rays, int64_t* hits_by_depth) {
    rto_hit h;
    (void)rto_closest_hit(sc, o, d, &h);
    if (h.hit && hits_by_depth) ++hits_by_depth[depth < 7 ? depth : 7];
    uint8_t local[3];
    shade_with(sc, &h, o, local, brute_occluded, sc, shadow_rays);
    rgb[0] = local[0]; rgb[1] = local[1]; rgb[2] = local[2];
    if (depth >= max_scatters) return;
    if (!h.hit) return;

    const rto_prim* m = &sc->prims[h.prim];
    const double metallic = m->metallic;
    const double ior = m->ior;
    const double rd_len = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]);
    const double inv_len = (rd_len > 1e-15) ? (1.0 / rd_len) : 1.0;
    const double vx = -d[0] * inv_len, vy = -d[1] * inv_len, vz = -d[2] * inv_len;
    const double ct = h.n[0]*vx + h.n[1]*vy + h.n[2]*vz;
    const double cosTheta = (0.0 < ct) ? ct : 0.0;
    double F0;
    uint8_t base[3]; albedo_at(sc, &h, base);
    if (metallic <= 0.0) {
        const double r0 = (ior - 1.0) / (ior + 1.0);
        F0 = clampd(r0 * r0, 0.0, 1.0);
    } else {
        const double avg = ((double)base[0] + (double)base[1] + (double)base[2]) / (3.0 * 255.0);
        F0 = clampd(avg, 0.0, 1.0);
    }
    const double Fr = F0 + (1.0 - F0) * pow(1.0 - cosTheta, 5.0);
    if (Fr <= 0.0 || Fr < min_si) return;

    const double eps = 1e-6;
    const double so[3] = { h.p[0] + h.r[0] * eps, h.p[1] + h.r[1] * eps, h.p[2] + h.r[2] * eps };
    uint8_t sec[3];
    trace_rec(sc, so, h.r, depth + 1, max_scatters, min_si, sec, shadow_rays, hits_by_depth);

    uint8_t tinted[3] = { sec[0], sec[1], sec[2] };
    if (metallic > 0.0)
        for (int c = 0; c < 3; ++c) tinted[c] = round_u8(((double)base[c] / 255.0) * (double)sec[c]);
    for (int c = 0; c < 3; ++c)
        rgb[c] = round_u8((1.0 - Fr) * (double)local[c] + Fr * (double)tinted[c]);
}

void rto_trace(const rto_scene* sc, const double o[3], const double d[3], int depth, int max_scatters,
               double min_scatter_intensity, uint8_t rgb[3]) {
    trace_rec(sc, o, d, depth, max_scatters, min_scatter_intensity, rgb, 0, 0);
}

/* gamma_encode lambda, s2899221/Code/raytracer.cpp:428-432, for every 8-bit input. */
void rto_gamma_lut(uint8_t lut[256]) {
    for (int c = 0; c < 256; ++c) {
        const double v = clampd((double)c, 0.0, 255.0) / 255.0;
        const double s = pow(v, 1.0 / 2.2);
        lut[c] = (uint8_t)(int)round(s * 255.0);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* camera                                                                                      */

static void cross3(const double a[3], const double b[3], double out[3]) {
    out[0] = a[1]*b[2] - a[2]*b[1];
    out[1] = a[2]*b[0] - a[0]*b[2];
    out[2] = a[0]*b[1] - a[1]*b[0];
}
static double len3(const double a[3]) { return sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }
/* Vec3::normalized, s2899221/Code/persp_camera.cpp:17 (exact zero length -> zero vector). */
static void cam_normalized(const double a[3], double out[3]) {
    const double len = len3(a);
    if (len == 0) { out[0] = out[1] = out[2] = 0.0; return; }
    out[0] = a[0]/len; out[1] = a[1]/len; out[2] = a[2]/len;
}

/* Camera::updateRotationMatrix (both overloads), s2899221/Code/persp_camera.cpp:35-66. */
void rto_camera_look(const double pos[3], const double dir[3], const double* up_or_null, rto_camera* cam) {
    cam->pos[0] = pos[0]; cam->pos[1] = pos[1]; cam->pos[2] = pos[2];
    double fwd[3]; cam_normalized(dir, fwd);
    double right[3], up[3], tmp[3];
    int have = 0;
    if (up_or_null) {
        double ug[3]; cam_normalized(up_or_null, ug);
        const double f_dot_u = fwd[0]*ug[0] + fwd[1]*ug[1] + fwd[2]*ug[2];
        if (!(len3(ug) < 1e-6 || fabs(f_dot_u) > 0.999)) {
            cross3(fwd, ug, tmp); cam_normalized(tmp, right);
            have = 1;
        }
    }
    if (!have) {
        double wu[3] = { 0, 0, 1 };
        cross3(fwd, wu, tmp); cam_normalized(tmp, right);
        if (len3(right) < 1e-6) { wu[0] = 0; wu[1] = 1; wu[2] = 0; cross3(fwd, wu, tmp); cam_normalized(tmp, right); }
    }
    cross3(right, fwd, up);
    for (int i = 0; i < 3; ++i) {
        cam->rot[i*3 + 0] = right[i];
        cam->rot[i*3 + 1] = up[i];
        cam->rot[i*3 + 2] = fwd[i];
    }
}

/* Camera::pixelToRay, s2899221/Code/persp_camera.cpp:94-118. */
void rto_pixel_to_ray(const rto_camera* cam, double px, double py, double o[3], double d[3]) {
    const double nx = px / (double)cam->width;
    const double ny = py / (double)cam->height;
    const double aspect_img = (double)cam->width / (double)cam->height;
    const double aspect_sensor = (cam->sensor_h > 0.0) ? (cam->sensor_w / cam->sensor_h) : aspect_img;
    double eff_w, eff_h;
    if (aspect_img >= aspect_sensor) { eff_w = cam->sensor_w; eff_h = eff_w / aspect_img; }
    else                             { eff_h = cam->sensor_h; eff_w = eff_h * aspect_img; }
    const double xs = eff_w * (nx - 0.5);
    const double ys = eff_h * (0.5 - ny);
    const double zs = cam->focal;
    const double* R = cam->rot;
    double w[3];
    w[0] = R[0]*xs + R[1]*ys + R[2]*zs;
    w[1] = R[3]*xs + R[4]*ys + R[5]*zs;
    w[2] = R[6]*xs + R[7]*ys + R[8]*zs;
    o[0] = cam->pos[0]; o[1] = cam->pos[1]; o[2] = cam->pos[2];
    cam_normalized(w, d);
}

/* ------------------------------------------------------------------------------------------ */
/* batched drivers                                                                             */

static int pick_threads(int threads) {
#ifdef _OPENMP
    return threads > 0 ? threads : omp_get_max_threads();
#else
    (void)threads; return 1;
#endif
}

void rto_trace_rays(const rto_scene* sc, const double* rays6, int64_t n, rto_hit* out, int threads) {
    const int nt = pick_threads(threads); (void)nt;
    #pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (int64_t i = 0; i < n; ++i)
        rto_closest_hit(sc, rays6 + i*6, rays6 + i*6 + 3, &out[i]);
}

void rto_shade_rays(const rto_scene* sc, const double* rays6, int64_t n, uint8_t* rgb, int threads) {
    const int nt = pick_threads(threads); (void)nt;
    #pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (int64_t i = 0; i < n; ++i) {
        rto_hit h; rto_closest_hit(sc, rays6 + i*6, rays6 + i*6 + 3, &h);
        rto_shade(sc, &h, rays6 + i*6, rgb + i*3);
    }
}

void rto_trace_rays_rgb(const rto_scene* sc, const double* rays6, int64_t n, int max_scatters, double min_scatter_intensity,
                        uint8_t* rgb, int threads) {
    const int nt = pick_threads(threads); (void)nt;
    #pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
    for (int64_t i = 0; i < n; ++i)
        rto_trace(sc, rays6 + i*6, rays6 + i*6 + 3, 0, max_scatters, min_scatter_intensity, rgb + i*3);
}

void rto_hit_bounds(const rto_prim* prims, int64_t n, double* out6) {
    for (int64_t i = 0; i < n; ++i) {
        const rto_prim* p = &prims[i];
        double* b = out6 + 6*i;
        const double* q = p->q;
        int bounded = 1;
        if (p->kind == RTO_PLANE) {
            /* the points Plane::intersect can return: P0 + u e1 + v e2 over the unit square (mesh.cpp:799-804, 878-886) */
            const double c[4][3] = { { q[0], q[1], q[2] }, { q[3], q[4], q[5] }, { q[9], q[10], q[11] },
                                     { q[3] + q[9] - q[0], q[4] + q[10] - q[1], q[5] + q[11] - q[2] } };
            for (int k = 0; k < 3; ++k) {
                b[k] = c[0][k]; b[3 + k] = c[0][k];
                for (int j = 1; j < 4; ++j) { if (c[j][k] < b[k]) b[k] = c[j][k]; if (c[j][k] > b[3 + k]) b[3 + k] = c[j][k]; }
            }
        } else {
            rto_aabb(p, b, b + 3);
            if (p->kind != RTO_TRIANGLE) {
                const double* s = (p->kind == RTO_SPHERE) ? q + 3 : q + 6;
                for (int k = 0; k < 3; ++k) if (s[k] == 0.0 || !isfinite(s[k])) bounded = 0;
            }
        }
        for (int k = 0; k < 6; ++k) if (!isfinite(b[k])) bounded = 0;
        if (!bounded) { b[0] = b[1] = b[2] = -RTO_INF; b[3] = b[4] = b[5] = RTO_INF; }
    }
}

/* rto_closest_hit for a batch, rays and primitives in blocks (64 rays x 512 primitives stay in cache); per ray the
 * primitives still come in ascending id order, so strict '<' resolves ties as one_pass_intersection does
 * (s2899221/Code/raytracer.cpp:143-157). */
static void closest_hits_blocked(const rto_scene* sc, const double* rays6, int64_t n, rto_hit* out, int nt) {
    if (sc->instanced) {
        #pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
        for (int64_t i = 0; i < n; ++i) rto_closest_hit(sc, rays6 + i*6, rays6 + i*6 + 3, &out[i]);
        return;
    }
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int64_t b0 = 0; b0 < n; b0 += 64) {
        const int64_t b1 = (b0 + 64 < n) ? b0 + 64 : n;
        for (int64_t i = b0; i < b1; ++i) hit_clear(&out[i]);
        for (int64_t c0 = 0; c0 < sc->n_prims; c0 += 512) {
            const int64_t c1 = (c0 + 512 < sc->n_prims) ? c0 + 512 : sc->n_prims;
            for (int64_t i = b0; i < b1; ++i) {
                const double* o = rays6 + i*6; const double* d = o + 3;
                for (int64_t j = c0; j < c1; ++j) {
                    if (sc->prim_bounds && !may_touch(o, d, sc->prim_bounds + 6*j)) continue;
                    rto_hit h;
                    if (rto_intersect(&sc->prims[j], o, d, &h) && h.dist < out[i].dist) {
                        out[i] = h; out[i].id = (int32_t)j; out[i].prim = (int32_t)j;
                    }
                }
            }
        }
    }
}

typedef struct { const uint8_t* lit; int next; } verdict_list;
static int listed_occluded(const void* ctx, const double o[3], const double d[3], double light_dist) {
    verdict_list* v = (verdict_list*)ctx; (void)o; (void)d; (void)light_dist;
    return v->lit[v->next++] ? 0 : 1;
}

void rto_brute_shade_rays(const rto_scene* sc, const double* rays6, int64_t n, rto_hit* hits, uint8_t* lit,
                          uint8_t* rgb, int threads) {
    const int nt = pick_threads(threads);
    const int64_t L = sc->n_lights;
    closest_hits_blocked(sc, rays6, n, hits, nt);
    if (L <= 0) { memset(rgb, 0, (size_t)n * 3u); return; }
    /* the shadow rays of shade() for every hit and light, as shade_with builds them */
    int64_t n_hit = 0;
    for (int64_t i = 0; i < n; ++i) n_hit += hits[i].hit ? 1 : 0;
    double* srays = (double*)malloc((size_t)(n_hit * L > 0 ? n_hit * L : 1) * 6u * sizeof(double));
    double* sdist = (double*)malloc((size_t)(n_hit * L > 0 ? n_hit * L : 1) * sizeof(double));
    rto_hit* shits = (rto_hit*)malloc((size_t)(n_hit * L > 0 ? n_hit * L : 1) * sizeof(rto_hit));
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!hits[i].hit) continue;
        double nn[3]; normalize3(hits[i].n[0], hits[i].n[1], hits[i].n[2], nn);
        const double* p = hits[i].p;
        for (int64_t li = 0; li < L; ++li, ++k) {
            const rto_light* Lt = &sc->lights[li];
            const double lx = Lt->pos[0] - p[0], ly = Lt->pos[1] - p[1], lz = Lt->pos[2] - p[2];
            const double len2 = lx*lx + ly*ly + lz*lz;
            const double dist2 = (1e-12 < len2) ? len2 : 1e-12;
            double ldir[3]; normalize3(lx, ly, lz, ldir);
            sdist[k] = shadow_origin(p, nn, ldir, dist2, srays + 6*k);
            srays[6*k + 3] = ldir[0]; srays[6*k + 4] = ldir[1]; srays[6*k + 5] = ldir[2];
        }
    }
    closest_hits_blocked(sc, srays, n_hit * L, shits, nt);
    k = 0;
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t li = 0; li < L; ++li) lit[i*L + li] = 0;
        if (hits[i].hit)
            for (int64_t li = 0; li < L; ++li, ++k) {
                const double t = shits[k].dist;
                const int blocked = shits[k].hit && t > 1e-4 && t < sdist[k] - 1e-4;      /* raytracer.cpp:230-235 */
                lit[i*L + li] = blocked ? 0 : 1;
            }
        verdict_list v = { lit + i*L, 0 };
        shade_with(sc, &hits[i], rays6 + i*6, rgb + i*3, listed_occluded, &v, 0);
    }
    free(srays); free(sdist); free(shits);
}

/* Pixel centres (x + 0.5, y + 0.5), row-major, y down: s2899221/Code/raytracer.cpp:131-134. */
void rto_trace_primary(const rto_scene* sc, const rto_camera* cam, int x0, int y0, int w, int h,
                       int32_t* ids, double* dist, int threads) {
    const int nt = pick_threads(threads); (void)nt;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int yy = 0; yy < h; ++yy)
        for (int xx = 0; xx < w; ++xx) {
            double o[3], d[3];
            rto_pixel_to_ray(cam, (double)(x0 + xx) + 0.5, (double)(y0 + yy) + 0.5, o, d);
            rto_hit hit; rto_closest_hit(sc, o, d, &hit);
            const int64_t k = (int64_t)yy * w + xx;
            if (ids) ids[k] = hit.id;
            if (dist) dist[k] = hit.dist;
        }
}

/* RayTracer::render_unaccelerated_ppm with samples_per_pixel = 1, s2899221/Code/raytracer.cpp:373-442. */
void rto_render(const rto_scene* sc, const rto_camera* cam, int x0, int y0, int w, int h,
                int max_scatters, double min_scatter_intensity, uint8_t* rgb,
                int64_t* n_hits, int64_t* shadow_rays, int threads) {
    if (n_hits) *n_hits = 0;
    if (shadow_rays) *shadow_rays = 0;
    if (sc->n_lights <= 0) { memset(rgb, 0, (size_t)w * (size_t)h * 3u); return; }
    uint8_t lut[256]; rto_gamma_lut(lut);
    int64_t tot_sh = 0, tot_hits = 0;
    const int nt = pick_threads(threads); (void)nt;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nt) reduction(+:tot_sh, tot_hits)
    for (int yy = 0; yy < h; ++yy)
        for (int xx = 0; xx < w; ++xx) {
            double o[3], d[3];
            rto_pixel_to_ray(cam, (double)(x0 + xx) + 0.5, (double)(y0 + yy) + 0.5, o, d);
            uint8_t c[3];
            int64_t sh = 0; int64_t hb[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            trace_rec(sc, o, d, 0, max_scatters, min_scatter_intensity, c, &sh, hb);
            tot_sh += sh; tot_hits += hb[0];
            uint8_t* px = rgb + ((int64_t)yy * w + xx) * 3;
            for (int k = 0; k < 3; ++k) {
                /* mean over one sample, rounded, then gamma: raytracer.cpp:424-435 */
                const int lin = (int)round((double)c[k] / 1.0);
                px[k] = lut[lin < 0 ? 0 : (lin > 255 ? 255 : lin)];
            }
        }
    if (n_hits) *n_hits = tot_hits;
    if (shadow_rays) *shadow_rays = tot_sh;
}

/* ------------------------------------------------------------------------------------------ */
/* flattened-BVH traversal (restates the product's traversal, not the reference's BVH)         */

typedef struct {
    const rto_scene* sc; const rto_bvh* bvh; int64_t* counts;
    /* shadow walks of the frame pipeline: the primitive the ray starts on and the unit normal there (have_self != 0) */
    int have_self; int32_t self_pid; double self_n[3];
} bvh_ctx;

typedef struct { float inv[3], np[3], nm[3]; } box_ray;   /* reciprocal direction, -(origin +/- pad) * reciprocal */

static void box_ray_init(box_ray* br, const double o[3], const double d[3], float radius) {
    /* padding = 2^-18 * max(|origin|, scene radius); reciprocal of the normalised direction as |d| / d in FP32, kept
     * finite (+-2^100) so that a ray along an axis yields huge distances of the right sign */
    const float ox = (float)o[0], oy = (float)o[1], oz = (float)o[2];
    const float m = fmaxf(fmaxf(radius, fabsf(ox)), fmaxf(fabsf(oy), fabsf(oz)));
    const float pad = ldexpf(m, -18);
    const float big = 1.2676506e30f;
    const double d_len = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]);
    const float of[3] = { ox, oy, oz };
    for (int k = 0; k < 3; ++k) {
        br->inv[k] = fminf(fmaxf((float)d_len / (float)d[k], -big), big);
        br->np[k] = -(of[k] + pad) * br->inv[k];
        br->nm[k] = -(of[k] - pad) * br->inv[k];
    }
}

/* Slab test against the box grown by pad on every side, each plane's distance as one fused multiply-add (the product's
 * arithmetic, so that both walks visit the same boxes); returns the entry distance clamped to >= 0, or +inf when the
 * ray misses or the box starts beyond `limit`. */
static inline float box_entry(const rto_node32* nd, const box_ray* br, float limit) {
    float tn = 0.0f, tf = limit;
    for (int k = 0; k < 3; ++k) {
        const float a = fmaf(nd->lo[k], br->inv[k], br->np[k]);
        const float b = fmaf(nd->hi[k], br->inv[k], br->nm[k]);
        const float t0 = fminf(a, b), t1 = fmaxf(a, b);
        tn = fmaxf(tn, t0);
        tf = fminf(tf, t1);
    }
    return (tn <= tf) ? tn : __builtin_inff();
}

static float f32_up(double x) {            /* smallest float >= max(x, 0); NaN -> +inf */
    if (!(x == x)) return __builtin_inff();
    float f = (float)x;
    if ((double)f < x) f = nextafterf(f, __builtin_inff());
    return (f > 0.0f) ? f : 0.0f;
}

#define RTO_STACK 64

typedef struct {
    const bvh_ctx* c; const double* o; const double* d; int mode; double light_dist;
    rto_hit best; double limit; float limit_f; int found_mid; int64_t vp;
    int32_t skip_pid;                  /* the ray's own primitive when its test is known to miss, else -1 */
} walk_state;

/* One primitive slot; returns 1 when the walk can stop (shadow mode: a hit at <= 1e-4). */
static int walk_slot(walk_state* w, uint32_t slot) {
    const uint32_t pid = w->c->bvh->slot_prim[slot];
    if ((int32_t)pid == w->skip_pid) return 0;
    rto_hit h; w->vp += 1;
    if (!rto_intersect(&w->c->sc->prims[pid], w->o, w->d, &h)) return 0;
    if (w->mode == 0) {
        if (h.dist < w->best.dist || (h.dist == w->best.dist && (int32_t)pid < w->best.id)) {
            w->best = h; w->best.id = (int32_t)pid; w->best.prim = (int32_t)pid;
            w->limit = h.dist; w->limit_f = f32_up(w->limit);
        }
    } else {
        if (h.dist <= 1e-4) return 1;                                   /* the closest hit is not beyond eps */
        if (h.dist < w->light_dist - 1e-4 && !w->found_mid) {
            w->found_mid = 1; w->limit = 1e-4; w->limit_f = f32_up(w->limit);   /* now only hits <= eps matter */
        }
    }
    return 0;
}

/* Generic walk.  mode 0: closest hit (tie -> lowest id).  mode 1: shadow verdict for a light at
 * light_dist - returns 1 when blocked by the reference's rule. */
#if defined(__x86_64__) && defined(__GNUC__)
/* fmaf() as one instruction where the CPU has it (function multiversioning; the plain clone calls libm); nothing else in
 * here is contracted (-ffp-contract=off) */
__attribute__((target_clones("fma", "default")))
#endif
static int bvh_walk(const bvh_ctx* c, const double o[3], const double d[3], int mode, double light_dist, rto_hit* out) {
    const rto_bvh* B = c->bvh;
    box_ray br; box_ray_init(&br, o, d, B->radius);
    walk_state w; w.c = c; w.o = o; w.d = d; w.mode = mode; w.light_dist = light_dist;
    hit_clear(&w.best); w.found_mid = 0; w.vp = 0;
    /* The product's frame pipeline passes over the test of the primitive a shadow ray starts on when it is known to miss:
     * the primitive is convex and lies behind the tangent plane at the hit, the ray starts 1e-4 in front of that plane
     * and does not point back at it.  True for a flat primitive on either side, for a solid with a proper scale when
     * the ray leaves along the outward normal.  (The verdict cannot differ; the count of primitive tests does.) */
    w.skip_pid = -1;
    if (mode == 1 && c->have_self && c->self_pid >= 0 && !c->sc->instanced) {
        const rto_prim* P = &c->sc->prims[c->self_pid];
        const int outward = c->self_n[0]*d[0] + c->self_n[1]*d[1] + c->self_n[2]*d[2] >= 0.0;
        if (P->kind == RTO_PLANE || P->kind == RTO_TRIANGLE) w.skip_pid = c->self_pid;
        else {
            const double* S = P->kind == RTO_SPHERE ? P->q + 3 : P->q + 6;
            const int proper = S[0] > 0.0 && S[1] > 0.0 && S[2] > 0.0 && S[0] < 1e300 && S[1] < 1e300 && S[2] < 1e300;
            if (proper && outward) w.skip_pid = c->self_pid;
        }
    }
    w.limit = (mode == 0) ? RTO_INF : (light_dist - 1e-4);   /* only hits with dist < limit matter */
    w.limit_f = f32_up(w.limit);
    /* The stack keeps, for every postponed child, its (first, count) and its box entry distance: a pop needs
     * no memory access and drops entries that now start beyond the search limit.  Node records fetched: the
     * root once, then both children of every internal node visited. */
    struct { uint32_t first, count; float t; } stack[RTO_STACK];
    int sp = 0;
    int64_t vn = 0;
    int closed = 0;

    if (B->nodes4 && B->n_nodes4 > 0 && B->n_nodes > 0) {
        /* the product's wide walk, restated: node records fetched = the root once, then four per wide node visited */
        const uint32_t NONE = 0xFFFFFFFFu;
        struct { uint32_t ref; float t; } wstack[96];
        int wsp = 0;
        uint32_t cur = NONE;
        const rto_node32 root = B->nodes[0]; vn += 1;
        if (box_entry(&root, &br, w.limit_f) != __builtin_inff()) cur = 0u;
        for (;;) {
            if (cur == NONE) {
                while (wsp > 0 && cur == NONE) { --wsp; if (wstack[wsp].t <= w.limit_f) cur = wstack[wsp].ref; }
                if (cur == NONE) break;
            }
            if (cur >> 28) {
                const uint32_t first = cur & 0x0FFFFFFFu, count = cur >> 28;
                for (uint32_t s = 0; s < count && !closed; ++s) closed = walk_slot(&w, first + s);
                if (closed) break;
                cur = NONE;
                continue;
            }
            const rto_node4* n4 = &B->nodes4[cur]; vn += 4;
            float t[4]; uint32_t r[4];
            for (int k = 0; k < 4; ++k) {
                rto_node32 nd;
                for (int a = 0; a < 3; ++a) { nd.lo[a] = n4->lo[a][k]; nd.hi[a] = n4->hi[a][k]; }
                t[k] = box_entry(&nd, &br, w.limit_f);
                r[k] = n4->ref[k];
            }
            if (r[2] == NONE) t[2] = __builtin_inff();
            if (r[3] == NONE) t[3] = __builtin_inff();
#define RTO_EXCHANGE(i, j) if (t[j] < t[i]) { const float tt = t[i]; t[i] = t[j]; t[j] = tt; const uint32_t rr = r[i]; r[i] = r[j]; r[j] = rr; }
            RTO_EXCHANGE(0, 1) RTO_EXCHANGE(2, 3) RTO_EXCHANGE(0, 2) RTO_EXCHANGE(1, 3) RTO_EXCHANGE(1, 2)
#undef RTO_EXCHANGE
            for (int k = 3; k >= 1; --k) if (t[k] != __builtin_inff()) { wstack[wsp].ref = r[k]; wstack[wsp].t = t[k]; ++wsp; }
            cur = (t[0] != __builtin_inff()) ? r[0] : NONE;
        }
    } else if (B->n_nodes > 0) {
        const rto_node32 root = B->nodes[0]; vn += 1;
        int have = (box_entry(&root, &br, w.limit_f) != __builtin_inff());
        uint32_t first = root.first, count = root.count;
        for (;;) {
            if (have) {
                if (count > 0) {
                    for (uint32_t s = 0; s < count && !closed; ++s) closed = walk_slot(&w, first + s);
                    if (closed) break;
                    have = 0;
                } else {
                    const rto_node32 l = B->nodes[first], r = B->nodes[first + 1]; vn += 2;
                    const float tl = box_entry(&l, &br, w.limit_f), tr = box_entry(&r, &br, w.limit_f);
                    const int hl = tl != __builtin_inff(), hr = tr != __builtin_inff();
                    if (hl && hr) {
                        if (tr < tl) { stack[sp].first = l.first; stack[sp].count = l.count; stack[sp].t = tl; ++sp; first = r.first; count = r.count; }
                        else         { stack[sp].first = r.first; stack[sp].count = r.count; stack[sp].t = tr; ++sp; first = l.first; count = l.count; }
                    } else if (hl) { first = l.first; count = l.count; }
                    else if (hr)   { first = r.first; count = r.count; }
                    else have = 0;
                }
            }
            if (!have) {
                while (sp > 0 && !have) {
                    --sp;
                    if (stack[sp].t <= w.limit_f) { first = stack[sp].first; count = stack[sp].count; have = 1; }
                }
                if (!have) break;
            }
        }
    }
    /* unbounded primitives are tested against every ray */
    for (uint32_t s = 0; s < B->n_loose && !closed; ++s) {
        if (B->loose_boxes) {          /* the product's cull: the padded box of the primitive, tested like a box of the tree */
            rto_node32 nd;
            for (int a = 0; a < 3; ++a) { nd.lo[a] = B->loose_boxes[8*s + a]; nd.hi[a] = B->loose_boxes[8*s + 4 + a]; }
            if (box_entry(&nd, &br, w.limit_f) == __builtin_inff()) continue;
        }
        closed = walk_slot(&w, (uint32_t)B->n_slots + s);
    }
    c->counts[0] += vn; c->counts[1] += w.vp;
    if (mode == 0) { if (out) *out = w.best; return w.best.hit; }
    return w.found_mid && !closed;
}

int rto_bvh_closest_hit(const rto_scene* sc, const rto_bvh* bvh, const double o[3], const double d[3],
                        rto_hit* out, int64_t counts[2]) {
    int64_t local[2] = { 0, 0 };
    bvh_ctx c = { sc, bvh, counts ? counts : local, 0, -1, { 0.0, 0.0, 0.0 } };
    return bvh_walk(&c, o, d, 0, 0.0, out);
}

int rto_bvh_occluded(const rto_scene* sc, const rto_bvh* bvh, const double o[3], const double d[3],
                     double light_dist, int64_t counts[2]) {
    int64_t local[2] = { 0, 0 };
    bvh_ctx c = { sc, bvh, counts ? counts : local, 0, -1, { 0.0, 0.0, 0.0 } };
    return bvh_walk(&c, o, d, 1, light_dist, 0);
}

static int bvh_occluded_cb(const void* ctx, const double o[3], const double d[3], double light_dist) {
    return bvh_walk((const bvh_ctx*)ctx, o, d, 1, light_dist, 0);
}

void rto_bvh_trace_rays(const rto_scene* sc, const rto_bvh* bvh, const double* rays6, int64_t n, rto_hit* out,
                        int64_t counts[2], int threads) {
    int64_t c0 = 0, c1 = 0;
    const int nt = pick_threads(threads); (void)nt;
    #pragma omp parallel for schedule(dynamic, 64) num_threads(nt) reduction(+:c0, c1)
    for (int64_t i = 0; i < n; ++i) {
        int64_t pc[2] = { 0, 0 };
        bvh_ctx c = { sc, bvh, pc, 0, -1, { 0.0, 0.0, 0.0 } };
        bvh_walk(&c, rays6 + i*6, rays6 + i*6 + 3, 0, 0.0, &out[i]);
        c0 += pc[0]; c1 += pc[1];
    }
    if (counts) { counts[0] += c0; counts[1] += c1; }
}

void rto_bvh_render(const rto_scene* sc, const rto_bvh* bvh, const rto_camera* cam, int x0, int y0, int w, int h,
                    int step, int32_t* ids, uint8_t* rgb, int64_t counts[4], int64_t* n_hits, int threads) {
    uint8_t lut[256]; rto_gamma_lut(lut);
    if (step < 1) step = 1;
    int64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, hits = 0;
    const int nt = pick_threads(threads); (void)nt;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nt) reduction(+:c0, c1, c2, c3, hits)
    for (int yy = 0; yy < h; yy += step)
        for (int xx = 0; xx < w; xx += step) {
            double o[3], d[3];
            rto_pixel_to_ray(cam, (double)(x0 + xx) + 0.5, (double)(y0 + yy) + 0.5, o, d);
            int64_t pc[2] = { 0, 0 }, sc_[2] = { 0, 0 };
            bvh_ctx pctx = { sc, bvh, pc, 0, -1, { 0.0, 0.0, 0.0 } };
            bvh_ctx sctx = { sc, bvh, sc_, 0, -1, { 0.0, 0.0, 0.0 } };
            rto_hit hit;
            bvh_walk(&pctx, o, d, 0, 0.0, &hit);
            if (hit.hit) { sctx.have_self = 1; sctx.self_pid = hit.prim; normalize3(hit.n[0], hit.n[1], hit.n[2], sctx.self_n); }
            uint8_t c[3] = { 0, 0, 0 };
            if (sc->n_lights > 0) shade_with(sc, &hit, o, c, bvh_occluded_cb, &sctx, 0);
            const int64_t k = (int64_t)(yy / step) * ((w + step - 1) / step) + (xx / step);
            if (ids) ids[k] = hit.id;
            if (rgb) { rgb[k*3+0] = lut[c[0]]; rgb[k*3+1] = lut[c[1]]; rgb[k*3+2] = lut[c[2]]; }
            c0 += pc[0]; c1 += pc[1]; c2 += sc_[0]; c3 += sc_[1]; hits += hit.hit;
        }
    if (counts) { counts[0] = c0; counts[1] = c1; counts[2] = c2; counts[3] = c3; }
    if (n_hits) *n_hits = hits;
}
