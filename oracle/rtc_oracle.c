/* TEST INFRASTRUCTURE ONLY -- see rtc_oracle.h.  Literal CPU restatement of the reference's
 * trace loop.  Citations are file:line under /root/reference.  All arithmetic is fp64 like the
 * reference (vector.py:79, canvas.py:37).  Points/vectors are kept as xyz: the reference's
 * 4th component is exactly 1 (points) or 0 (vectors) everywhere on this path, so its 4-wide dot
 * products (vector.py:180) equal the 3-wide ones here.
 */
#include "rtc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPSILON 0.000001 /* renderer.py:15, shapes.py:8 */

typedef struct { double x, y, z; } v3;
typedef struct { v3 o, d; } ray_t;
typedef struct { double t; int node; double u, v; } isect;
typedef struct { isect *h; int n, cap; } xs_t;

typedef struct {
    const orc_scene *s;
    double *pat_ab; /* mutable copy: NestedPattern writes p1.a / p1.b (shapes.py:1893-1894) */
    int64_t rays[4];
} ctx_t;

/* ---------------------------------------------------------------- small vector helpers */
static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 mul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vector.py:199-201 */
static v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* vector.py:155 */
static double magnitude(v3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
/* vector.py:170: v * (1.0 / magnitude(v)) */
static v3 normalize(v3 a) { return mul(a, 1.0 / magnitude(a)); }
/* renderer.py:893 */
static v3 reflect(v3 in, v3 n) { return sub(in, mul(n, 2 * dot(in, n))); }
/* matrix * point (w = 1) and matrix * vector (w = 0), matrix.py:172 */
static v3 mpoint(const double *m, v3 p) {
    return V(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
             m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
             m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
static v3 mvector(const double *m, v3 p) {
    return V(m[0] * p.x + m[1] * p.y + m[2] * p.z,
             m[4] * p.x + m[5] * p.y + m[6] * p.z,
             m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
/* transpose(m) * vector, xyz rows only (shapes.py:289-290 zeroes w afterwards) */
static v3 mtvector(const double *m, v3 p) {
    return V(m[0] * p.x + m[4] * p.y + m[8] * p.z,
             m[1] * p.x + m[5] * p.y + m[9] * p.z,
             m[2] * p.x + m[6] * p.y + m[10] * p.z);
}
/* np.isclose(a, b) with default rtol=1e-5, atol=1e-8 */
static int isclose(double a, double b) {
    if (isnan(a) || isnan(b)) return 0;
    if (isinf(a) || isinf(b)) return a == b;
    return fabs(a - b) <= 1e-8 + 1e-5 * fabs(b);
}
/* Python / numpy floor-mod for floats */
static double pymod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) { if ((b < 0) != (m < 0)) m += b; }
    else m = copysign(0.0, b);
    return m;
}
/* Python max(a,b,c) / min(a,b,c): first wins unless a later one compares strictly beyond it
 * (this is what lets the NaN of 0*inf through, shapes.py:577-578, 606-614) */
static double pymax3(double a, double b, double c) { double m = a; if (b > m) m = b; if (c > m) m = c; return m; }
static double pymin3(double a, double b, double c) { double m = a; if (b < m) m = b; if (c < m) m = c; return m; }

/* ---------------------------------------------------------------- hit lists */
static void xs_init(xs_t *x) { x->h = NULL; x->n = 0; x->cap = 0; }
static void xs_free(xs_t *x) { free(x->h); x->h = NULL; x->n = x->cap = 0; }
static void xs_push(xs_t *x, double t, int node, double u, double v) {
    if (x->n == x->cap) {
        x->cap = x->cap ? 2 * x->cap : 16;
        x->h = (isect *)realloc(x->h, (size_t)x->cap * sizeof(isect));
    }
    isect i = {t, node, u, v};
    x->h[x->n++] = i;
}
/* shapes.py:1384 sorted(..., key=t): stable.  Entries [from, n) are sorted in place. */
static void xs_sort(xs_t *x, int from) {
    for (int i = from + 1; i < x->n; i++) {
        isect k = x->h[i];
        int j = i - 1;
        while (j >= from && x->h[j].t > k.t) { x->h[j + 1] = x->h[j]; j--; }
        x->h[j + 1] = k;
    }
}

/* ---------------------------------------------------------------- local intersections */
/* shapes.py:597-618 (Cube) and 87-108 (BoundingBox): identical but for the slab bounds */
static void check_axis(double origin, double direction, double lo, double hi, double *tmin, double *tmax) {
    double tmin_numerator = lo - origin, tmax_numerator = hi - origin, a, b;
    if (fabs(direction) >= EPSILON) {
        a = tmin_numerator / direction;
        b = tmax_numerator / direction;
    } else {
        a = tmin_numerator * INFINITY; /* 0*inf = NaN is NOT repaired: `x == np.nan` is always False */
        b = tmax_numerator * INFINITY;
    }
    if (a > b) { *tmin = b; *tmax = a; } else { *tmin = a; *tmax = b; }
}
/* returns 1 and [tmin,tmax] unless tmin > tmax (shapes.py:75-85, 572-582) */
static int box_intersect(ray_t r, const double *lo, const double *hi, double *tmin, double *tmax) {
    double xa, xb, ya, yb, za, zb;
    check_axis(r.o.x, r.d.x, lo[0], hi[0], &xa, &xb);
    check_axis(r.o.y, r.d.y, lo[1], hi[1], &ya, &yb);
    check_axis(r.o.z, r.d.z, lo[2], hi[2], &za, &zb);
    *tmin = pymax3(xa, ya, za);
    *tmax = pymin3(xb, yb, zb);
    return !(*tmin > *tmax);
}

/* shapes.py:525-535 */
static void sphere_intersect(int node, ray_t r, xs_t *out) {
    v3 sphere_to_ray = r.o;
    double a = dot(r.d, r.d);
    double b = 2 * dot(r.d, sphere_to_ray);
    double c = dot(sphere_to_ray, sphere_to_ray) - 1;
    double discriminant = b * b - 4 * a * c;
    if (discriminant < 0) return;
    int from = out->n;
    xs_push(out, (-b - sqrt(discriminant)) / (2 * a), node, 0, 0);
    xs_push(out, (-b + sqrt(discriminant)) / (2 * a), node, 0, 0);
    xs_sort(out, from);
}
/* shapes.py:550-555 */
static void plane_intersect(int node, ray_t r, xs_t *out) {
    if (fabs(r.d.y) < EPSILON) return;
    xs_push(out, -r.o.y / r.d.y, node, 0, 0);
}
/* shapes.py:572-582 */
static void cube_intersect(int node, ray_t r, xs_t *out) {
    static const double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
    double tmin, tmax;
    if (!box_intersect(r, lo, hi, &tmin, &tmax)) return;
    int from = out->n;
    xs_push(out, tmin, node, 0, 0);
    xs_push(out, tmax, node, 0, 0);
    xs_sort(out, from);
}
/* shapes.py:734-792 */
static void cylinder_intersect(const orc_scene *s, int node, ray_t r, xs_t *out) {
    double minimum = s->fparam[3 * node], maximum = s->fparam[3 * node + 1];
    int closed = s->fparam[3 * node + 2] != 0;
    int from = out->n;
    double a = r.d.x * r.d.x + r.d.z * r.d.z;
    if (!isclose(a, 0)) {
        double b = 2 * r.o.x * r.d.x + 2 * r.o.z * r.d.z;
        double c = r.o.x * r.o.x + r.o.z * r.o.z - 1;
        double disc = b * b - 4 * a * c;
        if (disc < 0) return;
        double discsqrt = sqrt(disc);
        double t0 = (-b - discsqrt) / (2 * a), t1 = (-b + discsqrt) / (2 * a);
        if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
        double y0 = r.o.y + t0 * r.d.y;
        if (minimum < y0 && y0 < maximum) xs_push(out, t0, node, 0, 0);
        double y1 = r.o.y + t1 * r.d.y;
        if (minimum < y1 && y1 < maximum) xs_push(out, t1, node, 0, 0);
    }
    if (closed && !isclose(r.d.y, 0)) { /* _intersect_caps 782-792, _check_cap 776-780 */
        double lim[2] = {minimum, maximum};
        for (int k = 0; k < 2; k++) {
            double t = (lim[k] - r.o.y) / r.d.y;
            double x = r.o.x + t * r.d.x, z = r.o.z + t * r.d.z;
            if (!(x * x + z * z > 1)) xs_push(out, t, node, 0, 0);
        }
    }
    xs_sort(out, from);
}
/* shapes.py:636-705, quirks kept: the a~0 branch divides by 2b even when b~0 (647-650); the cap
 * radius test is against |y_cap|, not y_cap^2 (689-693) */
static void cone_intersect(const orc_scene *s, int node, ray_t r, xs_t *out) {
    double minimum = s->fparam[3 * node], maximum = s->fparam[3 * node + 1];
    int closed = s->fparam[3 * node + 2] != 0;
    int from = out->n;
    double a = r.d.x * r.d.x + r.d.z * r.d.z - r.d.y * r.d.y;
    double b = 2 * r.o.x * r.d.x + 2 * r.o.z * r.d.z - 2 * r.o.y * r.d.y;
    double c = r.o.x * r.o.x + r.o.z * r.o.z - r.o.y * r.o.y;
    if (isclose(a, 0)) {
        xs_push(out, -c / (2 * b), node, 0, 0);
    } else {
        double disc = b * b - 4 * a * c;
        if (disc < 0) return;
        double discsqrt = sqrt(disc);
        double t0 = (-b - discsqrt) / (2 * a), t1 = (-b + discsqrt) / (2 * a);
        if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
        double y0 = r.o.y + t0 * r.d.y;
        if (minimum < y0 && y0 < maximum) xs_push(out, t0, node, 0, 0);
        double y1 = r.o.y + t1 * r.d.y;
        if (minimum < y1 && y1 < maximum) xs_push(out, t1, node, 0, 0);
    }
    if (closed && !isclose(r.d.y, 0)) {
        double lim[2] = {minimum, maximum};
        for (int k = 0; k < 2; k++) {
            double t = (lim[k] - r.o.y) / r.d.y;
            double x = r.o.x + t * r.d.x, z = r.o.z + t * r.d.z;
            if (!(x * x + z * z > fabs(lim[k]))) xs_push(out, t, node, 0, 0);
        }
    }
    xs_sort(out, from);
}
/* shapes.py:812-830 and 860-878 (Moller-Trumbore) */
static void triangle_intersect(const orc_scene *s, int node, ray_t r, xs_t *out) {
    const double *T = s->tri + 18 * (size_t)s->tri_idx[node];
    v3 p1 = V(T[0], T[1], T[2]), p2 = V(T[3], T[4], T[5]), p3 = V(T[6], T[7], T[8]);
    v3 e1 = sub(p2, p1), e2 = sub(p3, p1);
    v3 dir_cross_e2 = cross(r.d, e2);
    double det = dot(e1, dir_cross_e2);
    if (fabs(det) < EPSILON) return;
    double f = 1.0 / det;
    v3 p1_to_origin = sub(r.o, p1);
    double u = f * dot(p1_to_origin, dir_cross_e2);
    if (u < 0 || u > 1) return;
    v3 origin_cross_e1 = cross(p1_to_origin, e1);
    double v = f * dot(r.d, origin_cross_e1);
    if (v < 0 || (u + v) > 1) return;
    double t = f * dot(e2, origin_cross_e1);
    xs_push(out, t, node, u, v);
}

/* ---------------------------------------------------------------- tree intersection */
/* CSG.includes / Group.includes / Shape.includes (shapes.py:268-269, 363-364, 449-450) */
static int includes(const orc_scene *s, int node, int other) {
    if (s->kind[node] == ORC_GROUP || s->kind[node] == ORC_CSG) {
        for (int k = 0; k < s->child_cnt[node]; k++)
            if (includes(s, s->child_idx[s->child_off[node] + k], other)) return 1;
        return 0;
    }
    return node == other;
}
/* shapes.py:409-422 */
static int intersection_allowed(int op, int lhit, int inl, int inr) {
    switch (op) {
    case ORC_UNION: return (lhit && !inr) || (!lhit && !inl);
    case ORC_INTERSECT: return (lhit && inr) || (!lhit && inl);
    default: return (lhit && !inr) || (!lhit && inl);
    }
}

/* Shape.intersect (shapes.py:271-273) + transform (1447-1448): the ray is carried into the
 * node's space by inverse(node.transform); the direction is NOT renormalised. */
static void intersect_node(const orc_scene *s, int node, ray_t rin, xs_t *out) {
    const double *m = s->inv + 16 * (size_t)node;
    ray_t r = {mpoint(m, rin.o), mvector(m, rin.d)};
    switch (s->kind[node]) {
    case ORC_SPHERE: sphere_intersect(node, r, out); break;
    case ORC_PLANE: plane_intersect(node, r, out); break;
    case ORC_CUBE: cube_intersect(node, r, out); break;
    case ORC_CYLINDER: cylinder_intersect(s, node, r, out); break;
    case ORC_CONE: cone_intersect(s, node, r, out); break;
    case ORC_TRIANGLE:
    case ORC_SMOOTH_TRIANGLE: triangle_intersect(s, node, r, out); break;
    case ORC_GROUP: { /* shapes.py:452-459: every child, the bounding-box test is disabled */
        int from = out->n;
        for (int k = 0; k < s->child_cnt[node]; k++)
            intersect_node(s, s->child_idx[s->child_off[node] + k], r, out);
        xs_sort(out, from);
        break;
    }
    case ORC_CSG: { /* shapes.py:366-388 */
        double tmin, tmax;
        const double *bb = s->bbox + 6 * (size_t)node;
        if (!box_intersect(r, bb, bb + 3, &tmin, &tmax)) break;
        int left = s->child_idx[s->child_off[node]], right = s->child_idx[s->child_off[node] + 1];
        xs_t xs; xs_init(&xs);
        intersect_node(s, left, r, &xs);
        intersect_node(s, right, r, &xs);
        xs_sort(&xs, 0);
        int inl = 0, inr = 0;
        for (int i = 0; i < xs.n; i++) {
            int lhit = includes(s, left, xs.h[i].node);
            if (intersection_allowed(s->csg_op[node], lhit, inl, inr))
                xs_push(out, xs.h[i].t, xs.h[i].node, xs.h[i].u, xs.h[i].v);
            if (lhit) inl = !inl; else inr = !inr;
        }
        xs_free(&xs);
        break;
    }
    }
}

/* renderer.py:75-79 */
static void intersect_world(const orc_scene *s, ray_t r, xs_t *out) {
    for (int k = 0; k < s->n_top; k++) intersect_node(s, s->top[k], r, out);
    xs_sort(out, 0);
}
/* shapes.py:1422-1425: first t > 0, strictly, no epsilon */
static int hit(const xs_t *xs) {
    for (int i = 0; i < xs->n; i++) if (xs->h[i].t > 0) return i;
    return -1;
}

/* ---------------------------------------------------------------- normals */
/* shapes.py:282-286 */
static v3 world_to_object(const orc_scene *s, int node, v3 p) {
    if (s->parent[node] >= 0) p = world_to_object(s, s->parent[node], p);
    return mpoint(s->inv + 16 * (size_t)node, p);
}
/* shapes.py:288-296: w zeroed and the vector renormalised at EVERY level */
static v3 normal_to_world(const orc_scene *s, int node, v3 n) {
    n = normalize(mtvector(s->inv + 16 * (size_t)node, n));
    if (s->parent[node] >= 0) n = normal_to_world(s, s->parent[node], n);
    return n;
}
static v3 local_normal_at(const orc_scene *s, int node, v3 p, double hu, double hv) {
    switch (s->kind[node]) {
    case ORC_SPHERE: return p; /* shapes.py:537-538 */
    case ORC_PLANE: return V(0, 1, 0); /* 557-558 */
    case ORC_CUBE: { /* 584-595 */
        double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
        double maxc = pymax3(ax, ay, az);
        if (isclose(maxc, ax)) return V(p.x, 0, 0);
        if (isclose(maxc, ay)) return V(0, p.y, 0);
        return V(0, 0, p.z);
    }
    case ORC_CYLINDER: { /* 767-774 */
        double dist = p.x * p.x + p.z * p.z;
        double minimum = s->fparam[3 * node], maximum = s->fparam[3 * node + 1];
        if (dist < 1 && (maximum - EPSILON) < p.y) return V(0, 1, 0);
        if (dist < 1 && (minimum + EPSILON) > p.y) return V(0, -1, 0);
        return V(p.x, 0, p.z);
    }
    case ORC_CONE: { /* 676-687 (dist < 1, as written) */
        double dist = p.x * p.x + p.z * p.z;
        double minimum = s->fparam[3 * node], maximum = s->fparam[3 * node + 1];
        if (dist < 1 && (maximum - EPSILON) < p.y) return V(0, 1, 0);
        if (dist < 1 && (minimum + EPSILON) > p.y) return V(0, -1, 0);
        double y = sqrt(dist);
        if (p.y > 0) y = -y;
        return V(p.x, y, p.z);
    }
    case ORC_TRIANGLE: { /* 805-810: normalize(cross(e2, e1)) */
        const double *T = s->tri + 18 * (size_t)s->tri_idx[node];
        v3 p1 = V(T[0], T[1], T[2]);
        v3 e1 = sub(V(T[3], T[4], T[5]), p1), e2 = sub(V(T[6], T[7], T[8]), p1);
        return normalize(cross(e2, e1));
    }
    case ORC_SMOOTH_TRIANGLE: { /* 857-858 */
        const double *T = s->tri + 18 * (size_t)s->tri_idx[node];
        v3 n1 = V(T[9], T[10], T[11]), n2 = V(T[12], T[13], T[14]), n3 = V(T[15], T[16], T[17]);
        return add(add(mul(n2, hu), mul(n3, hv)), mul(n1, 1 - hu - hv));
    }
    }
    return V(0, 0, 0);
}
/* shapes.py:275-280 */
static v3 normal_at(const orc_scene *s, int node, v3 world_point, double hu, double hv) {
    v3 local_point = world_to_object(s, node, world_point);
    v3 local_normal = local_normal_at(s, node, local_point, hu, hv);
    v3 world_normal = normal_to_world(s, node, local_normal);
    return normalize(world_normal);
}

/* ---------------------------------------------------------------- pnoise3 of PyPI `noise` 1.2.2
 * The package is a dependency of the reference (shapes.py:1496, called at :1916-1920) that is neither vendored in
 * /root/reference nor installed in this image, so what follows restates its published algorithm: Ken Perlin's
 * "improved noise" (2002) as packaged by Casey Duncan (_perlin.c / _noise.h) -- single-precision arithmetic,
 * Perlin's reference permutation, a 16-entry gradient table, fade 6t^5 - 15t^4 + 10t^3, and for octaves > 1 the sum of
 * octaves (persistence 0.5, lacunarity 2, repeat 1024 scaled per octave, base 0) divided by the sum of amplitudes.
 * PARITY UNPINNED: no file of the reference holds an output of this function; tests/ check the restatement only
 * against the algorithm's own invariants and against a second, independent transcription (oracle/_stubs/noise.py). */
static const unsigned char PN_PERM[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23,
    190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174,
    20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133,
    230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18,
    169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38,
    147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2,
    44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112,
    104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49,
    192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29,
    24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180};
static const float PN_GRAD3[16][3] = {{1, 1, 0}, {-1, 1, 0}, {1, -1, 0}, {-1, -1, 0}, {1, 0, 1}, {-1, 0, 1}, {1, 0, -1}, {-1, 0, -1},
                                      {0, 1, 1}, {0, -1, 1}, {0, 1, -1}, {0, -1, -1}, {1, 0, -1}, {-1, 0, -1}, {0, -1, 1}, {0, 1, 1}};
static int pn_perm(int i) { return PN_PERM[i & 255]; } /* the package doubles the table; indices here stay below 512 */
static float pn_grad3(int hash, float x, float y, float z) {
    const float *g = PN_GRAD3[hash & 15];
    return x * g[0] + y * g[1] + z * g[2];
}
static float pn_lerp(float t, float a, float b) { return a + t * (b - a); }
static float pn_noise3(float x, float y, float z, int repeatx, int repeaty, int repeatz, int base) {
    int i = (int)floorf(fmodf(x, (float)repeatx)), j = (int)floorf(fmodf(y, (float)repeaty)), k = (int)floorf(fmodf(z, (float)repeatz));
    int ii = (int)fmodf((float)(i + 1), (float)repeatx), jj = (int)fmodf((float)(j + 1), (float)repeaty), kk = (int)fmodf((float)(k + 1), (float)repeatz);
    i = (i & 255) + base; j = (j & 255) + base; k = (k & 255) + base;
    ii = (ii & 255) + base; jj = (jj & 255) + base; kk = (kk & 255) + base;
    x -= floorf(x); y -= floorf(y); z -= floorf(z);
    float fx = x * x * x * (x * (x * 6 - 15) + 10), fy = y * y * y * (y * (y * 6 - 15) + 10), fz = z * z * z * (z * (z * 6 - 15) + 10);
    int A = pn_perm(i), AA = pn_perm(A + j), AB = pn_perm(A + jj), B = pn_perm(ii), BA = pn_perm(B + j), BB = pn_perm(B + jj);
    return pn_lerp(fz,
                   pn_lerp(fy, pn_lerp(fx, pn_grad3(pn_perm(AA + k), x, y, z), pn_grad3(pn_perm(BA + k), x - 1, y, z)),
                           pn_lerp(fx, pn_grad3(pn_perm(AB + k), x, y - 1, z), pn_grad3(pn_perm(BB + k), x - 1, y - 1, z))),
                   pn_lerp(fy, pn_lerp(fx, pn_grad3(pn_perm(AA + kk), x, y, z - 1), pn_grad3(pn_perm(BA + kk), x - 1, y, z - 1)),
                           pn_lerp(fx, pn_grad3(pn_perm(AB + kk), x, y - 1, z - 1), pn_grad3(pn_perm(BB + kk), x - 1, y - 1, z - 1))));
}
/* noise.pnoise3(x, y, z, octaves): arguments narrowed to float on entry, the float result widened on return */
double orc_pnoise3(double xd, double yd, double zd, int octaves) {
    float x = (float)xd, y = (float)yd, z = (float)zd;
    if (octaves == 1) return (double)pn_noise3(x, y, z, 1024, 1024, 1024, 0);
    if (octaves < 1) return NAN; /* the package raises ValueError */
    float freq = 1.0f, amp = 1.0f, max = 0.0f, total = 0.0f;
    for (int o = 0; o < octaves; o++) {
        total += pn_noise3(x * freq, y * freq, z * freq, (int)(1024 * freq), (int)(1024 * freq), (int)(1024 * freq), 0) * amp;
        max += amp;
        freq *= 2.0f;
        amp *= 0.5f;
    }
    return (double)(total / max);
}

/* ---------------------------------------------------------------- patterns */
typedef struct { double r, g, b; } rgb_t;
static rgb_t RGB(double r, double g, double b) { rgb_t c = {r, g, b}; return c; }
static rgb_t rgb_add(rgb_t a, rgb_t b) { return RGB(a.r + b.r, a.g + b.g, a.b + b.b); }
static rgb_t rgb_sub(rgb_t a, rgb_t b) { return RGB(a.r - b.r, a.g - b.g, a.b - b.b); }
static rgb_t rgb_mul(rgb_t a, double s) { return RGB(a.r * s, a.g * s, a.b * s); }
static rgb_t rgb_mulc(rgb_t a, rgb_t b) { return RGB(a.r * b.r, a.g * b.g, a.b * b.b); }
static rgb_t pat_a(const ctx_t *c, int p) { const double *q = c->pat_ab + 6 * p; return RGB(q[0], q[1], q[2]); }
static rgb_t pat_b(const ctx_t *c, int p) { const double *q = c->pat_ab + 6 * p; return RGB(q[3], q[4], q[5]); }

static rgb_t uv_pattern_at(ctx_t *c, int p, double u, double v) {
    const orc_scene *s = c->s;
    switch (s->pat_type[p]) {
    case ORC_UV_CHECKER: { /* shapes.py:1939-1942 */
        double u2 = floor(u * s->pat_param[4 * p]), v2 = floor(v * s->pat_param[4 * p + 1]);
        return pymod(u2 + v2, 2) == 0 ? pat_a(c, p) : pat_b(c, p);
    }
    case ORC_UV_ALIGN: { /* 1954-1967 */
        const double *q = s->pat_colors + 15 * p;
        int k = 0;
        if (v > 0.8) { if (u < 0.2) k = 1; else if (u > 0.8) k = 2; }
        else if (v < 0.2) { if (u < 0.2) k = 3; else if (u > 0.8) k = 4; }
        return RGB(q[3 * k], q[3 * k + 1], q[3 * k + 2]);
    }
    case ORC_UV_IMAGE: { /* 1975-1979 + canvas.py:49-55; np.round = half-to-even = rint */
        int tx = s->pat_tex[p];
        double vv = 1 - v;
        double y = u * (s->tex_w[tx] - 1), x = vv * (s->tex_h[tx] - 1);
        long row = (long)rint(x), col = (long)rint(y);
        /* numpy negative indices wrap once; anything else is an IndexError in the reference */
        if (row < 0) row += s->tex_h[tx];
        if (col < 0) col += s->tex_w[tx];
        if (row < 0 || row >= s->tex_h[tx] || col < 0 || col >= s->tex_w[tx]) return RGB(NAN, NAN, NAN);
        const uint8_t *px = s->tex_data + s->tex_off[tx] + 3 * ((size_t)row * s->tex_w[tx] + col);
        double sc = s->tex_scale[tx];
        return RGB(px[0] / sc, px[1] / sc, px[2] / sc);
    }
    }
    return RGB(NAN, NAN, NAN);
}

static rgb_t pattern_at_shape(ctx_t *c, int p, int node, v3 world_pt);

/* pattern_at for pattern-space point `pt` */
static rgb_t pattern_at(ctx_t *c, int p, v3 pt) {
    const orc_scene *s = c->s;
    const int32_t *ch = s->pat_child + 6 * p;
    switch (s->pat_type[p]) {
    case ORC_PAT_TEST: return RGB(pt.x, pt.y, pt.z); /* shapes.py:1614-1615 */
    case ORC_PAT_CHECKERS: { /* 1991-1994 */
        double sum = floor(pt.x) + floor(pt.y) + floor(pt.z);
        return isclose(pymod(sum, 2), 0) ? pat_a(c, p) : pat_b(c, p);
    }
    case ORC_PAT_GRADIENT: { /* 2001-2004 */
        rgb_t distance = rgb_sub(pat_b(c, p), pat_a(c, p));
        double fraction = pt.x - floor(pt.x);
        return rgb_add(pat_a(c, p), rgb_mul(distance, fraction));
    }
    case ORC_PAT_RADIAL: { /* 2011-2016 */
        rgb_t distance = rgb_sub(pat_b(c, p), pat_a(c, p));
        double mag = sqrt(pt.x * pt.x + pt.z * pt.z);
        double fraction = mag - floor(mag);
        return rgb_add(pat_a(c, p), rgb_mul(distance, fraction));
    }
    case ORC_PAT_RING: /* 2023-2025 */
        return pymod(floor(sqrt(pt.x * pt.x + pt.z * pt.z)), 2) == 0 ? pat_a(c, p) : pat_b(c, p);
    case ORC_PAT_STRIPE: /* 2032-2033 */
        return pymod(floor(pt.x), 2) == 0 ? pat_a(c, p) : pat_b(c, p);
    case ORC_PAT_MAP_PLANAR: /* 1747-1749, 1838-1839 */
        return uv_pattern_at(c, ch[0], pymod(pt.x, 1), pymod(pt.z, 1));
    case ORC_PAT_MAP_SPHERICAL: { /* 1690-1697 */
        double theta = atan2(pt.x, pt.z);
        double radius = magnitude(pt);
        double phi = acos(pt.y / radius);
        double raw_u = theta / (2 * M_PI);
        return uv_pattern_at(c, ch[0], 1 - (raw_u + 0.5), 1 - phi / M_PI);
    }
    case ORC_PAT_MAP_CYLINDRICAL: { /* 1705-1741, 1859-1861: body and cap share one formula */
        double radius = sqrt(pt.x * pt.x + pt.z * pt.z), abs_y = fabs(pt.y);
        double coord = radius; if (abs_y > coord) coord = abs_y; /* max(radius, abs_y) */
        int face = (coord == radius) ? 0 : 1;
        double theta = atan2(pt.z, pt.x);
        double raw_u = theta / (2 * M_PI);
        return uv_pattern_at(c, ch[face], 1 - (raw_u + 0.5), pymod(pt.y, 1));
    }
    case ORC_PAT_MAP_CUBE: { /* 1772-1824, 1848-1850 */
        double coord = pymax3(fabs(pt.x), fabs(pt.y), fabs(pt.z));
        int face; double u, v;
        if (coord == pt.x) face = 0; else if (coord == -pt.x) face = 1;
        else if (coord == pt.y) face = 2; else if (coord == -pt.y) face = 3;
        else if (coord == pt.z) face = 4; else face = 5;
        switch (face) {
        case 0: u = pymod(1 - pt.z, 2.0) / 2.0; v = pymod(pt.y + 1, 2.0) / 2.0; break; /* right */
        case 1: u = pymod(pt.z + 1, 2.0) / 2.0; v = pymod(pt.y + 1, 2.0) / 2.0; break; /* left */
        case 2: u = pymod(pt.x + 1, 2.0) / 2.0; v = pymod(1 - pt.z, 2.0) / 2.0; break; /* up */
        case 3: u = pymod(pt.x + 1, 2.0) / 2.0; v = pymod(pt.z + 1, 2.0) / 2.0; break; /* down */
        case 4: u = pymod(pt.x + 1, 2.0) / 2.0; v = pymod(pt.y + 1, 2.0) / 2.0; break; /* front */
        default: u = pymod(1 - pt.x, 2.0) / 2.0; v = pymod(pt.y + 1, 2.0) / 2.0; break; /* back */
        }
        return uv_pattern_at(c, ch[face], u, v);
    }
    }
    return RGB(NAN, NAN, NAN);
}

/* Pattern.pattern_at_shape (shapes.py:1664-1666) and its three overrides (1874-1877, 1892-1895) */
static rgb_t pattern_at_shape(ctx_t *c, int p, int node, v3 world_pt) {
    const orc_scene *s = c->s;
    const int32_t *ch = s->pat_child + 6 * p;
    switch (s->pat_type[p]) {
    case ORC_PAT_BLENDED: {
        rgb_t p1 = pattern_at_shape(c, ch[0], node, world_pt);
        rgb_t p2 = pattern_at_shape(c, ch[1], node, world_pt);
        return rgb_mul(rgb_add(p1, p2), 0.5); /* (p1 + p2) / 2 */
    }
    case ORC_PAT_NESTED: {
        rgb_t a = pattern_at_shape(c, ch[1], node, world_pt);
        double *q = c->pat_ab + 6 * ch[0];
        q[0] = a.r; q[1] = a.g; q[2] = a.b;
        rgb_t b = pattern_at_shape(c, ch[2], node, world_pt);
        q[3] = b.r; q[4] = b.g; q[5] = b.b;
        return pattern_at_shape(c, ch[0], node, world_pt);
    }
    case ORC_PAT_PERTURBED: { /* shapes.py:1911-1925: the WORLD point is moved, then handed to p1.pattern_at_shape */
        const double *q = s->pat_param + 4 * p;
        double freq = q[0], scale = q[1];
        int octaves = (int)q[2];
        double x = world_pt.x / freq, y = world_pt.y / freq, z = world_pt.z / freq;
        double new_x = world_pt.x + orc_pnoise3(x, y, z, octaves) * scale;
        z += 1.0;
        double new_y = world_pt.y + orc_pnoise3(x, y, z, octaves) * scale;
        z += 1.0;
        double new_z = world_pt.z + orc_pnoise3(x, y, z, octaves) * scale;
        return pattern_at_shape(c, ch[0], node, V(new_x, new_y, new_z));
    }
    default: {
        v3 object_point = world_to_object(s, node, world_pt);
        v3 pattern_point = mpoint(s->pat_inv + 16 * (size_t)p, object_point);
        return pattern_at(c, p, pattern_point);
    }
    }
}

/* ---------------------------------------------------------------- shading */
static rgb_t color_at(ctx_t *c, ray_t r, int remaining);

/* renderer.py:552-560 */
static int is_shadowed(ctx_t *c, v3 light_position, v3 point) {
    const orc_scene *s = c->s;
    v3 v = sub(light_position, point);
    double distance = magnitude(v);
    ray_t r = {point, normalize(v)};
    c->rays[1]++;
    xs_t xs; xs_init(&xs);
    intersect_world(s, r, &xs);
    int shadowed = 0;
    for (int i = 0; i < xs.n; i++) { /* filter casts_shadow, then hit() */
        if (!s->mat_shadow[s->material[xs.h[i].node]]) continue;
        if (xs.h[i].t > 0) { shadowed = xs.h[i].t < distance; break; }
    }
    xs_free(&xs);
    return shadowed;
}

/* AreaLight._point_on_light with jitter off (renderer.py:947-955) */
static v3 point_on_light(const orc_scene *s, int l, int u, int v) {
    const double *g = s->light_geom + 9 * l;
    v3 corner = V(g[0], g[1], g[2]), uvec = V(g[3], g[4], g[5]), vvec = V(g[6], g[7], g[8]);
    return add(add(corner, mul(uvec, u + 0.5)), mul(vvec, v + 0.5));
}
/* renderer.py:919-922, 964-973 */
static double intensity_at(ctx_t *c, int l, v3 point) {
    const orc_scene *s = c->s;
    if (s->light_type[l] == 0) {
        const double *g = s->light_geom + 9 * l;
        return is_shadowed(c, V(g[0], g[1], g[2]), point) ? 0.0 : 1.0;
    }
    int usteps = s->light_steps[3 * l], vsteps = s->light_steps[3 * l + 1];
    double total = 0.0;
    for (int v = 0; v < vsteps; v++)
        for (int u = 0; u < usteps; u++)
            if (!is_shadowed(c, point_on_light(s, l, u, v), point)) total += 1.0;
    return total / (usteps * vsteps);
}

/* renderer.py:1066-1090 */
static rgb_t lighting(ctx_t *c, int material, int node, int l, v3 point, v3 eyev, v3 normalv, double shade_intensity) {
    const orc_scene *s = c->s;
    const double *m = s->mat + 10 * material;
    rgb_t pcolor = s->mat_pattern[material] >= 0 ? pattern_at_shape(c, s->mat_pattern[material], node, point)
                                                 : RGB(m[0], m[1], m[2]);
    const double *li = s->light_intensity + 3 * l;
    rgb_t light_intensity = RGB(li[0], li[1], li[2]);
    rgb_t effective_color = rgb_mulc(pcolor, light_intensity);
    rgb_t ambient = rgb_mul(effective_color, m[3]);
    if (shade_intensity == 0.0) return ambient;

    int usteps = 1, vsteps = 1;
    if (s->light_type[l] == 1) { usteps = s->light_steps[3 * l]; vsteps = s->light_steps[3 * l + 1]; }
    int samples = usteps * vsteps;
    rgb_t acc = RGB(0, 0, 0);
    for (int v = 0; v < vsteps; v++)
        for (int u = 0; u < usteps; u++) {
            v3 position;
            if (s->light_type[l] == 0) { const double *g = s->light_geom + 9 * l; position = V(g[0], g[1], g[2]); }
            else position = point_on_light(s, l, u, v);
            v3 lightv = normalize(sub(position, point));
            double light_dot_normal = dot(lightv, normalv);
            if (light_dot_normal >= 0) {
                acc = rgb_add(acc, rgb_mul(rgb_mul(effective_color, m[4]), light_dot_normal));
                v3 reflectv = reflect(neg(lightv), normalv);
                double reflect_dot_eye = dot(reflectv, eyev);
                if (reflect_dot_eye > 0) {
                    double factor = pow(reflect_dot_eye, m[6]);
                    acc = rgb_add(acc, rgb_mul(rgb_mul(light_intensity, m[5]), factor));
                }
            }
        }
    acc = RGB(acc.r / samples, acc.g / samples, acc.b / samples); /* acc / light.samples */
    return rgb_add(ambient, rgb_mul(acc, shade_intensity));
}

typedef struct {
    double t; int node; v3 point, eyev, normalv, reflectv, over_point, under_point;
    int inside; double n1, n2;
} comps_t;

/* renderer.py:206-244 */
static comps_t prepare_computations(const orc_scene *s, const xs_t *xs, int idx, ray_t r) {
    comps_t c;
    const isect *i = &xs->h[idx];
    c.t = i->t;
    c.node = i->node;
    c.point = add(r.o, mul(r.d, i->t)); /* shapes.py:1266 */
    c.normalv = normal_at(s, i->node, c.point, i->u, i->v);
    c.eyev = neg(r.d);
    c.reflectv = reflect(r.d, c.normalv); /* computed BEFORE the inside flip */
    c.inside = 0;
    if (dot(c.normalv, c.eyev) < 0) { c.inside = 1; c.normalv = neg(c.normalv); }
    c.over_point = add(c.point, mul(c.normalv, EPSILON));
    c.under_point = sub(c.point, mul(c.normalv, EPSILON));
    c.n1 = c.n2 = 1.0;

    int *containers = (int *)malloc(sizeof(int) * (size_t)(xs->n + 1));
    int nc = 0;
    for (int k = 0; k < xs->n; k++) {
        if (k == idx)
            c.n1 = nc == 0 ? 1.0 : s->mat[10 * s->material[containers[nc - 1]] + 9];
        int obj = xs->h[k].node, found = -1;
        for (int q = 0; q < nc; q++) if (containers[q] == obj) { found = q; break; } /* list.remove: first match */
        if (found >= 0) { memmove(containers + found, containers + found + 1, sizeof(int) * (size_t)(nc - found - 1)); nc--; }
        else containers[nc++] = obj;
        if (k == idx) {
            c.n2 = nc == 0 ? 1.0 : s->mat[10 * s->material[containers[nc - 1]] + 9];
            break;
        }
    }
    free(containers);
    return c;
}

/* renderer.py:272-283 */
static double schlick(const comps_t *c) {
    double cosv = dot(c->eyev, c->normalv);
    if (c->n1 > c->n2) {
        double n = c->n1 / c->n2;
        double sin2_t = n * n * (1.0 - cosv * cosv);
        if (sin2_t > 1.0) return 1.0;
        cosv = sqrt(1.0 - sin2_t);
    }
    double r0 = (c->n1 - c->n2) / (c->n1 + c->n2);
    r0 = r0 * r0;
    double m = 1.0 - cosv;
    return r0 + (1.0 - r0) * (m * m * m * m * m);
}

/* renderer.py:506-526 with reflected_color (400-406) and refracted_color (336-353) inline */
static rgb_t shade_hit(ctx_t *c, const comps_t *cp, int remaining) {
    const orc_scene *s = c->s;
    int material = s->material[cp->node];
    const double *m = s->mat + 10 * material;
    rgb_t surface = RGB(0, 0, 0);
    for (int l = 0; l < s->n_lights; l++) {
        double intensity = intensity_at(c, l, cp->over_point);
        surface = rgb_add(surface, lighting(c, material, cp->node, l, cp->over_point, cp->eyev, cp->normalv, intensity));
    }
    rgb_t reflected = RGB(0, 0, 0), refracted = RGB(0, 0, 0);
    double reflective = m[7], transparency = m[8];
    if (!(remaining == 0 || reflective == 0)) {
        ray_t rr = {cp->over_point, cp->reflectv};
        c->rays[2]++;
        reflected = rgb_mul(color_at(c, rr, remaining - 1), reflective);
    }
    if (!(transparency == 0 || remaining == 0)) {
        double n_ratio = cp->n1 / cp->n2;
        double cos_i = dot(cp->eyev, cp->normalv);
        double sin2_t = n_ratio * n_ratio * (1 - cos_i * cos_i);
        if (!(sin2_t > 1.0)) {
            double cos_t = sqrt(1.0 - sin2_t);
            v3 direction = sub(mul(cp->normalv, n_ratio * cos_i - cos_t), mul(cp->eyev, n_ratio));
            ray_t rr = {cp->under_point, direction};
            c->rays[3]++;
            refracted = rgb_mul(color_at(c, rr, remaining - 1), transparency);
        }
    }
    if (reflective > 0 && transparency > 0) {
        double reflectance = schlick(cp);
        surface = rgb_add(surface, rgb_add(rgb_mul(reflected, reflectance), rgb_mul(refracted, 1.0 - reflectance)));
    } else {
        surface = rgb_add(surface, rgb_add(reflected, refracted));
    }
    return surface;
}

/* renderer.py:606-611 */
static rgb_t color_at_ex(ctx_t *c, ray_t r, int remaining, int *hit_node, double *hit_tt) {
    xs_t xs; xs_init(&xs);
    intersect_world(c->s, r, &xs);
    int i = hit(&xs);
    rgb_t out = RGB(0, 0, 0);
    if (hit_node) *hit_node = -1;
    if (i >= 0) {
        if (hit_node) { *hit_node = xs.h[i].node; *hit_tt = xs.h[i].t; }
        comps_t cp = prepare_computations(c->s, &xs, i, r);
        out = shade_hit(c, &cp, remaining);
    }
    xs_free(&xs);
    return out;
}
static rgb_t color_at(ctx_t *c, ray_t r, int remaining) { return color_at_ex(c, r, remaining, NULL, NULL); }

/* renderer.py:742-751 */
static ray_t ray_for_pixel(const orc_camera *cam, int px, int py) {
    double xoffset = (px + 0.5) * cam->pixel_size, yoffset = (py + 0.5) * cam->pixel_size;
    double world_x = cam->half_width - xoffset, world_y = cam->half_height - yoffset;
    v3 pixel = mpoint(cam->inv_transform, V(world_x, world_y, -1));
    v3 origin = mpoint(cam->inv_transform, V(0, 0, 0));
    ray_t r = {origin, normalize(sub(pixel, origin))};
    return r;
}

/* ---------------------------------------------------------------- public entry points */
static void ctx_init(ctx_t *c, const orc_scene *s) {
    c->s = s;
    c->pat_ab = (double *)malloc(sizeof(double) * 6 * (size_t)(s->n_patterns > 0 ? s->n_patterns : 1));
    if (s->n_patterns > 0) memcpy(c->pat_ab, s->pat_ab, sizeof(double) * 6 * (size_t)s->n_patterns);
    memset(c->rays, 0, sizeof(c->rays));
}
static void ctx_free(ctx_t *c) { free(c->pat_ab); }
static ray_t mkray(const double *o, const double *d) { ray_t r = {V(o[0], o[1], o[2]), V(d[0], d[1], d[2])}; return r; }

int orc_render_pixels(const orc_scene *s, const orc_camera *cam, int64_t n_pix, const int32_t *px,
                      const int32_t *py, int remaining, int n_threads, double *rgb,
                      int32_t *hit_leaf, double *hit_t, int64_t *rays) {
    for (int l = 0; l < s->n_lights; l++)
        if (s->light_type[l] == 1 && s->light_steps[3 * l + 2]) return -1; /* jitter: not reproducible in the reference itself */
    int64_t total[4] = {0, 0, 0, 0};
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
#endif
    {
        ctx_t c; ctx_init(&c, s);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t i = 0; i < n_pix; i++) {
            ray_t r = ray_for_pixel(cam, px[i], py[i]);
            c.rays[0]++;
            int node; double tt = 0;
            rgb_t col = color_at_ex(&c, r, remaining, &node, &tt);
            rgb[3 * i] = col.r; rgb[3 * i + 1] = col.g; rgb[3 * i + 2] = col.b;
            if (hit_leaf) hit_leaf[i] = node >= 0 ? s->leaf_id[node] : -1;
            if (hit_t) hit_t[i] = node >= 0 ? tt : INFINITY;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int k = 0; k < 4; k++) total[k] += c.rays[k];
        ctx_free(&c);
    }
    if (rays) for (int k = 0; k < 4; k++) rays[k] = total[k];
    return 0;
}

void orc_ray_for_pixel(const orc_camera *c, int px, int py, double *origin3, double *dir3) {
    ray_t r = ray_for_pixel(c, px, py);
    origin3[0] = r.o.x; origin3[1] = r.o.y; origin3[2] = r.o.z;
    dir3[0] = r.d.x; dir3[1] = r.d.y; dir3[2] = r.d.z;
}
static int xs_export(xs_t *xs, int cap, double *t_out, int32_t *node_out) {
    int n = xs->n;
    for (int i = 0; i < n && i < cap; i++) { t_out[i] = xs->h[i].t; node_out[i] = xs->h[i].node; }
    xs_free(xs);
    return n;
}
int orc_intersect_world(const orc_scene *s, const double *o, const double *d, int cap, double *t_out, int32_t *node_out) {
    xs_t xs; xs_init(&xs);
    intersect_world(s, mkray(o, d), &xs);
    return xs_export(&xs, cap, t_out, node_out);
}
int orc_intersect_node(const orc_scene *s, int node, const double *o, const double *d, int cap, double *t_out, int32_t *node_out) {
    xs_t xs; xs_init(&xs);
    intersect_node(s, node, mkray(o, d), &xs);
    return xs_export(&xs, cap, t_out, node_out);
}
void orc_color_at(const orc_scene *s, const double *o, const double *d, int remaining, double *rgb) {
    ctx_t c; ctx_init(&c, s);
    rgb_t col = color_at(&c, mkray(o, d), remaining);
    rgb[0] = col.r; rgb[1] = col.g; rgb[2] = col.b;
    ctx_free(&c);
}
void orc_normal_at(const orc_scene *s, int node, const double *p, double u, double v, double *n3) {
    v3 n = normal_at(s, node, V(p[0], p[1], p[2]), u, v);
    n3[0] = n.x; n3[1] = n.y; n3[2] = n.z;
}
void orc_pattern_at_shape(const orc_scene *s, int pattern, int node, const double *p, double *rgb) {
    ctx_t c; ctx_init(&c, s);
    rgb_t col = pattern_at_shape(&c, pattern, node, V(p[0], p[1], p[2]));
    rgb[0] = col.r; rgb[1] = col.g; rgb[2] = col.b;
    ctx_free(&c);
}
void orc_lighting(const orc_scene *s, int material, int node, int light, const double *p, const double *eye,
                  const double *n, double intensity, double *rgb) {
    ctx_t c; ctx_init(&c, s);
    rgb_t col = lighting(&c, material, node, light, V(p[0], p[1], p[2]), V(eye[0], eye[1], eye[2]),
                         V(n[0], n[1], n[2]), intensity);
    rgb[0] = col.r; rgb[1] = col.g; rgb[2] = col.b;
    ctx_free(&c);
}
int orc_is_shadowed(const orc_scene *s, const double *lp, const double *p) {
    ctx_t c; ctx_init(&c, s);
    int r = is_shadowed(&c, V(lp[0], lp[1], lp[2]), V(p[0], p[1], p[2]));
    ctx_free(&c);
    return r;
}
int orc_prepare_computations(const orc_scene *s, const double *o, const double *d, int k, double *out) {
    ray_t r = mkray(o, d);
    xs_t xs; xs_init(&xs);
    intersect_world(s, r, &xs);
    if (k < 0 || k >= xs.n) { xs_free(&xs); return -1; }
    comps_t c = prepare_computations(s, &xs, k, r);
    double *q = out;
    *q++ = c.t;
    *q++ = c.point.x; *q++ = c.point.y; *q++ = c.point.z;
    *q++ = c.eyev.x; *q++ = c.eyev.y; *q++ = c.eyev.z;
    *q++ = c.normalv.x; *q++ = c.normalv.y; *q++ = c.normalv.z;
    *q++ = c.inside;
    *q++ = c.reflectv.x; *q++ = c.reflectv.y; *q++ = c.reflectv.z;
    *q++ = c.over_point.x; *q++ = c.over_point.y; *q++ = c.over_point.z;
    *q++ = c.under_point.x; *q++ = c.under_point.y; *q++ = c.under_point.z;
    *q++ = c.n1; *q++ = c.n2;
    *q++ = schlick(&c);
    int node = c.node;
    xs_free(&xs);
    return node;
}
