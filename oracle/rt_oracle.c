/* rt_oracle.c — CPU oracle (f64), a restatement of the reference's ray-casting path.
 * TEST INFRASTRUCTURE ONLY — see rt_oracle.h for the rules and for what pins it.
 * Every function names the reference lines it follows (paths relative to rust/src/).
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define EPS RTMI_EPSILON

/* ------------------------------------------------------------------------------------------- */
/* tuples (tuple.rs) — w is implicit: points have w=1, vectors w=0                              */
/* ------------------------------------------------------------------------------------------- */
typedef struct { double x, y, z; } v3;

static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vmul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* tuple.rs:122 */
static v3 vcross(v3 a, v3 b) {                                               /* tuple.rs:126 */
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static double vlen(v3 a) { return sqrt(vdot(a, a)); } /* tuple.rs:106 */
/* tuple.rs:118 -> vecmath::vec4_normalized = scale by the inverse length */
static v3 vnorm(v3 a) { return vmul(a, 1.0 / vlen(a)); }
/* tuple.rs:131 */
static v3 vreflect(v3 a, v3 n) { return vsub(a, vmul(n, 2.0 * vdot(a, n))); }
static v3 from3(const double *p) { return V(p[0], p[1], p[2]); }
static void to3(v3 a, double *p) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* Matrix * Point / Matrix * Vector on a flat row-major 4x4 (matrix.rs:311-345) */
static v3 mat_point(const double *m, v3 p) {
    return V(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
             m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
static v3 mat_vector(const double *m, v3 p) {
    return V(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
             m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
/* transpose(m) * vector, xyz only (shapes/mod.rs:319-322) */
static v3 mat_t_vector(const double *m, v3 p) {
    return V(m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
             m[2] * p.x + m[6] * p.y + m[10] * p.z);
}

typedef struct { v3 o, d; } ray_t;                 /* ray.rs:7-11 */
static v3 ray_position(ray_t r, double t) { return vadd(r.o, vmul(r.d, t)); } /* ray.rs:26-28 */
static ray_t ray_transform(ray_t r, const double *m) {                       /* ray.rs:30-35 */
    ray_t q = {mat_point(m, r.o), mat_vector(m, r.d)};
    return q;
}

/* ------------------------------------------------------------------------------------------- */
/* counter-based RNG: Philox4x32-10                                                             */
/* ------------------------------------------------------------------------------------------- */
void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uniform k of (pixel, sample): 24 bits so that f32 and f64 consumers see the same number */
double orc_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k) {
    uint32_t o[4];
    orc_philox(seed, pixel, sample, k >> 2, 0x52544D49u /* "RTMI" */, o);
    return (double)(o[k & 3] >> 8) * (1.0 / 16777216.0);
}

typedef struct {
    uint64_t seed;
    uint32_t pixel, sample, draw;
} rng_t;

static double rng_next(rng_t *g) { return orc_uniform(g->seed, g->pixel, g->sample, g->draw++); }

/* ------------------------------------------------------------------------------------------- */
/* intersections (ray.rs:38-70)                                                                 */
/* ------------------------------------------------------------------------------------------- */
typedef struct { double t; int leaf; double u, v; } isect;
typedef struct { isect *xs; int n, cap; } xlist;

static void xl_init(xlist *l) { l->xs = NULL; l->n = 0; l->cap = 0; }
static void xl_free(xlist *l) { free(l->xs); xl_init(l); }
static void xl_push(xlist *l, double t, int leaf, double u, double v) {
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 8;
        l->xs = (isect *)realloc(l->xs, (size_t)l->cap * sizeof(isect));
    }
    isect i = {t, leaf, u, v};
    l->xs[l->n++] = i;
}
static void xl_extend(xlist *l, const xlist *o) {
    for (int i = 0; i < o->n; ++i) xl_push(l, o->xs[i].t, o->xs[i].leaf, o->xs[i].u, o->xs[i].v);
}
/* sort_intersections (ray.rs:225-231): the reference's sort_unstable_by leaves the order of equal
 * t unspecified; this restatement uses a stable sort so ties keep insertion order. */
static void xl_sort(xlist *l) {
    if (l->n < 2) return;
    if (l->n <= 24) {
        for (int i = 1; i < l->n; ++i) {
            isect k = l->xs[i];
            int j = i - 1;
            while (j >= 0 && l->xs[j].t > k.t) { l->xs[j + 1] = l->xs[j]; --j; }
            l->xs[j + 1] = k;
        }
        return;
    }
    /* bottom-up merge sort */
    isect *tmp = (isect *)malloc((size_t)l->n * sizeof(isect));
    isect *a = l->xs, *b = tmp;
    for (int w = 1; w < l->n; w *= 2) {
        for (int lo = 0; lo < l->n; lo += 2 * w) {
            int mid = lo + w < l->n ? lo + w : l->n, hi = lo + 2 * w < l->n ? lo + 2 * w : l->n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) b[k++] = (a[j].t < a[i].t) ? a[j++] : a[i++];
            while (i < mid) b[k++] = a[i++];
            while (j < hi) b[k++] = a[j++];
        }
        isect *s = a; a = b; b = s;
    }
    if (a != l->xs) memcpy(l->xs, a, (size_t)l->n * sizeof(isect));
    free(tmp);
}

/* PhotonMap (photon_map.rs:224-227): the photons in kd-tree order (node i has children 2i+1, 2i+2), the flag
 * PhotonMap::balance gave each (photon_map.rs:215-221) and where each came from in the caller's array */
enum { KD_LEAF = 1, KD_SPLIT_X = 2, KD_SPLIT_Y = 3, KD_SPLIT_Z = 4 };
typedef struct {
    int64_t n;
    rtmi_photon *photons;
    uint8_t *flag;
    int64_t *orig;
    double max_search_radius;
} pmap_t;

/* per-thread context: scene, flags, counters, RNG */
typedef struct {
    const rtmi_scene_desc *sc;
    int flags;
    uint64_t stat[RTMI_N_STATS];
    rng_t rng;
    const pmap_t *pm; /* built at the first shade_hit that needs it, or handed in by orc_render */
    int pm_owned;
} ctx_t;

/* ------------------------------------------------------------------------------------------- */
/* Geometry::intersect per kind, object space                                                   */
/* ------------------------------------------------------------------------------------------- */
static double sq(double x) { return x * x; }

/* sphere.rs:47-63 */
static int sphere_intersect(ray_t r, double *ts) {
    v3 sphere_to_ray = r.o;
    double a = vdot(r.d, r.d);
    double b = 2.0 * vdot(r.d, sphere_to_ray);
    double c = vdot(sphere_to_ray, sphere_to_ray) - 1.0;
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return 0;
    double t0 = -(b + sqrt(discriminant)) / (2.0 * a);
    double t1 = -(b - sqrt(discriminant)) / (2.0 * a);
    if (t1 < t0) { double s = t0; t0 = t1; t1 = s; } /* intersections![] sorts */
    ts[0] = t0; ts[1] = t1;
    return 2;
}

/* plane.rs:35-44 */
static int plane_intersect(ray_t r, double *ts) {
    if (fabs(r.d.y) < EPS) return 0;
    ts[0] = -r.o.y / r.d.y;
    return 1;
}

/* cube.rs:74-85 */
static int cube_check_axis(double origin, double direction, double *tmin, double *tmax) {
    if (direction == 0.0 && (origin < -1.0 || origin > 1.0)) return 0;
    double t_min = (-1.0 - origin) / direction;
    double t_max = (1.0 - origin) / direction;
    if (t_min > t_max) { *tmin = t_max; *tmax = t_min; } else { *tmin = t_min; *tmax = t_max; }
    return 1;
}
/* cube.rs:33-46 */
static int cube_intersect(ray_t r, double *ts) {
    double xmin, xmax, ymin, ymax, zmin, zmax;
    if (!cube_check_axis(r.o.x, r.d.x, &xmin, &xmax)) return 0;
    if (!cube_check_axis(r.o.y, r.d.y, &ymin, &ymax)) return 0;
    double tmin = fmax(xmin, ymin), tmax = fmin(xmax, ymax);
    if (!cube_check_axis(r.o.z, r.d.z, &zmin, &zmax)) return 0;
    tmin = fmax(tmin, zmin); tmax = fmin(tmax, zmax);
    if (!(tmin <= tmax)) return 0;
    ts[0] = tmin; ts[1] = tmax;
    return 2;
}

static void sort_ts(double *ts, int n) {
    for (int i = 1; i < n; ++i) {
        double k = ts[i];
        int j = i - 1;
        while (j >= 0 && ts[j] > k) { ts[j + 1] = ts[j]; --j; }
        ts[j + 1] = k;
    }
}

/* cylinder.rs:61-64 / cone.rs:61-64 */
static int check_cap(ray_t r, double t, double radius) {
    v3 p = ray_position(r, t);
    return sq(p.x) + sq(p.z) <= sq(radius);
}
/* cylinder.rs:36-58 (radius 1 for both caps) / cone.rs:36-58 (radius = |y| of the cap) */
static int intersect_caps(const rtmi_leaf *g, ray_t r, int is_cone, double *ts, int n) {
    if (g->closed && fabs(r.d.y) > EPS) {
        double t_min = (g->minimum - r.o.y) / r.d.y;
        if (check_cap(r, t_min, is_cone ? g->minimum : 1.0)) ts[n++] = t_min;
        double t_max = (g->maximum - r.o.y) / r.d.y;
        if (check_cap(r, t_max, is_cone ? g->maximum : 1.0)) ts[n++] = t_max;
    }
    sort_ts(ts, n);
    return n;
}
/* cylinder.rs:79-109 */
static int cylinder_intersect(const rtmi_leaf *g, ray_t r, double *ts) {
    double a = sq(r.d.x) + sq(r.d.z);
    double b = 2.0 * r.o.x * r.d.x + 2.0 * r.o.z * r.d.z;
    double c = sq(r.o.x) + sq(r.o.z) - 1.0;
    if (fabs(a) < EPS) return intersect_caps(g, r, 0, ts, 0);
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return 0;
    int n = 0;
    double t0 = (b + sqrt(discriminant)) / (-2.0 * a);
    double y0 = r.o.y + t0 * r.d.y;
    if (y0 > g->minimum && y0 < g->maximum) ts[n++] = t0;
    double t1 = (b - sqrt(discriminant)) / (-2.0 * a);
    double y1 = r.o.y + t1 * r.d.y;
    if (y1 > g->minimum && y1 < g->maximum) ts[n++] = t1;
    return intersect_caps(g, r, 0, ts, n);
}
/* cone.rs:79-120 */
static int cone_intersect(const rtmi_leaf *g, ray_t r, double *ts) {
    double a = sq(r.d.x) - sq(r.d.y) + sq(r.d.z);
    double b = 2.0 * (r.o.x * r.d.x - r.o.y * r.d.y + r.o.z * r.d.z);
    double c = sq(r.o.x) - sq(r.o.y) + sq(r.o.z);
    if (fabs(a) < EPS) {
        if (fabs(b) < EPS) return intersect_caps(g, r, 1, ts, 0);
        ts[0] = c / (-2.0 * b);
        return intersect_caps(g, r, 1, ts, 1);
    }
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return 0;
    int n = 0;
    double t0 = (b + sqrt(discriminant)) / (-2.0 * a);
    double y0 = r.o.y + t0 * r.d.y;
    if (y0 > g->minimum && y0 < g->maximum) ts[n++] = t0;
    double t1 = (b - sqrt(discriminant)) / (-2.0 * a);
    double y1 = r.o.y + t1 * r.d.y;
    if (y1 > g->minimum && y1 < g->maximum) ts[n++] = t1;
    return intersect_caps(g, r, 1, ts, n);
}
/* triangle.rs:233-265 (Moeller-Trumbore; no sign test on t) */
static int triangle_intersect(const rtmi_leaf *g, ray_t r, double *ts, double *uv) {
    v3 p1 = from3(g->p1), e1 = vsub(from3(g->p2), p1), e2 = vsub(from3(g->p3), p1);
    v3 dir_cross_e2 = vcross(r.d, e2);
    double det = vdot(e1, dir_cross_e2);
    if (fabs(det) < EPS) return 0;
    double f = 1.0 / det;
    v3 p1_to_origin = vsub(r.o, p1);
    double u = f * vdot(p1_to_origin, dir_cross_e2);
    if (u < 0.0 || u > 1.0) return 0;
    v3 origin_cross_e1 = vcross(p1_to_origin, e1);
    double v = f * vdot(r.d, origin_cross_e1);
    if (v < 0.0 || (u + v) > 1.0) return 0;
    ts[0] = f * vdot(e2, origin_cross_e1);
    if (uv) { uv[0] = u; uv[1] = v; }
    return 1;
}

static int local_intersect(const rtmi_leaf *g, ray_t r, double *ts, double *uv) {
    switch (g->kind) {
    case RTMI_SPHERE: return sphere_intersect(r, ts);
    case RTMI_PLANE: return plane_intersect(r, ts);
    case RTMI_CUBE: return cube_intersect(r, ts);
    case RTMI_CYLINDER: return cylinder_intersect(g, r, ts);
    case RTMI_CONE: return cone_intersect(g, r, ts);
    case RTMI_TRIANGLE:
    case RTMI_SMOOTH_TRIANGLE: return triangle_intersect(g, r, ts, uv);
    default: return 0;
    }
}

int orc_local_intersect(const rtmi_leaf *leaf, const double *origin, const double *dir, double *ts,
                        double *uv) {
    ray_t r = {from3(origin), from3(dir)};
    double luv[2] = {0, 0};
    int n = local_intersect(leaf, r, ts, luv);
    if (uv) { uv[0] = luv[0]; uv[1] = luv[1]; }
    return n;
}

/* Geometry::normal_at per kind, object space */
static v3 local_normal(const rtmi_leaf *g, v3 p, double u, double v) {
    switch (g->kind) {
    case RTMI_SPHERE: return p; /* sphere.rs:65-67 */
    case RTMI_PLANE: return V(0, 1, 0); /* plane.rs:46-48 */
    case RTMI_CUBE: { /* cube.rs:48-67 */
        double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
        if (ax >= ay) return ax >= az ? V(p.x, 0, 0) : V(0, 0, p.z);
        return ay >= az ? V(0, p.y, 0) : V(0, 0, p.z);
    }
    case RTMI_CYLINDER: { /* cylinder.rs:111-123 */
        double dist = sq(p.x) + sq(p.z);
        if (dist < 1.0) {
            if (p.y >= g->maximum - EPS) return V(0, 1, 0);
            if (p.y <= g->minimum + EPS) return V(0, -1, 0);
        }
        return V(p.x, 0, p.z);
    }
    case RTMI_CONE: { /* cone.rs:122-142 */
        double dist = sq(p.x) + sq(p.z);
        if (dist < 1.0) {
            if (p.y >= g->maximum - EPS) return V(0, 1, 0);
            if (p.y <= g->minimum + EPS) return V(0, -1, 0);
        }
        return V(p.x, p.y < 0.0 ? sqrt(dist) : -sqrt(dist), p.z);
    }
    case RTMI_TRIANGLE: { /* triangle.rs:33-45,69-71 */
        v3 p1 = from3(g->p1), e1 = vsub(from3(g->p2), p1), e2 = vsub(from3(g->p3), p1);
        return vnorm(vcross(e2, e1));
    }
    case RTMI_SMOOTH_TRIANGLE: /* triangle.rs:141-143 */
        return vnorm(vadd(vadd(vmul(from3(g->n2), u), vmul(from3(g->n3), v)),
                          vmul(from3(g->n1), 1.0 - u - v)));
    default: return V(0, 0, 0);
    }
}

void orc_local_normal(const rtmi_leaf *leaf, const double *point, double u, double v, double *n) {
    to3(local_normal(leaf, from3(point), u, v), n);
}

/* Shape::normal_at (shapes/mod.rs:261-265) with normal_to_world (319-322) */
static v3 normal_at(const rtmi_leaf *g, v3 world_point, double u, double v) {
    v3 obj_point = mat_point(g->inv, world_point);
    v3 obj_normal = local_normal(g, obj_point, u, v);
    return vnorm(mat_t_vector(g->inv, obj_normal));
}

void orc_normal_at(const rtmi_leaf *leaf, const double *world_point, double u, double v, double *n) {
    to3(normal_at(leaf, from3(world_point), u, v), n);
}

/* ------------------------------------------------------------------------------------------- */
/* Aabb (aabb.rs)                                                                               */
/* ------------------------------------------------------------------------------------------- */
/* aabb.rs:130-141 */
static int aabb_check_axis(double min, double max, double origin, double direction, double *tmin,
                           double *tmax) {
    if (direction == 0.0 && (origin < min || origin > max)) return 0;
    double t_min = (min - origin) / direction;
    double t_max = (max - origin) / direction;
    if (t_min > t_max) { *tmin = t_max; *tmax = t_min; } else { *tmin = t_min; *tmax = t_max; }
    return 1;
}
/* aabb.rs:45-70 */
int orc_aabb_intersect(const double *bmin, const double *bmax, const double *origin,
                       const double *dir, double *tmin_out, double *tmax_out) {
    double x0, x1, y0, y1, z0, z1;
    if (!aabb_check_axis(bmin[0], bmax[0], origin[0], dir[0], &x0, &x1)) return 0;
    if (!aabb_check_axis(bmin[1], bmax[1], origin[1], dir[1], &y0, &y1)) return 0;
    if (!aabb_check_axis(bmin[2], bmax[2], origin[2], dir[2], &z0, &z1)) return 0;
    double tmin = fmax(fmax(x0, y0), z0), tmax = fmin(fmin(x1, y1), z1);
    if (!(tmin <= tmax)) return 0;
    if (tmin_out) *tmin_out = tmin;
    if (tmax_out) *tmax_out = tmax;
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* SceneItem::intersect (shapes/mod.rs:108-115)                                                 */
/* ------------------------------------------------------------------------------------------- */
static int is_csg(int kind) { return kind >= RTMI_NODE_CSG_UNION && kind <= RTMI_NODE_CSG_DIFFERENCE; }

/* Shape::intersect (shapes/mod.rs:252-255) */
static void leaf_intersect(ctx_t *cx, int leaf, ray_t world_ray, xlist *out) {
    const rtmi_leaf *g = &cx->sc->leaves[leaf];
    double ts[4], uv[2] = {0, 0};
    cx->stat[RTMI_STAT_LEAF_TESTS]++;
    int n = local_intersect(g, ray_transform(world_ray, g->inv), ts, uv);
    for (int i = 0; i < n; ++i) xl_push(out, ts[i], leaf, uv[0], uv[1]);
}

/* SceneItem::contains (shapes/mod.rs:171-179, csg.rs:126-128): pointer identity == leaf index */
static int node_contains(const rtmi_scene_desc *sc, int node, int leaf) {
    const rtmi_node *nd = &sc->nodes[node];
    if (nd->kind == RTMI_NODE_LEAF) return nd->first == leaf;
    if (is_csg(nd->kind)) return node_contains(sc, nd->first, leaf) || node_contains(sc, nd->count, leaf);
    for (int i = 0; i < nd->count; ++i)
        if (node_contains(sc, sc->children[nd->first + i], leaf)) return 1;
    return 0;
}

/* CsgOp::eval (csg.rs:26-32) */
int orc_csg_eval(int kind, int lhit, int inl, int inr) {
    switch (kind) {
    case RTMI_NODE_CSG_UNION: return (lhit && !inr) || (!lhit && !inl);
    case RTMI_NODE_CSG_INTERSECTION: return (lhit && inr) || (!lhit && inl);
    case RTMI_NODE_CSG_DIFFERENCE: return (lhit && !inr) || (!lhit && inl);
    default: return 0;
    }
}

static void node_intersect(ctx_t *cx, int node, ray_t ray, xlist *out) {
    const rtmi_scene_desc *sc = cx->sc;
    const rtmi_node *nd = &sc->nodes[node];
    if (nd->kind == RTMI_NODE_LEAF) {
        leaf_intersect(cx, nd->first, ray, out);
        return;
    }
    cx->stat[RTMI_STAT_NODE_VISITS]++;
    if (nd->kind == RTMI_NODE_GROUP || nd->kind == RTMI_NODE_BOUNDED) {
        if (nd->kind == RTMI_NODE_BOUNDED) { /* BoundingGroup::intersect, shapes/mod.rs:471-477 */
            double o[3], d[3];
            to3(ray.o, o); to3(ray.d, d);
            if (!orc_aabb_intersect(nd->aabb_min, nd->aabb_max, o, d, NULL, NULL)) return;
        }
        /* Group::intersect, shapes/mod.rs:399-405 */
        xlist xs; xl_init(&xs);
        for (int i = 0; i < nd->count; ++i) node_intersect(cx, sc->children[nd->first + i], ray, &xs);
        xl_sort(&xs);
        xl_extend(out, &xs);
        xl_free(&xs);
        return;
    }
    /* CsgPair::intersect + filter_intersections, csg.rs:97-124 */
    xlist xs; xl_init(&xs);
    node_intersect(cx, nd->first, ray, &xs);
    node_intersect(cx, nd->count, ray, &xs);
    xl_sort(&xs);
    int inl = 0, inr = 0;
    for (int i = 0; i < xs.n; ++i) {
        int lhit = node_contains(sc, nd->first, xs.xs[i].leaf);
        int filt = orc_csg_eval(nd->kind, lhit, inl, inr);
        if (lhit) inl = !inl; else inr = !inr;
        if (filt) xl_push(out, xs.xs[i].t, xs.xs[i].leaf, xs.xs[i].u, xs.xs[i].v);
    }
    xl_free(&xs);
}

/* World::intersect (world.rs:145-156) / intersect_shadow (158-170) */
static void world_intersect(ctx_t *cx, ray_t ray, int shadow_only, xlist *out) {
    const rtmi_scene_desc *sc = cx->sc;
    for (int i = 0; i < sc->n_top; ++i) {
        if (shadow_only && !sc->top_cast_shadow[i]) continue;
        node_intersect(cx, sc->top_nodes[i], ray, out);
    }
    xl_sort(out);
}

/* hit (ray.rs:143-154): first element with t >= 0 of the sorted list */
static int hit_index(const xlist *xs) {
    for (int i = 0; i < xs->n; ++i)
        if (xs->xs[i].t >= 0.0) return i;
    return -1;
}

static void ctx_init(ctx_t *cx, const rtmi_scene_desc *sc, int flags) {
    memset(cx, 0, sizeof *cx);
    cx->sc = sc;
    cx->flags = flags;
}

int orc_world_intersect(const rtmi_scene_desc *scene, const double *origin, const double *dir,
                        int shadow_only, int cap, double *ts, int32_t *leaves) {
    ctx_t cx; ctx_init(&cx, scene, 0);
    ray_t r = {from3(origin), from3(dir)};
    xlist xs; xl_init(&xs);
    world_intersect(&cx, r, shadow_only, &xs);
    for (int i = 0; i < xs.n && i < cap; ++i) {
        if (ts) ts[i] = xs.xs[i].t;
        if (leaves) leaves[i] = xs.xs[i].leaf;
    }
    int n = xs.n;
    xl_free(&xs);
    return n;
}

int orc_node_intersect(const rtmi_scene_desc *scene, int node, const double *origin,
                       const double *dir, int cap, double *ts, int32_t *leaves) {
    ctx_t cx; ctx_init(&cx, scene, 0);
    ray_t r = {from3(origin), from3(dir)};
    xlist xs; xl_init(&xs);
    node_intersect(&cx, node, r, &xs);
    for (int i = 0; i < xs.n && i < cap; ++i) {
        if (ts) ts[i] = xs.xs[i].t;
        if (leaves) leaves[i] = xs.xs[i].leaf;
    }
    int n = xs.n;
    xl_free(&xs);
    return n;
}

/* ------------------------------------------------------------------------------------------- */
/* IntersectionState (ray.rs:72-140,177-219)                                                    */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    double t;
    int leaf;
    int inside;
    v3 point, over_point, under_point, eyev, normalv, reflectv;
    double n1, n2;
} comps_t;

/* compute_materials + compute_refractive_indices (ray.rs:104-140) */
static void refractive_indices(ctx_t *cx, const xlist *xs, const isect *self, double *n1, double *n2) {
    const rtmi_scene_desc *sc = cx->sc;
    if (cx->flags & ORC_SCHEME_N1N2) { /* Scheme: list holds only the hit (world.sld:76) */
        *n1 = 1.0;
        *n2 = sc->materials[sc->leaves[self->leaf].material].refractive_index;
        return;
    }
    int *containers = (int *)malloc((size_t)(xs->n + 1) * sizeof(int));
    int nc = 0;
    *n1 = 1.0; *n2 = 1.0;
    for (int k = 0; k < xs->n; ++k) {
        const isect *i = &xs->xs[k];
        int is_self = (i->t == self->t && i->leaf == self->leaf); /* PartialEq, ray.rs:47-51 */
        if (is_self)
            *n1 = nc ? sc->materials[sc->leaves[containers[nc - 1]].material].refractive_index : 1.0;
        int idx = -1;
        for (int c = 0; c < nc; ++c)
            if (containers[c] == i->leaf) { idx = c; break; }
        if (idx >= 0) {
            memmove(containers + idx, containers + idx + 1, (size_t)(nc - idx - 1) * sizeof(int));
            --nc;
        } else {
            containers[nc++] = i->leaf;
        }
        if (is_self) {
            *n2 = nc ? sc->materials[sc->leaves[containers[nc - 1]].material].refractive_index : 1.0;
            free(containers);
            return;
        }
    }
    free(containers); /* reference panics "hit not found in intersections" */
}

/* Intersection::prepare_computations (ray.rs:72-102) */
static comps_t prepare_computations(ctx_t *cx, const isect *self, ray_t ray, const xlist *xs) {
    comps_t c;
    const rtmi_leaf *g = &cx->sc->leaves[self->leaf];
    c.t = self->t;
    c.leaf = self->leaf;
    c.point = ray_position(ray, self->t);
    c.eyev = vneg(ray.d);
    v3 normalv = normal_at(g, c.point, self->u, self->v);
    c.inside = vdot(normalv, c.eyev) < 0.0;
    c.normalv = c.inside ? vneg(normalv) : normalv;
    c.over_point = vadd(c.point, vmul(c.normalv, EPS));
    c.under_point = vsub(c.point, vmul(c.normalv, EPS));
    c.reflectv = vreflect(ray.d, c.normalv);
    refractive_indices(cx, xs, self, &c.n1, &c.n2);
    return c;
}

/* schlick (ray.rs:199-219) */
static double schlick(v3 eyev, v3 normalv, double n1, double n2) {
    double cos_en = vdot(eyev, normalv);
    if (n1 > n2) {
        double n = n1 / n2;
        double sin2_t = n * n * (1.0 - cos_en * cos_en);
        if (sin2_t > 1.0) return 1.0;
        cos_en = sqrt(1.0 - sin2_t);
    }
    double r0 = sq((n1 - n2) / (n1 + n2));
    double tmp = 1.0 - cos_en;
    double tmp_pow_2 = tmp * tmp;
    double tmp_pow_5 = tmp_pow_2 * tmp_pow_2 * tmp;
    return r0 + (1.0 - r0) * tmp_pow_5;
}

double orc_schlick(const double *eyev, const double *normalv, double n1, double n2) {
    return schlick(from3(eyev), from3(normalv), n1, n2);
}

/* ------------------------------------------------------------------------------------------- */
/* patterns (pattern.rs:7-35,51-53)                                                             */
/* ------------------------------------------------------------------------------------------- */
/* Rust's `%` on f64 is fmod (sign of the dividend): -1 % 2 == -1 != 0 */
static v3 pattern_func(const rtmi_pattern *p, v3 q) {
    v3 a = from3(p->a), b = from3(p->b);
    switch (p->kind) {
    case RTMI_PATTERN_STRIPE: return fmod(floor(q.x), 2.0) == 0.0 ? a : b;
    case RTMI_PATTERN_GRADIENT: return vadd(a, vmul(vsub(b, a), q.x - floor(q.x)));
    case RTMI_PATTERN_RING: return fmod(floor(sqrt(q.x * q.x + q.z * q.z)), 2.0) == 0.0 ? a : b;
    case RTMI_PATTERN_CHECKERS:
        return fmod(floor(q.x + EPS) + floor(q.y + EPS) + floor(q.z + EPS), 2.0) == 0.0 ? a : b;
    case RTMI_PATTERN_XYZ: return q;
    default: return V(0, 0, 0);
    }
}

void orc_pattern_at(const rtmi_pattern *p, const double *leaf_inv, const double *world_point,
                    double *rgb) {
    v3 obj = mat_point(leaf_inv, from3(world_point)); /* shapes/mod.rs:267-269 */
    to3(pattern_func(p, mat_point(p->inv, obj)), rgb); /* pattern.rs:51-53 */
}

/* Phong::color_at (materials.rs:239-244): evaluated at comps.point */
static v3 surface_color_at(ctx_t *cx, const comps_t *c) {
    const rtmi_leaf *g = &cx->sc->leaves[c->leaf];
    const rtmi_material *m = &cx->sc->materials[g->material];
    if (m->pattern < 0) return from3(m->color);
    const rtmi_pattern *p = &cx->sc->patterns[m->pattern];
    return pattern_func(p, mat_point(p->inv, mat_point(g->inv, c->point)));
}

/* ------------------------------------------------------------------------------------------- */
/* lights (lights.rs)                                                                           */
/* ------------------------------------------------------------------------------------------- */
enum { IL_RAY = 0, IL_OMNI = 1, IL_NOLIGHT = 2 };
typedef struct { int kind; v3 origin, direction, color; } incoming_t;

#define ORC_PI 3.14159265358979323846

static incoming_t incoming_at(const rtmi_light *l, v3 point, double u1, double u2) {
    incoming_t in;
    v3 pos = from3(l->position), I = from3(l->intensity);
    in.kind = IL_RAY;
    in.origin = pos; in.direction = V(0, 0, 0); in.color = I;
    switch (l->kind) {
    case RTMI_LIGHT_POINT: /* lights.rs:89-95 */
        in.direction = vnorm(vsub(pos, point));
        break;
    case RTMI_LIGHT_AMBIENT: /* lights.rs:292-294 */
        in.kind = IL_OMNI;
        break;
    case RTMI_LIGHT_REALISTIC_POINT: { /* lights.rs:153-160 */
        v3 d = vsub(pos, point);
        in.direction = vnorm(d);
        in.color = vmul(I, 1.0 / (4.0 * ORC_PI * vdot(d, d)));
        break;
    }
    case RTMI_LIGHT_BEAM: { /* lights.rs:230-243 */
        v3 dir = from3(l->direction), up = from3(l->up), right = from3(l->right);
        v3 d = vsub(point, pos);
        if (vdot(d, dir) < 0.0) { in.kind = IL_NOLIGHT; break; }
        double u = vdot(d, up) / vdot(up, up);
        double v = vdot(d, right) / vdot(right, right);
        in.origin = vadd(vadd(pos, vmul(up, u)), vmul(right, v));
        in.direction = vneg(dir);
        in.color = vmul(I, exp(-(u * u + v * v)));
        break;
    }
    case RTMI_LIGHT_SPHERE: { /* lights.rs:353-362; UnitSphere by inverse CDF: z uniform, phi uniform */
        double z = 1.0 - 2.0 * u1, phi = 2.0 * ORC_PI * u2;
        double rr = sqrt(fmax(0.0, 1.0 - z * z));
        v3 p = V(rr * cos(phi), rr * sin(phi), z);
        in.origin = vadd(pos, vmul(p, l->radius));
        in.direction = vnorm(vsub(in.origin, point));
        break;
    }
    case RTMI_LIGHT_DISC: { /* lights.rs:438-453; UnitDisc by inverse CDF: r = sqrt(u1) */
        double rr = sqrt(u1), phi = 2.0 * ORC_PI * u2;
        double u = rr * cos(phi), v = rr * sin(phi);
        in.origin = vadd(vadd(pos, vmul(from3(l->width), u)), vmul(from3(l->height), v));
        v3 d = vsub(in.origin, point);
        in.direction = vnorm(d);
        double distance2 = vdot(d, d);
        double cosn = fmax(vdot(vneg(in.direction), from3(l->normal)), 0.0);
        in.color = vmul(vmul(I, cosn), 1.0 / (4.0 * ORC_PI * distance2));
        break;
    }
    default:
        in.kind = IL_NOLIGHT;
    }
    return in;
}

static int light_draws(int kind) { return (kind == RTMI_LIGHT_SPHERE || kind == RTMI_LIGHT_DISC) ? 2 : 0; }

void orc_incoming_at(const rtmi_light *light, const double *point, double u1, double u2, int *kind_out,
                     double *origin, double *direction, double *color) {
    incoming_t in = incoming_at(light, from3(point), u1, u2);
    *kind_out = in.kind;
    to3(in.origin, origin); to3(in.direction, direction); to3(in.color, color);
}

/* Phong::lighting (materials.rs:246-288) */
static v3 vhad(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static v3 lighting(const rtmi_material *m, v3 surface_color, incoming_t light, v3 eyev, v3 normalv,
                   int in_shadow) {
    if (light.kind == IL_NOLIGHT) return V(0, 0, 0);
    if (light.kind == IL_OMNI) return vhad(surface_color, light.color);
    v3 effective_color = vhad(surface_color, light.color);
    if (in_shadow) return V(0, 0, 0);
    double light_dot_normal = vdot(light.direction, normalv);
    if (light_dot_normal <= 0.0) return V(0, 0, 0);
    v3 diffuse = vmul(effective_color, m->diffuse * light_dot_normal);
    v3 reflectv = vreflect(vneg(light.direction), normalv);
    double reflect_dot_eye = vdot(reflectv, eyev);
    v3 specular = V(0, 0, 0);
    if (reflect_dot_eye > 0.0) specular = vmul(light.color, m->specular * pow(reflect_dot_eye, m->shininess));
    return vadd(diffuse, specular);
}

void orc_lighting(const rtmi_material *m, const double *surface_color, const rtmi_light *light,
                  const double *point, const double *eyev, const double *normalv, int in_shadow,
                  double *rgb) {
    incoming_t in = incoming_at(light, from3(point), 0.5, 0.5);
    to3(lighting(m, from3(surface_color), in, from3(eyev), from3(normalv), in_shadow), rgb);
}


/* ------------------------------------------------------------------------------------------- */
/* Photon map: PhotonMap::from_vec / balance / find_nearest (photon_map.rs:230-380)             */
/* ------------------------------------------------------------------------------------------- */
/* compute_number_of_left_children (photon_map.rs:423-444) */
int64_t orc_left_children(int64_t n) {
    if (n <= 0) return 0;
    if (n <= 3) return 1;
    if (n == 4) return 2;
    if (n <= 7) return 3;
    int depth = (int)floor(log2((double)n));
    int64_t full = (int64_t)1 << depth;
    int64_t last = n - (full - 1);
    if (last <= full / 2) return (full - 2) / 2 + last;
    return (full - 2) / 2 + full / 2;
}

/* StoredPhoton::compress_color / decompress_color (photon_map.rs:32-47) */
void orc_compress_color(const double *c, float *scale, uint8_t *rgb) {
    float x = (float)fmax(fmax(c[0], c[1]), c[2]);
    for (int k = 0; k < 3; ++k) {
        float v = 255.0f * (float)c[k] / x;
        v = fminf(fmaxf(v, 0.0f), 255.0f); /* (f32::max / min drop a NaN: 0 / 0 becomes 0, as Rust's do) */
        rgb[k] = (uint8_t)v;               /* Rust `as u8`: truncation */
    }
    *scale = x;
}
void orc_decompress_color(float scale, const uint8_t *rgb, double *c) {
    for (int k = 0; k < 3; ++k) c[k] = (double)rgb[k] * (double)scale / 255.0;
}

typedef struct { const rtmi_photon *ph; int axis; } keyctx_t;
static const rtmi_photon *g_sort_ph; /* (qsort has no context argument; builds are serialised by pm_lock) */
static int g_sort_axis;
static int cmp_key(const void *a, const void *b) {
    int64_t i = *(const int64_t *)a, j = *(const int64_t *)b;
    float x = g_sort_ph[i].position[g_sort_axis], y = g_sort_ph[j].position[g_sort_axis];
    if (x < y) return -1;
    if (x > y) return 1;
    return i < j ? -1 : (i > j ? 1 : 0); /* equal keys: the reference's partition_by_key leaves their order to its
                                            partial sort (partial_sort.rs); index order here - unpinned for coincident coordinates */
}
static pthread_mutex_t pm_lock = PTHREAD_MUTEX_INITIALIZER;

/* PhotonMap::balance (photon_map.rs:252-313): the median by COUNT (compute_number_of_left_children keeps the tree
 * left-balanced) along the longest side of the cell; the cell is cut at the median photon's coordinate. */
static void balance(const rtmi_photon *ph, int64_t *idx, int64_t n, int64_t *tree, uint8_t *flag, int64_t node, double lo[3], double hi[3]) {
    if (n == 0) return;
    if (n == 1) { tree[node] = idx[0]; flag[idx[0]] = KD_LEAF; return; }
    int64_t n_left = orc_left_children(n);
    double sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
    int axis = (sx >= sy && sx >= sz) ? 0 : ((sy >= sx && sy >= sz) ? 1 : 2);
    g_sort_ph = ph; g_sort_axis = axis;
    qsort(idx, (size_t)n, sizeof(int64_t), cmp_key); /* (partition_by_key(n_left, ..) needs only the n_left smallest in front) */
    int64_t m = idx[n_left];
    double median = (double)ph[m].position[axis];
    tree[node] = m;
    flag[m] = (uint8_t)(KD_SPLIT_X + axis);
    double lhi[3] = {hi[0], hi[1], hi[2]}, rlo[3] = {lo[0], lo[1], lo[2]};
    lhi[axis] = median; rlo[axis] = median; /* Aabb::split_x/y/z (aabb.rs:105-127) */
    balance(ph, idx, n_left, tree, flag, 2 * (node + 1) - 1, lo, lhi);
    balance(ph, idx + n_left + 1, n - n_left - 1, tree, flag, 2 * (node + 1), rlo, hi);
}

/* the index tree PhotonMap::balance leaves (what the reference's balance2/5/10 tests compare); lo/hi: the extent to start from */
void orc_photon_balance(const rtmi_photon *ph, int64_t n, const double *lo, const double *hi, int64_t *tree) {
    int64_t *idx = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    uint8_t *flag = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int64_t i = 0; i < n; ++i) { idx[i] = i; tree[i] = -1; }
    double l[3] = {lo[0], lo[1], lo[2]}, h[3] = {hi[0], hi[1], hi[2]};
    pthread_mutex_lock(&pm_lock);
    balance(ph, idx, n, tree, flag, 0, l, h);
    pthread_mutex_unlock(&pm_lock);
    free(idx); free(flag);
}

/* PhotonMap::from_vec (photon_map.rs:230-243): extent (compute_extent, 415-421), balance, permute into tree order */
static pmap_t *pmap_build(const rtmi_photon *ph, int64_t n, double max_search_radius) {
    pmap_t *pm = (pmap_t *)calloc(1, sizeof(pmap_t));
    pm->n = n;
    pm->photons = (rtmi_photon *)malloc((size_t)(n > 0 ? n : 1) * sizeof(rtmi_photon));
    pm->flag = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    pm->orig = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    pm->max_search_radius = max_search_radius > 0.0 ? max_search_radius : INFINITY;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = fmin(lo[a], (double)ph[i].position[a]); hi[a] = fmax(hi[a], (double)ph[i].position[a]); }
    int64_t *idx = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    uint8_t *flag = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int64_t i = 0; i < n; ++i) { idx[i] = i; pm->orig[i] = -1; }
    pthread_mutex_lock(&pm_lock);
    balance(ph, idx, n, pm->orig, flag, 0, lo, hi);
    pthread_mutex_unlock(&pm_lock);
    for (int64_t k = 0; k < n; ++k) { pm->photons[k] = ph[pm->orig[k]]; pm->flag[k] = flag[pm->orig[k]]; }
    free(idx); free(flag);
    return pm;
}
static void pmap_free(pmap_t *pm) {
    if (!pm) return;
    free(pm->photons); free(pm->flag); free(pm->orig); free(pm);
}

/* std::collections::BinaryHeap<SearchedPhoton> as find_nearest uses it: a max-heap on the squared distance */
typedef struct { double d2; int64_t node; } near_t;
typedef struct { near_t *a; int n, cap; } nheap_t;
static void nh_push(nheap_t *h, near_t e) {
    int i = h->n++;
    h->a[i] = e;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!(h->a[p].d2 < h->a[i].d2)) break;
        near_t t = h->a[p]; h->a[p] = h->a[i]; h->a[i] = t;
        i = p;
    }
}
static void nh_pop(nheap_t *h) {
    h->a[0] = h->a[--h->n];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < h->n && h->a[m].d2 < h->a[l].d2) m = l;
        if (r < h->n && h->a[m].d2 < h->a[r].d2) m = r;
        if (m == i) break;
        near_t t = h->a[m]; h->a[m] = h->a[i]; h->a[i] = t;
        i = m;
    }
}

/* PhotonMap::locate_photons (photon_map.rs:323-364): near side, far side if the plane is within the search radius and
 * (the heap is not full or the plane is nearer than its farthest photon), then this node's own photon */
static void locate_photons(const pmap_t *pm, int64_t node, int k, const double p[3], nheap_t *h) {
    if (node >= pm->n) return;
    const rtmi_photon *ph = &pm->photons[node];
    if (2 * (node + 1) - 1 < pm->n) { /* !is_leaf (photon_map.rs:376-378) */
        int axis = pm->flag[node] - KD_SPLIT_X;
        double d = p[axis] - (double)ph->position[axis];
        if (d < 0.0) {
            locate_photons(pm, 2 * (node + 1) - 1, k, p, h);
            if (-d < pm->max_search_radius && (h->n < k || d * d < h->a[0].d2)) locate_photons(pm, 2 * (node + 1), k, p, h);
        } else {
            locate_photons(pm, 2 * (node + 1), k, p, h);
            if (d < pm->max_search_radius && (h->n < k || d * d < h->a[0].d2)) locate_photons(pm, 2 * (node + 1) - 1, k, p, h);
        }
    }
    double dx = p[0] - (double)ph->position[0], dy = p[1] - (double)ph->position[1], dz = p[2] - (double)ph->position[2];
    double d2 = dx * dx + dy * dy + dz * dz;
    if (h->n < k || d2 < h->a[0].d2) {
        if (h->n == k) nh_pop(h);
        near_t e = {d2, node};
        nh_push(h, e);
    }
}

/* PhotonMap::find_nearest (photon_map.rs:315-321): the photons found (as indices into the CALLER's array, in the heap's
 * order) and the squared distance of the farthest (1.0 if none).  Returns how many. */
int orc_find_nearest(const rtmi_photon *photons, int64_t n, double max_search_radius, int n_neighbors, const double *p, int64_t *found,
                     double *max_sq) {
    pmap_t *pm = pmap_build(photons, n, max_search_radius);
    nheap_t h = {(near_t *)malloc((size_t)(n_neighbors > 0 ? n_neighbors : 1) * sizeof(near_t)), 0, n_neighbors};
    if (n_neighbors > 0) locate_photons(pm, 0, n_neighbors, p, &h);
    *max_sq = h.n ? h.a[0].d2 : 1.0;
    for (int i = 0; i < h.n; ++i) found[i] = pm->orig[h.a[i].node];
    int cnt = h.n;
    free(h.a);
    pmap_free(pm);
    return cnt;
}

static const pmap_t *ctx_pmap(ctx_t *cx) {
    if (!cx->pm && cx->sc->n_photons > 0 && cx->sc->photons) {
        cx->pm = pmap_build(cx->sc->photons, cx->sc->n_photons, cx->sc->max_search_radius);
        cx->pm_owned = 1;
    }
    return cx->pm;
}
static void ctx_done(ctx_t *cx) {
    if (cx->pm_owned) pmap_free((pmap_t *)cx->pm);
    cx->pm = NULL; cx->pm_owned = 0;
}

/* the photon-map term of Phong::shade_hit (materials.rs:376-384) */
static v3 photon_estimate(ctx_t *cx, const comps_t *c, v3 surface_color) {
    const pmap_t *pm = ctx_pmap(cx);
    int k = cx->sc->n_nearest;
    if (!pm || k <= 0) return V(0, 0, 0);
    nheap_t h = {(near_t *)malloc((size_t)k * sizeof(near_t)), 0, k};
    double p[3] = {c->point.x, c->point.y, c->point.z};
    locate_photons(pm, 0, k, p, &h);
    double square_radius = h.n ? h.a[0].d2 : 1.0;
    v3 total = V(0, 0, 0);
    for (int i = 0; i < h.n; ++i) { /* (the fold runs over the heap's array, as BinaryHeap::into_iter does) */
        const rtmi_photon *ph = &pm->photons[h.a[i].node];
        double pw[3];
        orc_decompress_color(ph->power_scale, ph->power_rgb, pw);
        double w = fmax(vdot(c->normalv, V((double)ph->direction[0], (double)ph->direction[1], (double)ph->direction[2])), 0.0);
        total = vadd(total, vmul(V(pw[0], pw[1], pw[2]), w));
    }
    free(h.a);
    v3 sc = V(surface_color.x * total.x, surface_color.y * total.y, surface_color.z * total.z);
    return vmul(sc, 1.0 / (3.14159265358979323846 * square_radius)); /* (Color / f64 is scale(1 / x), color.rs:145-149) */
}

/* ------------------------------------------------------------------------------------------- */
/* World::color_at / is_shadowed / Phong::shade_hit                                             */
/* ------------------------------------------------------------------------------------------- */
static int color_at(ctx_t *cx, ray_t ray, int remaining, int ray_class, v3 *out);

/* World::is_shadowed (world.rs:172-180) */
static int is_shadowed(ctx_t *cx, const incoming_t *light, v3 p) {
    if (light->kind == IL_OMNI) return 0;
    if (light->kind == IL_NOLIGHT) return 1;
    cx->stat[RTMI_STAT_SHADOW]++;
    ray_t r = {p, light->direction};
    xlist xs; xl_init(&xs);
    world_intersect(cx, r, 1, &xs);
    int h = hit_index(&xs);
    int shadowed = 0;
    if (h >= 0) shadowed = xs.xs[h].t < vlen(vsub(light->origin, p));
    xl_free(&xs);
    return shadowed;
}

/* Color::clip (color.rs:62-68) */
static double clip1(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

/* Phong::reflected_color (materials.rs:290-308) */
static v3 reflected_color(ctx_t *cx, const rtmi_material *m, const comps_t *c, int remaining) {
    if (m->reflective == 0.0 || remaining == 0) return V(0, 0, 0);
    ray_t r = {c->over_point, c->reflectv};
    v3 col;
    if (!color_at(cx, r, remaining - 1, RTMI_STAT_REFLECTION, &col)) return V(0, 0, 0);
    return vmul(col, m->reflective);
}

/* Phong::refracted_color (materials.rs:310-336) */
static v3 refracted_color(ctx_t *cx, const rtmi_material *m, const comps_t *c, int remaining) {
    if (remaining == 0 || m->transparency == 0.0) return V(0, 0, 0);
    double n_ratio = c->n1 / c->n2;
    double cos_i = vdot(c->eyev, c->normalv);
    double sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
    if (sin2_t > 1.0) return V(0, 0, 0);
    double cos_t = sqrt(1.0 - sin2_t);
    v3 direction = vsub(vmul(c->normalv, n_ratio * cos_i - cos_t), vmul(c->eyev, n_ratio));
    ray_t r = {c->under_point, direction};
    v3 col;
    if (!color_at(cx, r, remaining - 1, RTMI_STAT_REFRACTION, &col)) return V(0, 0, 0);
    return vmul(col, m->transparency);
}

/* Phong::shade_hit (materials.rs:371-412), with the photon-map branch (376-384) when the scene carries a map */
static v3 shade_hit(ctx_t *cx, const comps_t *c, int remaining) {
    const rtmi_scene_desc *sc = cx->sc;
    const rtmi_material *m = &sc->materials[sc->leaves[c->leaf].material];
    v3 surface_color = surface_color_at(cx, c);
    v3 surface = V(0, 0, 0);
    if (sc->n_photons > 0 && sc->photons) surface = vadd(surface, photon_estimate(cx, c, surface_color));
    for (int li = 0; li < (sc->no_direct_illumination ? 0 : sc->n_lights); ++li) { /* world.direct_illumination_enabled(), materials.rs:386 */
        const rtmi_light *l = &sc->lights[li];
        double u1 = 0, u2 = 0;
        if (light_draws(l->kind)) { u1 = rng_next(&cx->rng); u2 = rng_next(&cx->rng); }
        incoming_t in = incoming_at(l, c->over_point, u1, u2);
        int in_shadow = is_shadowed(cx, &in, c->over_point);
        surface = vadd(surface, lighting(m, surface_color, in, c->eyev, c->normalv, in_shadow));
    }
    surface = V(clip1(surface.x, 0, 1), clip1(surface.y, 0, 1), clip1(surface.z, 0, 1));
    v3 emissive = vmul(surface_color, m->emissive);
    v3 reflected = reflected_color(cx, m, c, remaining);
    v3 refracted = refracted_color(cx, m, c, remaining);
    if (m->reflective > 0.0 && m->transparency > 0.0) {
        double reflectance = schlick(c->eyev, c->normalv, c->n1, c->n2);
        return vadd(vadd(vadd(surface, vmul(reflected, reflectance)), vmul(refracted, 1.0 - reflectance)),
                    emissive);
    }
    return vadd(vadd(vadd(surface, reflected), refracted), emissive);
}

/* World::color_at (world.rs:115-118) */
static int color_at(ctx_t *cx, ray_t ray, int remaining, int ray_class, v3 *out) {
    cx->stat[ray_class]++;
    xlist xs; xl_init(&xs);
    world_intersect(cx, ray, 0, &xs);
    int h = hit_index(&xs);
    if (h < 0) {
        xl_free(&xs);
        *out = V(0, 0, 0);
        return 0;
    }
    comps_t c = prepare_computations(cx, &xs.xs[h], ray, &xs);
    *out = shade_hit(cx, &c, remaining);
    xl_free(&xs);
    return 1;
}

int orc_color_at(const rtmi_scene_desc *scene, const double *origin, const double *dir, int remaining,
                 int flags, double *rgb) {
    ctx_t cx; ctx_init(&cx, scene, flags);
    ray_t r = {from3(origin), from3(dir)};
    v3 c;
    int hit = color_at(&cx, r, remaining, RTMI_STAT_PRIMARY, &c);
    to3(c, rgb);
    ctx_done(&cx);
    return hit;
}

static void fill_record(ctx_t *cx, const comps_t *c, rtmi_hit_record *rec) {
    rec->hit = 1;
    rec->leaf = c->leaf;
    rec->inside = c->inside;
    rec->t = c->t;
    to3(c->point, rec->point);
    to3(c->normalv, rec->normal);
    rec->n1 = c->n1;
    rec->n2 = c->n2;
    rec->shadowed = -1.0;
    int k0 = cx->sc->n_lights > 0 ? cx->sc->lights[0].kind : -1;
    if (k0 == RTMI_LIGHT_POINT || k0 == RTMI_LIGHT_REALISTIC_POINT || k0 == RTMI_LIGHT_BEAM) {
        incoming_t in = incoming_at(&cx->sc->lights[0], c->over_point, 0.5, 0.5);
        if (in.kind == IL_RAY) rec->shadowed = is_shadowed(cx, &in, c->over_point);
    }
}

int orc_hit_record(const rtmi_scene_desc *scene, const double *origin, const double *dir, int flags,
                   rtmi_hit_record *rec) {
    ctx_t cx; ctx_init(&cx, scene, flags);
    ray_t r = {from3(origin), from3(dir)};
    memset(rec, 0, sizeof *rec);
    xlist xs; xl_init(&xs);
    world_intersect(&cx, r, 0, &xs);
    int h = hit_index(&xs);
    if (h >= 0) {
        comps_t c = prepare_computations(&cx, &xs.xs[h], r, &xs);
        fill_record(&cx, &c, rec);
    }
    xl_free(&xs);
    return h >= 0;
}

int orc_shade_explicit(const rtmi_scene_desc *scene, const double *origin, const double *dir, int n_xs,
                       const double *ts, const int32_t *leaves, const double *uvs, double hit_t,
                       int hit_leaf, int remaining, int what, int flags, double *rgb,
                       rtmi_hit_record *rec) {
    ctx_t cx; ctx_init(&cx, scene, flags);
    ray_t r = {from3(origin), from3(dir)};
    xlist xs; xl_init(&xs);
    for (int i = 0; i < n_xs; ++i)
        xl_push(&xs, ts[i], leaves[i], uvs ? uvs[2 * i] : 0.0, uvs ? uvs[2 * i + 1] : 0.0);
    xl_sort(&xs);
    int h = -1;
    for (int i = 0; i < xs.n; ++i)
        if (xs.xs[i].t == hit_t && xs.xs[i].leaf == hit_leaf) { h = i; break; }
    if (h < 0) { xl_free(&xs); return 0; }
    comps_t c = prepare_computations(&cx, &xs.xs[h], r, &xs);
    const rtmi_material *m = &scene->materials[scene->leaves[c.leaf].material];
    v3 col;
    if (what == 1) col = reflected_color(&cx, m, &c, remaining);
    else if (what == 2) col = refracted_color(&cx, m, &c, remaining);
    else col = shade_hit(&cx, &c, remaining);
    if (rgb) to3(col, rgb);
    if (rec) { memset(rec, 0, sizeof *rec); fill_record(&cx, &c, rec); }
    xl_free(&xs);
    ctx_done(&cx);
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* Camera (camera.rs)                                                                           */
/* ------------------------------------------------------------------------------------------- */
/* Camera::ray_for_pixel (camera.rs:120-153) */
static ray_t ray_for_pixel(const rtmi_camera *cam, int px, int py, int randomize, rng_t *g) {
    double x_offset, y_offset;
    if (randomize) {
        x_offset = ((double)px + rng_next(g)) * cam->pixel_size;
        y_offset = ((double)py + rng_next(g)) * cam->pixel_size;
    } else {
        x_offset = ((double)px + 0.5) * cam->pixel_size;
        y_offset = ((double)py + 0.5) * cam->pixel_size;
    }
    double world_x = cam->half_width - x_offset;
    double world_y = cam->half_height - y_offset;
    v3 pixel = mat_point(cam->inv, V(world_x, world_y, -1.0));
    v3 origin = mat_point(cam->inv, V(0, 0, 0));
    ray_t primary = {origin, vnorm(vsub(pixel, origin))};
    if (cam->aperture_size == 0.0) return primary;
    v3 focal_point = ray_position(primary, cam->focal_distance);
    double a0 = rng_next(g), a1 = rng_next(g);
    v3 ap_pixel = vadd(pixel, V((a0 - 0.5) * cam->aperture_size, (a1 - 0.5) * cam->aperture_size, 0.0));
    ray_t secondary = {ap_pixel, vnorm(vsub(focal_point, ap_pixel))};
    return secondary;
}

void orc_camera_ray(const rtmi_camera *cam, uint64_t seed, int px, int py, int sample, double *origin,
                    double *dir) {
    rng_t g = {seed, (uint32_t)(py * cam->hsize + px), (uint32_t)sample, 0};
    ray_t r = ray_for_pixel(cam, px, py, sample != 0, &g);
    to3(r.o, origin); to3(r.d, dir);
}

/* color_variance_of_mean (camera.rs:305-310) */
static double color_variance_of_mean(double n, v3 sum, v3 sos) {
    v3 mean = vmul(sum, 1.0 / n);
    v3 var = vsub(vmul(sos, 1.0 / n), vhad(mean, mean));
    double variance = (var.x + var.y + var.z) / 3.0;
    return variance / n;
}

/* one sample: World::trace(ray).unwrap_or(BLACK) (camera.rs:274-276, world.rs:111-113) */
static v3 trace_sample(ctx_t *cx, const rtmi_camera *cam, int x, int y, uint32_t sample) {
    cx->rng.pixel = (uint32_t)(y * cam->hsize + x);
    cx->rng.sample = sample;
    cx->rng.draw = 0;
    cx->stat[RTMI_STAT_SAMPLES]++;
    ray_t r = ray_for_pixel(cam, x, y, sample != 0, &cx->rng);
    v3 c;
    color_at(cx, r, cx->sc->max_depth, RTMI_STAT_PRIMARY, &c);
    return c;
}

/* Camera::multisample (camera.rs:273-302) */
static v3 multisample(ctx_t *cx, const rtmi_camera *cam, int max_samples, int x, int y) {
    v3 c = trace_sample(cx, cam, x, y, 0);
    /* camera.rs:277-278: BOTH accumulators start from the first sample itself - the sum of squares from c, not
     * from c * c.  The variance estimate therefore carries an extra (c0 - c0^2) / n per channel, and a flat pixel
     * of value c takes sqrt(mean(c - c^2)) / SE samples (50 for c = 0.5 at SE = 1e-2) instead of min_samples. */
    v3 sos = c, sum = c;
    double n = 1.0;
    uint32_t s = 1;
    while (n < (double)cam->min_samples) {
        if (max_samples > 0 && (int)s >= max_samples) break;
        c = trace_sample(cx, cam, x, y, s++);
        sum = vadd(sum, c); sos = vadd(sos, vhad(c, c));
        n += 1.0;
    }
    double se2 = cam->allowed_standard_error * cam->allowed_standard_error;
    while (color_variance_of_mean(n, sum, sos) > se2) {
        if (max_samples > 0 && (int)s >= max_samples) break;
        c = trace_sample(cx, cam, x, y, s++);
        sum = vadd(sum, c); sos = vadd(sos, vhad(c, c));
        n += 1.0;
    }
    return vmul(sum, 1.0 / n);
}

/* Camera::render / trace_pixels (camera.rs:155-161,250-265): rayon's par_iter over pixels becomes a
 * pool of pthreads pulling rows from a shared counter. */
typedef struct {
    const rtmi_scene_desc *sc;
    const rtmi_camera *cam;
    const rtmi_render_opts *opts;
    int flags, row_stride, row_offset;
    double *rgb;
    volatile int *next_row;
    uint64_t stat[RTMI_N_STATS];
    const pmap_t *pm; /* the frame's photon map, built once by orc_render */
} job_t;

static void *render_worker(void *arg) {
    job_t *j = (job_t *)arg;
    ctx_t cx; ctx_init(&cx, j->sc, j->flags);
    cx.pm = j->pm;
    cx.rng.seed = j->opts ? j->opts->seed : 0;
    int max_samples = j->opts ? j->opts->max_samples : 0;
    for (;;) {
        int k = __sync_fetch_and_add(j->next_row, 1);
        int y = k * j->row_stride + j->row_offset;
        if (y >= j->cam->vsize) break;
        for (int x = 0; x < j->cam->hsize; ++x) {
            v3 c = multisample(&cx, j->cam, max_samples, x, y);
            to3(c, j->rgb + 3 * ((size_t)y * j->cam->hsize + x));
        }
    }
    memcpy(j->stat, cx.stat, sizeof cx.stat);
    return NULL;
}

int orc_render(const rtmi_scene_desc *scene, const rtmi_camera *cam, const rtmi_render_opts *opts,
               int n_threads, int flags, int row_stride, int row_offset, double *rgb_out,
               uint64_t *stats) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (row_stride < 1) row_stride = 1;
    volatile int next_row = 0;
    job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    pmap_t *pm = (scene->n_photons > 0 && scene->photons) ? pmap_build(scene->photons, scene->n_photons, scene->max_search_radius) : NULL;
    for (int i = 0; i < n_threads; ++i) {
        jobs[i].pm = pm;
        jobs[i].sc = scene; jobs[i].cam = cam; jobs[i].opts = opts; jobs[i].flags = flags;
        jobs[i].row_stride = row_stride; jobs[i].row_offset = row_offset;
        jobs[i].rgb = rgb_out; jobs[i].next_row = &next_row;
        if (n_threads == 1) render_worker(&jobs[i]);
        else pthread_create(&th[i], NULL, render_worker, &jobs[i]);
    }
    if (stats) memset(stats, 0, RTMI_N_STATS * sizeof(uint64_t));
    for (int i = 0; i < n_threads; ++i) {
        if (n_threads > 1) pthread_join(th[i], NULL);
        if (stats)
            for (int k = 0; k < RTMI_N_STATS; ++k) stats[k] += jobs[i].stat[k];
    }
    free(jobs); free(th);
    pmap_free(pm);
    return 0;
}

/* Color::to_u8 (color.rs:53-60): clip, x255, f64::round = half away from zero */
void orc_to_u8(const double *rgb, int64_t n, uint8_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)round(clip1(rgb[i], 0.0, 1.0) * 255.0);
}

/* ------------------------------------------------------------------------------------------- */
/* World::compute_photon_map's tracing half (world.rs:186-273)                                  */
/* ------------------------------------------------------------------------------------------- */
enum { PK_DIRECT = 0, PK_CAUSTIC = 1, PK_DIFFUSE = 2 }; /* PhotonKind, photon_map.rs:185-190 */

/* UnitSphere by inverse CDF (two draws), the map incoming_at uses for SphereLight */
static v3 unit_sphere(rng_t *g) {
    double u1 = rng_next(g), u2 = rng_next(g);
    double z = 1.0 - 2.0 * u1, phi = 2.0 * ORC_PI * u2;
    double rr = sqrt(fmax(0.0, 1.0 - z * z));
    return V(rr * cos(phi), rr * sin(phi), z);
}

/* CosineDistribution::sample (cosine_distribution.rs:16-28); the loop is cut after 16 rounds (direction itself) */
static v3 cosine_sample(v3 dir, rng_t *g) {
    for (int k = 0; k < 16; ++k) {
        v3 v = vadd(dir, unit_sphere(g));
        double l = vdot(v, v);
        if (l > EPS) return vmul(v, 1.0 / sqrt(l));
    }
    return dir;
}
void orc_cosine_sample(const double *dir, uint64_t seed, uint32_t index, double *out) {
    rng_t g = {seed, index, 0, 0};
    to3(cosine_sample(from3(dir), &g), out);
}

/* TravellingPhoton::reflect / refract (photon_map.rs:153-172): the new direction */
void orc_photon_reflect(const double *dir, const double *normv, double *out) { to3(vreflect(from3(dir), from3(normv)), out); }
int orc_photon_refract(const double *dir, const double *normv, double n1, double n2, double *out) {
    v3 d = from3(dir), n = from3(normv);
    double n_ratio = n1 / n2;
    double cos_i = -vdot(d, n);
    double sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
    if (sin2_t > 1.0) return 0; /* (the reference takes sqrt of a negative number here: no answer) */
    double cos_t = sqrt(1.0 - sin2_t);
    to3(vadd(vmul(n, n_ratio * cos_i - cos_t), vmul(d, n_ratio)), out);
    return 1;
}

/* Light::power (lights.rs:68-70,100-102,165-167,248-250,298-300,367-369,458-460) */
static double light_power(const rtmi_light *l) {
    if (l->kind == RTMI_LIGHT_AMBIENT) return 0.0;
    return (l->intensity[0] + l->intensity[1] + l->intensity[2]) / 3.0;
}

/* Light::emit_photon (lights.rs:104-111,169-176,252-261,371-380,462-473) */
static void light_emit(const rtmi_light *l, rng_t *g, ray_t *ray, v3 *power) {
    v3 pos = from3(l->position), I = from3(l->intensity);
    ray->o = pos; *power = I;
    switch (l->kind) {
    case RTMI_LIGHT_POINT: ray->d = unit_sphere(g); break;
    case RTMI_LIGHT_REALISTIC_POINT: ray->d = unit_sphere(g); *power = vmul(vmul(I, 2.0), 1.0 / ORC_PI); break;
    case RTMI_LIGHT_BEAM: { /* two StandardNormal draws: Box-Muller on two uniforms */
        double u1 = rng_next(g), u2 = rng_next(g);
        double r = sqrt(-2.0 * log(1.0 - u1)), phi = 2.0 * ORC_PI * u2;
        double u = r * cos(phi), v = r * sin(phi);
        ray->o = vadd(vadd(pos, vmul(from3(l->up), u)), vmul(from3(l->right), v));
        ray->d = from3(l->direction);
        *power = vmul(vmul(I, 2.0), 1.0 / ORC_PI);
        break;
    }
    case RTMI_LIGHT_SPHERE: {
        v3 p = unit_sphere(g);
        ray->d = cosine_sample(p, g);
        ray->o = vadd(pos, vmul(p, l->radius));
        break;
    }
    case RTMI_LIGHT_DISC: { /* UnitDisc by inverse CDF as in incoming_at */
        double u1 = rng_next(g), u2 = rng_next(g);
        double rr = sqrt(u1), phi = 2.0 * ORC_PI * u2;
        ray->o = vadd(vadd(pos, vmul(from3(l->width), rr * cos(phi))), vmul(from3(l->height), rr * sin(phi)));
        ray->d = cosine_sample(from3(l->normal), g);
        *power = vmul(vmul(vmul(I, 0.25), 2.0), 1.0 / ORC_PI);
        break;
    }
    default: ray->d = V(0, 0, 0); *power = V(0, 0, 0);
    }
}

/* rand 0.7.3 WeightedIndex: the first index whose cumulative weight exceeds x = u * total; a rounded-up x falls to the
 * last index with a weight */
static int weighted_index(const double *w, int n, double u) {
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += w[i];
    double x = u * total, cum = 0.0;
    int last = -1;
    for (int i = 0; i < n; ++i) {
        if (w[i] > 0.0) last = i;
        cum += w[i];
        if (x < cum) return i;
    }
    return last;
}

/* returns the number of photons the call stores (out holds the first min(that, capacity)); -1: no light with power;
 * -2 - leaf: Phong::photon_hit's assert on the absorption probability fails at that leaf (materials.rs:443-445) */
int64_t orc_trace_photons(const rtmi_scene_desc *sc, const rtmi_photon_opts *o, rtmi_photon *out, int64_t capacity, int64_t *n_rays,
                          int64_t *n_truncated) {
    ctx_t cx; ctx_init(&cx, sc, 0);
    int64_t stored = 0, rays = 0, truncated = 0;
    const int max_bounces = o->max_bounces > 0 ? o->max_bounces : RTMI_PHOTON_BOUNCES;
    const int direct_only = o->enable_direct && !o->enable_caustic && !o->enable_diffuse; /* world.rs:130-134 */
    double *lw = (double *)malloc((size_t)(sc->n_lights + 1) * sizeof(double));
    double total_power = 0.0;
    for (int i = 0; i < sc->n_lights; ++i) { lw[i] = light_power(&sc->lights[i]); total_power += lw[i]; }
    if (!(total_power > 0.0)) { free(lw); ctx_done(&cx); return -1; }
    const float inv_n = (float)(1.0 / (double)o->n_emit);
    int64_t result = 0;
    const int64_t idx_end = o->count > 0 ? o->first + o->count : o->n_emit; /* the call's share of the photons */
    for (int64_t idx = o->first; idx < idx_end && result >= 0; ++idx) {
        rng_t g = {o->seed ^ ((uint64_t)0x50484F54u << 32), (uint32_t)idx, 0, 0};
        ray_t ray; v3 power;
        int li = weighted_index(lw, sc->n_lights, rng_next(&g)); /* world.rs:211-215 */
        light_emit(&sc->lights[li], &g, &ray, &power);
        int kind = PK_DIRECT;
        for (int bounce = 0;; ++bounce) {
            if (bounce >= max_bounces) { truncated++; break; }
            if (power.x + power.y + power.z < EPS) break; /* world.rs:224-226 */
            if (kind != PK_DIRECT && direct_only) break;  /* world.rs:228-232 */
            xlist xs; xl_init(&xs);
            world_intersect(&cx, ray, 0, &xs);
            rays++;
            int h = hit_index(&xs);
            if (h < 0) { xl_free(&xs); break; }
            comps_t c = prepare_computations(&cx, &xs.xs[h], ray, &xs);
            xl_free(&xs);
            const rtmi_material *m = &sc->materials[sc->leaves[c.leaf].material];
            /* Phong::photon_hit (materials.rs:414-464) */
            v3 dr = vmul(surface_color_at(&cx, &c), m->diffuse);
            double spec = fmax(m->specular, m->reflective), trans = m->transparency;
            double pd_avg = (dr.x + dr.y + dr.z) / 3.0;
            int store = pd_avg > EPS;
            if (!o->enable_diffuse) pd_avg = 0.0;
            if (m->reflective > 0.0 && m->transparency > 0.0) {
                double r = schlick(c.eyev, c.normalv, c.n1, c.n2);
                spec *= r; trans *= 1.0 - r;
            }
            double w[4] = {1.0 - pd_avg - spec - trans, pd_avg, spec, trans};
            if (!(w[0] >= 0.0)) { result = -2 - c.leaf; break; }
            int choice = weighted_index(w, 4, rng_next(&g));
            if (store && ((kind == PK_DIRECT && o->enable_direct) || (kind == PK_DIFFUSE && o->enable_diffuse) ||
                          (kind == PK_CAUSTIC && o->enable_caustic))) { /* world.rs:256-265 */
                if (stored < capacity && out) {
                    rtmi_photon *p = &out[stored];
                    memset(p, 0, sizeof *p);
                    double pw[3] = {power.x, power.y, power.z};
                    p->position[0] = (float)c.point.x; p->position[1] = (float)c.point.y; p->position[2] = (float)c.point.z;
                    p->direction[0] = (float)-ray.d.x; p->direction[1] = (float)-ray.d.y; p->direction[2] = (float)-ray.d.z;
                    orc_compress_color(pw, &p->power_scale, p->power_rgb); /* StoredPhoton::new, photon_map.rs:23-30 */
                    p->power_scale *= inv_n;                               /* scale_power, photon_map.rs:61-64 */
                }
                stored++;
            }
            if (choice == 0) break;
            if (choice == 1) { /* TravellingPhoton::scatter (photon_map.rs:144-151) */
                double avg = (dr.x + dr.y + dr.z) / 3.0;
                ray.o = c.over_point; ray.d = cosine_sample(c.normalv, &g);
                power = vmul(vhad(power, dr), 1.0 / avg);
                kind = PK_DIFFUSE;
            } else if (choice == 2) { /* reflect (photon_map.rs:153-159) */
                ray.d = vreflect(ray.d, c.normalv); ray.o = c.over_point;
                if (kind == PK_DIRECT) kind = PK_CAUSTIC;
            } else { /* refract (photon_map.rs:161-172) */
                double nd[3], dd[3] = {ray.d.x, ray.d.y, ray.d.z}, nn[3] = {c.normalv.x, c.normalv.y, c.normalv.z};
                if (!orc_photon_refract(dd, nn, c.n1, c.n2, nd)) break;
                ray.o = c.under_point; ray.d = from3(nd);
                if (kind == PK_DIRECT) kind = PK_CAUSTIC;
            }
        }
    }
    free(lw);
    ctx_done(&cx);
    if (n_rays) *n_rays = rays;
    if (n_truncated) *n_truncated = truncated;
    return result < 0 ? result : stored;
}
