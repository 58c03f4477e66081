/* rt_oracle.h — CPU oracle: an f64 restatement of the reference's ray-casting path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker
 * (or, for cpu_baseline, as the thing timed on the host cores).  raytracing_amd/ never links,
 * imports or falls back to it.
 *
 * Pinned by: the reference's own known-answer unit tests (rust/src/{world,ray,materials,camera,
 * pattern,shapes/..}.rs `#[cfg(test)]` blocks, transcribed in tests/test_oracle_kat.py) and the
 * Scheme twin's deterministic renders scheme/chapterNN.ppm (tests/golden/, tests/test_golden_images.py).
 * The reference itself (Rust; no rustc/cargo here, crates not vendored) cannot be built or run in
 * this environment, so there is no oracle/_ref.
 *
 * Structure follows the reference on purpose (per-ray intersection lists, global sort, recursion):
 *   camera.rs:120-153,250-310   world.rs:111-180   ray.rs:26-35,72-219
 *   materials.rs:239-336,371-412   lights.rs:89-95,153-160,230-243,292-294,353-362,438-453
 *   shapes/mod.rs:108-124,252-322,399-477   sphere.rs:47-71  plane.rs:35-52  cube.rs:33-85
 *   cylinder.rs:36-128  cone.rs:36-148  triangle.rs:33-45,141-143,233-265  csg.rs:26-32,97-129
 *   aabb.rs:45-141  pattern.rs:7-35,51-53  color.rs:53-68  photon_map.rs:15-72,224-444 (the map and its gather; not the tracing of photons)
 * Randomness: the reference draws from rand::thread_rng (OS-seeded, unreproducible; "parity
 * unpinned" for sample positions).  Both this oracle and the HIP path draw from one seeded
 * Philox4x32-10 stream, counter = (pixel, sample, draw), and replace rand_distr's rejection
 * samplers UnitSphere / UnitDisc by fixed two-draw inverse-CDF maps of the same distribution.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rtmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* flags */
#define ORC_SCHEME_N1N2 1 /* Scheme twin's 2-argument prepare-computations (scheme/raytrace/world.sld:76):
                             the container list holds only the hit -> n1 = 1, n2 = ior(hit object).   */

/* Camera::render (camera.rs:155-161) over rows y with y % row_stride == row_offset (row_stride 1 =
 * whole frame).  rgb_out: hsize*vsize*3 doubles (rows not rendered stay untouched).  stats: RTMI_N_STATS. */
int orc_render(const rtmi_scene_desc *scene, const rtmi_camera *cam, const rtmi_render_opts *opts,
               int n_threads, int flags, int row_stride, int row_offset, double *rgb_out,
               uint64_t *stats);

/* Geometry::intersect in object space (local_intersect); returns count (<= 4), ts ascending as the
 * reference returns them; uv gets (u,v) per intersection for triangles. */
int orc_local_intersect(const rtmi_leaf *leaf, const double *origin, const double *dir, double *ts,
                        double *uv);
/* Geometry::normal_at in object space. */
void orc_local_normal(const rtmi_leaf *leaf, const double *point, double u, double v, double *n);
/* Shape::normal_at (shapes/mod.rs:261-265): world point -> world normal. */
void orc_normal_at(const rtmi_leaf *leaf, const double *world_point, double u, double v, double *n);
/* World::intersect / intersect_shadow (world.rs:145-170): sorted; returns count (may exceed cap). */
int orc_world_intersect(const rtmi_scene_desc *scene, const double *origin, const double *dir,
                        int shadow_only, int cap, double *ts, int32_t *leaves);
/* SceneItem::intersect on one node (shapes/mod.rs:108-115). */
int orc_node_intersect(const rtmi_scene_desc *scene, int node, const double *origin,
                       const double *dir, int cap, double *ts, int32_t *leaves);
/* World::color_at (world.rs:115-118); returns 1 on hit, 0 on None (rgb = BLACK). */
int orc_color_at(const rtmi_scene_desc *scene, const double *origin, const double *dir,
                 int remaining, int flags, double *rgb);
/* hit + prepare_computations (+ is_shadowed for light 0) of the closest hit. */
int orc_hit_record(const rtmi_scene_desc *scene, const double *origin, const double *dir, int flags,
                   rtmi_hit_record *rec);
/* The unit tests' idiom: an explicit intersection list xs (ts/leaves, sorted by the callee like
 * `intersections![]`), comps = xs[hit_index].prepare_computations(ray, xs), then
 * what = 0: World::shade_hit, 1: Phong::reflected_color, 2: Phong::refracted_color. */
int orc_shade_explicit(const rtmi_scene_desc *scene, const double *origin, const double *dir,
                       int n_xs, const double *ts, const int32_t *leaves, const double *uvs,
                       double hit_t, int hit_leaf, int remaining, int what, int flags, double *rgb,
                       rtmi_hit_record *rec);
/* Phong::lighting (materials.rs:246-288) with light.incoming_at(point). */
void orc_lighting(const rtmi_material *m, const double *surface_color, const rtmi_light *light,
                  const double *point, const double *eyev, const double *normalv, int in_shadow,
                  double *rgb);
/* Light::incoming_at: kind_out 0 = Ray, 1 = Omni, 2 = NoLight; u1,u2 are the two uniforms an area
 * light would draw. */
void orc_incoming_at(const rtmi_light *light, const double *point, double u1, double u2,
                     int *kind_out, double *origin, double *direction, double *color);
double orc_schlick(const double *eyev, const double *normalv, double n1, double n2);
/* Shape::pattern_at (shapes/mod.rs:267-269, pattern.rs:51-53). */
void orc_pattern_at(const rtmi_pattern *p, const double *leaf_inv, const double *world_point,
                    double *rgb);
/* Camera::ray_for_pixel (camera.rs:120-153) on the seeded stream. */
void orc_camera_ray(const rtmi_camera *cam, uint64_t seed, int px, int py, int sample,
                    double *origin, double *dir);
int orc_csg_eval(int op_node_kind, int lhit, int inl, int inr); /* csg.rs:26-32 */
/* Aabb::intersect (aabb.rs:45-70): returns 1 and fills tmin,tmax, or 0. */
int orc_aabb_intersect(const double *bmin, const double *bmax, const double *origin,
                       const double *dir, double *tmin, double *tmax);
/* Philox4x32-10 block; uniform k of (pixel, sample). */
void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out4);
double orc_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k);
/* Color::to_u8 (color.rs:53-60). */
/* Photon map (photon_map.rs): compute_number_of_left_children (423-444), compress / decompress_color (32-47), the index
 * tree PhotonMap::balance leaves (252-313; lo / hi: the extent it starts from), PhotonMap::from_vec + find_nearest
 * (230-243, 315-364; found: indices into `photons`, returns how many). */
int64_t orc_left_children(int64_t n);
void orc_compress_color(const double *c, float *scale, uint8_t *rgb);
void orc_decompress_color(float scale, const uint8_t *rgb, double *c);
void orc_photon_balance(const rtmi_photon *photons, int64_t n, const double *lo, const double *hi, int64_t *tree);
int orc_find_nearest(const rtmi_photon *photons, int64_t n, double max_search_radius, int n_neighbors, const double *p,
                     int64_t *found, double *max_sq);

/* World::compute_photon_map's tracing half (world.rs:186-273; materials.rs:414-464; photon_map.rs:144-172; the lights'
 * emit_photon): the number of photons stored (out: the first min(that, capacity), in the reference's order), -1 when no
 * light has power (world.rs:212 panics), -2 - leaf when a material's absorption probability is negative
 * (materials.rs:443-445 panics).  Draws: the Philox stream rtmi_trace_photons documents. */
int64_t orc_trace_photons(const rtmi_scene_desc *scene, const rtmi_photon_opts *opts, rtmi_photon *out, int64_t capacity,
                          int64_t *n_rays, int64_t *n_truncated);
void orc_photon_reflect(const double *dir, const double *normv, double *out);
int orc_photon_refract(const double *dir, const double *normv, double n1, double n2, double *out);
void orc_cosine_sample(const double *dir, uint64_t seed, uint32_t index, double *out);

void orc_to_u8(const double *rgb, int64_t n_values, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
