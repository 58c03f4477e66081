/* rtmi.h — C ABI of the MI355X ray-casting core.
 *
 * This is the drop-in boundary for the reference's per-pixel hot path
 *     impl Camera { pub fn render(&self, world: &World) -> Canvas }   rust/src/camera.rs:155-161
 *     pub fn trace_pixels(&self, world, cb) -> Vec<(u32,u32,Color)>    rust/src/camera.rs:250-265
 *     impl World  { pub fn trace / color_at(&self, ray, remaining) }   rust/src/world.rs:111-118
 * The reference has no FFI of its own; a host front end (Rust `World/Shape/Phong/Camera`, or the
 * C++ mirror in include/rtmi.hpp) flattens its scene tree into the plain structs below and calls
 * these entry points.  All pointers are caller-owned for the duration of the call; every function
 * returns RTMI_OK (0) or a negative error code and never throws; rtmi_last_error() gives the text.
 *
 * All scene numbers cross the boundary as f64 (the reference is f64 everywhere, tuple.rs:18); the
 * library converts to the arithmetic type selected in rtmi_render_opts.precision.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_ABI_VERSION 3
#define RTMI_EPSILON 1e-5 /* rust/src/approx_eq.rs:13 */

/* ---- error codes --------------------------------------------------------------------------- */
enum {
    RTMI_OK = 0,
    RTMI_ERR_INVALID = -1,     /* malformed descriptor (bad index, null pointer, bad enum)      */
    RTMI_ERR_UNSUPPORTED = -2, /* valid scene that exceeds a documented device limit             */
    RTMI_ERR_DEVICE = -3,      /* HIP runtime error, or no GPU present                           */
    RTMI_ERR_NOMEM = -4
};

/* ---- leaf primitives: the reference's `Shape` + `Geometry` (rust/src/shapes/mod.rs:218-225) -- */
enum {
    RTMI_SPHERE = 0,          /* shapes/sphere.rs:47-71   */
    RTMI_PLANE = 1,           /* shapes/plane.rs:35-52    */
    RTMI_CUBE = 2,            /* shapes/cube.rs:33-85     */
    RTMI_CYLINDER = 3,        /* shapes/cylinder.rs:36-128 */
    RTMI_CONE = 4,            /* shapes/cone.rs:36-148    */
    RTMI_TRIANGLE = 5,        /* shapes/triangle.rs:24-83 */
    RTMI_SMOOTH_TRIANGLE = 6  /* shapes/triangle.rs:91-155 */
};

typedef struct rtmi_leaf {
    int32_t kind;        /* RTMI_SPHERE ...                                                     */
    int32_t material;    /* index into rtmi_scene_desc.materials                                */
    int32_t closed;      /* cylinder / cone caps (cylinder.rs:18)                               */
    int32_t reserved;
    double inv[16];      /* Shape::inv_cumulative_transform as Matrix::into_flat gives it:
                            16 row-major f64, world -> object (matrix.rs:166-187)               */
    double minimum;      /* cylinder / cone truncation, +-inf allowed (cylinder.rs:16-17)       */
    double maximum;
    double p1[3], p2[3], p3[3]; /* triangle corners in object space (triangle.rs:24-31)         */
    double n1[3], n2[3], n3[3]; /* smooth-triangle vertex normals (triangle.rs:98-100)          */
} rtmi_leaf;

/* ---- scene tree: the reference's `SceneItem` enum (shapes/mod.rs:54-60) --------------------- */
enum {
    RTMI_NODE_LEAF = 0,             /* SceneItem::Primitive : first = leaf index                 */
    RTMI_NODE_GROUP = 1,            /* SceneItem::Compound  : children[first .. first+count)     */
    RTMI_NODE_BOUNDED = 2,          /* SceneItem::Bounded   : same, gated by aabb (mod.rs:471-477) */
    RTMI_NODE_CSG_UNION = 3,        /* SceneItem::CsgPair   : first = left node, count = right node */
    RTMI_NODE_CSG_INTERSECTION = 4, /*   (csg.rs:18-32)                                           */
    RTMI_NODE_CSG_DIFFERENCE = 5
};

typedef struct rtmi_node {
    int32_t kind;
    int32_t first;
    int32_t count;
    int32_t reserved;
    double aabb_min[3]; /* world-space box exactly as SceneItem::update_aabb left it            */
    double aabb_max[3]; /* (shapes/mod.rs:153-160,461-469). Only BOUNDED nodes are gated by it
                           in the reference; an inverted box (min > max) means "never finalized". */
} rtmi_node;

/* ---- materials: `Phong` (materials.rs:53-63) and `Pattern` (pattern.rs:7-67) ----------------- */
enum {
    RTMI_PATTERN_STRIPE = 0,   /* pattern.rs:7-9   */
    RTMI_PATTERN_GRADIENT = 1, /* pattern.rs:11-13 */
    RTMI_PATTERN_RING = 2,     /* pattern.rs:15-23 */
    RTMI_PATTERN_CHECKERS = 3, /* pattern.rs:25-35 */
    RTMI_PATTERN_XYZ = 4       /* the reference tests' `test_pattern`: color(p.x,p.y,p.z) (pattern.rs:90-92) */
};

typedef struct rtmi_pattern {
    int32_t kind;
    int32_t reserved;
    double a[3], b[3];
    double inv[16]; /* Pattern::inv_transform, row-major (pattern.rs:40,55-57) */
} rtmi_pattern;

typedef struct rtmi_material {
    double color[3]; /* SurfaceColor::Flat; ignored when pattern >= 0 */
    int32_t pattern; /* index into patterns, or -1                    */
    int32_t reserved;
    double emissive, diffuse, specular, shininess, reflective, transparency, refractive_index;
} rtmi_material;

/* ---- lights: `Light::incoming_at` (lights.rs) ------------------------------------------------ */
enum {
    RTMI_LIGHT_POINT = 0,           /* lights.rs:89-95   */
    RTMI_LIGHT_AMBIENT = 1,         /* lights.rs:292-294 -> IncomingLight::Omni */
    RTMI_LIGHT_SPHERE = 2,          /* lights.rs:353-362 */
    RTMI_LIGHT_DISC = 3,            /* lights.rs:438-453 */
    RTMI_LIGHT_REALISTIC_POINT = 4, /* lights.rs:153-160 */
    RTMI_LIGHT_BEAM = 5             /* lights.rs:230-243 */
};

typedef struct rtmi_light {
    int32_t kind;
    int32_t reserved;
    double position[3];
    double intensity[3];
    double radius;       /* SphereLight                                      */
    double normal[3];    /* DiscLight::normal                                */
    double width[3];     /* DiscLight::width  (already scaled by the radius) */
    double height[3];    /* DiscLight::height                                */
    double up[3];        /* Beam::up                                         */
    double right[3];     /* Beam::right                                      */
    double direction[3]; /* Beam::direction = normalize(right x up)          */
} rtmi_light;

/* ---- flattened `World` (world.rs:14-23 after finalize_scene, world.rs:98-105) ---------------- */
/* A stored photon as photon_map.rs:15-20 holds it (StoredPhoton::new, then scale_power): position
 * and direction (MINUS the direction of travel, TravellingPhoton::store, photon_map.rs:140-142) rounded to f32, the power as an f32 scale and three bytes
 * (compress_color / decompress_color, photon_map.rs:32-47: channel = byte * scale / 255).      */
typedef struct rtmi_photon {
    float position[3];
    float direction[3];
    float power_scale;
    uint8_t power_rgb[3];
    uint8_t reserved;
} rtmi_photon; /* 32 bytes */
#define RTMI_MAX_NEAREST 128 /* the device keeps the candidates of an estimate per lane */

typedef struct rtmi_scene_desc {
    int32_t abi_version; /* RTMI_ABI_VERSION */
    int32_t max_depth;   /* World::max_reflection_depth (world.rs:61: 10; the Scheme twin uses 5) */

    int32_t n_leaves;
    int32_t n_nodes;
    const rtmi_leaf *leaves;
    const rtmi_node *nodes;

    int32_t n_children;
    int32_t n_top;
    const int32_t *children;        /* node indices referenced by GROUP / BOUNDED nodes            */
    const int32_t *top_nodes;       /* World::objects in insertion order                          */
    const int32_t *top_cast_shadow; /* SceneItem::cast_shadow of each top-level item
                                       (world.rs:158-164, shapes/mod.rs:117-124)                   */
    int32_t n_materials;
    int32_t n_patterns;
    const rtmi_material *materials;
    const rtmi_pattern *patterns;

    int32_t n_lights;
    int32_t reserved;
    const rtmi_light *lights;       /* World::lights in insertion order                            */

    /* World::photon_map (world.rs:18,186-209) for the gather of Phong::shade_hit (materials.rs:376-384).
     * The photons as compute_photon_map collected them (world.rs:195-201), in any order: the library
     * balances them into the reference's kd-tree itself (PhotonMap::from_vec, photon_map.rs:230-313).
     * rtmi_trace_photons (below) produces such an array.  n_photons == 0 (or photon_map_enabled() false,
     * world.rs:124-128): no map, the branch is not taken.                                          */
    int64_t n_photons;
    const rtmi_photon *photons;
    int32_t n_nearest;              /* the usize next to the map (world.rs:18): photons per estimate, 1 .. RTMI_MAX_NEAREST */
    int32_t no_direct_illumination; /* != 0: World::enable_direct_illumination(false) (world.rs:74-76): the
                                       light loop of materials.rs:386-397 is skipped                 */
    double max_search_radius;       /* PhotonMap::set_max_search_radius (photon_map.rs:244-246);
                                       <= 0: unlimited (the INFINITY a new map has, photon_map.rs:241) */
} rtmi_scene_desc;

/* ---- `Camera` (camera.rs:14-32) --------------------------------------------------------------- */
typedef struct rtmi_camera {
    int32_t hsize, vsize;
    double inv[16]; /* Camera::inv_transform, row-major */
    double pixel_size, half_width, half_height; /* camera.rs:36-48 */
    double allowed_standard_error;              /* camera.rs:59, default 0.1 */
    int32_t min_samples;                        /* camera.rs:60, default 5   */
    int32_t reserved;
    double focal_distance;                      /* camera.rs:61 */
    double aperture_size;                       /* camera.rs:62 */
} rtmi_camera;

/* ---- render options (no counterpart in the reference: it uses thread_rng and f64) ------------ */
enum { RTMI_PRECISION_F32 = 0, RTMI_PRECISION_F64 = 1 };

typedef struct rtmi_render_opts {
    uint64_t seed;       /* key of the counter-based RNG (Philox4x32-10); counter = (pixel, sample, draw) */
    int32_t max_samples; /* cap on samples per pixel; 0 = unbounded like camera.rs:290-299         */
    int32_t precision;   /* RTMI_PRECISION_F32 (the fast path) or _F64 (validation mode)          */
    int32_t rank;        /* row-tile sharding: this process renders the tile rows t with          */
    int32_t nranks;      /*   t % nranks == rank; nranks <= 1 renders the whole frame             */
    int32_t tile_rows;   /* rows per tile, a multiple of 8; 0 = default (8)                       */
    int32_t reserved;
} rtmi_render_opts;

/* Scene-query counters ("rays" of the Mrays/s metric), one u64 each.                            */
enum {
    RTMI_STAT_PRIMARY = 0,    /* Camera rays given to World::trace (camera.rs:274-296)            */
    RTMI_STAT_REFLECTION = 1, /* color_at calls from reflected_color (materials.rs:300-304)       */
    RTMI_STAT_REFRACTION = 2, /* color_at calls from refracted_color (materials.rs:327-331)       */
    RTMI_STAT_SHADOW = 3,     /* is_shadowed queries with a Ray light (world.rs:174)              */
    RTMI_STAT_SAMPLES = 4,    /* multisample iterations (== primary rays)                         */
    RTMI_STAT_LEAF_TESTS = 5, /* primitive local_intersect evaluations                            */
    RTMI_STAT_NODE_VISITS = 6,/* Group / BoundingGroup / CSG nodes entered                        */
    RTMI_STAT_OVERFLOW = 7,   /* rays that hit a fixed device capacity (result may be wrong)      */
    RTMI_STAT_FLOPS = 8,      /* algorithmic FLOPs of the events above, priced with the constants
                                 of DESIGN.md "FLOP accounting"                         */
    RTMI_STAT_AABB_TESTS = 9, /* slab tests                                                        */
    RTMI_STAT_DISCARDED = 10, /* samples computed ahead of the stopping rule and thrown away (never
                                 counted in any of the above; the oracle reports 0)                 */
    RTMI_STAT_FLOPS_SECOND = 11, /* the part of RTMI_STAT_FLOPS that the second-pass kernels counted (samples taken from
                                    the work queues: resample_kernel, units_kernel via sum_up_kernel); the rest is
                                    render_kernel's.  Device path only (the oracle reports 0).                  */
    RTMI_N_STATS = 12
};

typedef struct rtmi_stats {
    uint64_t counter[RTMI_N_STATS];
    double kernel_ms; /* device time of the render kernel(s), HIP events (host entry points only) */
} rtmi_stats;

typedef struct rtmi_scene rtmi_scene; /* opaque; immutable after creation; device-resident */

/* Number of pixel rows rank `rank` of `nranks` owns for a frame of `vsize` rows. */
int32_t rtmi_rank_rows(int32_t vsize, int32_t rank, int32_t nranks, int32_t tile_rows);

/* Replaces what `World::new/add_item/finalize_scene` leave in memory (world.rs:56-105): validates
 * the descriptor, lays it out for the device and uploads it to the CURRENT HIP device.          */
int rtmi_scene_create(const rtmi_scene_desc *desc, rtmi_scene **out);
void rtmi_scene_destroy(rtmi_scene *scene);

/* Replaces `Camera::render(&self, &World) -> Canvas` (camera.rs:155-161).  rgb_out is a HOST
 * buffer of hsize*vsize*3 floats, row-major, un-clipped linear colour like Canvas::data
 * (canvas.rs:8-21).  Renders the whole frame on the current device (opts->rank/nranks ignored).
 * The frame is rendered into a device buffer kept with the scene and leaves through pinned staging
 * in 8 pieces, each copied into rgb_out by host threads while the next arrives (RTMI_HOST_COPY_THREADS
 * in the environment, default 4, at most 8); the buffers are created by a scene's first call
 * and freed with it.  Concurrent calls on ONE scene are allowed: the second caller uses buffers of
 * its own for that call.                                                                          */
int rtmi_render(const rtmi_scene *scene, const rtmi_camera *camera, const rtmi_render_opts *opts,
                float *rgb_out, rtmi_stats *stats_out);

/* Same path with everything device-resident and asynchronous, for callers that own the stream
 * and the memory (one process per GPU under torch.distributed, HIP graphs).  d_rgb_out receives
 * only the rows this rank owns, compacted in tile order: rtmi_rank_rows(...)*hsize*3 floats.
 * d_stats (may be NULL) points at RTMI_N_STATS device u64 that are ADDED to.  The work memory of
 * the adaptive second pass comes from a per-device pool of up to 4 slots of 2 GB; a slot belongs
 * to a stream only while that stream's last call is in flight.  RTMI_WORK_SLOTS_PREALLOC slots
 * (environment, read when the device's first scene is created; default 2) exist from scene
 * creation on: while no more streams than that render AT THE SAME TIME the call allocates
 * nothing; beyond that the first call that needs another slot makes it (one hipMalloc, once).
 * Any number of streams may render over the life of a process; a fifth stream that renders WHILE
 * four others have work in flight is refused with RTMI_ERR_UNSUPPORTED.  Work on one stream is
 * ordered, calls on different streams overlap on the GPU.  Enqueues a 32 KB memset and one to
 * nine kernels (camera.rs:273-291 first; 290-299 the others).                                     */
int rtmi_render_device(const rtmi_scene *scene, const rtmi_camera *camera,
                       const rtmi_render_opts *opts, float *d_rgb_out, uint64_t *d_stats,
                       void *hip_stream);

/* `Camera::render` on several GPUs of ONE process (the reference's only fan-out is rayon's, camera.rs:260-264; the
 * one-process-per-GPU form of this is rtmi_render_device with opts->rank/nranks plus an RCCL gather, bench.py).
 * scenes[r] is the same world created on the device that renders share r (rtmi_scene_create with that device
 * current; several or all may live on one device).  Share r = the row tiles t with t % n_scenes == r, tile_rows rows
 * each (0 = 8).  Every share is enqueued on a stream of its own before any is waited for; a share rendered on another
 * device than scenes[0]'s is copied there by hipMemcpyPeerAsync on that share's stream as soon as ITS kernels are done
 * (an event; the N - 1 copies run at the same time, each over its own xGMI link) into a staging buffer of its own, and
 * a small kernel per share puts the rows in place while the other copies are still in flight.  d_rgb_out: the WHOLE
 * frame, vsize*hsize*3 floats in memory of scenes[0]'s device.  The image is the single-GPU image bit for bit (RNG
 * keyed by global pixel and sample).  Blocking.  Streams, events, row and staging buffers are made at the first call
 * for a given (scenes[0], share, frame size) and kept with scenes[0] until it is destroyed: later calls allocate
 * nothing.  stats_out (may be NULL) = counters summed over the shares; kernel_ms = the WALL span of the whole call on
 * scenes[0]'s device (first enqueue to last row in place: kernels, peer copies and de-interleaving), not kernel time.
 * The cross-device path has not run on hardware yet (no multi-GPU box in any session so far): what is verified is the
 * same code with every share on one device (tests/test_gpu_multi.py) and, when two devices are visible, on two.     */
int rtmi_render_multi(const rtmi_scene *const *scenes, int32_t n_scenes, const rtmi_camera *camera,
                      const rtmi_render_opts *opts, int32_t tile_rows, float *d_rgb_out,
                      rtmi_stats *stats_out);

/* Diagnostic twin of rtmi_render_device for measurement (bench.py's roofline figures): the same launches with HIP
 * events recorded on `hip_stream` before the first pass (render_kernel), between it and the second pass
 * (resample / units / sum_up kernels) and after it; waits for the call to finish and returns the two durations in
 * milliseconds.  Synchronous; creates and destroys three events.  No counterpart in the reference.             */
int rtmi_render_device_timed(const rtmi_scene *scene, const rtmi_camera *camera,
                             const rtmi_render_opts *opts, float *d_rgb_out, uint64_t *d_stats,
                             void *hip_stream, float *ms_first_pass, float *ms_second_pass);

/* rtmi_render_device for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers
 * under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc / hipFree.  Same arguments,
 * same kernels, same image bit for bit (a pixel's result does not depend on where its samples are taken); the
 * work memory of the adaptive second pass is not the per-stream pool but ONE array in static device memory,
 * 96 MiB, which every GPU the library runs on holds until the process ends (its address is looked up with
 * hipGetSymbolAddress once per device, at rtmi_scene_create).  Calls no hipMalloc, hipFree or allocating
 * library routine; enqueues a 32 KB hipMemsetAsync on the array's counters and the same one to nine kernels.
 * Limit: the pixels of one call (this rank's rows x hsize) must fit one work queue, 524,288 pixels in f32 mode
 * (e.g. 960x540) and 349,525 in f64 mode; larger calls are refused with RTMI_ERR_UNSUPPORTED and a message -
 * render larger frames in row-tile shares (opts->rank / nranks), one call after the other.  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (different streams, or threads) RACE; order them.      */
int rtmi_render_device_noalloc(const rtmi_scene *scene, const rtmi_camera *camera,
                               const rtmi_render_opts *opts, float *d_rgb_out, uint64_t *d_stats,
                               void *hip_stream);

/* Replaces the refinement loop of `Camera::render_interactive` / `render_live` (camera.rs:163-248: pixels arrive while
 * the frame is still being sampled) for a headless caller: the frame pass by pass.  Every pass takes each pixel from
 * the sums the pass before left it with up to `max_samples` samples in all (0 = to the end of camera.rs:290-299) and
 * hands back the whole frame, so early passes are previews and the later ones cost only the samples they add: the
 * samples of all passes together are those of one rtmi_render call with the last cap, and the frame after a pass IS
 * the frame rtmi_render gives with that cap, bit for bit.  The handle owns device memory for the frame and 32 bytes of
 * state per pixel (56 in f64 mode); HOST buffer out; whole frame on the current device (opts->rank/nranks ignored).  */
typedef struct rtmi_progressive rtmi_progressive;
int rtmi_progressive_begin(const rtmi_scene *scene, const rtmi_camera *camera,
                           const rtmi_render_opts *opts, rtmi_progressive **out);
int rtmi_progressive_pass(rtmi_progressive *p, int32_t max_samples, float *rgb_out,
                          rtmi_stats *stats_out);
void rtmi_progressive_end(rtmi_progressive *p);

/* Replaces `Color::to_u8` over a canvas (color.rs:53-60, canvas.rs:101-107): clip to [0,1],
 * x255, round half away from zero.  Device buffers, asynchronous.                               */
int rtmi_tonemap_u8_device(const float *d_rgb, uint8_t *d_rgb8, int64_t n_values, void *hip_stream);

/* Replaces `World::color_at(&self, ray, remaining)` (world.rs:115-118) for a batch of arbitrary
 * rays; this is how the reference's known-answer tests reach the device path.  HOST buffers:
 * origins/directions n*3 doubles; remaining n ints (NULL = scene max_depth); outputs (any may be
 * NULL): rgb n*3 (BLACK on a miss, like camera.rs:276), and per ray an rtmi_hit_record.         */
typedef struct rtmi_hit_record {
    int32_t hit;       /* 0 = `None`                                                              */
    int32_t leaf;      /* index of the hit Shape in rtmi_scene_desc.leaves                        */
    int32_t inside;    /* IntersectionState::inside (ray.rs:80)                                   */
    int32_t reserved;
    double t;          /* Intersection::t                                                         */
    double point[3];   /* IntersectionState::point                                                */
    double normal[3];  /* IntersectionState::normalv (already flipped when inside)                */
    double n1, n2;     /* IntersectionState::n1/n2 (ray.rs:104-140)                               */
    double shadowed;   /* World::is_shadowed(light 0, over_point) as 0/1; -1 when there is no
                          Ray light 0 (world.rs:172-180)                                          */
} rtmi_hit_record;

int rtmi_trace_rays(const rtmi_scene *scene, const rtmi_render_opts *opts, int64_t n,
                    const double *origins, const double *directions, const int32_t *remaining,
                    double *rgb_out, rtmi_hit_record *hits_out);

/* Replaces `Camera::ray_for_pixel(px, py, randomize)` (camera.rs:120-153) for a batch of pixels;
 * sample 0 is the centred ray, samples >= 1 are jittered from the seeded stream.  HOST buffers. */
int rtmi_camera_rays(const rtmi_camera *camera, const rtmi_render_opts *opts, int64_t n,
                     const int32_t *px, const int32_t *py, const int32_t *sample,
                     double *origins_out, double *directions_out);

/* Replaces the tracing half of `World::compute_photon_map` (world.rs:186-201): n_emit times
 * `World::emit_photon` (world.rs:211-215: a light chosen by `Light::power`, its `emit_photon`,
 * lights.rs:104-111,169-176,252-261,371-380,462-473) and `World::trace_photon` (world.rs:217-273) with
 * `Phong::photon_hit` (materials.rs:414-464: store where the diffuse reflectance is above EPSILON, then
 * Russian roulette between absorb / `scatter` / `reflect` / `refract`, photon_map.rs:144-172), and
 * `scale_power(1 / n_emit)` (world.rs:200).  The photons come back as the reference collects them: in
 * order of emission, each photon's hits in order of its bounces (rayon's indexed collect keeps that order),
 * ready for rtmi_scene_desc.photons (the balancing half, PhotonMap::from_vec, happens there).
 * The reference draws from thread_rng; here the draws of photon i come from the Philox stream
 * (seed with its high word xor "PHOT", counter (i, 0, draw)), so a call is reproducible.  Where the
 * reference has no answer this library ends the photon: a path longer than max_bounces (the reference
 * loops for ever in a closed lossless scene; counted in n_truncated), a refraction beyond the critical
 * angle (photon_map.rs:165 takes the root of a negative number).  A material whose absorption
 * probability is negative (materials.rs:443-444 panics) gives RTMI_ERR_INVALID naming the leaf, and so
 * does a world whose lights all have power 0 (world.rs:212 panics).  `Material::transform_photon`
 * (materials.rs:30-36) is the default (None) for every material this library holds.
 * HOST buffers.  `out` receives the first min(n_stored, capacity) photons; n_stored is always the whole
 * count, so (NULL, 0) asks for the size (and runs the counting pass only).  A scene's own photon map,
 * if it has one, plays no part.                                                                       */
typedef struct rtmi_photon_opts {
    uint64_t seed;
    int64_t n_emit;          /* n_photons of compute_photon_map, 1 .. 2^31 - 1                          */
    int32_t precision;       /* RTMI_PRECISION_F32 / _F64                                              */
    int32_t enable_direct;   /* World::enable_direct_photon_map  (world.rs:64: false)                  */
    int32_t enable_diffuse;  /* World::enable_diffuse_photon_map (world.rs:65: true)                   */
    int32_t enable_caustic;  /* World::enable_caustic_photon_map (world.rs:66: true)                   */
    int32_t max_bounces;     /* 0 = RTMI_PHOTON_BOUNCES                                                */
    int32_t reserved;
    int64_t first;           /* this call traces the photons first .. first + count - 1 of the n_emit:  */
    int64_t count;           /*   a share of the map (count 0 = all from `first` on).  Photon i draws    */
                             /*   from its own stream and is scaled by 1 / n_emit, so the shares of any  */
                             /*   split, one after the other, ARE the whole call's photons: ranks trace  */
                             /*   disjoint shares and exchange the arrays, no other collective           */
} rtmi_photon_opts;
#define RTMI_PHOTON_BOUNCES 256
typedef struct rtmi_photon_result {
    int64_t n_stored;    /* photons the whole call stores                                              */
    int64_t n_rays;      /* World::intersect calls (world.rs:234)                                      */
    int64_t n_truncated; /* paths ended by max_bounces                                                 */
    double kernel_ms;    /* device time of the call's kernels (counting pass, scan, storing pass)     */
} rtmi_photon_result;
int rtmi_trace_photons(const rtmi_scene *scene, const rtmi_photon_opts *opts, rtmi_photon *out,
                       int64_t capacity, rtmi_photon_result *result);

/* Thread-local text of the last error returned on this thread ("" if none). */
const char *rtmi_last_error(void);

/* Library / device facts for reports: fills "rtmi <ver> gfx950 ..." into buf. */
int rtmi_describe(char *buf, int32_t buf_len);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
