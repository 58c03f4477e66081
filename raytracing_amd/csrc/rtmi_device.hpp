// rtmi_device.hpp — device-side ray casting for gfx950 (MI355X), templated on the arithmetic type.
//
// One lane = one pixel-sample stream; one wave64 = one 8x8 pixel tile.  The path is FP32 VALU- and
// divergence-bound (no MFMA).  R = float is the product path; R = double is a validation mode that
// reproduces the f64 oracle almost bit for bit.
//
// Reference semantics restated here (paths relative to rust/src/):
//   camera.rs:120-153,273-310   ray_for_pixel, multisample, color_variance_of_mean
//   world.rs:111-180            trace / color_at / intersect / intersect_shadow / is_shadowed
//   ray.rs:72-140,143-154,199-219  prepare_computations, n1/n2 container walk, hit, schlick
//   materials.rs:239-336,371-412   color_at, lighting, reflected_color, refracted_color, shade_hit
//   lights.rs                   incoming_at for all six light kinds
//   shapes/{sphere,plane,cube,cylinder,cone,triangle,csg}.rs  local_intersect + normal_at
//   shapes/mod.rs:252-265,319-322,399-405,471-477  ray/normal transforms, Group, BoundingGroup
//   aabb.rs:45-70               slab test          pattern.rs:7-35   the four patterns
//
// What is deliberately NOT like the reference (same results, different machine):
//   * no intersection lists: closest hit = running minimum over t >= 0; shadow = any hit in
//     [0, dist); the n1/n2 container walk becomes an order-free "open leaf with the latest entry"
//     search (see collect_n1n2);
//   * CSG trees are evaluated per root from one small sorted buffer with a bit-mask of in/out states,
//     walking each leaf's chain of CSG ancestors (csg.rs:97-112 applied bottom-up);
//   * recursion (materials.rs:290-336) is a loop with a bounded stack of pending refraction rays;
//   * a ray spawned on leaf L never re-hits L at its own starting point: L's root nearest to zero is
//     classified "behind" analytically instead of trusting the sign of a +-1e-5 quantity in f32.
#pragma once

#ifdef RTMI_HOST_EMUL
#include "hip_emul.h" // tests/emul: host stand-in used to run the lane-independent kernels under sanitizers (never shipped)
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "../../include/rtmi.h"

namespace rtmi_dev {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int MAX_DEPTH = 16;      // bound of the pending-ray stack (scene max_depth must be <= this)
constexpr int NODE_STACK = 64;     // traversal stack entries per lane
constexpr int CSG_BUF = 48;        // intersections buffered per CSG root and lane
constexpr int CSG_MAX_NODES = 32;  // CSG operator nodes under one root (2 state bits each in a u64)

// leaf.kind bit layout
constexpr uint32_t LEAF_KIND_MASK = 0xFFu;
constexpr uint32_t LEAF_CLOSED = 1u << 8;
constexpr uint32_t LEAF_CAST_SHADOW = 1u << 9;
constexpr uint32_t LEAF_FAST = 1u << 10; // f32 only: sphere under a similarity transform, tested in world space (fast[])
constexpr uint32_t LEAF_QUICK = 1u << 12; // f32 only, top-level leaves: the leaf's LDS record holds all that its intersection test and its normal need
                                          // (a LEAF_FAST sphere: centre and squared radius; a plane: row 1 of the world->object matrix)
constexpr uint32_t LEAF_XPAIR = 1u << 13; // f32 only, top-level leaves: a sphere under a general affine transform that the transformed-sphere pair table holds
constexpr uint32_t LEAF_BOUND = 1u << 11; // f32 only: fast[] = a world-space sphere (centre, radius^2, slightly inflated) that
                                          // contains the leaf; a ray whose line misses it has no roots (leaves below nodes only)

// node.flags
constexpr uint32_t NODE_CSGROOT = 1u << 0;
constexpr uint32_t NODE_GATED = 1u << 1;       // box is finite and proper: may be slab-tested
constexpr uint32_t NODE_CAST_SHADOW = 1u << 2; // some leaf below casts shadows
constexpr uint32_t NODE_AXIS_SHIFT = 4;         // bits 4-5: 1 + the axis along which the node's two children are sorted by centre (0: no order)
constexpr uint32_t NODE_TRIBVH = 1u << 6;      // f32 path: everything below this node (Groups / BoundingGroups of triangles, other primitives and CSG roots,
                                               // all with finite bounds) has been rebuilt as a BVH of its own (DBvh2): csg_count = the entry of its root
constexpr uint32_t NODE_PAIR_SHIFT = 8, NODE_PAIR_MASK = 7u;  // bits 8-10 of a CSG root: RTMI_NODE_CSG_* of its one operator if the root is that operator
                                                              // over two leaves of at most two roots each (spheres, cubes: csg_pair), else 0
constexpr uint32_t NODE_ALLTRI = 1u << 3;      // not a CSG root, every direct leaf is a triangle: wtris[csg_count + i] is leaf
                                               // leaf_first + i in world space (f32 path)

template <typename R> struct DLeaf {
    R inv[12];           // rows 0..2 of the world->object matrix
    R minimum, maximum;  // cylinder / cone
    R fast[4];           // LEAF_FAST: world-space centre and squared radius; LEAF_BOUND: bounding sphere
    uint32_t kind;       // kind | LEAF_CLOSED | LEAF_CAST_SHADOW | LEAF_FAST
    uint32_t material;
    uint32_t tri;        // index into tris (triangle kinds)
    uint32_t csg_parent; // innermost enclosing CSG operator, or NONE
};

template <typename R> struct DTri {
    R p1[3], e1[3], e2[3]; // triangle.rs:33-45
    R n1[3], n2[3], n3[3]; // flat triangles store normalize(e2 x e1) three times
};

template <typename R> struct DMaterial {
    R color[3];
    R emissive, diffuse, specular, shininess, reflective, transparency, ior;
    int32_t pattern;
    uint32_t pad;
};

template <typename R> struct DPattern {
    R inv[12];
    R a[3], b[3];
    int32_t kind;
    uint32_t pad;
};

template <typename R> struct DLight {
    R pos[3], intensity[3];
    R radius;
    R normal[3], width[3], height[3];
    R up[3], right[3], dir[3];
    R inv_up2, inv_right2;
    int32_t kind;
    uint32_t pad;
};

// A triangle leaf in world space (f32 product path): the ray parameter t and the barycentrics u, v are ratios of
// determinants, hence the same in object and world space; k = det of the world->object linear part turns the world-space
// determinant into the object-space one that triangle.rs:243 compares with EPSILON and whose sign the spawn-side rule uses.
struct DWTri { // 48 B
    float p1[3], e1[3], e2[3];
    float k;
    uint32_t smooth; // 1 for SmoothTriangle
    uint32_t pad;
};

// The device's own BVH over the triangles below a NODE_TRIBVH node (f32 path).  The reference's tree - median splits on
// centres down to ten triangles per BoundingGroup (shapes/mod.rs:489-563) - decides nothing about the image: the hit is
// the nearest root over the LEAF SET, and a box only ever rules triangles out.  So the triangles are re-split on the
// host (binned SAH, few triangles per leaf) into a binary tree whose nodes hold BOTH children's boxes in one 64-byte
// record, and a wave walks that tree TOGETHER (packet_tris): one shared stack, node and triangle records through
// scalar loads, every lane testing its own ray against the same box or triangle.  The rays of a wave - an 8x8 tile's
// primary rays, the samples of one pixel, shadow rays towards one light - are a narrow bundle, so the union of the
// nodes they need is close to what each needs alone; and a traversal without per-lane stacks, vector loads or
// divergent branches is what this machine runs well.
struct alignas(64) DBvh2 { // 64 B
    float lox[2], hix[2], loy[2], hiy[2], loz[2], hiz[2]; // the two children's bounds, a plane of both at a time (slab_pair)
    uint32_t child[2];                     // stack entries (ENT_BVH | node, ENT_TRIS | count - 1 << 26 | first triangle)
    uint32_t pad[2];
};
// entries (top two bits): a DBvh2 node, a run of triangles in btris, one primitive leaf (leaves[]), or - both bits clear -
// a scene node (in a DBvh2 tree: a CSG root)
constexpr uint32_t ENT_BVH = 0x80000000u, ENT_TRIS = 0xC0000000u, ENT_LEAF = 0x40000000u, ENT_KIND = 0xC0000000u;
constexpr uint32_t ENT_TRI_FIRST_BITS = 26, ENT_TRI_FIRST_MASK = (1u << ENT_TRI_FIRST_BITS) - 1u;
constexpr int BVH_LEAF_MAX = 8;   // triangles per DBvh2 leaf (3 bits of count - 1 are used of the 4 available)
constexpr int PACKET_STACK = 64;  // entries of the wave's shared stack (the host checks the trees' depth against it)

struct DNode { // 48 B
    float bmin[3];
    uint32_t leaf_first;
    float bmax[3];
    uint32_t leaf_count;
    uint32_t child_first, child_count;
    uint32_t flags;
    uint32_t csg_count; // CSGROOT: number of CSG operator nodes below (state bits used)
};

struct DCsg { // 16 B
    uint32_t op;       // RTMI_NODE_CSG_*
    uint32_t left_end; // leaves of this operator with index < left_end belong to its left child
    uint32_t parent;   // enclosing CSG operator or NONE
    uint32_t bit;      // index of this operator under its root (state bits 2*bit, 2*bit+1)
};

// What the ray's misses so far say about a CSG root's remaining leaves (the device's own shortcut; it changes no result).
// CsgOp::eval (csg.rs:26-32) lets a root of the RIGHT side of an Intersection or Difference through only while the ray is
// inside the left side, and one of the LEFT side of an Intersection only while inside the right: an operator other than
// Union whose left side gave the ray no root at all gives none itself, whatever its right side holds.  The record sits at
// the LAST leaf of such an operator's left side: if no leaf from `start` (the operator's first leaf) up to here had a root,
// go on at leaf `skip` - the end of the operator, or of the outermost enclosing operator it is the whole left side of and
// that is no Union either.  A die of chapter-16 is ((((((cube & sphere) - side) - side) ...): a ray that misses the cube
// is done with the die after one test instead of 23.
struct DGuard { uint32_t start, skip; }; // skip == 0: no record at this leaf
// A leaf below a CSG root as the FIRST pass of csg_root's leaf loop reads it (one 32-byte scalar load, the loads of a run of
// leaves all in flight together): the leaf's bounding sphere (f32 scenes; r2 < 0: none), its kind word, its guard record,
// and `run` - the leaves from this one up to and including the next that has a guard record (or the root's last), CSG_RUN
// at most.  Inside a run no jump can start, so whether a lane tests a leaf of it is known when the run begins.
struct alignas(32) DCsgLeaf {
    float c[3], r2;
    uint32_t kind;
    uint32_t gstart, gskip; // DGuard
    uint32_t run;
};
constexpr int CSG_RUN = 32; // (a bit per leaf in a 32-bit word)
// A leaf's CSG ancestors as two masks over the root's operator flags (bit 2b = inl, 2b + 1 = inr of operator b): the flags
// CsgOp::eval (csg.rs:26-32) needs clear / set for a root of this leaf to pass (see csg_root).
struct alignas(16) DChain { unsigned long long zero, one; };

// A photon of the map (photon_map.rs:15-20) as the device reads it: 32 bytes, the bytes of the compressed power and the
// flag PhotonMap::balance gave the photon (photon_map.rs:215-221) in one word.
enum { KD_LEAF = 1, KD_SPLIT_X = 2, KD_SPLIT_Y = 3, KD_SPLIT_Z = 4 };
struct alignas(16) DPhoton { // (what the search reads of a node - position and flag - is the record's first 16 bytes: one load)
    float pos[3];
    uint32_t rgbf; // r | g << 8 | b << 16 | kd flag << 24
    float dir[3], scale;
};
static_assert(sizeof(DPhoton) == 32, "DPhoton is 32 bytes");
constexpr int PHOTON_STACK = 64; // entries of the search's stack: two per level of a tree of at most 2^30 photons

template <typename R> struct DScene {
    const DLeaf<R> *leaves;
    const DTri<R> *tris;
    const DMaterial<R> *materials;
    const DPattern<R> *patterns;
    const DLight<R> *lights;
    const DNode *nodes;
    const DCsg *csgs;
    const DCsgLeaf *csg_leaf;  // one per leaf (those below CSG roots are filled in)
    const DChain *csg_chain;   // one per leaf
    const DPhoton *photons;    // World::photon_map in kd-tree order (node i: children 2i + 1, 2i + 2), or null
    uint32_t n_photons, n_nearest, no_direct, pad3;
    R photon_radius;           // PhotonMap::max_search_radius (+inf: unlimited)
    const uint32_t *leaf_orig; // device leaf -> index in rtmi_scene_desc.leaves
    const DWTri *wtris;        // one per entry of tris (f32 scene only, else null)
    const DBvh2 *bvh2;         // the NODE_TRIBVH trees (f32 scene only)
    const DWTri *btris;        // their triangles in leaf order, pad = the device leaf
    uint32_t n_bvh2, n_btris;  // (what packet_tris checks its entries against)
    uint32_t n_leaves, n_lights, n_nodes, max_depth;
    uint32_t root_leaf_count, root_child_first, root_child_count;
    uint32_t has_units;        // some DBvh2 tree holds primitives that are not triangles or CSG roots, or the scene has a CSG root at all (packet_tris<MODE, true>)
    // what stage_scene copies into LDS (f32 scene only, else null / 0)
    const uint32_t *quick;     // LDS_MATS_AT dwords: the image of the block's first two tables (quick records, pattern slots)
    uint32_t n_quick;          // min(root_leaf_count, LDS_QUICK)
    uint32_t n_lds_materials;  // all materials, or 0 if there are more than LDS_MATS
    uint32_t n_lds_lights;     // all lights, or 0 if there are more than LDS_LIGHTS
    uint32_t n_others;         // top-level leaves that are not in the pair tables (flat f32 scenes; else all of them)
    const uint32_t *pairs;     // LDS_PAIRS * PAIR_DW dwords: the image of the block's pair tables (quick_pairs)
    const uint32_t *others;    // n_others device leaf indices, ascending
    uint32_t pair_counts;      // pairs of the tables, a byte each: spheres | shadow-casting spheres << 8 | planes << 16 | shadow-casting planes << 24
    uint32_t real_counts;      // the leaves in them (a table of an odd number of leaves ends with a filler), same bytes
    uint32_t xpair_counts;     // the table of spheres under a general affine transform: pairs | shadow-casting pairs << 8 | leaves << 16 | shadow-casting leaves << 24
    uint32_t pad5;
};

// ---------------------------------------------------------------------------------------------
// The scene's small tables in LDS (f32 product path; "stage small scenes' shapes and transforms in LDS").
// Every wave (= workgroup) copies them once when its kernel starts.  A record that every lane reads at the same index
// comes back as one broadcast LDS read, about 64 cycles after issue and with the next record's read already in flight,
// where the scalar loads they replace cost two dependent trips to the scalar cache per leaf (kind, then the fields the
// kind needs), a few hundred cycles each time, which three waves per SIMD cannot cover.
//   quick records   the first LDS_QUICK top-level leaves (World::objects that are primitives), QUICK_DW dwords each:
//                   kind | flags, material, pattern slot, -, q[4].  LEAF_QUICK: q is all the intersection test and the
//                   normal need; other kinds go on to their full record in global memory.
//   pattern slots   for up to LDS_PATS of those leaves whose material has a pattern: a DPattern<float> whose matrix is
//                   pattern.inv x leaf.inv, i.e. world point -> pattern space in one step (shapes/mod.rs:268 and
//                   pattern.rs:52 folded on the host in f64), so that a patterned hit needs nothing from global memory
//   materials       DMaterial<float> as they are, if there are at most LDS_MATS
//   lights          DLight<float> as they are, if there are at most LDS_LIGHTS
// ---------------------------------------------------------------------------------------------
// The flat-scene kernels (World::objects all primitives) use it.  The tree-scene kernels do not: their time goes into
// node traversal, the top level is a leaf or two, and the second code path cost them registers they do not have (the
// compiler gave resample_kernel<float, true> 248 VGPRs = 2 waves per SIMD with it).
// The kind of world a kernel is compiled for (the template parameter TREE of everything below; `true` / `false` convert to
// the first two): KIND_FLAT World::objects are all primitives, KIND_TREE anything (Groups, CSG, meshes), KIND_FLATX a flat
// world that also has spheres under a general affine transform in the transformed-sphere table (a kernel of its own: the
// table's loop holds 24 more registers, which cost the other flat scenes 8 % when it was compiled into their kernel).
enum { KIND_FLAT = 0, KIND_TREE = 1, KIND_FLATX = 2 };
// (The tree kernels with the block staged as well - pair tables for the top-level spheres and planes, materials and lights
// from LDS - were measured in round 3: teapot 23.7 -> 25.5 ms, dice 196 -> 206 ms; the second code path costs them 16 to 28
// more spilled registers, and that outweighs the scalar loads it saves.)
template <typename R, int TREE> constexpr bool staged_v = sizeof(R) == 4 && TREE != KIND_TREE;
constexpr int LDS_QUICK = 16, LDS_PATS = 4, LDS_MATS = 16, LDS_LIGHTS = 4;
constexpr int QUICK_DW = 8;
constexpr int MAT_DW = (int)(sizeof(DMaterial<float>) / 4), LIGHT_DW = (int)(sizeof(DLight<float>) / 4), PAT_DW = (int)(sizeof(DPattern<float>) / 4);
constexpr int LDS_PATS_AT = (LDS_QUICK + 1) * QUICK_DW; // (the leaf loop fetches one record ahead: one spare slot)
constexpr int LDS_MATS_AT = LDS_PATS_AT + LDS_PATS * PAT_DW;
constexpr int LDS_LIGHTS_AT = LDS_MATS_AT + LDS_MATS * MAT_DW;
//   pair tables     the LEAF_QUICK leaves once more, two to a record and one kind to a table, as the intersection loops
//                   want them (quick_pairs): x0 x1 y0 y1 z0 z1 w0 w1 leaf0 leaf1 - -, i.e. the two centres / plane rows
//                   component by component, so that one packed instruction (v_pk_fma_f32 ...) works on both leaves.
//                   Order: spheres that cast shadows, other spheres, planes that cast shadows, other planes; a group of
//                   an odd number of leaves ends with a filler that no ray hits (radius^2 < 0 / a zero row).
//   transformed     spheres under a general affine transform (the squashed spheres chapter-07 builds its walls from) the same
//   spheres         way with their whole world->object matrix: m0a m0b m1a m1b ... m11a m11b leaf0 leaf1 - -; shadow casters
//                   first; a filler is the unit sphere 1e15 away.  (LEAF_XPAIR in the leaf's quick record.)
constexpr int LDS_PAIRS = 10, PAIR_DW = 12; // (16 leaves in four groups: at most 10 records)
//                   The table has an LDS array of its own (lds_xpairs), which only the KIND_FLATX kernels allocate: LDS is
//                   granted in 512-byte granules and 12 waves per CU leave 13,312 bytes each - the flat kernels' 12,560 fit,
//                   with 6 records of this table on top (13,232) they still do, with 8 they did not (11 waves, -7 %).
constexpr int LDS_XPAIRS = 6, XPAIR_DW = 28;
constexpr int LDS_PAIRS_AT = LDS_LIGHTS_AT + LDS_LIGHTS * LIGHT_DW;
constexpr int LDS_SCENE_DW = LDS_PAIRS_AT + LDS_PAIRS * PAIR_DW; // 644 dwords = 2.6 KB per wave
constexpr int PAIRS_IMAGE_DW = LDS_PAIRS * PAIR_DW + LDS_XPAIRS * XPAIR_DW; // what DScene::pairs points at
static_assert(sizeof(DMaterial<float>) % 4 == 0 && sizeof(DLight<float>) % 4 == 0 && sizeof(DPattern<float>) % 4 == 0 && LDS_PATS_AT % 4 == 0 && LDS_MATS_AT % 4 == 0 && LDS_PAIRS_AT % 4 == 0, "LDS scene block layout");
__device__ __forceinline__ uint32_t *lds_scene() {
    __shared__ __align__(16) uint32_t s_scene[LDS_SCENE_DW];
    return s_scene;
}
__device__ __forceinline__ uint32_t *lds_xpairs() {
    __shared__ __align__(16) uint32_t s_xpairs[LDS_XPAIRS * XPAIR_DW];
    return s_xpairs;
}
// Called once by every kernel that traces rays, before anything else, by all lanes of the wave.
template <typename R, int TREE> __device__ __forceinline__ void stage_scene(const DScene<R> &sc) {
    if constexpr (staged_v<R, TREE>) {
        uint32_t *s = lds_scene();
#ifdef RTMI_HOST_EMUL // (lanes run one after the other there: each fills the whole block)
        const uint32_t first = 0, step = 1;
#else
        const uint32_t first = threadIdx.x & 63u, step = 64;
#endif
        const uint32_t nq = (uint32_t)LDS_MATS_AT, nm = sc.n_lds_materials * (uint32_t)MAT_DW, nl = sc.n_lds_lights * (uint32_t)LIGHT_DW;
        const uint32_t *m = reinterpret_cast<const uint32_t *>(sc.materials), *l = reinterpret_cast<const uint32_t *>(sc.lights);
        for (uint32_t i = first; i < nq; i += step) s[i] = sc.quick[i];
        for (uint32_t i = first; i < nm; i += step) s[LDS_MATS_AT + i] = m[i];
        for (uint32_t i = first; i < nl; i += step) s[LDS_LIGHTS_AT + i] = l[i];
        for (uint32_t i = first; i < (uint32_t)(LDS_PAIRS * PAIR_DW); i += step) s[LDS_PAIRS_AT + i] = sc.pairs[i];
        if constexpr (TREE == KIND_FLATX) {
            uint32_t *x = lds_xpairs();
            for (uint32_t i = first; i < (uint32_t)(LDS_XPAIRS * XPAIR_DW); i += step) x[i] = sc.pairs[LDS_PAIRS * PAIR_DW + i];
        }
#ifndef RTMI_HOST_EMUL
        __syncthreads(); // (one wave per workgroup: its own writes, visible to all its lanes from here on)
#endif
    }
}

template <typename R> struct DCamera {
    R inv[12];
    R pixel_size, half_width, half_height, se2, focal_distance, aperture_size;
    int32_t hsize, vsize, min_samples, max_samples;
};

struct DRender {
    uint32_t seed_lo, seed_hi;
    int32_t rank, nranks, tile_rows;
    int32_t tiles_x, local_tile_rows8; // 8-row tile rows owned by this rank
    int32_t n_tiles;
    uint32_t first, group, ppw, n_visits; // first pass (render_kernel): samples before a pixel's fate is decided, lanes per pixel, pixels per wave visit = 64 / group, visits
    uint32_t keep_lanes, inlane_need, inlane_max, heap_slots; // (heap_slots: runs in the photon heap workspace, 0 = none) first pass: a lane keeps its pixel past `first` samples while at least keep_lanes lanes of its wave do, the
                                                        // pixel is predicted to need at most inlane_need more, and it has taken fewer than first + inlane_max
};

// ---------------------------------------------------------------------------------------------
// small vector algebra
// ---------------------------------------------------------------------------------------------
template <typename R> struct V3 { R x, y, z; };
template <typename R> __device__ __forceinline__ V3<R> mk(R x, R y, R z) { return {x, y, z}; }
template <typename R> __device__ __forceinline__ V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename R> __device__ __forceinline__ V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename R> __device__ __forceinline__ V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <typename R> __device__ __forceinline__ V3<R> operator*(V3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename R> __device__ __forceinline__ V3<R> had(V3<R> a, V3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename R> __device__ __forceinline__ R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename R> __device__ __forceinline__ V3<R> cross(V3<R> a, V3<R> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// f32: the hardware's 1-ulp v_rcp / v_sqrt / v_rsq instead of the ~10-instruction correctly rounded sequences
// (the path is VALU-bound and carries ~1e-7 relative error anyway); f64: IEEE division and sqrt, like the oracle.
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rcp_(double x) { return 1.0 / x; }
__device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double div_(double a, double b) { return a / b; }
__device__ __forceinline__ float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rsqrt_(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float abs_(float x) { return fabsf(x); }
__device__ __forceinline__ double abs_(double x) { return fabs(x); }
__device__ __forceinline__ float min_(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double min_(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double max_(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float floor_(float x) { return floorf(x); }
__device__ __forceinline__ double floor_(double x) { return floor(x); }
// tuple.rs:118 normalises by multiplying with the inverse length
template <typename R> __device__ __forceinline__ V3<R> normalize(V3<R> a) { return a * rsqrt_(dot(a, a)); }
template <typename R> __device__ __forceinline__ V3<R> ld3(const R *p) { return {p[0], p[1], p[2]}; }

// rows 0..2 of M times a point (w = 1) / a vector (w = 0) / M^T times a vector (matrix.rs:311-345)
template <typename R> __device__ __forceinline__ V3<R> xf_point(const R *m, V3<R> p) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
template <typename R> __device__ __forceinline__ V3<R> xf_vector(const R *m, V3<R> p) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z, m[8] * p.x + m[9] * p.y + m[10] * p.z};
}
template <typename R> __device__ __forceinline__ V3<R> xf_normal(const R *m, V3<R> p) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z, m[2] * p.x + m[6] * p.y + m[10] * p.z};
}

template <typename R> struct Ray { V3<R> o, d; };
__device__ __forceinline__ float u2f_(uint32_t x) { return __builtin_bit_cast(float, x); }

// Two floats that one packed instruction works on (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: both halves at the rate
// of one scalar-per-lane operation).  The host emulation has no such type: a struct with the same operators.
#ifdef RTMI_HOST_EMUL
struct f2 { float x, y; };
inline f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
inline f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
inline f2 operator*(f2 a, f2 b) { return {a.x * b.x, a.y * b.y}; }
inline f2 operator-(f2 a) { return {-a.x, -a.y}; }
#else
typedef float f2 __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 splat2(float a) { return mk2(a, a); }
__device__ __forceinline__ uint32_t f2u_(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(a, min(b, c)); } // v_min3_u32

template <typename R> struct Lim;
template <> struct Lim<float> { static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); } };
template <> struct Lim<double> { static __device__ __forceinline__ double inf() { return __builtin_huge_val(); } };

// ---------------------------------------------------------------------------------------------
// counter-based RNG: Philox4x32-10, counter = (pixel, sample, draw/4, tag), key = seed.
// The oracle (oracle/rt_oracle.c: orc_philox / orc_uniform) runs the identical integer code, so both
// sides draw identical 24-bit uniforms.
// ---------------------------------------------------------------------------------------------
struct Rng {
    uint32_t k0, k1, pixel, sample, draw;
    uint32_t cache[4];
    uint32_t cached_block;
};
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (two multiplies per multiplier: a single 64-bit product, v_mad_u64_u32, measured 0.5 - 1 % slower on whole frames)
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void rng_begin(Rng &g, uint32_t pixel, uint32_t sample) {
    g.pixel = pixel; g.sample = sample; g.draw = 0; g.cached_block = NONE;
}
template <typename R> __device__ __forceinline__ R rng_next(Rng &g) {
    uint32_t blk = g.draw >> 2;
    if (blk != g.cached_block) {
        philox4x32_10(g.k0, g.k1, g.pixel, g.sample, blk, 0x52544D49u, g.cache);
        g.cached_block = blk;
    }
    uint32_t lane = g.draw & 3u;
    uint32_t v = lane == 0 ? g.cache[0] : lane == 1 ? g.cache[1] : lane == 2 ? g.cache[2] : g.cache[3];
    g.draw++;
    return (R)(v >> 8) * (R)(1.0 / 16777216.0);
}

// The work counters of one lane.  Two homes (the choice is a constant of the kernel that owns the object, so every test
// of `in_lds` folds away once the callees are inlined):
//   registers  c[]: the trace / photon kernels, the second pass's per-wave totals;
//   LDS        one column of the wave's counter block ([slot][lane], stats_block): the sampling loops of the render
//              kernels.  Ten counters that live in registers through every traversal, shading and spawn are what the
//              register allocator pays for with spills (measured on the headline frame: leaving the counters out makes the
//              first pass 15 % faster, and their arithmetic is a third of that); an update in LDS is one ds_add_u32
//              with no result - no vector ALU slot, no register.
// A sample's FLOPs go to `fl` (32 bits: one full-rate addition with a literal) and from there to `flops` when the sample ends.
constexpr int STAT_SLOTS_FLAT = 7, STAT_SLOTS_TREE = 6, STAT_SLOTS_MAX = 10; // (slots in order of how seldom they count: a tree kernel keeps the per-step ones in registers)
__device__ __forceinline__ constexpr int stat_slot(int i) {
    return i == RTMI_STAT_PRIMARY ? 0 : i == RTMI_STAT_REFLECTION ? 1 : i == RTMI_STAT_REFRACTION ? 2 : i == RTMI_STAT_SHADOW ? 3 : i == RTMI_STAT_SAMPLES ? 4
         : i == RTMI_STAT_OVERFLOW ? 5 : i == RTMI_STAT_LEAF_TESTS ? 6 : i == RTMI_STAT_NODE_VISITS ? 7 : i == RTMI_STAT_AABB_TESTS ? 8 : i == RTMI_STAT_DISCARDED ? 9 : 99;
}
__device__ __forceinline__ void lds_add(uint32_t *p, uint32_t v) {
#ifdef RTMI_HOST_EMUL
    *p += v;
#else
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); // (result unused: ds_add_u32)
#endif
}
struct Stats {
    uint32_t c[RTMI_N_STATS]; // c[RTMI_STAT_FLOPS] is unused: flops are summed in 64 bits below
    uint32_t fl;              // the FLOPs of the sample (photon, probe ray) in hand (stats_sample_end)
    unsigned long long flops;
    uint32_t *lds;            // this lane's column of the wave's counter block, when in_lds
    int n_slots;              // slots of that block (counters beyond it stay in c[])
    bool in_lds;
#ifdef RTMI_CYCLES
    unsigned long long cyc[4]; // shader-clock cycles: 0 the tree walk (packet_scene), 1 csg_root's leaf loop, 2 its sort + filter, 3 the whole kernel
#endif
    __device__ __forceinline__ void add(int i, uint32_t v) {
        if (in_lds && stat_slot(i) < n_slots) lds_add(lds + stat_slot(i) * 64, v);
        else c[i] += v;
    }
    __device__ __forceinline__ void inc(int i) { add(i, 1u); }
    __device__ __forceinline__ void flag(int i) { // (RTMI_STAT_OVERFLOW: "happened", never a count that matters)
        if (in_lds && stat_slot(i) < n_slots) lds[stat_slot(i) * 64] = 1u;
        else c[i] = 1u;
    }
    __device__ __forceinline__ uint32_t get(int i) const { return (in_lds && stat_slot(i) < n_slots) ? lds[stat_slot(i) * 64] + c[i] : c[i]; }
    // registers as the home
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < RTMI_N_STATS; ++i) c[i] = 0;
        fl = 0; flops = 0; lds = nullptr; n_slots = 0; in_lds = false;
#ifdef RTMI_CYCLES
        cyc[0] = cyc[1] = cyc[2] = cyc[3] = 0;
#endif
    }
    // LDS as the home: `block` = the wave's [n][64] block, zeroed here (this lane's column)
    __device__ __forceinline__ void zero_in_lds(uint32_t *block, int n, int lane) {
        zero();
        lds = block + lane; n_slots = n; in_lds = true;
#pragma unroll
        for (int k = 0; k < STAT_SLOTS_MAX; ++k)
            if (k < n) lds[k * 64] = 0u;
    }
};
template <int TREE> __device__ __forceinline__ uint32_t *stats_block() { // (one wave per workgroup)
    __shared__ uint32_t s_stats[TREE == KIND_TREE ? STAT_SLOTS_TREE : STAT_SLOTS_FLAT][64];
    return &s_stats[0][0];
}
__device__ __forceinline__ void stats_sample_end(Stats &st) { st.flops += st.fl; st.fl = 0; }
// -DRTMI_CYCLES=1 or 2 (diagnostic builds, tools/cycles.py): where a wave's time goes, summed per wave into the counters
// aabb_tests, leaf_tests, node_visits and flops INSTEAD of what those normally count.  1: the tree walk and its CSG parts;
// 2: the three kinds of intersection query (closest hit, shadow, containers).
#if defined(RTMI_CYCLES) && RTMI_CYCLES == 3 // 3: camera ray + Philox, prepare_computations + material + pattern, lighting
#define CYC3_BEGIN(v) const unsigned long long v = __builtin_readcyclecounter()
#define CYC3_END(st, i, v) (st).cyc[i] += __builtin_readcyclecounter() - v
#else
#define CYC3_BEGIN(v)
#define CYC3_END(st, i, v)
#endif
#if defined(RTMI_CYCLES) && RTMI_CYCLES == 4 // 4: the spawn logic after a hit (weights, schlick, directions, pending stack), popping a pending ray, the sample loop's own work
#define CYC4_BEGIN(v) const unsigned long long v = __builtin_readcyclecounter()
#define CYC4_END(st, i, v) (st).cyc[i] += __builtin_readcyclecounter() - v
#else
#define CYC4_BEGIN(v)
#define CYC4_END(st, i, v)
#endif
#if defined(RTMI_CYCLES) && RTMI_CYCLES == 5 // 5: csg_root's bound pass over a run, its full leaf tests, its sort + filter
#define CYC5_BEGIN(v) const unsigned long long v = __builtin_readcyclecounter()
#define CYC5_END(st, i, v) (st).cyc[i] += __builtin_readcyclecounter() - v
#else
#define CYC5_BEGIN(v)
#define CYC5_END(st, i, v)
#endif
#if defined(RTMI_CYCLES) && RTMI_CYCLES == 6 // 6: packet_tris - its box-pair steps, its triangle runs, its pops
#define CYC6_BEGIN(v) const unsigned long long v = __builtin_readcyclecounter()
#define CYC6_END(st, i, v) (st).cyc[i] += __builtin_readcyclecounter() - v
#else
#define CYC6_BEGIN(v)
#define CYC6_END(st, i, v)
#endif
#if defined(RTMI_CYCLES) && RTMI_CYCLES == 1
#define CYC_BEGIN(v) const unsigned long long v = __builtin_readcyclecounter()
#define CYC_END(st, i, v) (st).cyc[i] += __builtin_readcyclecounter() - v
#else
#define CYC_BEGIN(v)
#define CYC_END(st, i, v)
#endif

// -DRTMI_SKIP=<bits> (diagnostic builds only: the IMAGE IS WRONG): a stage is left out so that an A/B run against the product
// build shows what the stage costs on a whole frame (the subtractive twin of the RTMI_CYCLES probes; tools/skip_ab.sh).
//   1 pattern lookups   2 Phong::lighting   4 the work counters   8 the n1 / n2 container query   16 shadow queries
#ifndef RTMI_SKIP
#define RTMI_SKIP 0
#endif

// Algorithmic FLOP prices per event (mul/add/sub/div/sqrt/exp2/log2 = 1, FMA = 2, compares = 0),
// derived from the cited reference lines.  They price the REFERENCE's arithmetic, not this kernel's.
constexpr uint32_t F_XFORM = 33;       // ray -> object space (ray.rs:30-35)
constexpr uint32_t F_SPHERE_MISS = 21, F_SPHERE_HIT = 29; // sphere.rs:50-60
constexpr uint32_t F_PLANE = 3;        // plane.rs:36-41
constexpr uint32_t F_CUBE = 20;        // cube.rs:34-44,78-79
constexpr uint32_t F_CYL_OPEN = 30, F_CYL_CLOSED = 50; // cylinder.rs:42-49,80-104 (cone priced alike)
constexpr uint32_t F_TRI_DET = 15, F_TRI_U = 24, F_TRI_V = 39, F_TRI_FULL = 45; // triangle.rs:240-264
constexpr uint32_t F_AABB = 20;        // aabb.rs:46-70,134-135
constexpr uint32_t F_PREPARE = 75, F_SMOOTH_NORMAL = 9; // ray.rs:77-84, shapes/mod.rs:262-264,320-321
constexpr uint32_t F_LIGHT_RAY = 60, F_LIGHT_SHADOWED = 20, F_LIGHT_OMNI = 3; // lights.rs:90-94, materials.rs:256-285
constexpr uint32_t F_COMBINE = 15;     // materials.rs:399-410
constexpr uint32_t F_PHOTON_VISIT = 9, F_PHOTON_GATHER = 17, F_PHOTON_ESTIMATE = 8; // photon_map.rs:355 / materials.rs:379-381 / 382
constexpr uint32_t F_SCHLICK = 20;     // ray.rs:200-218
constexpr uint32_t F_REFRACT_DIR = 25; // materials.rs:319-326
constexpr uint32_t F_WEIGHT = 3;       // materials.rs:305,332
constexpr uint32_t F_PATTERN_XF = 33, F_PATTERN_EVAL = 6; // shapes/mod.rs:268, pattern.rs:52,7-35
constexpr uint32_t F_CAMERA = 50, F_CAMERA_LENS = 35;     // camera.rs:125-149

// ---------------------------------------------------------------------------------------------
// local_intersect per primitive kind (object space).  Roots come back ascending in t[0..n).
// ---------------------------------------------------------------------------------------------
template <typename R> __device__ __forceinline__ void cswap(R &a, R &b) { R lo = min_(a, b), hi = max_(a, b); a = lo; b = hi; }

// Roots of a t^2 + 2 hb t + c = 0 given disc4 = hb^2 - a c >= 0.
// double: literally the reference's expressions (sphere.rs:59-60, cylinder.rs:94,100).
// float : the cancellation-free form q = -(hb + sign(hb) sqrt(disc4)); roots q/a and c/q.
__device__ __forceinline__ void quad_roots(double a, double hb, double c, double disc4, double &t0, double &t1) {
    (void)c;
    double b = 2.0 * hb, sq = sqrt(4.0 * disc4);
    t0 = -(b + sq) / (2.0 * a);
    t1 = -(b - sq) / (2.0 * a);
}
__device__ __forceinline__ void quad_roots(float a, float hb, float c, float disc4, float &t0, float &t1) {
    float sq = sqrt_(disc4);
    float q = -(hb + copysignf(sq, hb));
    t0 = div_(q, a);
    t1 = (q != 0.0f) ? div_(c, q) : t0;
}

template <typename R>
__device__ __forceinline__ int local_intersect(const DLeaf<R> &g, uint32_t gkind, const DTri<R> *tris, V3<R> o, V3<R> d, R t[4], R &u_out, R &v_out,
                                               uint32_t &fl, R &det_out) {
    const R EPS = (R)RTMI_EPSILON;
    const R INF = Lim<R>::inf();
    uint32_t kind = gkind & LEAF_KIND_MASK;
    int n = 0;
    fl = 0;
    if (kind == RTMI_SPHERE) { // sphere.rs:47-63
        R a = dot(d, d), hb = dot(d, o);
        R disc4, c = dot(o, o) - (R)1;
        if constexpr (sizeof(R) == 4) {
            // |o - (hb/a) d|^2 avoids the b^2 - 4ac cancellation for distant origins
            V3<R> l = o - d * div_(hb, a);
            disc4 = a * ((R)1 - dot(l, l));
        } else {
            disc4 = hb * hb - a * c;
        }
        fl = F_SPHERE_MISS;
        if (!(disc4 < (R)0)) {
            quad_roots(a, hb, c, disc4, t[0], t[1]);
            cswap(t[0], t[1]);
            n = 2;
            fl = F_SPHERE_HIT;
        }
    } else if (kind == RTMI_PLANE) { // plane.rs:35-44
        fl = F_PLANE;
        if (!(abs_(d.y) < EPS)) { t[0] = div_(-o.y, d.y); n = 1; }
    } else if (kind == RTMI_CUBE) { // cube.rs:33-46,74-85
        bool ok = true;
        R tmin = -INF, tmax = INF;
        R oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            if (dd[ax] == (R)0 && (oo[ax] < (R)-1 || oo[ax] > (R)1)) ok = false;
            R a0 = div_((R)-1 - oo[ax], dd[ax]), a1 = div_((R)1 - oo[ax], dd[ax]);
            R lo = a0 > a1 ? a1 : a0, hi = a0 > a1 ? a0 : a1;
            tmin = ax == 0 ? lo : max_(tmin, lo);
            tmax = ax == 0 ? hi : min_(tmax, hi);
        }
        fl = F_CUBE;
        if (ok && tmin <= tmax) { t[0] = tmin; t[1] = tmax; n = 2; }
    } else if (kind == RTMI_CYLINDER || kind == RTMI_CONE) { // cylinder.rs:36-109, cone.rs:36-120
        const bool cone = kind == RTMI_CONE;
        fl = (gkind & LEAF_CLOSED) ? F_CYL_CLOSED : F_CYL_OPEN;
        R a = d.x * d.x + d.z * d.z, hb = o.x * d.x + o.z * d.z, c = o.x * o.x + o.z * o.z;
        if (cone) { a -= d.y * d.y; hb -= o.y * d.y; c -= o.y * o.y; } else { c -= (R)1; }
        R w0 = INF, w1 = INF, k0 = INF, k1 = INF; // wall roots, cap roots; +inf = absent
        bool caps = true;
        if (abs_(a) < EPS) {
            // cone only: one root when b != 0 (cone.rs:88-97); the reference's b is 2*hb
            if (cone && !(abs_((R)2 * hb) < EPS)) { w0 = div_(c, (R)-2 * ((R)2 * hb)); ++n; }
        } else {
            R disc4 = hb * hb - a * c;
            if (disc4 < (R)0) {
                caps = false; // cylinder.rs:89-90: no wall roots -> caps are skipped too
            } else {
                R t0, t1;
                quad_roots(a, hb, c, disc4, t0, t1);
                R y0 = o.y + t0 * d.y;
                if (y0 > g.minimum && y0 < g.maximum) { w0 = t0; ++n; }
                R y1 = o.y + t1 * d.y;
                if (y1 > g.minimum && y1 < g.maximum) { w1 = t1; ++n; }
            }
        }
        if (caps && (gkind & LEAF_CLOSED) && abs_(d.y) > EPS) { // intersect_caps
            R tc0 = div_(g.minimum - o.y, d.y), tc1 = div_(g.maximum - o.y, d.y);
            R r0 = cone ? g.minimum * g.minimum : (R)1, r1 = cone ? g.maximum * g.maximum : (R)1;
            R x0 = o.x + tc0 * d.x, z0 = o.z + tc0 * d.z;
            if (x0 * x0 + z0 * z0 <= r0) { k0 = tc0; ++n; }
            R x1 = o.x + tc1 * d.x, z1 = o.z + tc1 * d.z;
            if (x1 * x1 + z1 * z1 <= r1) { k1 = tc1; ++n; }
        }
        // ascending; absent slots (+inf) sink to the end
        t[0] = w0; t[1] = w1; t[2] = k0; t[3] = k1;
        cswap(t[0], t[1]); cswap(t[2], t[3]); cswap(t[0], t[2]); cswap(t[1], t[3]); cswap(t[1], t[2]);
    } else { // triangle.rs:233-265
        const DTri<R> &tr = tris[g.tri];
        V3<R> e1 = ld3(tr.e1), e2 = ld3(tr.e2);
        V3<R> dir_cross_e2 = cross(d, e2);
        R det = dot(e1, dir_cross_e2);
        det_out = det; // = d . (e2 x e1): the side of the geometric normal the ray travels towards
        fl = F_TRI_DET;
        if (!(abs_(det) < EPS)) {
            R f = rcp_(det);
            V3<R> p1_to_origin = o - ld3(tr.p1);
            R u = f * dot(p1_to_origin, dir_cross_e2);
            fl = F_TRI_U;
            if (!(u < (R)0 || u > (R)1)) {
                V3<R> origin_cross_e1 = cross(p1_to_origin, e1);
                R v = f * dot(d, origin_cross_e1);
                fl = F_TRI_V;
                if (!(v < (R)0 || (u + v) > (R)1)) {
                    t[0] = f * dot(e2, origin_cross_e1);
                    u_out = u; v_out = v;
                    n = 1;
                    fl = F_TRI_FULL;
                }
            }
        }
    }
    return n;
}

// Geometry::normal_at per kind (object space)
template <typename R>
__device__ __forceinline__ V3<R> local_normal(const DLeaf<R> &g, const DTri<R> *tris, V3<R> p, R u, R v) {
    const R EPS = (R)RTMI_EPSILON;
    uint32_t kind = g.kind & LEAF_KIND_MASK;
    if (kind == RTMI_SPHERE) return p;               // sphere.rs:65-67
    if (kind == RTMI_PLANE) return mk<R>(0, 1, 0);   // plane.rs:46-48
    if (kind == RTMI_CUBE) {                         // cube.rs:48-67
        R ax = abs_(p.x), ay = abs_(p.y), az = abs_(p.z);
        if (ax >= ay) return ax >= az ? mk<R>(p.x, 0, 0) : mk<R>(0, 0, p.z);
        return ay >= az ? mk<R>(0, p.y, 0) : mk<R>(0, 0, p.z);
    }
    if (kind == RTMI_CYLINDER || kind == RTMI_CONE) { // cylinder.rs:111-123, cone.rs:122-142
        R dist = p.x * p.x + p.z * p.z;
        if (dist < (R)1) {
            if (p.y >= g.maximum - EPS) return mk<R>(0, 1, 0);
            if (p.y <= g.minimum + EPS) return mk<R>(0, -1, 0);
        }
        if (kind == RTMI_CYLINDER) return mk<R>(p.x, 0, p.z);
        R s = sqrt_(dist);
        return mk<R>(p.x, p.y < (R)0 ? s : -s, p.z);
    }
    const DTri<R> &tr = tris[g.tri]; // triangle.rs:69-71,141-143
    return normalize(ld3(tr.n2) * u + ld3(tr.n3) * v + ld3(tr.n1) * ((R)1 - u - v));
}

// How a ray was spawned on its origin leaf (bits above the leaf index in Ray bookkeeping)
constexpr uint32_t SIDE_PLUS = 1u; // smooth triangles: the spawn point lies on the +(e2 x e1) side of the triangle's plane

// squared distance of the ray's line from a sphere's centre against its squared radius
template <typename R> __device__ __forceinline__ bool line_misses_sphere(const Ray<R> &ray, const R *sph) {
    const V3<R> oc = ray.o - ld3(sph);
    const R a = dot(ray.d, ray.d), hb = dot(ray.d, oc);
    const V3<R> l = oc - ray.d * (hb * rcp_(a));
    return dot(l, l) > sph[3];
}

// World-space intersection of one leaf: Shape::intersect (shapes/mod.rs:252-255).  When the ray was spawned on
// this very leaf, its root nearest to zero IS the spawn point (|t| ~ 1e-5) and its sign is known analytically:
//  * geometric normals (everything but smooth triangles): the spawn point was pushed off the surface along the
//    normal the ray leaves on, so that root is behind the origin;
//  * smooth triangles: the push is along the INTERPOLATED normal, so a ray may start on one side of the flat
//    triangle and head for the other (the reference's shadow-terminator artefact, which must be kept): the root is
//    in front iff the ray travels towards the plane, i.e. sign(d . n_g) differs from the side it started on.
// What a leaf that its bounding sphere ruled out is priced at: the reference's own (missing) test of it.
__device__ __forceinline__ uint32_t bound_reject_flops(uint32_t gkind) {
    const uint32_t kind = gkind & LEAF_KIND_MASK;
    return kind == RTMI_SPHERE ? F_SPHERE_MISS : kind == RTMI_CUBE ? F_CUBE : kind >= RTMI_TRIANGLE ? F_TRI_U : (gkind & LEAF_CLOSED) ? F_CYL_CLOSED : F_CYL_OPEN;
}
// BOUND_DONE: the caller has tested the leaf's bounding sphere already (csg_root's first pass)
template <typename R, bool BOUND_DONE = false>
__device__ __forceinline__ int leaf_roots(const DLeaf<R> &g, uint32_t gkind, const DTri<R> *tris, const Ray<R> &ray, bool is_origin, uint32_t origin_side,
                                          R t[4], R &u, R &v, Stats &st) {
    // (gkind = g.kind, handed in by callers that know it already - the top-level loop has it from the leaf's LDS record -
    // so that the record's other fields need not wait for it)
    const uint32_t kind = gkind & LEAF_KIND_MASK;
    uint32_t fl;
    R det = 0;
    int n = 0;
    if (sizeof(R) == 4 && (gkind & LEAF_FAST)) {
        // f32 product path: a sphere whose transform is a similarity is the world-space sphere |p - c|^2 = r^2; the
        // ray parameter t is the same in both spaces (ray.rs:30-35 does not renormalise), so these are the roots of
        // sphere.rs:47-63 at a third of the arithmetic.  Roots as closest approach -+ half chord (no cancellation).
        const V3<R> oc = ray.o - ld3(g.fast);
        const R a = dot(ray.d, ray.d), inv_a = rcp_(a), hb = dot(ray.d, oc);
        const V3<R> l = oc - ray.d * (hb * inv_a);
        const R disc4 = a * (g.fast[3] - dot(l, l));
        fl = F_SPHERE_MISS;
        if (!(disc4 < (R)0)) {
            const R tm = -hb * inv_a, hc = sqrt_(disc4) * inv_a;
            t[0] = tm - hc; t[1] = tm + hc;
            n = 2;
            fl = F_SPHERE_HIT;
        }
    } else if (sizeof(R) == 4 && !BOUND_DONE && (gkind & LEAF_BOUND) && !is_origin && line_misses_sphere(ray, g.fast)) {
        fl = bound_reject_flops(gkind); // no roots
    } else if (kind == RTMI_PLANE) { // plane.rs:35-44 needs the y row of the object-space ray only
        const R oy = g.inv[4] * ray.o.x + g.inv[5] * ray.o.y + g.inv[6] * ray.o.z + g.inv[7];
        const R dy = g.inv[4] * ray.d.x + g.inv[5] * ray.d.y + g.inv[6] * ray.d.z;
        fl = F_PLANE;
        if (!(abs_(dy) < (R)RTMI_EPSILON)) { t[0] = div_(-oy, dy); n = 1; }
    } else {
        V3<R> o = xf_point(g.inv, ray.o), d = xf_vector(g.inv, ray.d); // ray.rs:30-35 (direction not renormalised)
        n = local_intersect(g, gkind, tris, o, d, t, u, v, fl, det);
    }
    st.inc(RTMI_STAT_LEAF_TESTS);
    st.fl += F_XFORM + fl;
    if (is_origin && n > 0) {
        int k = 0;
        R best = abs_(t[0]);
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < n && abs_(t[i]) < best) { best = abs_(t[i]); k = i; }
        R forced = -max_(best, (R)1e-30);
        if (kind == RTMI_SMOOTH_TRIANGLE) {
            bool plus = (origin_side & SIDE_PLUS) != 0;
            bool towards = plus ? det < (R)0 : det > (R)0;
            if (towards) forced = -forced;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i == k) t[i] = forced;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// Aabb::intersect (aabb.rs:45-70,130-141) on a conservatively padded f32 box
// ---------------------------------------------------------------------------------------------
// inv_d: 1/d per axis (f32 path: computed once per ray; (lo - o) * (1/d) has the same infinities and NaNs as the division)
template <typename R> __device__ __forceinline__ bool slab(const DNode &nd, const Ray<R> &ray, V3<R> inv_d, R t_lo, R t_hi) {
    R o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z}, id[3] = {inv_d.x, inv_d.y, inv_d.z};
    R tmin = -Lim<R>::inf(), tmax = Lim<R>::inf();
    bool ok = true;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        R lo = (R)nd.bmin[ax], hi = (R)nd.bmax[ax];
        if (d[ax] == (R)0 && (o[ax] < lo || o[ax] > hi)) ok = false;
        R a0, a1;
        if constexpr (sizeof(R) == 4) { a0 = (lo - o[ax]) * id[ax]; a1 = (hi - o[ax]) * id[ax]; }
        else { a0 = div_(lo - o[ax], d[ax]); a1 = div_(hi - o[ax], d[ax]); }
        tmin = max_(tmin, min_(a0, a1));
        tmax = min_(tmax, max_(a0, a1));
    }
    return ok && tmin <= tmax && tmin <= t_hi && tmax >= t_lo;
}

// ---------------------------------------------------------------------------------------------
// CSG: CsgOp::eval (csg.rs:26-32)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool csg_eval(uint32_t op, bool lhit, bool inl, bool inr) {
    if (op == RTMI_NODE_CSG_UNION) return (lhit && !inr) || (!lhit && !inl);
    if (op == RTMI_NODE_CSG_INTERSECTION) return (lhit && inr) || (!lhit && inl);
    return (lhit && !inr) || (!lhit && inl);
}

enum { MODE_CLOSEST = 0, MODE_ANY = 1, MODE_COLLECT = 2 };

// id word of a buffered CSG intersection
constexpr uint32_t ID_LEAF_MASK = 0x0FFFFFFFu;
constexpr uint32_t ID_ORD_SHIFT = 28;
constexpr uint32_t ID_DONE = 1u << 31;
constexpr uint32_t ID_KEEP = 1u << 30;

template <typename R> struct Hit {
    R t;
    uint32_t leaf; // NONE = miss
    uint32_t ord;  // which of the leaf's ascending roots
};

// result of the n1/n2 search
template <typename R> struct Open {
    R best_t, other_t;         // latest entry t among open leaves / among open leaves other than the hit leaf
    uint32_t best, other;      // their leaves (NONE if none)
    uint32_t hit_parity;       // number of hit-leaf roots in front of the hit (odd = hit leaf is open)
};

template <typename R> __device__ __forceinline__ void open_candidate(Open<R> &op, uint32_t leaf, uint32_t hit_leaf, R t_entry) {
    if (op.best == NONE || t_entry > op.best_t) { op.best = leaf; op.best_t = t_entry; }
    if (leaf != hit_leaf && (op.other == NONE || t_entry > op.other_t)) { op.other = leaf; op.other_t = t_entry; }
}

// The intersections csg_root buffers, per lane: scratch memory.  (Keeping the first 4 or 8 of them in LDS, [entry][lane],
// was measured on the dice frame twice: 333 ms without, 345 and 380 ms with (round 2), 195 / 224 / 238 ms (round 3, with LDS to
// spare) - the per-lane scratch lines stay in the vector cache; the index arithmetic and the branch per access cost more.  Nor did reading up to eight of them into registers in one batch before the
// filter, with every leaf's DChain packed into one word beside them, change the frame's time: 196.4 and 200.9 ms against
// 196.0 - the filter's time is its own arithmetic and branches, not the waits for the buffer.)
template <typename R> struct CsgBuf {
    R deep_t[CSG_BUF];
    uint32_t deep_id[CSG_BUF];
    __device__ __forceinline__ void put(int i, R t, uint32_t id) { deep_t[i] = t; deep_id[i] = id; }
    __device__ __forceinline__ R t(int i) const { return deep_t[i]; }
    __device__ __forceinline__ uint32_t id(int i) const { return deep_id[i]; }
    __device__ __forceinline__ void set_id(int i, uint32_t id) { deep_id[i] = id; }
};

// A CSG root that is one operator over two leaves of at most two roots each (CsgPair::intersect + filter_intersections,
// csg.rs:97-124, for that case), f32 path: no buffer, no sort, no loop - the four roots stay in registers and each passes
// or not by where it lies in the OTHER leaf's interval, which is what the filter's inl / inr flags say at its turn.  Ties go
// as the reference's stable sort of left ++ right has them: a left root at a right root's parameter comes first (it sees
// inr as it was before that right root toggled it), a right root at a left root's parameter comes after it.
// Returns false for a lane whose leaf gave another number of roots than 0 or 2 (never, for spheres and cubes): that lane
// takes the general path.
template <int MODE>
__device__ __forceinline__ bool csg_pair(const DScene<float> &sc, uint32_t op, uint32_t la, const Ray<float> &ray, uint32_t origin_leaf, uint32_t origin_side,
                                         Hit<float> &best, float t_limit, uint32_t hit_leaf, uint32_t hit_ord, Open<float> &open, Stats &st) {
    const uint32_t lb = la + 1u;
    const DLeaf<float> &ga = sc.leaves[la], &gb = sc.leaves[lb];
    const uint32_t ka = ga.kind, kb = gb.kind;
    float a[4], b[4], u, v;
    int na = 0, nb = 0;
    if (!(MODE == MODE_ANY && !(ka & LEAF_CAST_SHADOW))) na = leaf_roots(ga, ka, sc.tris, ray, la == origin_leaf, origin_side, a, u, v, st);
    // (the guard record of the general path: without a root on its left side an Intersection or Difference is empty)
    const bool need_b = !(MODE == MODE_ANY && !(kb & LEAF_CAST_SHADOW)) && (op == (uint32_t)RTMI_NODE_CSG_UNION || na > 0);
    if (need_b) nb = leaf_roots(gb, kb, sc.tris, ray, lb == origin_leaf, origin_side, b, u, v, st);
    if ((na != 0 && na != 2) || (nb != 0 && nb != 2)) return false;
    const bool ha = na == 2, hb = nb == 2;
    // inside the other leaf at a root's turn
    auto in_b = [&](float x) { return hb && b[0] < x && x <= b[1]; };
    auto in_a = [&](float y) { return ha && a[0] <= y && y < a[1]; };
    bool ka0, ka1, kb0, kb1; // does the root pass (CsgOp::eval, csg.rs:26-32)
    if (op == (uint32_t)RTMI_NODE_CSG_UNION) { ka0 = !in_b(a[0]); ka1 = !in_b(a[1]); kb0 = !in_a(b[0]); kb1 = !in_a(b[1]); }
    else if (op == (uint32_t)RTMI_NODE_CSG_INTERSECTION) { ka0 = in_b(a[0]); ka1 = in_b(a[1]); kb0 = in_a(b[0]); kb1 = in_a(b[1]); }
    else { ka0 = !in_b(a[0]); ka1 = !in_b(a[1]); kb0 = in_a(b[0]); kb1 = in_a(b[1]); }
    ka0 = ka0 && ha; ka1 = ka1 && ha; kb0 = kb0 && hb; kb1 = kb1 && hb;
    if (MODE == MODE_COLLECT) {
        // survivors in front of the hit, per leaf: odd count = still open, entered at its last such root
        const bool fa0 = ka0 && ((la == hit_leaf) ? (0u < hit_ord) : (a[0] < t_limit)), fa1 = ka1 && ((la == hit_leaf) ? (1u < hit_ord) : (a[1] < t_limit));
        const bool fb0 = kb0 && ((lb == hit_leaf) ? (0u < hit_ord) : (b[0] < t_limit)), fb1 = kb1 && ((lb == hit_leaf) ? (1u < hit_ord) : (b[1] < t_limit));
        const uint32_t ca = (fa0 ? 1u : 0u) + (fa1 ? 1u : 0u), cb = (fb0 ? 1u : 0u) + (fb1 ? 1u : 0u);
        if (ca) { if (la == hit_leaf) open.hit_parity = ca; if (ca & 1u) open_candidate(open, la, hit_leaf, fa1 ? a[1] : a[0]); }
        if (cb) { if (lb == hit_leaf) open.hit_parity = cb; if (cb & 1u) open_candidate(open, lb, hit_leaf, fb1 ? b[1] : b[0]); }
        return true;
    }
    // the first survivor with t >= 0 in ascending order (ties: the left leaf's root first, the lower ordinal first)
    float bt = Lim<float>::inf();
    uint32_t bl = NONE, bo = 0u;
    auto take = [&](bool keep, float t, uint32_t leaf, uint32_t ord) { if (keep && t >= 0.0f && t < bt) { bt = t; bl = leaf; bo = ord; } };
    take(ka0, a[0], la, 0u); take(ka1, a[1], la, 1u); take(kb0, b[0], lb, 0u); take(kb1, b[1], lb, 1u);
    if (bl != NONE && bt < t_limit && bt < best.t) { best.t = bt; best.leaf = bl; best.ord = bo; }
    return true;
}

// Evaluate one CSG root (CsgPair::intersect + filter_intersections, csg.rs:97-124, recursively):
// buffer every root of every leaf below, visit them in ascending t, and let each pass the filters of
// its CSG ancestors bottom-up.  MODE_CLOSEST / MODE_ANY stop at the first survivor with t >= 0.
template <typename R, int MODE>
__device__ __forceinline__ void csg_root(const DScene<R> &sc, uint32_t node_flags, uint32_t leaf_first, uint32_t leaf_count, const Ray<R> &ray, uint32_t origin_leaf,
                         uint32_t origin_side, Hit<R> &best, R t_limit, uint32_t hit_leaf, uint32_t hit_ord, Open<R> &open, Stats &st) {
    if constexpr (sizeof(R) == 4) {
        const uint32_t pair_op = (node_flags >> NODE_PAIR_SHIFT) & NODE_PAIR_MASK; // (wave-uniform)
        if (pair_op != 0u && csg_pair<MODE>(sc, pair_op, leaf_first, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st)) return;
    }
    CsgBuf<R> buf;
    int k = 0, first_i = -1; // (first_*: the smallest root so far, the earliest of equals)
    R first_t = (R)0;
    uint32_t first_id = 0;
    CYC_BEGIN(c_leaf);
    const uint32_t leaf_end = leaf_first + leaf_count;
    uint32_t skip_until = leaf_first, last_root = NONE; // leaves below skip_until are not tested (DGuard); the last leaf that had a root
    // The leaves in runs (DCsgLeaf): a first pass tests the ray's line against the bounding sphere of every leaf of the run -
    // short records through scalar loads that do not wait for each other, a dozen operations a leaf - and the second pass
    // goes through the full test of only those leaves that some lane of the wave still needs.  (A ray through a die of
    // chapter-16 passes the spheres of two or three of its 21 dimples.)
    for (uint32_t base = leaf_first; base < leaf_end;) {
        uint32_t run = sc.csg_leaf[base].run;
        run = min(run == 0u ? 1u : run, min((uint32_t)CSG_RUN, leaf_end - base));
        uint32_t near = 0u, visit = 0u; // per lane: leaves to test in full; wave: leaves some lane tests in full
        CYC5_BEGIN(c_p1);
#pragma unroll 4
        for (uint32_t b = 0; b < run; ++b) {
            const DCsgLeaf &c = sc.csg_leaf[base + b];
            const uint32_t li = base + b, ckind = c.kind;
            const bool test = li >= skip_until && !(MODE == MODE_ANY && !(ckind & LEAF_CAST_SHADOW));
            bool out = false;
            if constexpr (sizeof(R) == 4) out = (ckind & LEAF_BOUND) != 0u && li != origin_leaf && line_misses_sphere(ray, c.c);
            if (test && out) { st.inc(RTMI_STAT_LEAF_TESTS); st.fl += F_XFORM + bound_reject_flops(ckind); }
            const bool full = test && !out;
            near |= full ? 1u << b : 0u;
            visit |= __ballot(full) != 0ull ? 1u << b : 0u;
        }
        CYC5_END(st, 0, c_p1);
        CYC5_BEGIN(c_p2);
        while (visit != 0u) {
            const uint32_t b = (uint32_t)__builtin_ctz(visit);
            visit &= visit - 1u;
            if (!((near >> b) & 1u)) continue;
            const uint32_t li = base + b;
            const DLeaf<R> &g = sc.leaves[li];
            R t[4], u, v;
            int n = leaf_roots<R, true>(g, g.kind, sc.tris, ray, li == origin_leaf, origin_side, t, u, v, st);
            if (n > 0) last_root = li;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < n) {
                    if (k < CSG_BUF) {
                        const uint32_t id = li | ((uint32_t)i << ID_ORD_SHIFT);
                        if (k == 0 || t[i] < first_t) { first_t = t[i]; first_i = k; first_id = id; }
                        buf.put(k, t[i], id);
                        ++k;
                    } else st.flag(RTMI_STAT_OVERFLOW);
                }
            }
        }
        CYC5_END(st, 1, c_p2);
        const DCsgLeaf &e = sc.csg_leaf[base + run - 1u];
        base += run;
        if (e.gskip != 0u && base >= skip_until && (last_root == NONE || last_root < e.gstart)) skip_until = max(skip_until, e.gskip);
        if (__ballot(skip_until < leaf_end) == 0ull) break; // every lane that is here is done with this root
    }
    CYC_END(st, 1, c_leaf);
    CYC_BEGIN(c_sort);
    CYC5_BEGIN(c_sort5);
    // filter_intersections (csg.rs:107-124) for a root of leaf li, all its CSG ancestors at once.  CsgOp::eval depends, for
    // a given operator and side, on ONE of the operator's two flags (a left root on inr, a right root on inl), so "passes
    // every ancestor" is: the flags in chain.zero are clear and those in chain.one are set (DChain, built on the host).
    // Every ancestor the root reaches toggles its other flag; it reaches them bottom-up until the first that stops it, and
    // an ancestor's flags sit below its descendants' (operators are numbered parent first), so that is the HIGHEST
    // failing flag and the toggles are those at or above its pair.
    // Ascending order (ties: buffer order, as the reference's stable sort of left ++ right, csg.rs:100-104) without
    // marking entries: a step takes the smallest (t, index) after the step before.  The first step's entry is known from
    // the leaf loop; later steps scan the buffer four entries at a time (the loads of a scan do not wait for each other).
    unsigned long long state = 0; // bit 2b = inl, 2b+1 = inr of CSG operator b under this root
    R prev_t = (R)0;
    int prev_i = -1;
    static_assert(CSG_BUF % 4 == 0, "the scan reads whole groups of four entries");
    for (int step = 0; step < k; ++step) {
        int m = -1;
        R mt = Lim<R>::inf();
        uint32_t id = 0;
        if (step == 0) { m = first_i; mt = first_t; id = first_id; }
        else {
            for (int i0 = 0; i0 < k; i0 += 4) {
                R tj[4];
                uint32_t dj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { tj[j] = buf.t(i0 + j); dj[j] = buf.id(i0 + j); }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + j;
                    const bool after = tj[j] > prev_t || (tj[j] == prev_t && i > prev_i);
                    if (i < k && after && (m < 0 || tj[j] < mt)) { m = i; mt = tj[j]; id = dj[j]; }
                }
            }
        }
        if (m < 0) break; // (a root that is not a number: no order to go on with)
        prev_t = mt; prev_i = m;
        const uint32_t li = id & ID_LEAF_MASK;
        const DChain ch = sc.csg_chain[li];
        const unsigned long long tested = ch.zero | ch.one;
        const unsigned long long fail = (state & ch.zero) | (~state & ch.one);
        unsigned long long toggle = ((tested & 0x5555555555555555ull) << 1) | ((tested & 0xAAAAAAAAAAAAAAAAull) >> 1);
        const bool keep = fail == 0ull;
        if (!keep) toggle &= ~((1ull << ((63 - __clzll((long long)fail)) & ~1)) - 1ull);
        state ^= toggle;
        if (MODE == MODE_COLLECT) { if (keep) buf.set_id(m, id | ID_KEEP); continue; }
        if (keep && mt >= (R)0) {
            if (mt < t_limit && mt < best.t) { best.t = mt; best.leaf = li; best.ord = (id >> ID_ORD_SHIFT) & 3u; }
            break; // sorted: nothing later can be closer
        }
    }
    CYC_END(st, 2, c_sort);
    CYC5_END(st, 2, c_sort5);
    if (MODE == MODE_COLLECT) {
        // survivors in front of the hit, per leaf: odd count = still open, entered at its last such root.  A leaf's roots
        // sit next to each other in the buffer, in ascending order (the leaf loop writes them so): one pass.
        uint32_t cur = NONE, cnt = 0u;
        R last_t = (R)0;
        auto close_leaf = [&]() {
            if (cnt == 0u) return;
            if (cur == hit_leaf) open.hit_parity = cnt;
            if (cnt & 1u) open_candidate(open, cur, hit_leaf, last_t);
        };
        for (int i = 0; i < k; ++i) {
            const uint32_t id = buf.id(i), li = id & ID_LEAF_MASK;
            if (li != cur) { close_leaf(); cur = li; cnt = 0u; }
            if (!(id & ID_KEEP)) continue;
            const R ti = buf.t(i);
            const bool front = (li == hit_leaf) ? (((id >> ID_ORD_SHIFT) & 3u) < hit_ord) : (ti < t_limit);
            if (front) { ++cnt; last_t = ti; }
        }
        close_leaf();
    }
}

// What a traversal does with the ascending roots t[0..n) of leaf li (a leaf that is not under any CSG operator).
template <typename R, int MODE>
__device__ __forceinline__ void apply_roots(uint32_t li, int n, const R t[4], Hit<R> &best, R t_limit, uint32_t hit_leaf, uint32_t hit_ord, Open<R> &open) {
    if (MODE == MODE_COLLECT) {
        int cnt = 0;
        R last_t = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool front = i < n && ((li == hit_leaf) ? ((uint32_t)i < hit_ord) : (t[i] < t_limit));
            if (front) { ++cnt; last_t = t[i]; }
        }
        if (li == hit_leaf) open.hit_parity = (uint32_t)cnt;
        if (cnt & 1) open_candidate(open, li, hit_leaf, last_t);
    } else {
        // hit (ray.rs:143-154): the first root with t >= 0 of this leaf, kept if it beats the running minimum
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            if (i < n && t[i] >= (R)0 && t[i] < t_limit && t[i] < best.t) { best.t = t[i]; best.leaf = li; best.ord = (uint32_t)i; }
        }
    }
}

// One leaf that is not under any CSG operator.
template <typename R, int MODE>
__device__ __forceinline__ void plain_leaf(const DScene<R> &sc, uint32_t li, const DLeaf<R> &g, uint32_t gkind, const Ray<R> &ray, uint32_t origin_leaf,
                                           uint32_t origin_side, Hit<R> &best, R t_limit, uint32_t hit_leaf, uint32_t hit_ord, Open<R> &open, Stats &st) {
    if (MODE == MODE_ANY && !(gkind & LEAF_CAST_SHADOW)) return;
    R t[4], u = 0, v = 0;
    int n = leaf_roots(g, gkind, sc.tris, ray, li == origin_leaf, origin_side, t, u, v, st);
    apply_roots<R, MODE>(li, n, t, best, t_limit, hit_leaf, hit_ord, open);
}

// Spheres under a general affine transform (LEAF_XPAIR), two per step from the transformed-sphere table: both rays into
// object space (ray.rs:30-35: 18 packed multiply-adds for the pair), then sphere.rs:47-63 with the roots as closest
// approach -+ half chord.  `take(id, t0, t1, true)` gets the ascending roots of every leaf that has any; `hits` counts them,
// `tested` the leaves (fillers excluded) - see quick_pairs.
template <int MODE, typename TAKE>
__device__ __forceinline__ void xsphere_pairs(const DScene<float> &sc, const Ray<float> &ray, f2 ox, f2 oy, f2 oz, f2 dx, f2 dy, f2 dz, uint32_t origin_leaf,
                                              TAKE &take, uint32_t &hits, uint32_t &tested) {
    const uint32_t xc = sc.xpair_counts;
    const uint32_t n = MODE == MODE_ANY ? (xc >> 8) & 255u : xc & 255u;
    if (n == 0u) return;
    tested += MODE == MODE_ANY ? xc >> 24 : (xc >> 16) & 255u;
    const uint32_t *tab = lds_xpairs();
    for (uint32_t p = 0; p < n; ++p) {
        const uint32_t *r = tab + p * (uint32_t)XPAIR_DW;
        f2 m[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = mk2(u2f_(r[2 * k]), u2f_(r[2 * k + 1]));
        const uint32_t id0 = r[24], id1 = r[25];
        const f2 px = m[0] * ox + m[1] * oy + m[2] * oz + m[3], py = m[4] * ox + m[5] * oy + m[6] * oz + m[7], pz = m[8] * ox + m[9] * oy + m[10] * oz + m[11];
        const f2 vx = m[0] * dx + m[1] * dy + m[2] * dz, vy = m[4] * dx + m[5] * dy + m[6] * dz, vz = m[8] * dx + m[9] * dy + m[10] * dz;
        const f2 a = vx * vx + vy * vy + vz * vz;
        const f2 ia = mk2(rcp_(a.x), rcp_(a.y));
        const f2 k = (vx * px + vy * py + vz * pz) * ia;
        const f2 lx = px - vx * k, ly = py - vy * k, lz = pz - vz * k;
        const f2 one = splat2(1.0f);
        const f2 x = (one - (lx * lx + ly * ly + lz * lz)) * ia;
        const bool h0 = !(x.x < 0.0f), h1 = !(x.y < 0.0f);
        if (h0 || h1) {
            const float s0 = sqrt_(x.x), s1 = sqrt_(x.y);
            float t00 = -k.x - s0, t01 = s0 - k.x, t10 = -k.y - s1, t11 = s1 - k.y;
            hits += (h0 ? 1u : 0u) + (h1 ? 1u : 0u);
            if (origin_leaf == id0 && h0) { if (abs_(t01) < abs_(t00)) t01 = -max_(abs_(t01), 1e-30f); else t00 = -max_(abs_(t00), 1e-30f); }
            if (origin_leaf == id1 && h1) { if (abs_(t11) < abs_(t10)) t11 = -max_(abs_(t11), 1e-30f); else t10 = -max_(abs_(t10), 1e-30f); }
            if (MODE != MODE_COLLECT || h0) take(id0, t00, t01, true);
            if (MODE != MODE_COLLECT || h1) take(id1, t10, t11, true);
        }
    }
}

// The LEAF_QUICK top-level leaves - spheres under a similarity transform as world-space spheres |p - c|^2 = r^2, planes as
// row 1 of their world->object matrix (see leaf_roots: the ray parameter is the same in both spaces) - from the pair
// tables of the LDS block, TWO leaves per step.  What the loops are built around:
//   * no branch per leaf and no list: the two leaves' arithmetic is the same instruction stream on packed pairs
//     (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: (o - c), d . oc, the foot of the perpendicular, its squared length);
//     only sqrt / rcp and the update are per leaf;
//   * hit (ray.rs:143-154: the first root with t >= 0) and "nearer than the best so far" are ONE unsigned minimum: the
//     bit patterns of the non-negative floats ascend like the numbers, and every negative number, -0 and every NaN is
//     above +inf's pattern - so min3_u32(best, t0, t1) is the running minimum over the roots that count, and a missed
//     leaf (the root of a negative number: NaN) needs no test of its own.  (-0, a root exactly at the origin of a ray that
//     was not spawned there, counts as behind: a set of measure zero.)
//   * the shadow query (MODE_ANY; world.rs:158-180) keeps no leaf and no order: the minimum over all shadow-casting
//     leaves, compared with the distance once at the end;
//   * the leaf a ray was spawned on is the only per-lane branch: its root nearest zero is the spawn point (see leaf_roots).
// Counters: every leaf of the tables is tested by every ray that comes here (as World::intersect tests every object,
// world.rs:145-170: the reference has no early exit either), so tests and their FLOPs are wave-uniform sums plus the
// per-lane count of sphere hits (F_SPHERE_HIT - F_SPHERE_MISS each).
template <int MODE, bool XP>
__device__ __forceinline__ void quick_pairs(const DScene<float> &sc, const Ray<float> &ray, uint32_t origin_leaf, Hit<float> &best, float t_limit,
                                            uint32_t hit_leaf, uint32_t hit_ord, Open<float> &open, Stats &st) {
    const uint32_t *tab = lds_scene() + LDS_PAIRS_AT;
    const uint32_t pc = sc.pair_counts, rc = sc.real_counts;
    const uint32_t sph_all = pc & 255u, n_sph = MODE == MODE_ANY ? (pc >> 8) & 255u : sph_all;
    const uint32_t n_pl = MODE == MODE_ANY ? pc >> 24 : (pc >> 16) & 255u;
    const uint32_t real_sph = MODE == MODE_ANY ? (rc >> 8) & 255u : rc & 255u, real_pl = MODE == MODE_ANY ? rc >> 24 : (rc >> 16) & 255u;
    if (n_sph + n_pl + (XP ? sc.xpair_counts & 255u : 0u) == 0u) return;
    const float inv_a = rcp_(dot(ray.d, ray.d));
    const f2 ox = splat2(ray.o.x), oy = splat2(ray.o.y), oz = splat2(ray.o.z), dx = splat2(ray.d.x), dy = splat2(ray.d.y), dz = splat2(ray.d.z), ia = splat2(inv_a);
    uint32_t bt = f2u_(MODE == MODE_ANY ? Lim<float>::inf() : best.t), bl = best.leaf, sphere_hits = 0;
    auto take = [&](uint32_t id, float t0, float t1, bool two) { // the roots of one leaf, ascending
        if (MODE == MODE_COLLECT) {
            const float t[4] = {t0, t1, 0.0f, 0.0f};
            apply_roots<float, MODE_COLLECT>(id, two ? 2 : 1, t, best, t_limit, hit_leaf, hit_ord, open);
        } else if (MODE == MODE_ANY) {
            bt = two ? umin3(bt, f2u_(t0), f2u_(t1)) : min(bt, f2u_(t0));
        } else {
            const uint32_t c = two ? umin3(bt, f2u_(t0), f2u_(t1)) : min(bt, f2u_(t0));
            bl = c != bt ? id : bl;
            bt = c;
        }
    };
    if (MODE == MODE_COLLECT) {
        // The container query (ray.rs:111-140) asks of every leaf whether an ODD number of its roots lies in front of the hit
        // (the ray is inside it there) and, if so, where it went in.  For a sphere that is "the hit point lies inside"
        // (t0 < T <= t1), for a plane "the hit point lies beyond it along the ray" (t < T, behind the origin included:
        // (n . P + w) and n . d have the same sign) - a few packed operations per pair and no root.  Only a leaf that IS
        // open (the glass sphere around the bubble, the sphere the ray is leaving) goes on to its roots and the
        // reference's rule on them (apply_roots), which also orders the open leaves by where they were entered.
        const V3<float> P = ray.o + ray.d * t_limit;
        const f2 px = splat2(P.x), py = splat2(P.y), pz = splat2(P.z);
        auto sphere_roots = [&](uint32_t id, float cx, float cy, float cz, float r2) {
            const V3<float> oc = ray.o - mk<float>(cx, cy, cz);
            const float k = dot(ray.d, oc) * inv_a;
            const V3<float> l = oc - ray.d * k;
            const float x = (r2 - dot(l, l)) * inv_a;
            if (x < 0.0f) return;
            const float s = sqrt_(x);
            float t[4] = {-k - s, s - k, 0.0f, 0.0f};
            if (origin_leaf == id) { if (abs_(t[1]) < abs_(t[0])) t[1] = -max_(abs_(t[1]), 1e-30f); else t[0] = -max_(abs_(t[0]), 1e-30f); }
            apply_roots<float, MODE_COLLECT>(id, 2, t, best, t_limit, hit_leaf, hit_ord, open);
        };
        for (uint32_t p = 0; p < n_sph; ++p) {
            const uint32_t *r = tab + p * (uint32_t)PAIR_DW;
            const f2 cx = mk2(u2f_(r[0]), u2f_(r[1])), cy = mk2(u2f_(r[2]), u2f_(r[3])), cz = mk2(u2f_(r[4]), u2f_(r[5])), r2 = mk2(u2f_(r[6]), u2f_(r[7]));
            const uint32_t id0 = r[8], id1 = r[9];
            const f2 ex = px - cx, ey = py - cy, ez = pz - cz;
            const f2 q = r2 - (ex * ex + ey * ey + ez * ez);
            const bool hit0 = id0 == hit_leaf, hit1 = id1 == hit_leaf;
            if (hit0 || hit1) open.hit_parity = hit_ord; // (a sphere's roots in front of its own hit: the number of the hit root)
            if (hit0 ? (hit_ord & 1u) != 0u : q.x > 0.0f) sphere_roots(id0, cx.x, cy.x, cz.x, r2.x);
            if (hit1 ? (hit_ord & 1u) != 0u : q.y > 0.0f) sphere_roots(id1, cx.y, cy.y, cz.y, r2.y);
        }
        for (uint32_t p = 0; p < n_pl; ++p) {
            const uint32_t *r = tab + (sph_all + p) * (uint32_t)PAIR_DW;
            const f2 nx = mk2(u2f_(r[0]), u2f_(r[1])), ny = mk2(u2f_(r[2]), u2f_(r[3])), nz = mk2(u2f_(r[4]), u2f_(r[5])), w = mk2(u2f_(r[6]), u2f_(r[7]));
            const uint32_t id0 = r[8], id1 = r[9];
            const f2 ddy = nx * dx + ny * dy + nz * dz, fp = nx * px + ny * py + nz * pz + w, side = fp * ddy;
            if (id0 == hit_leaf || id1 == hit_leaf) open.hit_parity = 0u; // (a plane's one root is the hit itself)
            const bool o0 = id0 != hit_leaf && side.x > 0.0f && !(abs_(ddy.x) < (float)RTMI_EPSILON);
            const bool o1 = id1 != hit_leaf && side.y > 0.0f && !(abs_(ddy.y) < (float)RTMI_EPSILON);
            if (o0 || o1) {
                const f2 ooy = nx * ox + ny * oy + nz * oz + w;
                if (o0) { float t[4] = {-ooy.x * rcp_(ddy.x), 0.0f, 0.0f, 0.0f}; if (origin_leaf == id0) t[0] = -max_(abs_(t[0]), 1e-30f);
                          apply_roots<float, MODE_COLLECT>(id0, 1, t, best, t_limit, hit_leaf, hit_ord, open); }
                if (o1) { float t[4] = {-ooy.y * rcp_(ddy.y), 0.0f, 0.0f, 0.0f}; if (origin_leaf == id1) t[0] = -max_(abs_(t[0]), 1e-30f);
                          apply_roots<float, MODE_COLLECT>(id1, 1, t, best, t_limit, hit_leaf, hit_ord, open); }
            }
        }
        uint32_t xs = real_sph, xhits = 0;
        auto take_collect = [&](uint32_t id, float t0, float t1, bool) {
            const float t[4] = {t0, t1, 0.0f, 0.0f};
            apply_roots<float, MODE_COLLECT>(id, 2, t, best, t_limit, hit_leaf, hit_ord, open);
        };
        if constexpr (XP) xsphere_pairs<MODE_COLLECT>(sc, ray, ox, oy, oz, dx, dy, dz, origin_leaf, take_collect, xhits, xs);
        st.add(RTMI_STAT_LEAF_TESTS, xs + real_pl);
        st.fl += xs * (F_XFORM + F_SPHERE_MISS) + real_pl * (F_XFORM + F_PLANE) + xhits * (F_SPHERE_HIT - F_SPHERE_MISS); // (priced as the reference's tests of these leaves; the quick spheres' hits are not told apart here)
        return;
    }
    for (uint32_t p = 0; p < n_sph; ++p) { // sphere.rs:47-63 in world space, roots as closest approach -+ half chord
        const uint32_t *r = tab + p * (uint32_t)PAIR_DW;
        const f2 cx = mk2(u2f_(r[0]), u2f_(r[1])), cy = mk2(u2f_(r[2]), u2f_(r[3])), cz = mk2(u2f_(r[4]), u2f_(r[5])), r2 = mk2(u2f_(r[6]), u2f_(r[7]));
        const uint32_t id0 = r[8], id1 = r[9];
        const f2 ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
        const f2 k = (dx * ocx + dy * ocy + dz * ocz) * ia;          // (d . oc) / (d . d): minus the parameter of closest approach
        const f2 lx = ocx - dx * k, ly = ocy - dy * k, lz = ocz - dz * k;
        const f2 x = (r2 - (lx * lx + ly * ly + lz * lz)) * ia;      // (half chord)^2 in units of t; negative: the line misses
        const bool h0 = !(x.x < 0.0f), h1 = !(x.y < 0.0f);
        if (h0 || h1) {
            const float s0 = sqrt_(x.x), s1 = sqrt_(x.y); // (NaN for the one that missed: never a minimum)
            float t00 = -k.x - s0, t01 = s0 - k.x, t10 = -k.y - s1, t11 = s1 - k.y;
            sphere_hits += (h0 ? 1u : 0u) + (h1 ? 1u : 0u);
            if (origin_leaf == id0 && h0) { // the spawn point: the root nearest zero lies behind the origin
                if (abs_(t01) < abs_(t00)) t01 = -max_(abs_(t01), 1e-30f); else t00 = -max_(abs_(t00), 1e-30f);
            }
            if (origin_leaf == id1 && h1) {
                if (abs_(t11) < abs_(t10)) t11 = -max_(abs_(t11), 1e-30f); else t10 = -max_(abs_(t10), 1e-30f);
            }
            if (MODE != MODE_COLLECT || h0) take(id0, t00, t01, true);
            if (MODE != MODE_COLLECT || h1) take(id1, t10, t11, true);
        }
    }
    for (uint32_t p = 0; p < n_pl; ++p) { // plane.rs:35-44 on the y row of the object-space ray
        const uint32_t *r = tab + (sph_all + p) * (uint32_t)PAIR_DW;
        const f2 nx = mk2(u2f_(r[0]), u2f_(r[1])), ny = mk2(u2f_(r[2]), u2f_(r[3])), nz = mk2(u2f_(r[4]), u2f_(r[5])), w = mk2(u2f_(r[6]), u2f_(r[7]));
        const uint32_t id0 = r[8], id1 = r[9];
        const f2 ddy = nx * dx + ny * dy + nz * dz, ooy = nx * ox + ny * oy + nz * oz + w;
        const bool h0 = !(abs_(ddy.x) < (float)RTMI_EPSILON), h1 = !(abs_(ddy.y) < (float)RTMI_EPSILON);
        const float nan = __builtin_nanf("");
        float t0 = h0 ? -ooy.x * rcp_(ddy.x) : nan, t1 = h1 ? -ooy.y * rcp_(ddy.y) : nan;
        if (origin_leaf == id0) t0 = -max_(abs_(t0), 1e-30f); // (of a NaN: a NaN still)
        if (origin_leaf == id1) t1 = -max_(abs_(t1), 1e-30f);
        if (MODE != MODE_COLLECT || h0) take(id0, t0, 0.0f, false);
        if (MODE != MODE_COLLECT || h1) take(id1, t1, 0.0f, false);
    }
    uint32_t tested_sph = real_sph;
    if constexpr (XP) xsphere_pairs<MODE>(sc, ray, ox, oy, oz, dx, dy, dz, origin_leaf, take, sphere_hits, tested_sph);
    st.add(RTMI_STAT_LEAF_TESTS, tested_sph + real_pl);
    st.fl += tested_sph * (F_XFORM + F_SPHERE_MISS) + real_pl * (F_XFORM + F_PLANE) + sphere_hits * (F_SPHERE_HIT - F_SPHERE_MISS);
    if (MODE == MODE_ANY) { if (bt < f2u_(t_limit)) best.leaf = 0u; } // (which leaf: nobody asks)
    else if (MODE == MODE_CLOSEST) { best.t = u2f_(bt); best.leaf = bl; }  // (best.ord of a leaf of the tables: prepare knows it)
}

// A triangle leaf of a NODE_ALLTRI node, tested in world space from its DWTri record alone: no leaf record, no ray
// transform (triangle.rs:233-265 with the determinants expressed in world space; priced like the reference's test).
template <int MODE>
__device__ __forceinline__ void tri_leaf(const DWTri &w, uint32_t li, const Ray<float> &ray, uint32_t origin_leaf, uint32_t origin_side,
                                         Hit<float> &best, float t_limit, uint32_t hit_leaf, uint32_t hit_ord, Open<float> &open, Stats &st) {
    const V3<float> e1 = ld3(w.e1), e2 = ld3(w.e2);
    const V3<float> dir_cross_e2 = cross(ray.d, e2);
    const float detw = dot(e1, dir_cross_e2), det = detw * w.k;
    uint32_t fl = F_TRI_DET;
    bool hit = false;
    float t = 0;
    if (!(abs_(det) < (float)RTMI_EPSILON)) {
        const float f = rcp_(detw);
        const V3<float> p1_to_origin = ray.o - ld3(w.p1);
        const float u = f * dot(p1_to_origin, dir_cross_e2);
        fl = F_TRI_U;
        if (!(u < 0.0f || u > 1.0f)) {
            const V3<float> origin_cross_e1 = cross(p1_to_origin, e1);
            const float v = f * dot(ray.d, origin_cross_e1);
            fl = F_TRI_V;
            if (!(v < 0.0f || (u + v) > 1.0f)) {
                t = f * dot(e2, origin_cross_e1);
                hit = true;
                fl = F_TRI_FULL;
            }
        }
    }
    st.inc(RTMI_STAT_LEAF_TESTS);
    st.fl += F_XFORM + fl;
    if (hit && li == origin_leaf) { // the spawn point itself (see leaf_roots)
        float forced = -max_(abs_(t), 1e-30f);
        if (w.smooth) {
            const bool plus = (origin_side & SIDE_PLUS) != 0;
            const bool towards = plus ? det < 0.0f : det > 0.0f;
            if (towards) forced = -forced;
        }
        t = forced;
    }
    if (MODE == MODE_COLLECT) {
        const bool front = hit && ((li == hit_leaf) ? (0u < hit_ord) : (t < t_limit));
        if (li == hit_leaf) open.hit_parity = front ? 1u : 0u;
        if (front) open_candidate(open, li, hit_leaf, t);
    } else if (hit && t >= 0.0f && t < t_limit && (t < best.t || (t == best.t && li < best.leaf))) {
        // (equal roots - a ray through an edge two triangles share - go to the lower leaf whatever order the triangles
        // are met in: a wave that walks the tree together meets them in an order that depends on its other rays)
        best.t = t; best.leaf = li; best.ord = 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// A wave walks a DBvh2 tree together (f32 path).  `mine`: this lane's ray needs the tree (it passed the NODE_TRIBVH node's
// own box).  Everything that steers the walk is wave-uniform: the current entry, the shared stack (LDS, one per wave;
// each entry carries the mask of the lanes whose ray hit that node's box), the node and triangle records (scalar loads).
// A lane tests a node's child boxes only if it hit the node itself - exactly the tests it would make walking alone,
// which is also what the counters count - and a subtree is entered if any lane's ray can still find a root in it.
// ---------------------------------------------------------------------------------------------
#ifdef RTMI_HOST_EMUL
inline bool lanes_of(unsigned long long mask) { return mask & 1ull; } // (a wave of one)
#else
__device__ __forceinline__ bool lanes_of(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
#endif
__device__ __forceinline__ uint32_t (*lds_packet_stack())[PACKET_STACK] {
    __shared__ uint32_t s_packet[3][PACKET_STACK]; // entry, mask low, mask high
    return s_packet;
}
// What the box tests of a DBvh2 walk need of a ray, once per query: a slab plane's parameter is fma(plane, id, b) with
// id = 1 / d and b = -o / d.  An axis the ray does not move along (|d| < 1e-30) gets id = 0, b = NaN: its parameters are all
// NaN, the NaN-ignoring min / max below drop them, and the axis rules no box out - such a ray enters some boxes beside it
// that Aabb::intersect (aabb.rs:45-70) would not; what it HITS is decided by the leaves, as ever.  e bounds the rounding of
// the fma form (|b| 2^-24 per plane from b's own rounding, as much again from the sum) so that a box is never missed for it.
struct RaySlab { float idx, idy, idz, bx, by, bz, e; };
__device__ __forceinline__ RaySlab ray_slab(const Ray<float> &ray) {
    RaySlab r;
    const bool zx = !(abs_(ray.d.x) >= 1e-30f), zy = !(abs_(ray.d.y) >= 1e-30f), zz = !(abs_(ray.d.z) >= 1e-30f);
    const float nan = __builtin_nanf("");
    r.idx = zx ? 0.0f : rcp_(ray.d.x); r.idy = zy ? 0.0f : rcp_(ray.d.y); r.idz = zz ? 0.0f : rcp_(ray.d.z);
    r.bx = zx ? nan : -ray.o.x * r.idx; r.by = zy ? nan : -ray.o.y * r.idy; r.bz = zz ? nan : -ray.o.z * r.idz;
    r.e = ((zx ? 0.0f : abs_(r.bx)) + (zy ? 0.0f : abs_(r.by)) + (zz ? 0.0f : abs_(r.bz))) * 0x1p-22f;
    return r;
}
// Both children's boxes of a DBvh2 node against one ray (aabb.rs:45-70,130-141): six packed multiply-adds give the twelve
// plane parameters (the record holds a plane of both boxes side by side: one scalar register pair per instruction), then
// per box 6 min / max, a max3 and a min3, the slack and the clamp to the ray's interval.  t0 / t1: where the ray enters.
__device__ __forceinline__ void slab_pair(const DBvh2 &n, const RaySlab &rs, float t_lo, float t_hi, bool &h0, bool &h1, float &t0, float &t1) {
    const f2 ix = splat2(rs.idx), iy = splat2(rs.idy), iz = splat2(rs.idz), bx = splat2(rs.bx), by = splat2(rs.by), bz = splat2(rs.bz);
    const f2 lx = mk2(n.lox[0], n.lox[1]) * ix + bx, hx = mk2(n.hix[0], n.hix[1]) * ix + bx;
    const f2 ly = mk2(n.loy[0], n.loy[1]) * iy + by, hy = mk2(n.hiy[0], n.hiy[1]) * iy + by;
    const f2 lz = mk2(n.loz[0], n.loz[1]) * iz + bz, hz = mk2(n.hiz[0], n.hiz[1]) * iz + bz;
    const float near0 = max_(max_(min_(lx.x, hx.x), min_(ly.x, hy.x)), min_(lz.x, hz.x));
    const float far0 = min_(min_(max_(lx.x, hx.x), max_(ly.x, hy.x)), max_(lz.x, hz.x)) + rs.e;
    const float near1 = max_(max_(min_(lx.y, hx.y), min_(ly.y, hy.y)), min_(lz.y, hz.y));
    const float far1 = min_(min_(max_(lx.y, hx.y), max_(ly.y, hy.y)), max_(lz.y, hz.y)) + rs.e;
    t0 = max_(near0, t_lo); t1 = max_(near1, t_lo);
    h0 = t0 <= min_(far0, t_hi); h1 = t1 <= min_(far1, t_hi);
}
// UNITS: entries other than DBvh2 nodes and triangle runs can occur (a scene of meshes only runs the loop without them)
template <int MODE, bool UNITS>
__device__ __forceinline__ void packet_tris(const DScene<float> &sc, uint32_t root, bool mine, int base, const Ray<float> &ray, const RaySlab &rs, uint32_t origin_leaf,
                            uint32_t origin_side, Hit<float> &best, float t_limit, uint32_t hit_leaf, uint32_t hit_ord, Open<float> &open, Stats &st) {
    // (base: the entries of the shared stack below it belong to the caller, packet_scene)
    uint32_t (*stk)[PACKET_STACK] = lds_packet_stack();
#ifdef RTMI_HOST_EMUL
    const int lane = 0; // (the emulation runs one lane at a time: a wave of one)
#else
    const int lane = threadIdx.x & 63;
#endif
    // the lane that writes the shared stack: the first lane that is here at all (lane 0 may not be - its pixel finished,
    // its ray hit nothing - and a push nobody made would be popped as whatever the LDS held)
    const int writer = __ffsll((long long)__ballot(true)) - 1;
    uint32_t cur = root;
    uint32_t nv = 0; // box-pair steps of this lane: the counters get them on the way out (one add a step instead of three)
    auto leave = [&]() { st.add(RTMI_STAT_NODE_VISITS, nv); st.add(RTMI_STAT_AABB_TESTS, 2u * nv); st.fl += nv * (2u * F_AABB); };
    unsigned long long mask = __ballot(mine);
    int sp = base;
    const float lo = (MODE == MODE_COLLECT) ? -Lim<float>::inf() : 0.0f;
    for (;;) {
        const bool in = lanes_of(mask); // (the mask is wave-uniform: it becomes the execution mask as it is)
        bool pop = false;
        if ((cur & ENT_KIND) == ENT_TRIS) {
            const uint32_t first = cur & ENT_TRI_FIRST_MASK, count = ((cur >> ENT_TRI_FIRST_BITS) & 15u) + 1u;
            if (first + count > sc.n_btris) { st.flag(RTMI_STAT_OVERFLOW); return; } // (never: an entry that is not the builder's must not become an address)
            CYC6_BEGIN(c_tri);
            if (in && !(MODE == MODE_ANY && best.leaf != NONE)) {
                for (uint32_t i = 0; i < count; ++i) {
                    const DWTri &w = sc.btris[first + i]; // (wave-uniform index: a scalar load)
                    tri_leaf<MODE>(w, w.pad, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
                }
            }
            CYC6_END(st, 1, c_tri);
            if (MODE == MODE_ANY && __ballot(mine && best.leaf == NONE) == 0ull) { leave(); return; } // every ray of the wave is shadowed
            pop = true;
        } else if (UNITS && (cur & ENT_KIND) == ENT_LEAF) { // a primitive that is not a triangle
            const uint32_t li = cur & ~ENT_KIND;
            if (li >= sc.n_leaves) { st.flag(RTMI_STAT_OVERFLOW); return; } // (never: see above)
            if (in && !(MODE == MODE_ANY && best.leaf != NONE))
                plain_leaf<float, MODE>(sc, li, sc.leaves[li], sc.leaves[li].kind, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
            if (MODE == MODE_ANY && __ballot(mine && best.leaf == NONE) == 0ull) { leave(); return; }
            pop = true;
        } else if (UNITS && (cur & ENT_KIND) == 0u) { // a CSG root
            if (cur >= sc.n_nodes) { st.flag(RTMI_STAT_OVERFLOW); return; } // (never: see above)
            const DNode &nd = sc.nodes[cur];
            const uint32_t leaf_first = nd.leaf_first, leaf_count = nd.leaf_count;
            if (!(nd.flags & NODE_CSGROOT) || leaf_first + leaf_count > sc.n_leaves) { st.flag(RTMI_STAT_OVERFLOW); return; } // (never)
            if (in && !(MODE == MODE_ANY && best.leaf != NONE)) {
                st.inc(RTMI_STAT_NODE_VISITS);
                csg_root<float, MODE>(sc, nd.flags, leaf_first, leaf_count, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
            }
            if (MODE == MODE_ANY && __ballot(mine && best.leaf == NONE) == 0ull) { leave(); return; }
            pop = true;
        } else {
            if ((cur & ENT_KIND) != ENT_BVH || (cur & ~ENT_KIND) >= sc.n_bvh2) { st.flag(RTMI_STAT_OVERFLOW); return; } // (never: see above)
            CYC6_BEGIN(c_box);
            const DBvh2 &n = sc.bvh2[cur & ~ENT_KIND];
            const float hi = (MODE == MODE_CLOSEST) ? best.t : t_limit;
            float t0 = 0, t1 = 0;
            bool h0 = false, h1 = false;
            if (in && !(MODE == MODE_ANY && best.leaf != NONE)) {
                slab_pair(n, rs, lo, hi, h0, h1, t0, t1);
                ++nv;
            }
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
            if ((m0 | m1) == 0ull) pop = true;
            else if (m0 == 0ull) { cur = n.child[1]; mask = m1; }
            else if (m1 == 0ull) { cur = n.child[0]; mask = m0; }
            else {
                // both: the child most rays meet first goes first (closest hit: what it finds prunes the other)
                const bool one_first = MODE == MODE_CLOSEST && __popcll(__ballot(h0 && h1 && t1 < t0)) * 2 > __popcll(m0 & m1);
                const uint32_t far = one_first ? n.child[0] : n.child[1];
                const unsigned long long fm = one_first ? m0 : m1;
                if (sp < PACKET_STACK) { // (the host checks the depth: always true)
                    if (lane == writer) { stk[0][sp] = far; stk[1][sp] = (uint32_t)fm; stk[2][sp] = (uint32_t)(fm >> 32); }
                    ++sp;
                } else st.flag(RTMI_STAT_OVERFLOW);
                cur = one_first ? n.child[1] : n.child[0];
                mask = one_first ? m1 : m0;
            }
            CYC6_END(st, 0, c_box);
        }
        if (pop) {
            if (sp == base) { leave(); return; }
            CYC6_BEGIN(c_pop);
            --sp;
            cur = __builtin_amdgcn_readfirstlane(stk[0][sp]);
            mask = (unsigned long long)__builtin_amdgcn_readfirstlane(stk[1][sp]) | ((unsigned long long)__builtin_amdgcn_readfirstlane(stk[2][sp]) << 32);
            CYC6_END(st, 2, c_pop);
        }
    }
}

// The scene tree below World::objects (Groups, BoundingGroups, CSG roots, triangle meshes), walked by the whole wave together
// like packet_tris (f32 path): one shared stack of node indices, node and leaf records through scalar loads.  A node is
// entered if any lane's ray passes its box (aabb.rs:45-70 on the padded f32 box) and can still find a root there; only
// those lanes test its leaves, evaluate its CSG tree or go on into its triangle BVH.
template <int MODE>
__device__ __forceinline__ void packet_scene(const DScene<float> &sc, const Ray<float> &ray, uint32_t origin_leaf, uint32_t origin_side, Hit<float> &best,
                                             float t_limit, uint32_t hit_leaf, uint32_t hit_ord, Open<float> &open, Stats &st) {
    uint32_t (*stk)[PACKET_STACK] = lds_packet_stack();
#ifdef RTMI_HOST_EMUL
    const int lane = 0;
#else
    const int lane = threadIdx.x & 63;
#endif
    const int writer = __ffsll((long long)__ballot(true)) - 1; // (see packet_tris)
    const V3<float> inv_d = mk<float>(rcp_(ray.d.x), rcp_(ray.d.y), rcp_(ray.d.z));
    const RaySlab rs = ray_slab(ray);
    const float lo = (MODE == MODE_COLLECT) ? -Lim<float>::inf() : 0.0f;
    int sp = 0;
    auto push = [&](uint32_t v) {
        if (sp < PACKET_STACK) { if (lane == writer) stk[0][sp] = v; ++sp; }
        else st.flag(RTMI_STAT_OVERFLOW); // (the host checks the tree's need against PACKET_STACK: never)
    };
    for (uint32_t c = sc.root_child_count; c-- > 0;) push(sc.root_child_first + c); // popped first to last
    while (sp > 0) {
        if (MODE == MODE_ANY && __ballot(best.leaf == NONE) == 0ull) return; // every ray of the wave is shadowed
        --sp;
        const uint32_t top = __builtin_amdgcn_readfirstlane(stk[0][sp]);
        if (top >= sc.n_nodes) { st.flag(RTMI_STAT_OVERFLOW); return; } // (never: a stack entry must not become an address unchecked)
        const DNode &nd = sc.nodes[top];
        const uint32_t flags = nd.flags;
        if (MODE == MODE_ANY && !(flags & NODE_CAST_SHADOW)) continue;
        // closest / any: some root with 0 <= t <= hi must be possible (a CSG root then still evaluates its negative-t
        // roots for the in/out state); collect: anything in front of the hit counts
        const float hi = (MODE == MODE_CLOSEST) ? best.t : t_limit;
        bool h = !(MODE == MODE_ANY && best.leaf != NONE);
        if ((flags & NODE_GATED) && h) {
            st.inc(RTMI_STAT_AABB_TESTS);
            st.fl += F_AABB;
            h = slab(nd, ray, inv_d, lo, hi);
        }
        const unsigned long long hm = __ballot(h);
        if (hm == 0ull) continue;
        if (h) st.inc(RTMI_STAT_NODE_VISITS);
        if (flags & (NODE_TRIBVH | NODE_CSGROOT)) { // the node's own BVH, or the CSG root itself as a one-entry tree (its NODE_VISITS is counted there)
            // (csg_root has ONE call site in this walk, inside packet_tris: a second one here - the root evaluated at once, without
            // the trip through the entry loop - made every tree kernel slower, teapot 23.4 -> 27.7 ms with no CSG in it: three
            // more inlined copies of it cost registers everywhere)
            if (h && (flags & NODE_CSGROOT)) st.add(RTMI_STAT_NODE_VISITS, 0xFFFFFFFFu);
            const uint32_t entry = (flags & NODE_TRIBVH) ? nd.csg_count : top;
            if (sc.has_units) packet_tris<MODE, true>(sc, entry, h, sp, ray, rs, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
            else packet_tris<MODE, false>(sc, entry, h, sp, ray, rs, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
            continue;
        }
        // children are popped first to last; sorted children (two, by centre along an axis) last to first when most of the
        // rays run down that axis: the nearer one first
        const uint32_t ax = (flags >> NODE_AXIS_SHIFT) & 3u, n_children = nd.child_count, child_first = nd.child_first;
        bool rev = false;
        if (ax != 0) rev = __popcll(__ballot(h && (ax == 1 ? ray.d.x : ax == 2 ? ray.d.y : ray.d.z) < 0.0f)) * 2 > __popcll(hm);
        for (uint32_t c = n_children; c-- > 0;) push(child_first + (rev ? n_children - 1u - c : c));
        const uint32_t leaf_first = nd.leaf_first, leaf_count = nd.leaf_count;
        if (leaf_count && h) {
            if (flags & NODE_ALLTRI) {
                const uint32_t w0 = nd.csg_count;
                for (uint32_t i = 0; i < leaf_count; ++i)
                    tri_leaf<MODE>(sc.wtris[w0 + i], leaf_first + i, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
            } else {
                for (uint32_t li = leaf_first; li < leaf_first + leaf_count; ++li)
                    plain_leaf<float, MODE>(sc, li, sc.leaves[li], sc.leaves[li].kind, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
            }
        }
    }
}

// LDS part of the traversal stack (one array per kernel: the three traversal modes of a wave never run at the same time)
constexpr int LDS_NODES = 16;
__device__ __forceinline__ uint32_t (*lds_node_stack())[64] {
    __shared__ uint32_t s_nodes[LDS_NODES][64];
    return s_nodes;
}

// World::intersect + hit (world.rs:145-156, ray.rs:143-154) / intersect_shadow + is_shadowed
// (world.rs:158-180) / the container walk of ray.rs:111-140, as one traversal in three modes.
//   MODE_CLOSEST: best = min t >= 0.            t_limit = +inf
//   MODE_ANY    : best.leaf != NONE iff some shadow-casting root lies in [0, t_limit)
//   MODE_COLLECT: open = which leaves the ray is inside of just in front of (t_limit, hit_leaf, hit_ord)
template <typename R, int MODE, int TREE>
__device__ __forceinline__ void traverse_body(const DScene<R> &sc, const Ray<R> &ray, uint32_t origin_leaf, uint32_t origin_side, Hit<R> &best, R t_limit,
                         uint32_t hit_leaf, uint32_t hit_ord, Open<R> &open, Stats &st) {
    // Top-level primitives: wave-uniform loop bounds.  Flat f32 scenes: the LEAF_QUICK ones among the first n_quick from
    // the pair tables of the LDS block (quick_pairs), two per step and without a branch per leaf; the others - kinds the
    // tables do not hold, leaves past the block - from their record in global memory through scalar loads, with the kind
    // from the leaf's LDS record where it has one (one round trip instead of two).  f64 and tree scenes: scalar loads.
    if constexpr (staged_v<R, TREE>) {
        quick_pairs<MODE, TREE == KIND_FLATX>(sc, ray, origin_leaf, best, t_limit, hit_leaf, hit_ord, open, st);
        if (MODE == MODE_ANY && best.leaf != NONE) return;
        const uint32_t nq = sc.n_quick;
        for (uint32_t i = 0; i < sc.n_others; ++i) {
            const uint32_t li = sc.others[i];
            const uint32_t kind = li < nq ? __builtin_amdgcn_readfirstlane(lds_scene()[li * (uint32_t)QUICK_DW]) : sc.leaves[li].kind;
            plain_leaf<R, MODE>(sc, li, sc.leaves[li], kind, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
            if (MODE == MODE_ANY && best.leaf != NONE) return;
        }
    } else {
        for (uint32_t li = 0; li < sc.root_leaf_count; ++li) {
            plain_leaf<R, MODE>(sc, li, sc.leaves[li], sc.leaves[li].kind, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
            if (MODE == MODE_ANY && best.leaf != NONE) return;
        }
    }
    if (TREE != KIND_TREE) return; // flat scenes (World::objects are all primitives) compile without the tree / CSG machinery
    if (sc.root_child_count == 0) return;
    if constexpr (sizeof(R) == 4) { // the product path: the wave walks the tree together
        CYC_BEGIN(c_walk);
        packet_scene<MODE>(sc, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
        CYC_END(st, 0, c_walk);
        return;
    }
    // f64 validation mode: every lane walks the reference's tree on its own.
    // node stack: the first LDS_NODES entries of every lane in LDS ([entry][lane]), deeper ones in scratch
    uint32_t (*lds)[64] = lds_node_stack();
    const int lane = threadIdx.x & 63;
    uint32_t deep[NODE_STACK - LDS_NODES];
    int sp = 0;
    for (uint32_t c = sc.root_child_count; c-- > 0;) {
        const uint32_t v = sc.root_child_first + c;
        if (sp < LDS_NODES) lds[sp][lane] = v; else deep[sp - LDS_NODES] = v;
        ++sp;
    }
    V3<R> inv_d = mk<R>(0, 0, 0);
    if constexpr (sizeof(R) == 4) inv_d = mk<R>(rcp_(ray.d.x), rcp_(ray.d.y), rcp_(ray.d.z));
    // "while-while" traversal: every lane first walks inner nodes until it holds a node with leaves (lanes that found one
    // wait for the others), then the wave tests leaves together - instead of lanes at inner nodes idling through their
    // neighbours' triangle loops.
    auto push = [&](uint32_t v) {
        if (sp < LDS_NODES) { lds[sp][lane] = v; ++sp; }
        else if (sp < NODE_STACK) { deep[sp - LDS_NODES] = v; ++sp; }
        else st.flag(RTMI_STAT_OVERFLOW);
    };
    for (;;) {
        uint32_t w_first = 0, w_count = 0, w_flags = 0, w_aux = 0;
        bool have = false;
        while (sp > 0 && !have) {
            --sp;
            const uint32_t top = sp < LDS_NODES ? lds[sp][lane] : deep[sp - LDS_NODES];
            // closest / any: some root with 0 <= t <= hi must be possible (a CSG root then still evaluates its
            // negative-t roots for the in/out state); collect: anything in front of the hit counts
            const R hi = (MODE == MODE_CLOSEST) ? best.t : t_limit;
            const R lo = (MODE == MODE_COLLECT) ? -Lim<R>::inf() : (R)0;
            const DNode nd = sc.nodes[top];
            if (MODE == MODE_ANY && !(nd.flags & NODE_CAST_SHADOW)) continue;
            if (nd.flags & NODE_GATED) {
                st.inc(RTMI_STAT_AABB_TESTS);
                st.fl += F_AABB;
                if (!slab(nd, ray, inv_d, lo, hi)) continue;
            }
            st.inc(RTMI_STAT_NODE_VISITS);
            if (!(nd.flags & NODE_CSGROOT)) {
                // children are popped first to last; sorted children (two, by centre along an axis) last to first when the
                // ray runs down that axis: the nearer one first
                const uint32_t ax = (nd.flags >> NODE_AXIS_SHIFT) & 3u;
                const bool rev = ax != 0 && (ax == 1 ? ray.d.x : ax == 2 ? ray.d.y : ray.d.z) < (R)0;
                for (uint32_t c = nd.child_count; c-- > 0;) push(nd.child_first + (rev ? nd.child_count - 1u - c : c));
            }
            if ((nd.flags & NODE_CSGROOT) || nd.leaf_count) {
                w_first = nd.leaf_first; w_count = nd.leaf_count; w_flags = nd.flags; w_aux = nd.csg_count;
                have = true;
            }
        }
        if (!have) break;
        if (w_flags & NODE_CSGROOT) {
            csg_root<R, MODE>(sc, w_flags, w_first, w_count, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
        } else {
            bool done = false;
            if constexpr (sizeof(R) == 4) {
                if (w_flags & NODE_ALLTRI) {
                    for (uint32_t i = 0; i < w_count; ++i)
                        tri_leaf<MODE>(sc.wtris[w_aux + i], w_first + i, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
                    done = true;
                }
            }
            if (!done)
                for (uint32_t li = w_first; li < w_first + w_count; ++li)
                    plain_leaf<R, MODE>(sc, li, sc.leaves[li], sc.leaves[li].kind, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
        }
        if (MODE == MODE_ANY && best.leaf != NONE) return;
    }
}

// ---------------------------------------------------------------------------------------------
// patterns (pattern.rs:7-35).  Rust's `%` keeps the sign of the dividend, so "x % 2 == 0" is
// "x is an even integer" for integral x of either sign.
// ---------------------------------------------------------------------------------------------
template <typename R> __device__ __forceinline__ bool is_even(R x) { R h = x * (R)0.5; return floor_(h) == h; }

template <typename R> __device__ __forceinline__ V3<R> pattern_at(const DPattern<R> &p, V3<R> obj_point) {
    const R EPS = (R)RTMI_EPSILON;
    V3<R> q = xf_point(p.inv, obj_point); // pattern.rs:51-53
    V3<R> a = ld3(p.a), b = ld3(p.b);
    switch (p.kind) {
    case RTMI_PATTERN_STRIPE: return is_even(floor_(q.x)) ? a : b;
    case RTMI_PATTERN_GRADIENT: return a + (b - a) * (q.x - floor_(q.x));
    case RTMI_PATTERN_RING: return is_even(floor_(sqrt_(q.x * q.x + q.z * q.z))) ? a : b;
    case RTMI_PATTERN_CHECKERS: return is_even(floor_(q.x + EPS) + floor_(q.y + EPS) + floor_(q.z + EPS)) ? a : b;
    default: return q; // RTMI_PATTERN_XYZ
    }
}

// ---------------------------------------------------------------------------------------------
// lights (lights.rs)
// ---------------------------------------------------------------------------------------------
enum { IL_RAY = 0, IL_OMNI = 1, IL_NOLIGHT = 2 };
template <typename R> struct Incoming { int kind; V3<R> origin, direction, color; };

__device__ __forceinline__ void sincos2pi(float u, float &s, float &c) { sincospif(2.0f * u, &s, &c); }
__device__ __forceinline__ void sincos2pi(double u, double &s, double &c) { sincospi(2.0 * u, &s, &c); }
__device__ __forceinline__ float exp_(float x) { return expf(x); }
__device__ __forceinline__ double exp_(double x) { return exp(x); }
// x in (0, 1], y = shininess: materials.rs:281 `powf`
__device__ __forceinline__ float pow_(float x, float y) { return __builtin_exp2f(y * __builtin_log2f(x)); }
__device__ __forceinline__ double pow_(double x, double y) { return pow(x, y); }

template <typename R> __device__ __forceinline__ Incoming<R> incoming_at(const DLight<R> &l, V3<R> p, Rng &rng) {
    const R FOUR_PI = (R)(4.0 * 3.14159265358979323846);
    Incoming<R> in;
    V3<R> pos = ld3(l.pos);
    in.kind = IL_RAY; in.origin = pos; in.color = ld3(l.intensity); in.direction = mk<R>(0, 0, 0);
    switch (l.kind) {
    case RTMI_LIGHT_POINT: // lights.rs:89-95
        in.direction = normalize(pos - p);
        break;
    case RTMI_LIGHT_AMBIENT: // lights.rs:292-294
        in.kind = IL_OMNI;
        break;
    case RTMI_LIGHT_REALISTIC_POINT: { // lights.rs:153-160
        V3<R> d = pos - p;
        in.direction = normalize(d);
        in.color = in.color * rcp_(FOUR_PI * dot(d, d));
        break;
    }
    case RTMI_LIGHT_BEAM: { // lights.rs:230-243
        V3<R> dir = ld3(l.dir), up = ld3(l.up), right = ld3(l.right);
        V3<R> d = p - pos;
        if (dot(d, dir) < (R)0) { in.kind = IL_NOLIGHT; break; }
        R u = dot(d, up) * l.inv_up2, v = dot(d, right) * l.inv_right2;
        in.origin = pos + up * u + right * v;
        in.direction = -dir;
        in.color = in.color * exp_(-(u * u + v * v));
        break;
    }
    case RTMI_LIGHT_SPHERE: { // lights.rs:353-362; UnitSphere by inverse CDF (two draws)
        R u1 = rng_next<R>(rng), u2 = rng_next<R>(rng);
        R z = (R)1 - (R)2 * u1, s, c;
        sincos2pi(u2, s, c);
        R rr = sqrt_(max_((R)0, (R)1 - z * z));
        in.origin = pos + mk<R>(rr * c, rr * s, z) * l.radius;
        in.direction = normalize(in.origin - p);
        break;
    }
    case RTMI_LIGHT_DISC: { // lights.rs:438-453; UnitDisc by inverse CDF (two draws)
        R u1 = rng_next<R>(rng), u2 = rng_next<R>(rng);
        R rr = sqrt_(u1), s, c;
        sincos2pi(u2, s, c);
        in.origin = pos + ld3(l.width) * (rr * c) + ld3(l.height) * (rr * s);
        V3<R> d = in.origin - p;
        in.direction = normalize(d);
        R cosn = max_(dot(-in.direction, ld3(l.normal)), (R)0);
        in.color = (in.color * cosn) * rcp_(FOUR_PI * dot(d, d));
        break;
    }
    default: in.kind = IL_NOLIGHT;
    }
    return in;
}

// Phong::lighting (materials.rs:246-288)
template <typename R>
__device__ __forceinline__ V3<R> lighting(const DMaterial<R> &m, V3<R> surface_color, const Incoming<R> &light, V3<R> eyev, V3<R> normalv,
                                          bool in_shadow) {
    if (light.kind == IL_NOLIGHT) return mk<R>(0, 0, 0);
    if (light.kind == IL_OMNI) return had(surface_color, light.color);
    if (in_shadow) return mk<R>(0, 0, 0);
    R light_dot_normal = dot(light.direction, normalv);
    if (light_dot_normal <= (R)0) return mk<R>(0, 0, 0);
    V3<R> result = had(surface_color, light.color) * (m.diffuse * light_dot_normal);
    V3<R> ml = -light.direction;
    V3<R> reflectv = ml - normalv * ((R)2 * dot(ml, normalv));
    R reflect_dot_eye = dot(reflectv, eyev);
    if (reflect_dot_eye > (R)0) result = result + light.color * (m.specular * pow_(reflect_dot_eye, m.shininess));
    return result;
}

// schlick (ray.rs:199-219)
template <typename R> __device__ __forceinline__ R schlick(V3<R> eyev, V3<R> normalv, R n1, R n2) {
    R cos_en = dot(eyev, normalv);
    if (n1 > n2) {
        R n = div_(n1, n2);
        R sin2_t = n * n * ((R)1 - cos_en * cos_en);
        if (sin2_t > (R)1) return (R)1;
        cos_en = sqrt_((R)1 - sin2_t);
    }
    R r0 = div_(n1 - n2, n1 + n2);
    r0 = r0 * r0;
    R tmp = (R)1 - cos_en, tmp2 = tmp * tmp;
    return r0 + ((R)1 - r0) * (tmp2 * tmp2 * tmp);
}

// ---------------------------------------------------------------------------------------------
// prepare_computations (ray.rs:72-102) for the closest hit
// ---------------------------------------------------------------------------------------------
template <typename R> struct Comps {
    V3<R> point, eyev, normalv;
    bool inside;
    bool side_plus; // smooth triangles: normalv points to the +(e2 x e1) side of the flat triangle
    R n1, n2;
};

// (kind | flags and material of the hit leaf come back too: from the leaf's LDS record if it has one)
template <typename R, int TREE>
__device__ __forceinline__ void prepare(const DScene<R> &sc, const Ray<R> &ray, Hit<R> &h, Comps<R> &c, uint32_t &kind, uint32_t &material) {
    c.point = ray.o + ray.d * h.t; // Ray::position, ray.rs:26-28
    c.eyev = -ray.d;
    c.n1 = (R)1; c.n2 = (R)1;
    if constexpr (staged_v<R, TREE>) {
        if (h.leaf < sc.n_quick) {
            const uint32_t *q = lds_scene() + h.leaf * (uint32_t)QUICK_DW;
            kind = q[0]; material = q[1];
            if (kind & LEAF_QUICK) {
                const V3<R> qv = mk<R>(u2f_(q[4]), u2f_(q[5]), u2f_(q[6]));
                // sphere under a similarity: inv^T inv (p - c) is parallel to p - c; plane: inv^T (0, 1, 0) is row 1 of inv
                const V3<R> n = normalize((kind & LEAF_FAST) ? c.point - qv : qv);
                c.inside = dot(n, c.eyev) < (R)0;
                c.normalv = c.inside ? -n : n;
                c.side_plus = false;
                // which of the leaf's ascending roots the hit is (quick_pairs does not keep it): a sphere's second root is
                // where the ray leaves it, i.e. seen from inside; a plane has one
                h.ord = ((kind & LEAF_FAST) && c.inside) ? 1u : 0u;
                return;
            }
        }
    }
    const DLeaf<R> &g = sc.leaves[h.leaf];
    kind = g.kind; material = g.material;
    V3<R> obj_point = xf_point(g.inv, c.point);
    R u = 0, v = 0;
    V3<R> ng = mk<R>(0, 0, 0);
    if ((g.kind & LEAF_KIND_MASK) >= RTMI_TRIANGLE) { // barycentrics of the hit (triangle.rs:250,257)
        const DTri<R> &tr = sc.tris[g.tri];
        V3<R> o = xf_point(g.inv, ray.o), d = xf_vector(g.inv, ray.d);
        V3<R> e1 = ld3(tr.e1), e2 = ld3(tr.e2);
        ng = cross(e2, e1);
        V3<R> dir_cross_e2 = cross(d, e2);
        R f = rcp_(dot(e1, dir_cross_e2));
        V3<R> p1_to_origin = o - ld3(tr.p1);
        u = f * dot(p1_to_origin, dir_cross_e2);
        v = f * dot(d, cross(p1_to_origin, e1));
    }
    V3<R> n;
    if (sizeof(R) == 4 && (g.kind & LEAF_FAST)) n = normalize(c.point - ld3(g.fast)); // similarity: inv^T inv (p - c) is parallel to p - c
    else n = normalize(xf_normal(g.inv, local_normal(g, sc.tris, obj_point, u, v))); // shapes/mod.rs:261-265,319-322
    c.inside = dot(n, c.eyev) < (R)0;
    c.normalv = c.inside ? -n : n;
    c.side_plus = dot(xf_vector(g.inv, c.normalv), ng) > (R)0;
    if constexpr (staged_v<R, TREE>) { // a sphere of the transformed-sphere table: its second root is where the ray leaves it (see the quick leaves above)
        if (h.leaf < sc.n_quick && (lds_scene()[h.leaf * (uint32_t)QUICK_DW] & LEAF_XPAIR)) h.ord = c.inside ? 1u : 0u;
    }
}

// a material / the refractive index of a leaf's material: from the LDS block where it holds them
template <typename R, int TREE> __device__ __forceinline__ DMaterial<R> load_material(const DScene<R> &sc, uint32_t id) {
    if constexpr (staged_v<R, TREE>) {
        // (n_lds_materials is all of them or none: a wave-uniform choice between an LDS read and a global load - a per-lane
        // choice would make every access a flat load through both paths)
        if (sc.n_lds_materials != 0u) return *reinterpret_cast<const DMaterial<float> *>(lds_scene() + LDS_MATS_AT + min(id, (uint32_t)LDS_MATS - 1u) * (uint32_t)MAT_DW);
    }
    return sc.materials[id];
}
template <typename R, int TREE> __device__ __forceinline__ R leaf_ior(const DScene<R> &sc, uint32_t leaf) {
    uint32_t id = 0;
    bool have = false;
    if constexpr (staged_v<R, TREE>) {
        if (leaf < sc.n_quick) { id = lds_scene()[leaf * (uint32_t)QUICK_DW + 1u]; have = true; }
    }
    if (!have) id = sc.leaves[leaf].material;
    if constexpr (staged_v<R, TREE>) {
        if (sc.n_lds_materials != 0u) return u2f_(lds_scene()[LDS_MATS_AT + min(id, (uint32_t)LDS_MATS - 1u) * (uint32_t)MAT_DW + 9u]); // DMaterial::ior
    }
    return sc.materials[id].ior;
}

// (-DRTMI_CYCLES=2: the cycles of the closest-hit, shadow and container queries go to counters 0, 1, 2 - tools/cycles.py)
template <typename R, int MODE, int TREE>
__device__ __forceinline__ void traverse(const DScene<R> &sc, const Ray<R> &ray, uint32_t origin_leaf, uint32_t origin_side, Hit<R> &best, R t_limit,
                         uint32_t hit_leaf, uint32_t hit_ord, Open<R> &open, Stats &st) {
#if defined(RTMI_CYCLES) && RTMI_CYCLES == 2
    const unsigned long long c0 = __builtin_readcyclecounter();
#endif
    traverse_body<R, MODE, TREE>(sc, ray, origin_leaf, origin_side, best, t_limit, hit_leaf, hit_ord, open, st);
#if defined(RTMI_CYCLES) && RTMI_CYCLES == 2
    st.cyc[MODE == MODE_CLOSEST ? 0 : MODE == MODE_ANY ? 1 : 2] += __builtin_readcyclecounter() - c0;
#endif
}

// compute_refractive_indices (ray.rs:104-140) without the sorted list: a leaf is "open" in front of
// the hit iff an odd number of its intersections precede the hit, and containers.last() is the open
// leaf that was entered last (largest entry t).
template <typename R, int TREE>
__device__ __forceinline__ void collect_n1n2(const DScene<R> &sc, const Ray<R> &ray, uint32_t origin_leaf, uint32_t origin_side,
                                             const Hit<R> &h, Comps<R> &c, Stats &st) {
    Open<R> open;
    open.best = NONE; open.other = NONE; open.best_t = 0; open.other_t = 0; open.hit_parity = 0;
    Hit<R> dummy = {Lim<R>::inf(), NONE, 0};
    traverse<R, MODE_COLLECT, TREE>(sc, ray, origin_leaf, origin_side, dummy, h.t, h.leaf, h.ord, open, st);
    if ((open.best != NONE && open.best >= sc.n_leaves) || (open.other != NONE && open.other >= sc.n_leaves)) { // (never: see color_at)
        st.flag(RTMI_STAT_OVERFLOW); open.best = NONE; open.other = NONE;
    }
    c.n1 = open.best != NONE ? leaf_ior<R, TREE>(sc, open.best) : (R)1;
    if (open.hit_parity & 1u) c.n2 = open.other != NONE ? leaf_ior<R, TREE>(sc, open.other) : (R)1;
    else c.n2 = leaf_ior<R, TREE>(sc, h.leaf);
}

template <typename R> struct Pending { // a refraction ray waiting for its turn (materials.rs:327-331)
    V3<R> o, d;
    R weight;
    int32_t remaining;
    uint32_t origin_leaf, origin_side;
};

// Where the pending rays live.  f32 (product path): the first LDS_PEND entries of every lane sit in LDS, 8 dwords per
// entry laid out [entry][dword][lane] (conflict-free), 12 KB per wave = 12 waves per CU, which is the occupancy the
// register budget allows anyway; deeper entries and the f64 validation mode use a per-lane array (scratch memory).
// Recursion depth 5/6 scenes therefore push nothing to HBM.  One wave per workgroup (blockDim.x == 64) is assumed.
#ifndef RTMI_LDS_PEND
#define RTMI_LDS_PEND 4
#endif
constexpr int LDS_PEND = RTMI_LDS_PEND; // flat scenes (a fifth pending ray - five refractions waiting at once, one per level of a depth-5 tree - goes to
                                 // scratch: rare, and the 2 KB are what the counter block and the pair tables live in; 12.3 KB per wave in all)
constexpr int LDS_PEND_TREE = 4; // tree scenes share the 12 KB with the node stack (LDS_NODES); they do not stage the scene block
constexpr uint32_t PEND_LEAF_BITS = 26, PEND_LEAF_MASK = (1u << PEND_LEAF_BITS) - 1u; // leaf | side << 26 | remaining << 27
__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }

template <typename R, int LP> struct PendingStack {
    static constexpr int LDS_PEND = LP; // (shadows the namespace constant inside this class)
    Pending<R> deep[sizeof(R) == 4 ? MAX_DEPTH - LP : MAX_DEPTH];
    int sp = 0;
    __device__ __forceinline__ bool empty() const { return sp == 0; }
    __device__ __forceinline__ bool full() const { return sp >= MAX_DEPTH; }
    __device__ __forceinline__ void push(uint32_t (*lds)[8][64], const Pending<R> &e) {
        if constexpr (sizeof(R) == 4) {
            if (sp < LDS_PEND) {
                const int lane = threadIdx.x & 63;
                lds[sp][0][lane] = f2u(e.o.x); lds[sp][1][lane] = f2u(e.o.y); lds[sp][2][lane] = f2u(e.o.z);
                lds[sp][3][lane] = f2u(e.d.x); lds[sp][4][lane] = f2u(e.d.y); lds[sp][5][lane] = f2u(e.d.z);
                lds[sp][6][lane] = f2u(e.weight);
                lds[sp][7][lane] = (e.origin_leaf & PEND_LEAF_MASK) | ((e.origin_side & 1u) << PEND_LEAF_BITS) | ((uint32_t)e.remaining << (PEND_LEAF_BITS + 1));
            } else {
                deep[sp - LDS_PEND] = e;
            }
        } else {
            deep[sp] = e;
        }
        ++sp;
    }
    __device__ __forceinline__ Pending<R> pop(uint32_t (*lds)[8][64]) {
        --sp;
        if constexpr (sizeof(R) == 4) {
            if (sp < LDS_PEND) {
                const int lane = threadIdx.x & 63;
                Pending<R> e;
                e.o = mk<R>(u2f(lds[sp][0][lane]), u2f(lds[sp][1][lane]), u2f(lds[sp][2][lane]));
                e.d = mk<R>(u2f(lds[sp][3][lane]), u2f(lds[sp][4][lane]), u2f(lds[sp][5][lane]));
                e.weight = u2f(lds[sp][6][lane]);
                const uint32_t m = lds[sp][7][lane];
                const uint32_t leaf = m & PEND_LEAF_MASK;
                e.origin_leaf = leaf == PEND_LEAF_MASK ? NONE : leaf;
                e.origin_side = (m >> PEND_LEAF_BITS) & 1u;
                e.remaining = (int32_t)(m >> (PEND_LEAF_BITS + 1));
                return e;
            }
            return deep[sp - LDS_PEND];
        } else {
            return deep[sp];
        }
    }
};

// World::is_shadowed (world.rs:172-180)
template <typename R, int TREE>
__device__ __forceinline__ bool is_shadowed(const DScene<R> &sc, const Incoming<R> &in, V3<R> p, uint32_t origin_leaf, uint32_t origin_side,
                                            Stats &st) {
    if (in.kind == IL_OMNI) return false;
    if (in.kind == IL_NOLIGHT) return true;
    st.inc(RTMI_STAT_SHADOW);
    if (RTMI_SKIP & 16) return false;
    Ray<R> sray = {p, in.direction};
    V3<R> to_light = in.origin - p;
    R dist = sqrt_(dot(to_light, to_light));
    Hit<R> sh = {Lim<R>::inf(), NONE, 0};
    Open<R> unused;
    traverse<R, MODE_ANY, TREE>(sc, sray, origin_leaf, origin_side, sh, dist, NONE, 0, unused, st);
    return sh.leaf != NONE;
}

// The photon-map term of Phong::shade_hit (materials.rs:376-384): PhotonMap::find_nearest (photon_map.rs:315-364) with the
// recursion of locate_photons unrolled into a walk with stages per node (0: descend to the near side; 1: back from it, the
// far side if its plane is within the search radius and nearer than the farthest candidate; 2: the node's own photon,
// after both sides as in the reference) and its BinaryHeap as two per-lane arrays; then the fold over the photons found.
template <typename R> struct HeapEntry { R d; uint32_t n; }; // a candidate of find_nearest: squared distance, node
#ifndef RTMI_HEAP_ARITY
#define RTMI_HEAP_ARITY 4
#endif
constexpr int HEAP_ARITY = RTMI_HEAP_ARITY;
#ifndef RTMI_HEAP_LDS
#define RTMI_HEAP_LDS 21
#endif
constexpr int HEAP_LDS = RTMI_HEAP_LDS; // entries of the heap held in LDS (21: the top three levels at arity 4; measured 5 / 21 / 37 / 53 / 85: 356 / 256 / 317 / 394 / 425 ms on the 4K photon frame, 382 with none - more LDS per wave costs waves per CU)
// The candidates beyond the LDS-resident top of the heap.  Per-lane scratch interleaves the lanes dword by dword, so a heap
// walk - every lane at a different index - touches one 64-byte sector per lane and dword for four bytes of use (the 270 GB a
// 1080p frame moved).  The render kernels therefore get a workspace in the call's work memory, LANE-major: a lane's
// entries are one contiguous run of 8-byte pairs, the four children of a heap entry share a 64-byte line (heap_ws: the
// lane's own run, or null - the probe kernels, the f64 mode, work memory too small - for the scratch array).
constexpr bool HEAP_LEVELS_WHOLE = HEAP_ARITY == 4 && (HEAP_LDS == 1 || HEAP_LDS == 5 || HEAP_LDS == 21 || HEAP_LDS == 85); // the LDS part ends with a level
__device__ __host__ constexpr int heap_level_start(int level) { int at = 0, w = 1; for (int l = 0; l < level; ++l) { at += w; w *= HEAP_ARITY; } return at; } // 0, 1, 5, 21, 85, ...
__device__ __host__ constexpr int heap_parent_levels() { int l = 0; while (heap_level_start(l + 1) < RTMI_MAX_NEAREST) ++l; return l; }
constexpr int HEAP_PARENT_LEVELS = heap_parent_levels(); // levels whose entries can have children (4 at 128 candidates)
constexpr int HEAP_WS_ENTRIES = RTMI_MAX_NEAREST - HEAP_LDS; // pairs per lane
constexpr uint32_t HEAP_WS_SLOTS = 4096;                     // waves per launch that can have a run each (the host caps the grids of photon kernels at this)
template <typename R>
__device__ __forceinline__ V3<R> photon_estimate(const DScene<R> &sc, V3<R> p, V3<R> normalv, V3<R> surface_color, Stats &st, HeapEntry<R> *heap_ws = nullptr) {
    const uint32_t n = sc.n_photons;
    const int k = (int)min(sc.n_nearest, (uint32_t)RTMI_MAX_NEAREST);
    HeapEntry<R> he[RTMI_MAX_NEAREST];
    // The top levels - where most passes through the heap begin and end (a photon that replaces the farthest is seldom
    // much nearer than it) - live in LDS, [entry][lane]: these are the lines that every pass would otherwise pull through
    // the memory system from per-lane scratch (PMC: 515 GB moved per 1920x1080 frame, 12 x the photons read, before).  (One wave per workgroup: the lane is threadIdx.x & 63.)
    __shared__ HeapEntry<R> heap_top[HEAP_LDS][64];
    const int heap_lane = (int)(threadIdx.x & 63u);
    auto heap_ld = [&](int i) -> HeapEntry<R> { return i < HEAP_LDS ? heap_top[i][heap_lane] : (heap_ws ? heap_ws[i - HEAP_LDS] : he[i]); };
    auto heap_st = [&](int i, const HeapEntry<R> &e) {
        if (i < HEAP_LDS) heap_top[i][heap_lane] = e;
        else if (heap_ws) heap_ws[i - HEAP_LDS] = e; // (wave-uniform choice)
        else he[i] = e;
    };
    int cnt = 0;
    R top = (R)0; // the root's distance while cnt > 0, in a register: it is looked at for every photon visited
    const R pc[3] = {p.x, p.y, p.z};
    uint32_t visited = 0;
    // No stack: the tree is in heap order (parent of i: (i - 1) / 2), so the way back is arithmetic, and what is left to do
    // at each ancestor takes one bit per level (set: the walk is in its far side, nothing left) - the tree of 2^30 - 1
    // photons has 30 levels.  The reference looks at a node's own photon AFTER both sides (locate_photons, photon_map.rs:
    // 323-363).  Here a node inside the tree is looked at BETWEEN them, in the trip that comes back from the near side and
    // decides about the far side - two loads of its record and two trips of this loop instead of three; and a node whose
    // children are leaves (`low`: three quarters of the nodes a search looks at are in the tree's last two levels) BEFORE
    // them, in the trip that arrives at it, its distance to the plane kept in a register for the near leaf's trip to
    // decide about the far leaf - three trips and three loads for the three.
    // The photons found are the same: whatever the order, a photon enters the heap iff it is nearer than the k-th nearest
    // so far, and a side is skipped only if its plane is at least that far away - the k nearest within the radius either
    // way; the number of photons LOOKED AT on the way (RTMI_STAT_FLOPS' share) is not the reference's.
    uint32_t node = 0, depth = 0, in_far = 0;
    bool arriving = true; // from above (else: back from the near side of `node`)
    R kept_d = (R)0;      // of the last low node arrived at
    bool walking = n > 0 && k > 0;
    while (walking) { // (node < n here)
        const DPhoton &ph = sc.photons[node];
        const uint32_t left = 2u * node + 1u;
        const bool inner = left < n; // (!is_leaf, photon_map.rs:376-378)
        const bool low = inner && 2u * left + 1u >= n;
        const R dx = pc[0] - (R)ph.pos[0], dy = pc[1] - (R)ph.pos[1], dz = pc[2] - (R)ph.pos[2];
        const R d2 = dx * dx + dy * dy + dz * dz;
        const uint32_t axis = min((ph.rgbf >> 24) - (uint32_t)KD_SPLIT_X, 2u);
        const R d = axis == 0u ? dx : axis == 1u ? dy : dz;
        if (arriving && inner && !low) { // down the near side (a child beyond the photons returns at once, photon_map.rs:330-332)
            const uint32_t child = d < (R)0 ? left : left + 1u;
            in_far &= ~(1u << depth);
            if (child < n && depth < 31u) { node = child; ++depth; continue; }
        }
        {
            ++visited;
            // BinaryHeap (photon_map.rs:354-362) as one array of (distance, node) pairs, a max-heap with HEAP_ARITY
            // children per entry (half the levels of a binary one: every level is a round trip to per-lane memory).
            // Not full: push = a hole at the end that climbs while its parent is nearer.  Full and this photon nearer
            // than the farthest: pop + push = the farthest (the root) leaves and the hole it leaves sinks, the farthest
            // child moving up, to where the new photon belongs - one pass.  The photons kept and the farthest distance
            // are those of the reference's heap; their order inside the array - the order of the sum below - may
            // differ from it, i.e. the sum in its last bits.
            if (cnt < k) {
                int i = cnt++;
                while (i > 0) {
                    const int q = (i - 1) / HEAP_ARITY;
                    const HeapEntry<R> e = heap_ld(q);
                    if (!(e.d < d2)) break;
                    heap_st(i, e);
                    i = q;
                }
                heap_st(i, HeapEntry<R>{d2, node});
                top = i == 0 ? d2 : top;
            } else if (d2 < top) {
                int i = 0;
                R new_top = d2;
                if (HEAP_LEVELS_WHOLE && heap_ws) {
                    // The hole sinks one level per step, in every lane: step by step unrolled, where an entry and its
                    // children live is known when the code is written, not looked up per access.
                    bool sinking = true;
#pragma unroll
                    for (int lvl = 0; lvl < HEAP_PARENT_LEVELS; ++lvl) {
                        const int c0 = HEAP_ARITY * i + 1;
                        sinking = sinking && c0 < cnt;
                        if (sinking) {
                            const int last = cnt - 1;
                            HeapEntry<R> c;
                            int m = c0;
                            if (heap_level_start(lvl + 1) < HEAP_LDS) {
                                c = heap_top[c0][heap_lane];
#pragma unroll
                                for (int j = 1; j < HEAP_ARITY; ++j) {
                                    const int at = min(c0 + j, last);
                                    const HeapEntry<R> e = heap_top[at][heap_lane];
                                    if (c.d < e.d) { c = e; m = at; }
                                }
                            } else {
                                c = heap_ws[c0 - HEAP_LDS];
#pragma unroll
                                for (int j = 1; j < HEAP_ARITY; ++j) {
                                    const int at = min(c0 + j, last);
                                    const HeapEntry<R> e = heap_ws[at - HEAP_LDS];
                                    if (c.d < e.d) { c = e; m = at; }
                                }
                            }
                            if (!(d2 < c.d)) sinking = false;
                            else {
                                if (heap_level_start(lvl) < HEAP_LDS) heap_top[i][heap_lane] = c;
                                else heap_ws[i - HEAP_LDS] = c;
                                if (lvl == 0) new_top = c.d;
                                i = m;
                            }
                        }
                    }
                } else
                for (;;) {
                    const int c0 = HEAP_ARITY * i + 1;
                    if (c0 >= cnt) break;
                    const int last = cnt - 1;
                    HeapEntry<R> c;
                    int m = c0;
                    // (independent loads: one round trip per level; a clamped index repeats the last child.  With whole
                    // levels in LDS the children of one entry are all on the same side of HEAP_LDS: one choice of home
                    // per level instead of one per load)
                    if (HEAP_LEVELS_WHOLE && c0 < HEAP_LDS) {
                        c = heap_top[c0][heap_lane];
#pragma unroll
                        for (int j = 1; j < HEAP_ARITY; ++j) {
                            const int at = min(c0 + j, last);
                            const HeapEntry<R> e = heap_top[at][heap_lane];
                            if (c.d < e.d) { c = e; m = at; }
                        }
                    } else if (HEAP_LEVELS_WHOLE && heap_ws) {
                        c = heap_ws[c0 - HEAP_LDS];
#pragma unroll
                        for (int j = 1; j < HEAP_ARITY; ++j) {
                            const int at = min(c0 + j, last);
                            const HeapEntry<R> e = heap_ws[at - HEAP_LDS];
                            if (c.d < e.d) { c = e; m = at; }
                        }
                    } else {
                        c = heap_ld(c0);
#pragma unroll
                        for (int j = 1; j < HEAP_ARITY; ++j) {
                            const int at = min(c0 + j, last);
                            const HeapEntry<R> e = heap_ld(at);
                            if (c.d < e.d) { c = e; m = at; }
                        }
                    }
                    if (!(d2 < c.d)) break;
                    heap_st(i, c);
                    if (i == 0) new_top = c.d;
                    i = m;
                }
                heap_st(i, HeapEntry<R>{d2, node});
                top = new_top;
            }
        }
        if (low && arriving) { // its near leaf
            const uint32_t child = d < (R)0 ? left : left + 1u;
            in_far &= ~(1u << depth);
            kept_d = d;
            if (child < n && depth < 31u) { node = child; ++depth; continue; }
        }
        if (inner && (d < (R)0 ? -d : d) < sc.photon_radius && (cnt < k || d * d < top)) { // the far side
            const uint32_t child = d < (R)0 ? left + 1u : left;
            if (child < n && depth < 31u) { in_far |= 1u << depth; node = child; ++depth; arriving = true; continue; }
        }
        if (!inner && depth > 0u && (in_far >> (depth - 1u) & 1u) == 0u) {
            // the near leaf of a low node (4 * parent + 3, the first grandchild, is no photon): the far leaf is its sibling,
            // and the parent's kept distance decides about it; either way the parent is done
            const uint32_t parent = (node - 1u) >> 1;
            if (4u * parent + 3u >= n) {
                in_far |= 1u << (depth - 1u);
                const uint32_t sibling = (node & 1u) ? node + 1u : node - 1u;
                if ((kept_d < (R)0 ? -kept_d : kept_d) < sc.photon_radius && (cnt < k || kept_d * kept_d < top) && sibling < n) {
                    node = sibling; arriving = true; continue;
                }
            }
        }
        // back up to the nearest ancestor whose near side this was (those in their far side are done)
        const uint32_t open = ~in_far & ((1u << depth) - 1u);
        if (open == 0u) walking = false;
        else {
            const uint32_t level = 31u - (uint32_t)__builtin_clz(open);
            node = ((node + 1u) >> (depth - level)) - 1u;
            depth = level;
            arriving = false;
        }
    }
    const R square_radius = cnt ? top : (R)1; // (unwrap_or(1.0), photon_map.rs:318)
    V3<R> total = mk<R>(0, 0, 0);
    for (int i = 0; i < cnt; ++i) {
        const DPhoton &ph = sc.photons[min(heap_ld(i).n, n - 1u)];
        const R w = max_(dot(normalv, mk<R>((R)ph.dir[0], (R)ph.dir[1], (R)ph.dir[2])), (R)0);
        const R x = (R)ph.scale;
        // decompress_color (photon_map.rs:41-47): byte * scale / 255
        const V3<R> power = mk<R>(div_((R)(ph.rgbf & 255u) * x, (R)255), div_((R)((ph.rgbf >> 8) & 255u) * x, (R)255), div_((R)((ph.rgbf >> 16) & 255u) * x, (R)255));
        total = total + power * w;
    }
    st.fl += visited * F_PHOTON_VISIT + (uint32_t)cnt * F_PHOTON_GATHER + F_PHOTON_ESTIMATE;
    return had(surface_color, total) * rcp_((R)3.14159265358979323846 * square_radius); // (Color / f64 is scale(1 / x), color.rs:145-149)
}

// World::color_at (world.rs:115-118) with Phong::shade_hit (materials.rs:371-412) unrolled into a loop:
// shade_hit is linear in its children's colours, so each ray carries the product of the factors
// above it ("weight") and adds weight * (clipped direct light + emissive) to the sample.
// PHOTON: the scene may carry a photon map (kernels of their own: the search's per-lane arrays are not for the others to pay)
template <typename R, int TREE, bool PHOTON = false>
__device__ __forceinline__ V3<R> color_at(const DScene<R> &sc, Ray<R> ray, int remaining, Rng &rng, Stats &st, Hit<R> *first_hit, Comps<R> *first_comps,
                                          HeapEntry<R> *heap_ws = nullptr) {
    const R EPS = (R)RTMI_EPSILON;
    V3<R> acc = mk<R>(0, 0, 0);
    R weight = (R)1;
    uint32_t origin_leaf = NONE, origin_side = 0;
    int cls = RTMI_STAT_PRIMARY;
    constexpr int LP = TREE == KIND_TREE ? LDS_PEND_TREE : LDS_PEND;
    PendingStack<R, LP> pend;
    __shared__ uint32_t lds_pend[sizeof(R) == 4 ? LP : 1][8][64];
    bool first = true;
    for (;;) {
        if (cls == RTMI_STAT_PRIMARY) st.inc(RTMI_STAT_PRIMARY);
        else if (cls == RTMI_STAT_REFLECTION) st.inc(RTMI_STAT_REFLECTION);
        else st.inc(RTMI_STAT_REFRACTION);
        Hit<R> h = {Lim<R>::inf(), NONE, 0};
        Open<R> unused;
        traverse<R, MODE_CLOSEST, TREE>(sc, ray, origin_leaf, origin_side, h, Lim<R>::inf(), NONE, 0, unused, st);
        bool spawned = false;
        if (h.leaf != NONE && h.leaf >= sc.n_leaves) { st.flag(RTMI_STAT_OVERFLOW); h.leaf = NONE; } // (never: a leaf index must not become an address unchecked)
        if (h.leaf != NONE) {
            Comps<R> c;
            uint32_t kind, material;
            CYC3_BEGIN(c_prep);
            prepare<R, TREE>(sc, ray, h, c, kind, material);
            const DMaterial<R> m = load_material<R, TREE>(sc, material);
            V3<R> over_point = c.point + c.normalv * EPS;
            // Phong::color_at (materials.rs:239-244): patterns are looked up at comps.point
            V3<R> surface_color = ld3(m.color);
            st.fl += F_PREPARE + (((kind & LEAF_KIND_MASK) == RTMI_SMOOTH_TRIANGLE) ? F_SMOOTH_NORMAL : 0u) + F_COMBINE;
            if (m.pattern >= 0 && !(RTMI_SKIP & 1)) {
                bool done = false;
                if constexpr (staged_v<R, TREE>) {
                    const uint32_t slot = h.leaf < sc.n_quick ? lds_scene()[h.leaf * (uint32_t)QUICK_DW + 2u] : NONE;
                    if (slot != NONE) { // world point -> pattern space in one step, from the LDS block
                        const DPattern<float> &pat = *reinterpret_cast<const DPattern<float> *>(lds_scene() + LDS_PATS_AT + slot * (uint32_t)PAT_DW);
                        surface_color = pattern_at(pat, c.point);
                        st.fl += F_PATTERN_XF + F_PATTERN_EVAL + (pat.pad ? F_PATTERN_XF : 0u); // (priced like the reference's two steps)
                        done = true;
                    }
                }
                if (!done) {
                    const DPattern<R> &pat = sc.patterns[m.pattern];
                    surface_color = pattern_at(pat, xf_point(sc.leaves[h.leaf].inv, c.point));
                    st.fl += F_PATTERN_XF + F_PATTERN_EVAL + (pat.pad ? F_PATTERN_XF : 0u);
                }
            }
            CYC3_END(st, 1, c_prep);
            V3<R> surface = mk<R>(0, 0, 0);
            if constexpr (PHOTON) {
                if (sc.n_photons) surface = photon_estimate(sc, c.point, c.normalv, surface_color, st, heap_ws); // materials.rs:376-384
            }
            const uint32_t n_direct = (PHOTON && sc.no_direct) ? 0u : sc.n_lights; // world.direct_illumination_enabled(), materials.rs:386
            for (uint32_t li = 0; li < n_direct; ++li) { // materials.rs:386-397
                Incoming<R> in;
                bool staged = false;
                if constexpr (staged_v<R, TREE>) {
                    if (sc.n_lds_lights) {
                        in = incoming_at(*reinterpret_cast<const DLight<float> *>(lds_scene() + LDS_LIGHTS_AT + li * (uint32_t)LIGHT_DW), over_point, rng);
                        staged = true;
                    }
                }
                if (!staged) in = incoming_at(sc.lights[li], over_point, rng);
                bool in_shadow = is_shadowed<R, TREE>(sc, in, over_point, h.leaf, c.side_plus ? SIDE_PLUS : 0u, st);
                st.fl += in.kind == IL_RAY ? (in_shadow ? F_LIGHT_SHADOWED : F_LIGHT_RAY) : (in.kind == IL_OMNI ? F_LIGHT_OMNI : 0u);
                CYC3_BEGIN(c_light);
                if (RTMI_SKIP & 2) surface = surface + surface_color * (in_shadow ? (R)0 : (R)0.5); else
                surface = surface + lighting(m, surface_color, in, c.eyev, c.normalv, in_shadow);
                CYC3_END(st, 2, c_light);
            }
            surface = mk<R>(min_(max_(surface.x, (R)0), (R)1), min_(max_(surface.y, (R)0), (R)1), min_(max_(surface.z, (R)0), (R)1)); // :399
            acc = acc + (surface + surface_color * m.emissive) * weight;

            bool want_n = first_comps && first;
            bool do_refl = !(m.reflective == (R)0) && remaining > 0;   // materials.rs:297
            bool do_refr = !(m.transparency == (R)0) && remaining > 0; // materials.rs:316
            if ((do_refr || want_n) && !(RTMI_SKIP & 8)) collect_n1n2<R, TREE>(sc, ray, origin_leaf, origin_side, h, c, st);
            CYC4_BEGIN(c_spawn);
            if (first) {
                if (first_hit) *first_hit = h;
                if (first_comps) *first_comps = c;
            }
            R w_refl = weight * m.reflective, w_refr = weight * m.transparency;
            if (m.reflective > (R)0 && m.transparency > (R)0 && remaining > 0) { // materials.rs:406-408
                R reflectance = schlick(c.eyev, c.normalv, c.n1, c.n2);
                st.fl += F_SCHLICK;
                w_refl *= reflectance;
                w_refr *= (R)1 - reflectance;
            }
            Ray<R> refr_ray = ray;
            st.fl += (do_refl ? F_WEIGHT : 0u) + (do_refr ? F_WEIGHT + F_REFRACT_DIR : 0u);
            if (do_refr) { // materials.rs:319-326
                R n_ratio = div_(c.n1, c.n2);
                R cos_i = dot(c.eyev, c.normalv);
                R sin2_t = n_ratio * n_ratio * ((R)1 - cos_i * cos_i);
                if (sin2_t > (R)1) {
                    do_refr = false;
                } else {
                    R cos_t = sqrt_((R)1 - sin2_t);
                    refr_ray.o = c.point - c.normalv * EPS;
                    refr_ray.d = c.normalv * (n_ratio * cos_i - cos_t) - c.eyev * n_ratio;
                }
            }
            if (do_refl) {
                if (do_refr && pend.full()) st.flag(RTMI_STAT_OVERFLOW); // (unreachable while max_depth <= MAX_DEPTH: the ray would be lost)
                if (do_refr && !pend.full()) {
                    Pending<R> e;
                    e.o = refr_ray.o; e.d = refr_ray.d; e.weight = w_refr;
                    e.remaining = remaining - 1; e.origin_leaf = h.leaf;
                    e.origin_side = c.side_plus ? 0u : SIDE_PLUS; // under_point: the other side
                    pend.push(lds_pend, e);
                }
                ray.d = ray.d - c.normalv * ((R)2 * dot(ray.d, c.normalv)); // reflectv, ray.rs:84
                ray.o = over_point;
                weight = w_refl; remaining -= 1; origin_leaf = h.leaf; origin_side = c.side_plus ? SIDE_PLUS : 0u; cls = RTMI_STAT_REFLECTION;
                spawned = true;
            } else if (do_refr) {
                ray = refr_ray;
                weight = w_refr; remaining -= 1; origin_leaf = h.leaf; origin_side = c.side_plus ? 0u : SIDE_PLUS; cls = RTMI_STAT_REFRACTION;
                spawned = true;
            }
            CYC4_END(st, 0, c_spawn);
        } else if (first && first_hit) {
            *first_hit = h;
        }
        first = false;
        if (!spawned) {
            if (pend.empty()) break;
            CYC4_BEGIN(c_pop);
            const Pending<R> e = pend.pop(lds_pend);
            ray.o = e.o; ray.d = e.d; weight = e.weight;
            remaining = e.remaining; origin_leaf = e.origin_leaf; origin_side = e.origin_side;
            cls = RTMI_STAT_REFRACTION;
            CYC4_END(st, 1, c_pop);
        }
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// World::compute_photon_map's tracing half (world.rs:186-273) - rtmi_trace_photons
// ---------------------------------------------------------------------------------------------
struct DPhotonTrace {
    uint32_t seed_lo, seed_hi; // (the photon stream's key: the caller's seed with "PHOT" xored into the high word)
    uint32_t first, end, max_bounces, pad; // this call's photons: first .. end - 1 (counts / offsets are indexed from `first`)
    uint32_t enable_direct, enable_diffuse, enable_caustic, direct_only; // world.rs:64-66,130-134
    float inv_n;  // scale_power(1 / n_photons), world.rs:200
    uint32_t cap; // photons `out` holds (the storing pass)
};
enum { PT_RAYS = 0, PT_TRUNCATED = 1, PT_BAD_LEAF = 2, PT_COUNTERS = 4 };
enum { PK_DIRECT = 0, PK_CAUSTIC = 1, PK_DIFFUSE = 2 }; // PhotonKind, photon_map.rs:185-190

__device__ __forceinline__ float log_(float x) { return logf(x); }
__device__ __forceinline__ double log_(double x) { return log(x); }
__device__ __forceinline__ float third_(float x) { return x * (1.0f / 3.0f); }
__device__ __forceinline__ double third_(double x) { return x / 3.0; } // Color::sum() / 3.0 as the reference rounds it

// UnitSphere by inverse CDF (two draws), the map incoming_at uses for SphereLight
template <typename R> __device__ __forceinline__ V3<R> unit_sphere(Rng &rng) {
    R u1 = rng_next<R>(rng), u2 = rng_next<R>(rng);
    R z = (R)1 - (R)2 * u1, s, c;
    sincos2pi(u2, s, c);
    R rr = sqrt_(max_((R)0, (R)1 - z * z));
    return mk<R>(rr * c, rr * s, z);
}
// CosineDistribution::sample (cosine_distribution.rs:16-28); the loop is cut after 16 rounds (the direction itself)
template <typename R> __device__ __forceinline__ V3<R> cosine_sample(V3<R> dir, Rng &rng) {
    for (int k = 0; k < 16; ++k) {
        V3<R> v = dir + unit_sphere<R>(rng);
        R l = dot(v, v);
        if (l > (R)RTMI_EPSILON) return v * rsqrt_(l);
    }
    return dir;
}
// Light::power (lights.rs:68-70 and each light's `power`): the ambient light has none
template <typename R> __device__ __forceinline__ R light_power(const DLight<R> &l) {
    return l.kind == RTMI_LIGHT_AMBIENT ? (R)0 : third_(l.intensity[0] + l.intensity[1] + l.intensity[2]);
}
// Light::emit_photon (lights.rs:104-111,169-176,252-261,371-380,462-473)
template <typename R> __device__ __forceinline__ void light_emit(const DLight<R> &l, Rng &rng, Ray<R> &ray, V3<R> &power) {
    const R TWO_OVER_PI = (R)(2.0 / 3.14159265358979323846);
    V3<R> pos = ld3(l.pos);
    ray.o = pos; power = ld3(l.intensity); ray.d = mk<R>(0, 0, 0);
    switch (l.kind) {
    case RTMI_LIGHT_POINT: ray.d = unit_sphere<R>(rng); break;
    case RTMI_LIGHT_REALISTIC_POINT: ray.d = unit_sphere<R>(rng); power = power * TWO_OVER_PI; break;
    case RTMI_LIGHT_BEAM: { // two StandardNormal draws: Box-Muller on two uniforms
        R u1 = rng_next<R>(rng), u2 = rng_next<R>(rng), s, c;
        R r = sqrt_((R)-2 * log_((R)1 - u1));
        sincos2pi(u2, s, c);
        ray.o = pos + ld3(l.up) * (r * c) + ld3(l.right) * (r * s);
        ray.d = ld3(l.dir);
        power = power * TWO_OVER_PI;
        break;
    }
    case RTMI_LIGHT_SPHERE: {
        V3<R> p = unit_sphere<R>(rng);
        ray.d = cosine_sample(p, rng);
        ray.o = pos + p * l.radius;
        break;
    }
    case RTMI_LIGHT_DISC: {
        R u1 = rng_next<R>(rng), u2 = rng_next<R>(rng);
        R rr = sqrt_(u1), s, c;
        sincos2pi(u2, s, c);
        ray.o = pos + ld3(l.width) * (rr * c) + ld3(l.height) * (rr * s);
        ray.d = cosine_sample(ld3(l.normal), rng);
        power = power * ((R)0.25 * TWO_OVER_PI);
        break;
    }
    default: power = mk<R>(0, 0, 0);
    }
}
// StoredPhoton::new + scale_power (photon_map.rs:23-39,61-64): f32 position and direction, the power as an f32 scale
// and three bytes (f32 arithmetic whatever R is; a NaN quotient, 0 / 0, becomes 0 as Rust's f32::max makes it)
template <typename R> __device__ __forceinline__ void store_photon(rtmi_photon *p, V3<R> point, V3<R> travel, V3<R> power, float inv_n) {
    p->position[0] = (float)point.x; p->position[1] = (float)point.y; p->position[2] = (float)point.z;
    p->direction[0] = (float)-travel.x; p->direction[1] = (float)-travel.y; p->direction[2] = (float)-travel.z;
    const float x = (float)max_(max_(power.x, power.y), power.z);
    const float ch[3] = {(float)power.x, (float)power.y, (float)power.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = 255.0f * ch[k] / x;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        p->power_rgb[k] = (uint8_t)v;
    }
    p->reserved = 0;
    p->power_scale = x * inv_n;
}

// One lane traces the photons first, first + stride, ... : emission and bounces are ONE loop, so a lane whose photon
// has ended emits its next one while the wave's other lanes are still bouncing theirs.  Every turn of the loop either
// ends a photon or spends one of its max_bounces, and the photon index only grows: every lane leaves the loop.
// The call runs the photons twice (same stream, same paths): STORE false counts the hits photon i keeps into counts[i];
// after an exclusive scan of the counts, STORE true writes them at offsets[i], i + 1, ... - the reference's order
// (world.rs:195-201: by photon, then by bounce) with no sorting and no guess at the buffer's size.
template <typename R, bool STORE>
__device__ __forceinline__ void trace_photons(const DScene<R> &sc, const DPhotonTrace &pt, uint32_t first, uint32_t stride, rtmi_photon *out,
                                              uint32_t *counts, const uint32_t *offsets, unsigned long long *counters, Stats &st) {
    const R EPS = (R)RTMI_EPSILON;
    unsigned long long rays = 0, truncated = 0;
    uint32_t next = pt.end - pt.first > first ? pt.first + first : pt.end, idx = 0, bounce = 0, origin_leaf = NONE, origin_side = 0, mine = 0, base = 0;
    bool flush = false; // (counting: photon idx has ended and its count is not written yet)
    int kind = PK_DIRECT;
    bool alive = false;
    Ray<R> ray = {mk<R>(0, 0, 0), mk<R>(0, 0, 1)};
    V3<R> power = mk<R>(0, 0, 0);
    Rng rng;
    rng.k0 = pt.seed_lo; rng.k1 = pt.seed_hi;
    R total_power = 0;
    for (uint32_t i = 0; i < sc.n_lights; ++i) total_power += light_power(sc.lights[i]);
    for (;;) {
        if (!alive) {
            if constexpr (!STORE) {
                if (flush) counts[idx - pt.first] = mine;
                flush = false;
            }
            if (next >= pt.end) break;
            idx = next;
            mine = 0; flush = true;
            if constexpr (STORE) base = offsets[idx - pt.first];
            next = (pt.end - next > stride) ? next + stride : pt.end; // (no wrap-around near 2^32)
            rng_begin(rng, idx, 0);
            // World::emit_photon (world.rs:211-215): WeightedIndex over the lights' power
            const R x = rng_next<R>(rng) * total_power;
            R cum = 0;
            uint32_t li = 0;
            for (uint32_t i = 0; i < sc.n_lights; ++i) {
                const R w = light_power(sc.lights[i]);
                if (w > (R)0) li = i; // (a rounded-up x falls to the last light with power)
                cum += w;
                if (x < cum) break;
            }
            light_emit(sc.lights[li], rng, ray, power);
            kind = PK_DIRECT; bounce = 0; origin_leaf = NONE; origin_side = 0;
            alive = true;
        }
        // World::trace_photon (world.rs:217-273), one turn of its loop
        if (bounce >= pt.max_bounces) { truncated++; alive = false; continue; }
        bounce++;
        if (power.x + power.y + power.z < EPS) { alive = false; continue; }      // world.rs:224-226
        if (kind != PK_DIRECT && pt.direct_only) { alive = false; continue; }     // world.rs:228-232
        Hit<R> h = {Lim<R>::inf(), NONE, 0};
        Open<R> unused;
        traverse<R, MODE_CLOSEST, KIND_TREE>(sc, ray, origin_leaf, origin_side, h, Lim<R>::inf(), NONE, 0, unused, st);
        rays++;
        if (h.leaf != NONE && h.leaf >= sc.n_leaves) h.leaf = NONE; // (never: see color_at)
        if (h.leaf == NONE) { alive = false; continue; }
        Comps<R> c;
        uint32_t lkind, material;
        prepare<R, KIND_TREE>(sc, ray, h, c, lkind, material);
        const DMaterial<R> m = sc.materials[material];
        V3<R> surface_color = ld3(m.color);
        if (m.pattern >= 0) surface_color = pattern_at(sc.patterns[m.pattern], xf_point(sc.leaves[h.leaf].inv, c.point));
        // Phong::photon_hit (materials.rs:414-464)
        const V3<R> dr = surface_color * m.diffuse;
        R spec = max_(m.specular, m.reflective), trans = m.transparency;
        const R dr_avg = third_(dr.x + dr.y + dr.z);
        R pd_avg = dr_avg;
        const bool store = pd_avg > EPS;
        if (!pt.enable_diffuse) pd_avg = 0;
        if (m.transparency > (R)0) collect_n1n2<R, KIND_TREE>(sc, ray, origin_leaf, origin_side, h, c, st); // (n1, n2 are read by schlick and refract only)
        if (m.reflective > (R)0 && m.transparency > (R)0) {
            const R r = schlick(c.eyev, c.normalv, c.n1, c.n2);
            spec *= r; trans *= (R)1 - r;
        }
        const R p_absorb = (R)1 - pd_avg - spec - trans;
        if (!(p_absorb >= (R)0) || p_absorb > (R)1) { // materials.rs:443-445 asserts 0 <= p_absorb <= 1 and panics: the call fails and names the leaf
            if constexpr (!STORE) counters[PT_BAD_LEAF] = (unsigned long long)sc.leaf_orig[h.leaf] + 1ull;
            alive = false;
            continue;
        }
        // WeightedIndex over [absorb, diffuse, specular, transmit]: the first whose cumulative weight exceeds x
        const R c0 = p_absorb, c1 = c0 + pd_avg, c2 = c1 + spec, c3 = c2 + trans;
        const R x = rng_next<R>(rng) * c3;
        int choice = x < c0 ? 0 : x < c1 ? 1 : x < c2 ? 2 : 3;
        if (choice == 3 && !(trans > (R)0)) choice = spec > (R)0 ? 2 : pd_avg > (R)0 ? 1 : 0; // (x rounded up to the total)
        if (store && ((kind == PK_DIRECT && pt.enable_direct) || (kind == PK_DIFFUSE && pt.enable_diffuse) || (kind == PK_CAUSTIC && pt.enable_caustic))) {
            if constexpr (STORE) { // world.rs:256-265
                const unsigned long long slot = (unsigned long long)base + mine;
                if (slot < (unsigned long long)pt.cap) store_photon(&out[slot], c.point, ray.d, power, pt.inv_n);
            }
            mine++;
        }
        const uint32_t side_over = c.side_plus ? SIDE_PLUS : 0u, side_under = c.side_plus ? 0u : SIDE_PLUS;
        if (choice == 0) { alive = false; continue; }
        if (choice == 1) { // TravellingPhoton::scatter (photon_map.rs:144-151)
            ray.o = c.point + c.normalv * EPS;
            ray.d = cosine_sample(c.normalv, rng);
            power = had(power, dr) * rcp_(dr_avg);
            kind = PK_DIFFUSE; origin_side = side_over;
        } else if (choice == 2) { // reflect (photon_map.rs:153-159)
            ray.d = ray.d - c.normalv * ((R)2 * dot(ray.d, c.normalv));
            ray.o = c.point + c.normalv * EPS;
            if (kind == PK_DIRECT) kind = PK_CAUSTIC;
            origin_side = side_over;
        } else { // refract (photon_map.rs:161-172)
            const R n_ratio = div_(c.n1, c.n2);
            const R cos_i = -dot(ray.d, c.normalv);
            const R sin2_t = n_ratio * n_ratio * ((R)1 - cos_i * cos_i);
            if (sin2_t > (R)1) { alive = false; continue; } // (the reference takes the root of a negative number: no answer)
            const R cos_t = sqrt_((R)1 - sin2_t);
            ray.d = c.normalv * (n_ratio * cos_i - cos_t) + ray.d * n_ratio;
            ray.o = c.point - c.normalv * EPS;
            if (kind == PK_DIRECT) kind = PK_CAUSTIC;
            origin_side = side_under;
        }
        origin_leaf = h.leaf;
    }
    if constexpr (!STORE) {
        if (rays) atomicAdd(&counters[PT_RAYS], rays);
        if (truncated) atomicAdd(&counters[PT_TRUNCATED], truncated);
    }
}

// Camera::ray_for_pixel (camera.rs:120-153)
template <typename R> __device__ __forceinline__ uint32_t camera_flops(const DCamera<R> &cam) {
    return F_CAMERA + (cam.aperture_size == (R)0 ? 0u : F_CAMERA_LENS);
}
template <typename R> __device__ __forceinline__ Ray<R> ray_for_pixel(const DCamera<R> &cam, int px, int py, bool randomize, Rng &rng);
// The camera ray is always generated in f64 and then rounded to R: world_y = half_height - y_offset and the view
// rotation cancel to ~1e-4 for rays that graze the horizon, and an f32 error there moves the hit on a ground plane by
// whole checker cells (t ~ 1/d_y).  ~85 f64 FLOPs per sample against thousands of f32 FLOPs of tracing.
template <typename R> __device__ __forceinline__ Ray<R> camera_ray(const DCamera<double> &cam, int px, int py, bool randomize, Rng &rng) {
    Ray<double> r = ray_for_pixel<double>(cam, px, py, randomize, rng);
    Ray<R> q = {mk<R>((R)r.o.x, (R)r.o.y, (R)r.o.z), mk<R>((R)r.d.x, (R)r.d.y, (R)r.d.z)};
    return q;
}
template <typename R> __device__ __forceinline__ Ray<R> ray_for_pixel(const DCamera<R> &cam, int px, int py, bool randomize, Rng &rng) {
    R jx = (R)0.5, jy = (R)0.5;
    if (randomize) { jx = rng_next<R>(rng); jy = rng_next<R>(rng); }
    R x_offset = ((R)px + jx) * cam.pixel_size;
    R y_offset = ((R)py + jy) * cam.pixel_size;
    R world_x = cam.half_width - x_offset, world_y = cam.half_height - y_offset;
    V3<R> pixel = xf_point(cam.inv, mk<R>(world_x, world_y, (R)-1));
    V3<R> origin = mk<R>(cam.inv[3], cam.inv[7], cam.inv[11]);
    Ray<R> r = {origin, normalize(pixel - origin)};
    if (cam.aperture_size == (R)0) return r;
    V3<R> focal_point = r.o + r.d * cam.focal_distance;
    R a0 = rng_next<R>(rng), a1 = rng_next<R>(rng);
    V3<R> ap_pixel = pixel + mk<R>((a0 - (R)0.5) * cam.aperture_size, (a1 - (R)0.5) * cam.aperture_size, (R)0);
    Ray<R> s = {ap_pixel, normalize(focal_point - ap_pixel)};
    return s;
}

} // namespace rtmi_dev
