// rtmi.hip — C ABI (include/rtmi.h) of the MI355X ray-casting core: scene layout + upload, kernels, launches.
// gfx950 only.  No CPU fallback: without a usable HIP device every compute entry point fails with
// RTMI_ERR_DEVICE and a message.
#include "rtmi_launch.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <atomic>
#include <chrono>
#include <vector>

using namespace rtmi_dev;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
extern "C" const char *rtmi_last_error(void) { return g_err.c_str(); }

static long env_long(const char *name, long fallback) {
    const char *e = getenv(name);
    return e && *e ? atol(e) : fallback;
}

#ifndef RTMI_BVH_LEAF
#define RTMI_BVH_LEAF 4 // most triangles in a leaf of the device's own triangle BVH (RTMI_BVH_LEAF in the environment at scene creation replaces it)
#endif

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// kernels: rtmi_kernels.hip (built once per arithmetic type); only the type-free tone map lives here
// ---------------------------------------------------------------------------------------------

// Color::to_u8 (color.rs:53-60): clip, x255, round half away from zero.  4 values per lane.
__global__ void tonemap_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, long long n) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n && (((uintptr_t)(in + i)) & 15) == 0 && (((uintptr_t)(out + i)) & 3) == 0) {
        float4 v = *reinterpret_cast<const float4 *>(in + i);
        uint32_t a = (uint32_t)floorf(fminf(fmaxf(v.x, 0.f), 1.f) * 255.f + 0.5f);
        uint32_t b = (uint32_t)floorf(fminf(fmaxf(v.y, 0.f), 1.f) * 255.f + 0.5f);
        uint32_t c = (uint32_t)floorf(fminf(fmaxf(v.z, 0.f), 1.f) * 255.f + 0.5f);
        uint32_t d = (uint32_t)floorf(fminf(fmaxf(v.w, 0.f), 1.f) * 255.f + 0.5f);
        *reinterpret_cast<uint32_t *>(out + i) = a | (b << 8) | (c << 16) | (d << 24);
    } else {
        for (long long k = i; k < n && k < i + 4; ++k) out[k] = (uint8_t)floorf(fminf(fmaxf(in[k], 0.f), 1.f) * 255.f + 0.5f);
    }
}

// rank r's compact rows (tile t of the rank = tile t * nranks + r of the frame) to their places in the whole frame
__global__ void deinterleave_kernel(const float *__restrict__ src, float *__restrict__ dst, long long n_values, int row_values, int tile_rows, int rank,
                                    int nranks) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / row_values, col = i - row * row_values;
        const long long tile = row / tile_rows, within = row - tile * tile_rows;
        dst[((tile * nranks + rank) * tile_rows + within) * row_values + col] = src[i];
    }
}

// Exclusive scan of the per-photon hit counts (rtmi_trace_photons): blocks of 1024 counts (256 lanes x 4), their sums
// scanned by one block, then the offsets.  `sums` holds n_blocks + 1 values; the last one is the total.
#ifndef RTMI_HOST_EMUL
constexpr uint32_t SCAN_BLOCK = 1024;
template <bool APPLY>
__global__ void __launch_bounds__(256) scan_counts_kernel(const uint32_t *__restrict__ counts, uint32_t n, unsigned long long *__restrict__ sums,
                                                          uint32_t *__restrict__ offsets) {
    __shared__ uint32_t part[256];
    const uint32_t t = threadIdx.x;
    const unsigned long long at = (unsigned long long)blockIdx.x * SCAN_BLOCK + t * 4u;
    uint32_t v[4], own = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = at + k < n ? counts[at + k] : 0u; own += v[k]; }
    part[t] = own;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) { // inclusive scan of the lanes' sums (a block's counts stay far below 2^32)
        const uint32_t x = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    if constexpr (!APPLY) {
        if (t == 255) sums[blockIdx.x] = part[255];
    } else {
        unsigned long long run = sums[blockIdx.x] + (part[t] - own);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (at + k < n) offsets[at + k] = (uint32_t)run; // (the host refuses totals beyond 32 bits before anything reads these)
            run += v[k];
        }
    }
}
__global__ void __launch_bounds__(1024) scan_sums_kernel(unsigned long long *sums, uint32_t n_blocks) {
    __shared__ unsigned long long part[1024];
    const uint32_t t = threadIdx.x;
    unsigned long long carry = 0;
    for (uint32_t first = 0; first < n_blocks; first += 1024) { // (uniform trip count: every lane meets every barrier)
        const uint32_t i = first + t;
        const unsigned long long own = i < n_blocks ? sums[i] : 0ull;
        part[t] = own;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const unsigned long long x = t >= off ? part[t - off] : 0ull;
            __syncthreads();
            part[t] += x;
            __syncthreads();
        }
        if (i < n_blocks) sums[i] = carry + part[t] - own;
        carry += part[1023];
        __syncthreads();
    }
    if (t == 0) sums[n_blocks] = carry;
}
static void scan_counts(const uint32_t *counts, uint32_t n, unsigned long long *sums, uint32_t *offsets, hipStream_t st) {
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(scan_counts_kernel<false>, dim3(nb), dim3(256), 0, st, counts, n, sums, offsets);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, nb);
    hipLaunchKernelGGL(scan_counts_kernel<true>, dim3(nb), dim3(256), 0, st, counts, n, sums, offsets);
}
#else
constexpr uint32_t SCAN_BLOCK = 1024;
static void scan_counts(const uint32_t *counts, uint32_t n, unsigned long long *sums, uint32_t *offsets, hipStream_t) { // (host emulation: "device" memory is host memory)
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    unsigned long long run = 0;
    for (uint32_t i = 0; i < n; ++i) { offsets[i] = (uint32_t)run; run += counts[i]; }
    sums[nb] = run;
}
#endif

// ---------------------------------------------------------------------------------------------
// host: scene layout
// ---------------------------------------------------------------------------------------------
template <typename R> struct DeviceArrays {
    DLeaf<R> *leaves = nullptr;
    DTri<R> *tris = nullptr;
    DMaterial<R> *materials = nullptr;
    DPattern<R> *patterns = nullptr;
    DLight<R> *lights = nullptr;
    DCsgLeaf *csg_leaf = nullptr;
};

// Work-queue memory of the adaptive second pass (rtmi_launch.hpp DWork): a per-device pool of up to WORK_SLOTS buffers.
// RTMI_WORK_SLOTS_PREALLOC of them (environment, read when the device's first scene is created; default 2, at most
// WORK_SLOTS) are created with the first scene on the device - a caller that will render on k streams at once sets it to
// k and its enqueue path (rtmi_render_device: "asynchronous, allocates nothing") never meets a hipMalloc; a caller that did
// not reserve enough gets the missing slots the first time that many streams render at once (a one-off stall at that call,
// and RTMI_ERR_NOMEM there if the device is full - reserve up front, or use rtmi_render_device_noalloc, to rule that out).
// All are freed with the device's last scene.  A slot belongs to a stream while work of that stream is in flight on it
// (work on one stream is ordered, so one slot per stream is race-free); after every call an event is recorded on the
// stream, and a slot whose event has completed may be given to another stream.  A process may therefore use any number
// of streams over its life, WORK_SLOTS of them rendering at the same time.
#ifndef RTMI_WORK_SLOTS
#define RTMI_WORK_SLOTS 4
#endif
constexpr int WORK_SLOTS = RTMI_WORK_SLOTS;
constexpr size_t WORK_BYTES = (size_t)2 << 30; // 2 x 11 M queued pixels (f32: a whole 3840x2160 frame per queue) + 370 k units of 64 samples
// RTMI_WORK_MB (read once per process): the slot size in MiB instead, for tests that want queues, unit lists and the
// heavy table to overflow (every overflow path is exact, only slower)
static size_t work_bytes() {
    static const size_t v = [] {
        const char *e = getenv("RTMI_WORK_MB");
        const long mb = e ? atol(e) : 0;
        return mb >= 2 && mb <= 65536 ? (size_t)mb << 20 : WORK_BYTES;
    }();
    return v;
}
constexpr int MAX_DEVICES = 64;
struct WorkPool {
    int refs = 0;
    void *buf[WORK_SLOTS] = {};
    hipEvent_t done[WORK_SLOTS] = {}; // recorded after the last call that used the slot
    hipStream_t stream[WORK_SLOTS] = {};
    bool bound[WORK_SLOTS] = {};      // stream[i] has (or had, until done[i] completes) work on the slot
    unsigned long long stamp[WORK_SLOTS] = {};
    unsigned long long clock = 0;
};
static std::mutex g_pool_mutex;
static WorkPool g_pool[MAX_DEVICES];

static int pool_make_slot(WorkPool &p, int i) { // (g_pool_mutex held, the pool's device current)
    if (p.buf[i]) return RTMI_OK;
    hipError_t e = hipMalloc(&p.buf[i], work_bytes());
    if (e != hipSuccess) { p.buf[i] = nullptr; return fail(RTMI_ERR_NOMEM, std::string("work-queue pool: ") + hipGetErrorString(e)); }
    e = hipEventCreateWithFlags(&p.done[i], hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipFree(p.buf[i]); p.buf[i] = nullptr; return fail(RTMI_ERR_DEVICE, std::string("work-queue pool event: ") + hipGetErrorString(e)); }
    p.bound[i] = false;
    return RTMI_OK;
}

static int pool_acquire(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return fail(RTMI_ERR_UNSUPPORTED, "device ordinal beyond 63");
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    WorkPool &p = g_pool[dev];
    if (p.refs == 0) {
        const long want = std::max(1l, std::min((long)WORK_SLOTS, env_long("RTMI_WORK_SLOTS_PREALLOC", 2)));
        for (int i = 0; i < (int)want; ++i) {
            int rc = pool_make_slot(p, i);
            if (rc != RTMI_OK) { // (none or all: a half-made pool is freed again)
                for (int k = 0; k < i; ++k) { (void)hipFree(p.buf[k]); (void)hipEventDestroy(p.done[k]); p.buf[k] = nullptr; p.done[k] = nullptr; }
                return rc;
            }
        }
    }
    ++p.refs;
    return RTMI_OK;
}

static void pool_release(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    WorkPool &p = g_pool[dev];
    if (p.refs > 0 && --p.refs == 0)
        for (int i = 0; i < WORK_SLOTS; ++i) {
            if (p.buf[i]) { (void)hipFree(p.buf[i]); (void)hipEventDestroy(p.done[i]); }
            p.buf[i] = nullptr; p.done[i] = nullptr; p.bound[i] = false;
        }
}

// The slot for a call on stream `st` (g_pool_mutex held): the one the stream's earlier work is on, else an idle one -
// never used, or whose last call has completed (the least recently used of those) - else a new one; -1 when WORK_SLOTS
// streams have work in flight.
static int pool_slot(int dev, hipStream_t st, int *slot) {
    WorkPool &p = g_pool[dev];
    *slot = -1;
    for (int i = 0; i < WORK_SLOTS; ++i)
        if (p.buf[i] && p.bound[i] && p.stream[i] == st) { *slot = i; return RTMI_OK; }
    int idle = -1;
    for (int i = 0; i < WORK_SLOTS; ++i) {
        if (!p.buf[i]) continue;
        if (p.bound[i] && hipEventQuery(p.done[i]) != hipSuccess) continue; // its stream is still at work
        p.bound[i] = false;
        if (idle < 0 || p.stamp[i] < p.stamp[idle]) idle = i;
    }
    if (idle < 0)
        for (int i = 0; i < WORK_SLOTS && idle < 0; ++i)
            if (!p.buf[i]) {
                int rc = pool_make_slot(p, i);
                if (rc != RTMI_OK) return rc;
                idle = i;
            }
    if (idle < 0)
        return fail(RTMI_ERR_UNSUPPORTED, "more than " + std::to_string(WORK_SLOTS) + " streams are rendering on this device at the same time (one work-queue slot per stream with work in flight)");
    p.bound[idle] = true; p.stream[idle] = st;
    *slot = idle;
    return RTMI_OK;
}

// rtmi_render_device_noalloc: the same work memory as a file-scope array in device memory, for callers that cannot
// allocate (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Every GPU the library runs on
// holds its own copy of the array, at its own address, until the process ends; `scratch` itself is only the host-side
// stand-in of the symbol, never a device address: host code uses the per-device pointer hipGetSymbolAddress gives.
#ifndef RTMI_STATIC_SCRATCH_MB
#define RTMI_STATIC_SCRATCH_MB 96
#endif
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)RTMI_STATIC_SCRATCH_MB << 20;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
static void *g_scratch_ptr[MAX_DEVICES] = {};

// the address of `scratch` on device `dev`, which must be the current device (looked up once per device)
static int static_scratch(int dev, void **out) {
    *out = nullptr;
    if (dev < 0 || dev >= MAX_DEVICES) return fail(RTMI_ERR_UNSUPPORTED, "device ordinal beyond 63");
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (!g_scratch_ptr[dev]) {
        int cur = -1;
        HIP_TRY(hipGetDevice(&cur));
        if (cur != dev) return fail(RTMI_ERR_DEVICE, "the scene's device is not the current HIP device");
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RTMI_ERR_DEVICE, "hipGetSymbolAddress gave no address for the static work memory");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return RTMI_OK;
}

// What rtmi_render (host frame out) keeps between calls: the frame on the device, a pinned staging buffer the copy engine
// writes into, a stream and events.  Made at the first call for a frame size, freed with the scene.
constexpr int HOST_CHUNKS = 8; // the frame crosses PCIe in this many pieces; a piece is copied on to the caller's memory while the next one is in flight
struct HostFrameCache {
    std::mutex m;
    size_t values = 0;
    float *d_out = nullptr, *pinned = nullptr;
    uint64_t *d_stats = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, chunk[HOST_CHUNKS] = {};
};
// What rtmi_render_multi keeps with scenes[0]: per device a stream; per share its rows and counters on its device, a
// staging buffer and a stream on scenes[0]'s device, and two events.  Made at the first call for a set of scenes and a
// frame size, freed with scenes[0] (or when a call comes with another set / size).
struct MultiShare {
    int device = -1;
    size_t values = 0;
    float *rows = nullptr, *stage = nullptr;
    uint64_t *stats = nullptr;
    hipStream_t gather = nullptr;       // on scenes[0]'s device: waits for the share, copies it over, puts its rows in place
    hipEvent_t rendered = nullptr, placed = nullptr;
};
struct MultiCache {
    std::mutex m;
    int hsize = 0, vsize = 0, tile_rows = 0;
    std::vector<const void *> key;      // the scene handles of the call the cache was made for
    std::vector<MultiShare> shares;
    std::vector<std::pair<int, hipStream_t>> render_streams; // one per distinct device
};

struct rtmi_scene {
    int device = -1;
    bool pooled = false;
    mutable HostFrameCache host_cache;
    mutable MultiCache multi_cache;
    DeviceArrays<float> f32;
    DeviceArrays<double> f64;
    DNode *nodes = nullptr;
    DCsg *csgs = nullptr;
    DChain *csg_chain = nullptr;
    DPhoton *photons = nullptr; // World::photon_map in kd-tree order, behind one unused record (rtmi_scene_create)
    uint32_t n_photons = 0;
    uint32_t *leaf_orig = nullptr;
    DWTri *wtris = nullptr;
    DBvh2 *bvh2 = nullptr;  // the device's own triangle BVHs (NODE_TRIBVH) and their triangles in leaf order
    DWTri *btris = nullptr;
    uint32_t n_bvh2 = 0, n_btris = 0;
    bool bvh_has_units = false; // some DBvh2 tree has an ENT_LEAF entry
    uint32_t *quick = nullptr; // LDS records of the first top-level leaves (rtmi_device.hpp stage_scene)
    uint32_t n_quick = 0;
    uint32_t *pairs = nullptr, *others = nullptr; // the pair tables of the LDS block; the top-level leaves that are not in them
    uint32_t n_others = 0, pair_counts = 0, real_counts = 0, xpair_counts = 0;
    DScene<float> h32;
    DScene<double> h64;
    int num_cu = 256;
    double emit_power = 0.0; // sum of Light::power over World::lights (world.rs:212)
};

static bool is_csg_kind(int k) { return k >= RTMI_NODE_CSG_UNION && k <= RTMI_NODE_CSG_DIFFERENCE; }

struct Layout {
    const rtmi_scene_desc *d;
    std::vector<uint32_t> leaf_orig;   // device leaf -> desc leaf
    std::vector<uint32_t> leaf_csg;    // device leaf -> innermost CSG operator
    std::vector<uint8_t> leaf_shadow;  // device leaf -> casts shadows
    std::vector<DNode> nodes;
    std::vector<int> node_desc;        // device node -> desc node (-1 for the virtual root)
    std::vector<DCsg> csgs;
    std::vector<uint32_t> csg_l0, csg_l2; // first leaf / one past the last leaf of every CSG operator
    std::string err;
    int err_code = RTMI_OK;
    std::vector<uint8_t> seen_leaf, seen_node;

    bool bad(int code, const std::string &m) { if (err_code == RTMI_OK) { err_code = code; err = m; } return false; }

    bool check_node(int n) {
        if (n < 0 || n >= d->n_nodes) return bad(RTMI_ERR_INVALID, "node index out of range");
        if (seen_node[n]) return bad(RTMI_ERR_INVALID, "node referenced twice (the scene must be a tree)");
        seen_node[n] = 1;
        return true;
    }

    void push_leaf(int desc_leaf, uint32_t csg, bool shadow) {
        if (desc_leaf < 0 || desc_leaf >= d->n_leaves) { bad(RTMI_ERR_INVALID, "leaf index out of range"); return; }
        if (seen_leaf[desc_leaf]) { bad(RTMI_ERR_INVALID, "leaf referenced twice"); return; }
        seen_leaf[desc_leaf] = 1;
        leaf_orig.push_back((uint32_t)desc_leaf);
        leaf_csg.push_back(csg);
        leaf_shadow.push_back(shadow ? 1 : 0);
    }

    // conservative f32 box around the host's f64 box; returns false if the box cannot be used as a gate
    static bool make_box(const rtmi_node &n, DNode &o) {
        for (int a = 0; a < 3; ++a) {
            double lo = n.aabb_min[a], hi = n.aabb_max[a];
            if (!std::isfinite(lo) || !std::isfinite(hi) || lo > hi) return false;
            double pad = 1e-5 * std::fmax(std::fmax(std::fabs(lo), std::fabs(hi)), hi - lo) + 1e-6;
            float flo = (float)(lo - pad), fhi = (float)(hi + pad);
            if ((double)flo > lo - pad) flo = std::nextafterf(flo, -INFINITY);
            if ((double)fhi < hi + pad) fhi = std::nextafterf(fhi, INFINITY);
            o.bmin[a] = flo; o.bmax[a] = fhi;
        }
        return true;
    }

    // leaves below a CSG operator side; Groups are transparent to the filter (csg.rs:102, contains())
    void csg_side(int n, uint32_t parent, bool shadow, uint32_t &bit_counter) {
        if (!check_node(n)) return;
        const rtmi_node &nd = d->nodes[n];
        if (nd.kind == RTMI_NODE_LEAF) { push_leaf(nd.first, parent, shadow); return; }
        if (nd.kind == RTMI_NODE_GROUP || nd.kind == RTMI_NODE_BOUNDED) {
            if (nd.first < 0 || nd.count < 0 || nd.first + nd.count > d->n_children) { bad(RTMI_ERR_INVALID, "children range out of bounds"); return; }
            for (int i = 0; i < nd.count; ++i) csg_side(d->children[nd.first + i], parent, shadow, bit_counter);
            return;
        }
        if (!is_csg_kind(nd.kind)) { bad(RTMI_ERR_INVALID, "bad node kind"); return; }
        csg_op(n, parent, shadow, bit_counter);
    }
    void csg_op(int n, uint32_t parent, bool shadow, uint32_t &bit_counter) {
        const rtmi_node &nd = d->nodes[n];
        uint32_t me = (uint32_t)csgs.size();
        DCsg c;
        c.op = (uint32_t)nd.kind; c.left_end = 0; c.parent = parent; c.bit = bit_counter++;
        csgs.push_back(c);
        csg_l0.push_back((uint32_t)leaf_orig.size());
        csg_l2.push_back(0);
        csg_side(nd.first, me, shadow, bit_counter);
        csgs[me].left_end = (uint32_t)leaf_orig.size();
        csg_side(nd.count, me, shadow, bit_counter);
        csg_l2[me] = (uint32_t)leaf_orig.size();
    }

    // DGuard records (rtmi_device.hpp): one at the last leaf of the left side of every operator that is not a Union and has
    // leaves on both sides; an operator that is the WHOLE left side of such an operator passes the jump on to its end.
    // DChain records (rtmi_device.hpp): what CsgOp::eval (csg.rs:26-32) asks of the flags of every ancestor of a leaf
    std::vector<DChain> chains() const {
        std::vector<DChain> out(leaf_orig.size(), DChain{0ull, 0ull});
        for (size_t li = 0; li < leaf_orig.size(); ++li) {
            for (uint32_t c = leaf_csg[li]; c != NONE; c = csgs[c].parent) {
                const DCsg &cn = csgs[c];
                const bool lhit = li < cn.left_end;
                const unsigned long long bit = 1ull << (2 * cn.bit + (lhit ? 1 : 0)); // a left root looks at inr, a right one at inl
                bool want_set;
                if (cn.op == (uint32_t)RTMI_NODE_CSG_UNION) want_set = false;              // (lhit && !inr) || (!lhit && !inl)
                else if (cn.op == (uint32_t)RTMI_NODE_CSG_INTERSECTION) want_set = true;   // (lhit && inr) || (!lhit && inl)
                else want_set = !lhit;                                                     // (lhit && !inr) || (!lhit && inl)
                (want_set ? out[li].one : out[li].zero) |= bit;
            }
        }
        return out;
    }

    std::vector<DGuard> guards() const {
        std::vector<DGuard> g(leaf_orig.size(), DGuard{0u, 0u});
        // (two records for one leaf: the one with the weaker condition - the later `start` - stays)
        auto put = [&g](uint32_t leaf, DGuard r) {
            DGuard &o = g[leaf];
            if (o.skip == 0u || r.start > o.start || (r.start == o.start && r.skip > o.skip)) o = r;
        };
        for (size_t me = 0; me < csgs.size(); ++me) {
            const DCsg &c = csgs[me];
            if (c.op == (uint32_t)RTMI_NODE_CSG_UNION || c.left_end <= csg_l0[me] || csg_l2[me] <= c.left_end) continue;
            size_t top = me;
            while (csgs[top].parent != NONE) {
                size_t p = csgs[top].parent;
                if (csgs[p].op == (uint32_t)RTMI_NODE_CSG_UNION || csg_l0[p] != csg_l0[top] || csgs[p].left_end != csg_l2[top]) break;
                top = p;
            }
            put(c.left_end - 1, DGuard{csg_l0[me], csg_l2[top]});
            // an Intersection is as empty when its RIGHT side gave the ray no root (a left root passes only while inr)
            if (c.op == (uint32_t)RTMI_NODE_CSG_INTERSECTION) put(csg_l2[me] - 1, DGuard{c.left_end, csg_l2[top]});
        }
        // A record that can never fire is dropped (the leaf loop of csg_root works in runs that end at a record): one with
        // an earlier record e' inside its own range whose range starts no earlier and whose jump goes at least as far - if
        // e' fired, the ray is past this leaf; if it did not, the root that stopped it stops this one too.
        for (size_t e = 0; e < g.size(); ++e) {
            if (g[e].skip == 0u) continue;
            for (size_t f = g[e].start; f < e; ++f)
                if (g[f].skip != 0u && g[f].start >= g[e].start && g[f].skip >= g[e].skip) { g[e] = DGuard{0u, 0u}; break; }
        }
        return g;
    }

    // fill device node `slot` from desc node n (a GROUP/BOUNDED or a CSG root)
    void fill(uint32_t slot, int n, bool shadow) {
        if (!check_node(n)) return;
        node_desc[slot] = n;
        const rtmi_node &nd = d->nodes[n];
        DNode o;
        memset(&o, 0, sizeof o);
        o.flags = shadow ? NODE_CAST_SHADOW : 0;
        if (make_box(nd, o)) o.flags |= NODE_GATED;
        if (is_csg_kind(nd.kind)) {
            o.flags |= NODE_CSGROOT;
            o.leaf_first = (uint32_t)leaf_orig.size();
            uint32_t bits = 0;
            csg_op(n, NONE, shadow, bits);
            o.leaf_count = (uint32_t)leaf_orig.size() - o.leaf_first;
            o.csg_count = bits;
            if (bits > (uint32_t)CSG_MAX_NODES)
                bad(RTMI_ERR_UNSUPPORTED, "more than " + std::to_string(CSG_MAX_NODES) + " CSG operators under one CSG root");
            nodes[slot] = o;
            return;
        }
        if (nd.kind != RTMI_NODE_GROUP && nd.kind != RTMI_NODE_BOUNDED) { bad(RTMI_ERR_INVALID, "bad node kind"); return; }
        if (nd.first < 0 || nd.count < 0 || nd.first + nd.count > d->n_children) { bad(RTMI_ERR_INVALID, "children range out of bounds"); return; }
        std::vector<int> kids(d->children + nd.first, d->children + nd.first + nd.count);
        inner(slot, o, kids, std::vector<uint8_t>(kids.size(), shadow ? 1 : 0));
    }

    // direct primitive children first (contiguous leaves), then contiguous slots for the inner children
    void inner(uint32_t slot, DNode o, const std::vector<int> &kids, const std::vector<uint8_t> &shadow) {
        o.leaf_first = (uint32_t)leaf_orig.size();
        std::vector<std::pair<int, uint8_t>> sub;
        for (size_t i = 0; i < kids.size(); ++i) {
            int k = kids[i];
            if (k < 0 || k >= d->n_nodes) { bad(RTMI_ERR_INVALID, "node index out of range"); return; }
            if (d->nodes[k].kind == RTMI_NODE_LEAF) {
                if (!check_node(k)) return;
                push_leaf(d->nodes[k].first, NONE, shadow[i]);
            } else {
                sub.push_back({k, shadow[i]});
            }
        }
        o.leaf_count = (uint32_t)leaf_orig.size() - o.leaf_first;
        // Two boxed inner children (every split of a BoundingGroup, aabb.rs:45-70): put the one with the smaller centre
        // along the axis that separates them most first and note the axis, so that the traversal can visit the child
        // nearer to the ray's origin first and the running closest hit prunes the other more often.
        if (sub.size() == 2) {
            const rtmi_node &a = d->nodes[sub[0].first], &b = d->nodes[sub[1].first];
            DNode ta, tb;
            if (make_box(a, ta) && make_box(b, tb)) {
                int axis = 0;
                double sep = -1;
                for (int k = 0; k < 3; ++k) {
                    double ca = a.aabb_min[k] + a.aabb_max[k], cb = b.aabb_min[k] + b.aabb_max[k];
                    if (std::fabs(ca - cb) > sep) { sep = std::fabs(ca - cb); axis = k; }
                }
                if (a.aabb_min[axis] + a.aabb_max[axis] > b.aabb_min[axis] + b.aabb_max[axis]) std::swap(sub[0], sub[1]);
                o.flags |= (uint32_t)(axis + 1) << NODE_AXIS_SHIFT;
            }
        }
        o.child_first = (uint32_t)nodes.size();
        o.child_count = (uint32_t)sub.size();
        nodes.resize(nodes.size() + sub.size());
        node_desc.resize(nodes.size(), -1);
        nodes[slot] = o;
        for (size_t i = 0; i < sub.size(); ++i) fill(o.child_first + (uint32_t)i, sub[i].first, sub[i].second != 0);
    }

    // worst-case number of traversal-stack entries in use while the subtree of device node `slot` is processed:
    // its k children are pushed together and popped first child first, so k - 1 - i siblings wait below child i
    int stack_need(uint32_t slot) const {
        const DNode &o = nodes[slot];
        if (o.flags & NODE_CSGROOT) return 0;
        int need = (int)o.child_count;
        for (uint32_t i = 0; i < o.child_count; ++i)
            need = std::max(need, (int)(o.child_count - 1 - i) + stack_need(o.child_first + i));
        return need;
    }

    bool run() {
        seen_leaf.assign((size_t)std::max(0, d->n_leaves), 0);
        seen_node.assign((size_t)std::max(0, d->n_nodes), 0);
        nodes.resize(1); // slot 0 = virtual root = World::objects
        node_desc.assign(1, -1);
        DNode root;
        memset(&root, 0, sizeof root);
        root.flags = NODE_CAST_SHADOW;
        std::vector<int> kids(d->top_nodes, d->top_nodes + d->n_top);
        std::vector<uint8_t> sh((size_t)d->n_top);
        for (int i = 0; i < d->n_top; ++i) sh[(size_t)i] = d->top_cast_shadow ? (d->top_cast_shadow[i] ? 1 : 0) : 1;
        inner(0, root, kids, sh);
        if (err_code != RTMI_OK) return false;
        if (stack_need(0) > NODE_STACK) return bad(RTMI_ERR_UNSUPPORTED, "scene tree needs a deeper traversal stack than " + std::to_string(NODE_STACK));
        // CSG roots that are ONE operator over two leaves of at most two roots each - spheres, cubes: the operator goes into
        // the node's flags (csg_pair of the f32 kernels; the f64 kernels ignore it)
        for (DNode &nd : nodes) {
            if (!(nd.flags & NODE_CSGROOT) || nd.leaf_count != 2 || nd.csg_count != 1) continue;
            const int k0 = d->leaves[leaf_orig[nd.leaf_first]].kind, k1 = d->leaves[leaf_orig[nd.leaf_first + 1]].kind;
            const uint32_t c = leaf_csg[nd.leaf_first];
            const bool two = (k0 == RTMI_SPHERE || k0 == RTMI_CUBE) && (k1 == RTMI_SPHERE || k1 == RTMI_CUBE);
            if (two && c != NONE && c == leaf_csg[nd.leaf_first + 1] && csgs[c].left_end == nd.leaf_first + 1)
                nd.flags |= csgs[c].op << NODE_PAIR_SHIFT;
        }
        return true;
    }
};

template <typename R> static void cp3(R *dst, const double *src) { dst[0] = (R)src[0]; dst[1] = (R)src[1]; dst[2] = (R)src[2]; }
template <typename R> static void cp12(R *dst, const double *src) { for (int i = 0; i < 12; ++i) dst[i] = (R)src[i]; }

template <typename T> static int upload(T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void **)dst, bytes));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RTMI_OK;
}

// forward (object -> world) affine map = inverse of the 3x4 world -> object matrix `m`, by cofactors; returns its determinant
static double affine_forward(const double *m, double F[9], double T[3]) {
    const double a = m[0], b = m[1], c = m[2], e = m[4], f = m[5], g = m[6], p = m[8], q = m[9], r = m[10];
    const double det = a * (f * r - g * q) - b * (e * r - g * p) + c * (e * q - f * p);
    const double id = 1.0 / det;
    const double G[9] = {(f * r - g * q) * id, (c * q - b * r) * id, (b * g - c * f) * id,
                         (g * p - e * r) * id, (a * r - c * p) * id, (c * e - a * g) * id,
                         (e * q - f * p) * id, (b * p - a * q) * id, (a * f - b * e) * id};
    for (int k = 0; k < 9; ++k) F[k] = G[k];
    for (int k = 0; k < 3; ++k) T[k] = -(G[3 * k] * m[3] + G[3 * k + 1] * m[7] + G[3 * k + 2] * m[11]);
    return det;
}

// World-space bounds (f64) the device computes for itself: of a primitive (the shape's own extent, as Shape::aabb takes it
// from sphere.rs / cube.rs / cylinder.rs / cone.rs / triangle.rs, under the leaf's transform) and of a CSG tree - a Union is
// within the union of its sides' bounds, an Intersection within their intersection, a Difference within its left side's.
// A bound only ever rules rays out, so it may be tighter than the reference's (aabb.rs) without changing a hit.
struct Bound {
    double lo[3], hi[3];
    static Bound none() { Bound b; for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; } return b; }  // holds nothing
    static Bound all() { Bound b; for (int a = 0; a < 3; ++a) { b.lo[a] = -INFINITY; b.hi[a] = INFINITY; } return b; }   // rules nothing out
    bool empty() const { for (int a = 0; a < 3; ++a) if (!(lo[a] <= hi[a])) return true; return false; }
    bool finite() const { for (int a = 0; a < 3; ++a) if (!std::isfinite(lo[a]) || !std::isfinite(hi[a])) return false; return true; }
    void join(const Bound &o) { if (o.empty()) return; if (empty()) *this = none(); for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], o.lo[a]); hi[a] = std::fmax(hi[a], o.hi[a]); } }
    void meet(const Bound &o) { for (int a = 0; a < 3; ++a) { lo[a] = std::fmax(lo[a], o.lo[a]); hi[a] = std::fmin(hi[a], o.hi[a]); } }
};
static Bound primitive_bound(const rtmi_leaf &s) {
    double F[9], T[3];
    const double det = affine_forward(s.inv, F, T);
    if (!std::isfinite(det) || det == 0.0) return Bound::all();
    double l[3] = {-1, -1, -1}, h[3] = {1, 1, 1};
    std::vector<std::array<double, 3>> pts;
    switch (s.kind) {
    case RTMI_SPHERE: case RTMI_CUBE: break;
    case RTMI_CYLINDER: l[1] = s.minimum; h[1] = s.maximum; break;
    case RTMI_CONE: { const double r = std::fmax(std::fabs(s.minimum), std::fabs(s.maximum)); l[0] = l[2] = -r; h[0] = h[2] = r; l[1] = s.minimum; h[1] = s.maximum; break; }
    case RTMI_TRIANGLE: case RTMI_SMOOTH_TRIANGLE:
        pts.push_back({s.p1[0], s.p1[1], s.p1[2]}); pts.push_back({s.p2[0], s.p2[1], s.p2[2]}); pts.push_back({s.p3[0], s.p3[1], s.p3[2]});
        break;
    default: return Bound::all(); // a plane
    }
    if (pts.empty()) {
        for (int a = 0; a < 3; ++a) if (!std::isfinite(l[a]) || !std::isfinite(h[a]) || l[a] > h[a]) return Bound::all();
        for (int k = 0; k < 8; ++k) pts.push_back({(k & 1) ? h[0] : l[0], (k & 2) ? h[1] : l[1], (k & 4) ? h[2] : l[2]});
    }
    Bound b = Bound::none();
    for (const auto &v : pts)
        for (int a = 0; a < 3; ++a) {
            const double w = F[3 * a] * v[0] + F[3 * a + 1] * v[1] + F[3 * a + 2] * v[2] + T[a];
            if (!std::isfinite(w)) return Bound::all();
            b.lo[a] = std::fmin(b.lo[a], w); b.hi[a] = std::fmax(b.hi[a], w);
        }
    return b;
}
// (the descriptor has been validated by Layout::run by the time this is called: indices are in range, the nodes form a tree)
static Bound subtree_bound(const rtmi_scene_desc *d, int n) {
    const rtmi_node &nd = d->nodes[n];
    if (nd.kind == RTMI_NODE_LEAF) return primitive_bound(d->leaves[nd.first]);
    if (nd.kind == RTMI_NODE_GROUP || nd.kind == RTMI_NODE_BOUNDED) {
        Bound b = Bound::none();
        for (int i = 0; i < nd.count; ++i) {
            const Bound c = subtree_bound(d, d->children[nd.first + i]);
            if (!c.empty() && !c.finite()) return Bound::all();
            b.join(c);
        }
        return b;
    }
    const Bound l = subtree_bound(d, nd.first);
    if (nd.kind == RTMI_NODE_CSG_DIFFERENCE) return l;
    const Bound r = subtree_bound(d, nd.count);
    Bound b = l;
    if (nd.kind == RTMI_NODE_CSG_UNION) {
        if ((!l.empty() && !l.finite()) || (!r.empty() && !r.finite())) return Bound::all();
        b.join(r);
    } else b.meet(r);
    return b;
}

// World-space copies of the triangle leaves (DWTri, f32 path) in the order upload_typed numbers them (device leaf order),
// and the NODE_ALLTRI marking of the nodes whose direct leaves are all triangles.
static std::vector<DWTri> world_space_triangles(const rtmi_scene_desc *d, Layout &L, std::vector<uint32_t> &tri_of_leaf) {
    std::vector<DWTri> out;
    tri_of_leaf.assign(L.leaf_orig.size(), 0xFFFFFFFFu);
    for (size_t i = 0; i < L.leaf_orig.size(); ++i) {
        const rtmi_leaf &s = d->leaves[L.leaf_orig[i]];
        if (s.kind != RTMI_TRIANGLE && s.kind != RTMI_SMOOTH_TRIANGLE) continue;
        double F[9], T[3];
        const double det = affine_forward(s.inv, F, T);
        auto lin = [&](const double *v, float *o3) { for (int k = 0; k < 3; ++k) o3[k] = (float)(F[3 * k] * v[0] + F[3 * k + 1] * v[1] + F[3 * k + 2] * v[2]); };
        DWTri w;
        memset(&w, 0, sizeof w);
        double e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { e1[k] = s.p2[k] - s.p1[k]; e2[k] = s.p3[k] - s.p1[k]; }
        for (int k = 0; k < 3; ++k) w.p1[k] = (float)(F[3 * k] * s.p1[0] + F[3 * k + 1] * s.p1[1] + F[3 * k + 2] * s.p1[2] + T[k]);
        lin(e1, w.e1); lin(e2, w.e2);
        w.k = (float)det;
        w.smooth = s.kind == RTMI_SMOOTH_TRIANGLE ? 1u : 0u;
        tri_of_leaf[i] = (uint32_t)out.size();
        out.push_back(w);
    }
    for (size_t n = 1; n < L.nodes.size(); ++n) { // node 0 is the world itself: its leaves run through the top-level loop
        DNode &nd = L.nodes[n];
        if ((nd.flags & NODE_CSGROOT) || nd.leaf_count == 0) continue;
        bool all = true;
        for (uint32_t i = 0; i < nd.leaf_count; ++i) {
            const uint32_t t = tri_of_leaf[nd.leaf_first + i];
            if (t == 0xFFFFFFFFu || t != tri_of_leaf[nd.leaf_first] + i || !std::isfinite(out[t].k) || out[t].k == 0.0f) all = false;
        }
        if (all) { nd.flags |= NODE_ALLTRI; nd.csg_count = tri_of_leaf[nd.leaf_first]; }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// host: the device's own BVH (DBvh2, rtmi_device.hpp) in place of a subtree of Groups / BoundingGroups.  Its items are
// the subtree's triangles, its other primitives and its CSG roots (evaluated as a whole by csg_root)
// ---------------------------------------------------------------------------------------------
struct UnitBvhBuilder {
    struct Box {
        float lo[3], hi[3];
        void clear() { for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
        void add(const Box &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], b.lo[a]); hi[a] = std::fmax(hi[a], b.hi[a]); } }
        double area() const {
            const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
            return x < 0 ? 0.0 : 2.0 * (x * y + y * z + z * x);
        }
    };
    enum { TRI = 0, LEAF = 1, CSG = 2 };
    struct Item { Box b; float c[3]; uint32_t kind, wtri, ref; }; // ref: device leaf (TRI, LEAF) or device node (CSG)
    struct Bin2 { Box b; int left = -1, right = -1; uint32_t first = 0, count = 0; }; // binary tree: inner (left, right) or leaf (first, count)
    std::vector<Item> items;
    std::vector<Bin2> bin;
    std::vector<DBvh2> &out2;
    std::vector<DWTri> &out_tris;
    const std::vector<DWTri> &wtris;
    int leaf_max;
    int depth = 0;

    UnitBvhBuilder(std::vector<DBvh2> &o2, std::vector<DWTri> &ot, const std::vector<DWTri> &w, int lm) : out2(o2), out_tris(ot), wtris(w), leaf_max(lm) {}

    // a bound as the device uses it: float corners, padded like the scene nodes' boxes (Layout::make_box)
    static Box padded(const double lo3[3], const double hi3[3]) {
        Box b;
        for (int a = 0; a < 3; ++a) {
            const double lo = lo3[a], hi = hi3[a];
            const double pad = 1e-5 * std::fmax(std::fmax(std::fabs(lo), std::fabs(hi)), hi - lo) + 1e-6;
            float flo = (float)(lo - pad), fhi = (float)(hi + pad);
            if ((double)flo > lo - pad) flo = std::nextafterf(flo, -INFINITY);
            if ((double)fhi < hi + pad) fhi = std::nextafterf(fhi, INFINITY);
            b.lo[a] = flo; b.hi[a] = fhi;
        }
        return b;
    }
    // bounds of a world-space triangle as the device sees it
    static Box tri_box(const DWTri &w) {
        double lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            const double v0 = w.p1[a], v1 = (double)w.p1[a] + w.e1[a], v2 = (double)w.p1[a] + w.e2[a];
            lo[a] = std::fmin(v0, std::fmin(v1, v2)); hi[a] = std::fmax(v0, std::fmax(v1, v2));
        }
        return padded(lo, hi);
    }

    void add(const Box &b, uint32_t kind, uint32_t wtri, uint32_t ref) {
        Item it;
        it.b = b;
        for (int a = 0; a < 3; ++a) it.c[a] = 0.5f * (it.b.lo[a] + it.b.hi[a]);
        it.kind = kind; it.wtri = wtri; it.ref = ref;
        items.push_back(it);
    }
    void add_tri(uint32_t wtri, uint32_t leaf) { add(tri_box(wtris[wtri]), TRI, wtri, leaf); }
    void add_unit(const Bound &b, uint32_t kind, uint32_t ref) { add(padded(b.lo, b.hi), kind, 0, ref); }

    // a leaf of the tree: one item of any kind, or up to leaf_max triangles
    bool leaf_ok(uint32_t first, uint32_t count) const {
        if (count <= 1) return true;
        if (count > (uint32_t)leaf_max) return false;
        for (uint32_t i = first; i < first + count; ++i) if (items[i].kind != TRI) return false;
        return true;
    }

    // binned surface-area heuristic, 16 bins on the axis where the centres spread most; a median split when that fails
    int build2(uint32_t first, uint32_t count) {
        const int me = (int)bin.size();
        bin.emplace_back();
        Box b; b.clear();
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = first; i < first + count; ++i) {
            b.add(items[i].b);
            for (int a = 0; a < 3; ++a) { clo[a] = std::fmin(clo[a], items[i].c[a]); chi[a] = std::fmax(chi[a], items[i].c[a]); }
        }
        bin[me].b = b; bin[me].first = first; bin[me].count = count;
        if (count <= 1) return me;
        const bool may_stay = leaf_ok(first, count);
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
        const float ext = chi[axis] - clo[axis];
        uint32_t mid = first + count / 2;
        bool split = false;
        if (ext > 0 && std::isfinite(ext)) {
            constexpr int NB = 16;
            Box bb[NB]; uint32_t bn[NB] = {};
            for (auto &x : bb) x.clear();
            const float scale = (float)NB / ext;
            auto bin_of = [&](const Item &it) { int k = (int)((it.c[axis] - clo[axis]) * scale); return k < 0 ? 0 : (k >= NB ? NB - 1 : k); };
            for (uint32_t i = first; i < first + count; ++i) { const int k = bin_of(items[i]); bb[k].add(items[i].b); ++bn[k]; }
            double best = INFINITY; int best_k = -1;
            Box acc; double la[NB]; uint32_t ln[NB];
            acc.clear(); uint32_t n = 0;
            for (int k = 0; k < NB - 1; ++k) { acc.add(bb[k]); n += bn[k]; la[k] = acc.area(); ln[k] = n; }
            acc.clear(); n = 0;
            for (int k = NB - 1; k > 0; --k) {
                acc.add(bb[k]); n += bn[k];
                if (ln[k - 1] == 0 || n == 0) continue;
                const double cost = la[k - 1] * ln[k - 1] + acc.area() * n;
                if (cost < best) { best = cost; best_k = k; }
            }
            const double leaf_cost = b.area() * count;
            if (best_k > 0 && (!may_stay || best + 0.5 * b.area() < leaf_cost)) {
                auto it = std::partition(items.begin() + first, items.begin() + first + count, [&](const Item &x) { return bin_of(x) < best_k; });
                mid = (uint32_t)(it - items.begin());
                split = mid > first && mid < first + count;
            }
        }
        if (!split) {
            if (may_stay) return me;   // a leaf
            mid = first + count / 2;   // coincident centres: any split will do
            std::nth_element(items.begin() + first, items.begin() + mid, items.begin() + first + count,
                             [&](const Item &x, const Item &y) { return x.c[axis] < y.c[axis]; });
        }
        const int l = build2(first, mid - first);
        const int r = build2(mid, first + count - mid);
        bin[me].left = l; bin[me].right = r;
        return me;
    }

    uint32_t tri_base = 0;
    std::vector<uint32_t> tri_pos; // item -> its position among the triangle items
    uint32_t leaf_entry(const Bin2 &n) const {
        const Item &it = items[n.first];
        if (it.kind == LEAF) return ENT_LEAF | it.ref;
        if (it.kind == CSG) return it.ref; // a scene node
        return ENT_TRIS | ((n.count - 1u) << ENT_TRI_FIRST_BITS) | (tri_base + tri_pos[n.first]);
    }

    // one DBvh2 record per inner node: both children's boxes and entries
    uint32_t emit2(int node, int d) {
        depth = std::max(depth, d);
        const Bin2 &n = bin[node];
        if (n.left < 0) return leaf_entry(n);
        const uint32_t me = (uint32_t)out2.size();
        out2.emplace_back();
        DBvh2 rec;
        memset(&rec, 0, sizeof rec);
        const Box *cb[2] = {&bin[n.left].b, &bin[n.right].b};
        for (int c = 0; c < 2; ++c) {
            rec.lox[c] = cb[c]->lo[0]; rec.hix[c] = cb[c]->hi[0];
            rec.loy[c] = cb[c]->lo[1]; rec.hiy[c] = cb[c]->hi[1];
            rec.loz[c] = cb[c]->lo[2]; rec.hiz[c] = cb[c]->hi[2];
        }
        rec.child[0] = emit2(n.left, d + 1);
        rec.child[1] = emit2(n.right, d + 1);
        out2[me] = rec;
        return ENT_BVH | me;
    }

    // builds the tree over the items added so far and returns the stack entry of its root (an inner node: two items or more)
    uint32_t finish() {
        if (items.size() < 2 || (size_t)out_tris.size() + items.size() > (size_t)ENT_TRI_FIRST_MASK) return 0;
        tri_base = (uint32_t)out_tris.size();
        const int root = build2(0, (uint32_t)items.size());
        tri_pos.assign(items.size(), 0);
        uint32_t n = 0;
        for (size_t i = 0; i < items.size(); ++i) {
            tri_pos[i] = n;
            if (items[i].kind == TRI) { DWTri w = wtris[items[i].wtri]; w.pad = items[i].ref; out_tris.push_back(w); ++n; }
        }
        return emit2(root, 1);
    }
};

// Marks every maximal subtree of Groups / BoundingGroups whose contents all have finite bounds NODE_TRIBVH and builds its
// DBvh2 tree: (usable) triangles, other primitives with their own world-space bounds, CSG roots with the bounds their
// operators allow.  `tri_of_leaf`: device leaf -> index into wtris (or none).  Returns the deepest tree's depth in nodes.
static int build_unit_bvhs(const rtmi_scene_desc *d, Layout &L, const std::vector<DWTri> &wtris, const std::vector<uint32_t> &tri_of_leaf,
                           std::vector<DBvh2> &bvh2, std::vector<DWTri> &btris, std::vector<int> &bvh_depth_of_node) {
    const long leaf_max = std::max(1l, std::min((long)BVH_LEAF_MAX, env_long("RTMI_BVH_LEAF", RTMI_BVH_LEAF)));
    bvh_depth_of_node.assign(L.nodes.size(), 0);
    // what the tree would hold for a direct leaf / a CSG root; false: no finite bound
    auto leaf_bound = [&](uint32_t li, Bound &b) -> bool {
        const uint32_t t = tri_of_leaf[li];
        if (t != 0xFFFFFFFFu) return std::isfinite(wtris[t].k) && wtris[t].k != 0.0f;
        b = primitive_bound(d->leaves[L.leaf_orig[li]]);
        return b.finite() && !b.empty();
    };
    auto csg_bound = [&](uint32_t n, Bound &b) -> bool {
        b = subtree_bound(d, L.node_desc[n]);
        const DNode &nd = L.nodes[n];
        if (nd.flags & NODE_GATED) { // the reference's own box is a bound too
            Bound r;
            for (int a = 0; a < 3; ++a) { r.lo[a] = nd.bmin[a]; r.hi[a] = nd.bmax[a]; }
            b.meet(r);
        }
        return b.empty() || b.finite(); // (empty: no ray can hit it - the tree leaves it out)
    };
    // able[n]: everything below node n has a finite bound
    std::vector<int8_t> able(L.nodes.size(), -1);
    std::function<bool(uint32_t)> scan = [&](uint32_t n) -> bool {
        const DNode &nd = L.nodes[n];
        Bound b;
        bool ok;
        if (nd.flags & NODE_CSGROOT) ok = L.node_desc[n] >= 0 && csg_bound(n, b);
        else {
            ok = (nd.leaf_count + nd.child_count) > 0;
            for (uint32_t i = 0; i < nd.leaf_count; ++i) ok = leaf_bound(nd.leaf_first + i, b) && ok;
            for (uint32_t c = 0; c < nd.child_count; ++c) ok = scan(nd.child_first + c) && ok; // (scan every child: the marks are used below)
        }
        able[n] = ok ? 1 : 0;
        return ok;
    };
    for (uint32_t c = 0; c < L.nodes[0].child_count; ++c) scan(L.nodes[0].child_first + c);
    int deepest = 0;
    std::function<void(uint32_t, UnitBvhBuilder &)> gather = [&](uint32_t n, UnitBvhBuilder &b) {
        const DNode &nd = L.nodes[n];
        Bound bd;
        if (nd.flags & NODE_CSGROOT) {
            csg_bound(n, bd);
            if (!bd.empty()) b.add_unit(bd, UnitBvhBuilder::CSG, n);
            return;
        }
        for (uint32_t i = 0; i < nd.leaf_count; ++i) {
            const uint32_t li = nd.leaf_first + i;
            if (tri_of_leaf[li] != 0xFFFFFFFFu) b.add_tri(tri_of_leaf[li], li);
            else { leaf_bound(li, bd); b.add_unit(bd, UnitBvhBuilder::LEAF, li); }
        }
        for (uint32_t c = 0; c < nd.child_count; ++c) gather(nd.child_first + c, b);
    };
    std::function<void(uint32_t)> mark = [&](uint32_t n) {
        DNode &nd = L.nodes[n];
        if (nd.flags & NODE_CSGROOT) return;
        if (able[n] == 1) {
            UnitBvhBuilder b(bvh2, btris, wtris, (int)leaf_max);
            gather(n, b);
            const uint32_t entry = b.items.size() >= 4 ? b.finish() : 0u; // (a handful of items: the reference's node will do)
            if (entry && b.depth < PACKET_STACK) {
                nd.flags = (nd.flags | NODE_TRIBVH) & ~NODE_ALLTRI;
                nd.csg_count = entry;
                bvh_depth_of_node[n] = b.depth;
                deepest = std::max(deepest, b.depth);
                return;
            }
        }
        for (uint32_t c = 0; c < nd.child_count; ++c) mark(nd.child_first + c);
    };
    for (uint32_t c = 0; c < L.nodes[0].child_count; ++c) mark(L.nodes[0].child_first + c);
    return deepest;
}

// ---------------------------------------------------------------------------------------------
// host: PhotonMap::from_vec (photon_map.rs:230-313) - the photons into the reference's left-balanced kd-tree, heap order
// ---------------------------------------------------------------------------------------------
static long long kd_left_children(long long n) { // compute_number_of_left_children, photon_map.rs:423-444
    if (n <= 0) return 0;
    if (n <= 3) return 1;
    if (n == 4) return 2;
    if (n <= 7) return 3;
    int depth = 0;
    while ((2ll << depth) <= n) ++depth; // floor(log2 n)
    const long long full = 1ll << depth, last = n - (full - 1);
    return last <= full / 2 ? (full - 2) / 2 + last : (full - 2) / 2 + full / 2;
}
template <typename F> static void on_threads(int n_threads, F f) { // f(0) .. f(n_threads - 1), the first on the calling thread
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(f, t);
    f(0);
    for (auto &th : pool) th.join();
}
struct KdKey { float p[3]; uint32_t idx; }; // position beside the index: the selection reads 16-byte records in place, no chase into the photons
struct KdLess { // the order of partition_by_key's keys; equal keys by index (the reference leaves them to its partial sort)
    int axis;
    bool operator()(const KdKey &a, const KdKey &b) const { return a.p[axis] < b.p[axis] || (a.p[axis] == b.p[axis] && a.idx < b.idx); }
};
// std::nth_element for the few long ranges at the top of the tree: quickselect whose partition passes run on all threads
// (each thread counts, then moves its chunk's elements to their side in `tmp`), down to a range one thread finishes.
// The order is total (indices are distinct), so the k-th element and the SETS on its two sides are the ones
// std::nth_element gives; the order inside the sides differs and does not matter: every later choice is by key again.
static void parallel_select(KdKey *base, long long count, long long k, int axis, KdKey *tmp, int n_threads, long long serial_below) {
    const KdLess less{axis};
    long long lo = 0, hi = count;
    while (hi - lo > serial_below && n_threads > 1) {
        const long long len = hi - lo, chunk = (len + n_threads - 1) / n_threads;
        KdKey sample[9];
        for (int i = 0; i < 9; ++i) sample[i] = base[lo + (len - 1) * i / 8];
        std::sort(sample, sample + 9, less);
        const KdKey pivot = sample[4];
        std::vector<long long> n_less((size_t)n_threads, 0), at_less((size_t)n_threads, 0), at_rest((size_t)n_threads, 0);
        on_threads(n_threads, [&](int t) {
            const long long a = std::min(hi, lo + t * chunk), b = std::min(hi, a + chunk);
            long long c = 0;
            for (long long i = a; i < b; ++i) c += less(base[i], pivot) ? 1 : 0;
            n_less[(size_t)t] = c;
        });
        long long total_less = 0;
        for (int t = 0; t < n_threads; ++t) total_less += n_less[(size_t)t];
        if (total_less == 0 || total_less == len) break; // (the pivot is an end of the range: let one thread finish it)
        long long run_less = 0, run_rest = total_less;
        for (int t = 0; t < n_threads; ++t) {
            const long long a = std::min(hi, lo + t * chunk), b = std::min(hi, a + chunk);
            at_less[(size_t)t] = run_less; at_rest[(size_t)t] = run_rest;
            run_less += n_less[(size_t)t]; run_rest += (b - a) - n_less[(size_t)t];
        }
        on_threads(n_threads, [&](int t) {
            const long long a = std::min(hi, lo + t * chunk), b = std::min(hi, a + chunk);
            KdKey *pl = tmp + at_less[(size_t)t], *pr = tmp + at_rest[(size_t)t];
            for (long long i = a; i < b; ++i) {
                if (less(base[i], pivot)) *pl++ = base[i];
                else *pr++ = base[i];
            }
        });
        on_threads(n_threads, [&](int t) {
            const long long a = std::min(hi, lo + t * chunk), b = std::min(hi, a + chunk);
            if (b > a) memcpy(base + a, tmp + (a - lo), (size_t)(b - a) * sizeof(KdKey));
        });
        if (k < lo + total_less) hi = lo + total_less;
        else lo = lo + total_less;
    }
    std::nth_element(base + lo, base + k, base + hi, less);
}
// The recursion of PhotonMap::balance works on disjoint ranges of the photons and writes disjoint nodes: the top of the
// tree is split level by level (its long selections on all host threads, parallel_select) until there are enough subtrees,
// which the host threads (at most 16, RTMI_HOST_THREADS) then take one at a time.
static std::vector<DPhoton> balance_photons(const rtmi_photon *ph, long long n) {
    int n_threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    const char *env_threads = getenv("RTMI_HOST_THREADS");
    if (env_threads) n_threads = std::max(1, std::min(64, atoi(env_threads)));
    else if (n < 200000) n_threads = 1;
    long long pselect_min = 400000; // ranges this long are selected on all threads
    if (const char *e = getenv("RTMI_PSELECT_MIN")) pselect_min = std::max(64ll, atoll(e));
    std::vector<DPhoton> out((size_t)n);
    using KI = KdKey;
    std::vector<KI> ki((size_t)n), tmp;
    if (n_threads > 1 && n > pselect_min) tmp.resize((size_t)n);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}; // compute_extent, photon_map.rs:415-421
    {
        const int T = n >= 200000 ? n_threads : 1;
        std::vector<std::array<double, 6>> ext((size_t)T);
        on_threads(T, [&](int t) {
            const long long a = n * t / T, b = n * (t + 1) / T;
            std::array<double, 6> e = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (long long i = a; i < b; ++i) {
                KI &q = ki[(size_t)i];
                q.idx = (uint32_t)i;
                for (int c = 0; c < 3; ++c) { q.p[c] = ph[i].position[c]; e[(size_t)c] = std::fmin(e[(size_t)c], (double)q.p[c]); e[(size_t)c + 3] = std::fmax(e[(size_t)c + 3], (double)q.p[c]); }
            }
            ext[(size_t)t] = e;
        });
        for (const auto &e : ext)
            for (int c = 0; c < 3; ++c) { lo[c] = std::fmin(lo[c], e[(size_t)c]); hi[c] = std::fmax(hi[c], e[(size_t)c + 3]); }
    }
    struct Job { long long first, count, node; double lo[3], hi[3]; };
    auto put = [&](long long node, uint32_t i, uint32_t flag) {
        DPhoton &q = out[(size_t)node];
        for (int k = 0; k < 3; ++k) { q.pos[k] = ph[i].position[k]; q.dir[k] = ph[i].direction[k]; }
        q.scale = ph[i].power_scale;
        q.rgbf = (uint32_t)ph[i].power_rgb[0] | ((uint32_t)ph[i].power_rgb[1] << 8) | ((uint32_t)ph[i].power_rgb[2] << 16) | (flag << 24);
    };
    // one node: the split photon into its place, the two sides appended to `todo`
    auto step = [&](const Job &j, std::vector<Job> &todo, bool alone) { // alone: no other thread is at work, the long selections may use them all
        if (j.count == 0 || j.node >= n) return;
        if (j.count == 1) { put(j.node, ki[(size_t)j.first].idx, KD_LEAF); return; }
        const long long n_left = kd_left_children(j.count);
        const double sx = j.hi[0] - j.lo[0], sy = j.hi[1] - j.lo[1], sz = j.hi[2] - j.lo[2];
        const int axis = (sx >= sy && sx >= sz) ? 0 : ((sy >= sx && sy >= sz) ? 1 : 2); // photon_map.rs:270-294
        KI *base = ki.data() + j.first;
        // partition_by_key(n_left, ..): the n_left smallest keys in front, the next one at n_left
        if (alone && !tmp.empty() && j.count > pselect_min) parallel_select(base, j.count, n_left, axis, tmp.data(), n_threads, pselect_min / 4);
        else std::nth_element(base, base + n_left, base + j.count, KdLess{axis});
        const uint32_t m = base[n_left].idx;
        const double median = (double)base[n_left].p[axis];
        put(j.node, m, (uint32_t)(KD_SPLIT_X + axis));
        Job l = j, r = j;
        l.count = n_left; l.node = 2 * (j.node + 1) - 1; l.hi[axis] = median;            // Aabb::split_*, aabb.rs:105-127
        r.first = j.first + n_left + 1; r.count = j.count - n_left - 1; r.node = 2 * (j.node + 1); r.lo[axis] = median;
        todo.push_back(l); todo.push_back(r);
    };
    const bool timing = getenv("RTMI_TIMING") != nullptr; // (diagnostic: the phases of the build to stderr)
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "balance_photons %lld: %s at %.1f ms\n", n, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    std::vector<Job> level, next;
    Job root = {0, n, 0, {lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
    level.push_back(root);
    // breadth first until every thread has a few subtrees to take (one thread: the loop below does it all)
    while (n_threads > 1 && !level.empty() && level.size() < (size_t)n_threads * 4) {
        next.clear();
        for (const Job &j : level) step(j, next, true);
        level.swap(next);
    }
    lap("top levels split");
    auto subtree = [&](const Job &top) {
        std::vector<Job> todo;
        todo.push_back(top);
        while (!todo.empty()) {
            const Job j = todo.back();
            todo.pop_back();
            step(j, todo, false);
        }
    };
    if (n_threads <= 1) {
        for (const Job &j : level) subtree(j);
    } else {
        std::atomic<size_t> taken{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t)
            pool.emplace_back([&]() { for (size_t k; (k = taken.fetch_add(1)) < level.size();) subtree(level[k]); });
        for (auto &th : pool) th.join();
    }
    lap("subtrees done");
    return out;
}

// what the flat f32 kernels stage in LDS (rtmi_device.hpp stage_scene) and the list that goes with it
// Largest eigenvalue of F^T F for a 3x3 matrix F (row major): the square of F's largest singular value.  Closed form for a
// symmetric 3x3 matrix (trigonometric solution of the characteristic cubic), never below the true value by more than rounding
// (callers pad the radius they derive from it).
static double largest_eigenvalue_ftf(const double F[9]) {
    double A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = F[i] * F[j] + F[3 + i] * F[3 + j] + F[6 + i] * F[6 + j];
    const double p1 = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double q = (A[0][0] + A[1][1] + A[2][2]) / 3.0;
    if (p1 <= 1e-300) return std::max(A[0][0], std::max(A[1][1], A[2][2]));
    const double p2 = (A[0][0] - q) * (A[0][0] - q) + (A[1][1] - q) * (A[1][1] - q) + (A[2][2] - q) * (A[2][2] - q) + 2.0 * p1;
    const double p = std::sqrt(p2 / 6.0);
    double B[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) B[i][j] = (A[i][j] - (i == j ? q : 0.0)) / p;
    const double detB = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]) - B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0]) +
                        B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
    const double r = std::min(1.0, std::max(-1.0, 0.5 * detB));
    const double lam = q + 2.0 * p * std::cos(std::acos(r) / 3.0);
    double fro = 0; // (never above the Frobenius norm, never below a third of it: a guard against a bad cubic)
    for (int k = 0; k < 9; ++k) fro += F[k] * F[k];
    if (!(lam >= fro / 3.0 - 1e-12 * fro) || !(lam <= fro * (1.0 + 1e-12))) return fro;
    return lam;
}

struct QuickTables {
    std::vector<uint32_t> quick, pairs, others;
    uint32_t pair_counts = 0, real_counts = 0, xpair_counts = 0;
};

template <typename R>
static int upload_typed(const rtmi_scene_desc *d, const Layout &L, DeviceArrays<R> &A, QuickTables *qt = nullptr) {
    std::vector<uint32_t> *quick = qt ? &qt->quick : nullptr;
    std::vector<DLeaf<R>> leaves(L.leaf_orig.size());
    std::vector<DTri<R>> tris;
    for (size_t i = 0; i < leaves.size(); ++i) {
        const rtmi_leaf &s = d->leaves[L.leaf_orig[i]];
        DLeaf<R> &g = leaves[i];
        memset(&g, 0, sizeof g);
        cp12(g.inv, s.inv);
        g.minimum = (R)s.minimum; g.maximum = (R)s.maximum;
        g.kind = (uint32_t)s.kind | (s.closed ? LEAF_CLOSED : 0u) | (L.leaf_shadow[i] ? LEAF_CAST_SHADOW : 0u);
        g.material = (uint32_t)s.material;
        g.csg_parent = L.leaf_csg[i];
        g.tri = 0;
        if (sizeof(R) == 4 && s.kind == RTMI_SPHERE) {
            // similarity test on the world->object matrix M (rows of inv): M^T M = k I  =>  radius^2 = 1/k and the
            // world centre is -M^-1 t = -(1/k) M^T t
            const double *m = s.inv;
            double k = m[0] * m[0] + m[4] * m[4] + m[8] * m[8];
            bool sim = k > 0 && std::isfinite(k) && m[12] == 0 && m[13] == 0 && m[14] == 0 && m[15] == 1;
            for (int a = 0; a < 3 && sim; ++a)
                for (int b = 0; b < 3; ++b) {
                    double e = m[a] * m[b] + m[4 + a] * m[4 + b] + m[8 + a] * m[8 + b] - (a == b ? k : 0.0);
                    if (std::fabs(e) > 1e-12 * k) sim = false;
                }
            if (sim) {
                for (int a = 0; a < 3; ++a) g.fast[a] = (R)(-(m[a] * m[3] + m[4 + a] * m[7] + m[8 + a] * m[11]) / k);
                g.fast[3] = (R)(1.0 / k);
                g.kind |= LEAF_FAST;
            }
        }
        // bounding sphere of a finite leaf below a node (top-level leaves are few and usually large: not worth a test)
        if (sizeof(R) == 4 && !(g.kind & LEAF_FAST) && i >= L.nodes[0].leaf_count) {
            // object-space bound: centre c_o, radius r_o
            double co[3] = {0, 0, 0}, ro = -1;
            if (s.kind == RTMI_SPHERE) ro = 1.0;
            else if (s.kind == RTMI_CUBE) ro = std::sqrt(3.0);
            else if ((s.kind == RTMI_CYLINDER || s.kind == RTMI_CONE) && std::isfinite(s.minimum) && std::isfinite(s.maximum) && s.maximum >= s.minimum) {
                co[1] = 0.5 * (s.minimum + s.maximum);
                const double hh = 0.5 * (s.maximum - s.minimum);
                const double rad = s.kind == RTMI_CYLINDER ? 1.0 : std::max(std::fabs(s.minimum), std::fabs(s.maximum));
                ro = std::sqrt(hh * hh + rad * rad);
            } else if (s.kind == RTMI_TRIANGLE || s.kind == RTMI_SMOOTH_TRIANGLE) {
                for (int a = 0; a < 3; ++a) co[a] = (s.p1[a] + s.p2[a] + s.p3[a]) / 3.0;
                for (const double *p : {s.p1, s.p2, s.p3}) {
                    double d2 = 0;
                    for (int a = 0; a < 3; ++a) d2 += (p[a] - co[a]) * (p[a] - co[a]);
                    ro = std::max(ro, std::sqrt(d2));
                }
            }
            // to world space through the forward map F = inverse of the 3x4 part of s.inv; |F v| <= ||F||_F |v|
            const double *m = s.inv;
            const double a0 = m[0], b0 = m[1], c0 = m[2], e0 = m[4], f0 = m[5], g0 = m[6], p0 = m[8], q0 = m[9], r0 = m[10];
            const double det = a0 * (f0 * r0 - g0 * q0) - b0 * (e0 * r0 - g0 * p0) + c0 * (e0 * q0 - f0 * p0);
            if (ro >= 0 && std::isfinite(det) && det != 0) {
                const double id = 1.0 / det;
                const double F[9] = {(f0 * r0 - g0 * q0) * id, (c0 * q0 - b0 * r0) * id, (b0 * g0 - c0 * f0) * id,
                                     (g0 * p0 - e0 * r0) * id, (a0 * r0 - c0 * p0) * id, (c0 * e0 - a0 * g0) * id,
                                     (e0 * q0 - f0 * p0) * id, (b0 * p0 - a0 * q0) * id, (a0 * f0 - b0 * e0) * id};
                double cw[3];
                for (int k = 0; k < 3; ++k) {
                    const double x = co[0] - m[3], y = co[1] - m[7], z = co[2] - m[11]; // F (c_o - t)
                    cw[k] = F[3 * k] * x + F[3 * k + 1] * y + F[3 * k + 2] * z;
                }
                // |F v| <= s |v| with s the largest singular value of F (a pip of a die - a sphere scaled by 0.2, 0.1, 0.2 -
                // gets its longest half-axis, where the Frobenius norm gave 1.5 times that); a cube's corners are mapped one by one
                double stretch = ro * std::sqrt(largest_eigenvalue_ftf(F));
                if (s.kind == RTMI_CUBE) {
                    stretch = 0;
                    for (int c8 = 0; c8 < 8; ++c8) {
                        const double x = (c8 & 1) ? 1.0 : -1.0, y = (c8 & 2) ? 1.0 : -1.0, z = (c8 & 4) ? 1.0 : -1.0;
                        double d2 = 0;
                        for (int k = 0; k < 3; ++k) { const double w = F[3 * k] * x + F[3 * k + 1] * y + F[3 * k + 2] * z; d2 += w * w; }
                        stretch = std::max(stretch, std::sqrt(d2));
                    }
                }
                const double rw = stretch * (1.0 + 1e-4) + 1e-6;
                if (std::isfinite(rw) && std::isfinite(cw[0]) && std::isfinite(cw[1]) && std::isfinite(cw[2])) {
                    for (int k = 0; k < 3; ++k) g.fast[k] = (R)cw[k];
                    g.fast[3] = (R)(rw * rw);
                    g.kind |= LEAF_BOUND;
                }
            }
        }
        if (s.kind == RTMI_TRIANGLE || s.kind == RTMI_SMOOTH_TRIANGLE) {
            DTri<R> t;
            double e1[3], e2[3];
            for (int a = 0; a < 3; ++a) { e1[a] = s.p2[a] - s.p1[a]; e2[a] = s.p3[a] - s.p1[a]; } // triangle.rs:35-36
            cp3(t.p1, s.p1); cp3(t.e1, e1); cp3(t.e2, e2);
            if (s.kind == RTMI_SMOOTH_TRIANGLE) { cp3(t.n1, s.n1); cp3(t.n2, s.n2); cp3(t.n3, s.n3); }
            else { // normal = normalize(e2 x e1), triangle.rs:43
                double nx = e2[1] * e1[2] - e2[2] * e1[1], ny = e2[2] * e1[0] - e2[0] * e1[2], nz = e2[0] * e1[1] - e2[1] * e1[0];
                double inv = 1.0 / std::sqrt(nx * nx + ny * ny + nz * nz);
                double nn[3] = {nx * inv, ny * inv, nz * inv};
                cp3(t.n1, nn); cp3(t.n2, nn); cp3(t.n3, nn);
            }
            g.tri = (uint32_t)tris.size();
            tris.push_back(t);
        }
    }
    if (quick && sizeof(R) == 4) {
        // image of the LDS block's first two tables (rtmi_device.hpp stage_scene): records of the first top-level leaves -
        // kind | flags, material, pattern slot, -, q[4] - and the pattern slots of those of them whose material has a
        // pattern: world point -> pattern space in one step, pattern.inv x leaf.inv (pattern.rs:52, shapes/mod.rs:268)
        const size_t nq = std::min<size_t>(L.nodes[0].leaf_count, (size_t)LDS_QUICK);
        quick->assign((size_t)LDS_MATS_AT, 0u);
        uint32_t n_slots = 0;
        for (size_t i = 0; i < nq; ++i) {
            const DLeaf<R> &g = leaves[i];
            const rtmi_leaf &src = d->leaves[L.leaf_orig[i]];
            uint32_t *q = quick->data() + i * QUICK_DW;
            uint32_t kind = g.kind;
            float v[4] = {0, 0, 0, 0};
            if (kind & LEAF_FAST) { kind |= LEAF_QUICK; for (int a = 0; a < 4; ++a) v[a] = (float)g.fast[a]; }
            else if ((kind & LEAF_KIND_MASK) == RTMI_PLANE) { kind |= LEAF_QUICK; for (int a = 0; a < 4; ++a) v[a] = (float)g.inv[4 + a]; }
            q[0] = kind; q[1] = g.material; q[2] = NONE;
            memcpy(q + 4, v, sizeof v);
            const int pat = d->materials[src.material].pattern;
            const double *li = src.inv;
            const bool affine = li[12] == 0 && li[13] == 0 && li[14] == 0 && li[15] == 1;
            if (pat >= 0 && affine && n_slots < (uint32_t)LDS_PATS) {
                const rtmi_pattern &ps = d->patterns[pat];
                DPattern<float> slot;
                memset(&slot, 0, sizeof slot);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 4; ++c) {
                        double acc = c == 3 ? ps.inv[4 * r + 3] : 0.0;
                        for (int k = 0; k < 3; ++k) acc += ps.inv[4 * r + k] * li[4 * k + c];
                        slot.inv[4 * r + c] = (float)acc;
                    }
                for (int a = 0; a < 3; ++a) { slot.a[a] = (float)ps.a[a]; slot.b[a] = (float)ps.b[a]; }
                slot.kind = ps.kind;
                static const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                slot.pad = memcmp(ps.inv, ident, sizeof ident) != 0; // (as DPattern::pad: priced as a second transform)
                memcpy(quick->data() + LDS_PATS_AT + n_slots * PAT_DW, &slot, sizeof slot);
                q[2] = n_slots++;
            }
        }
        // the pair tables (quick_pairs): the LEAF_QUICK leaves by kind and by whether they cast shadows, two to a record,
        // component by component; a group of an odd number of leaves ends with a filler no ray hits.  Every other
        // top-level leaf goes on the `others` list.
        std::vector<uint32_t> group[4]; // spheres casting / not, planes casting / not
        std::vector<uint32_t> xgroup[2]; // spheres under a general affine transform (LEAF_XPAIR): casting / not
        for (size_t i = 0; i < L.nodes[0].leaf_count; ++i) {
            uint32_t kind = i < nq ? (*quick)[i * QUICK_DW] : 0u;
            if (i < nq && !(kind & LEAF_QUICK) && (kind & LEAF_KIND_MASK) == RTMI_SPHERE) {
                const double *li = d->leaves[L.leaf_orig[i]].inv;
                const bool affine = li[12] == 0 && li[13] == 0 && li[14] == 0 && li[15] == 1;
                std::vector<uint32_t> &xg = xgroup[(kind & LEAF_CAST_SHADOW) ? 0 : 1];
                // (room: LDS_XPAIRS records for both groups, each rounded up to whole pairs)
                const size_t pairs_if = (xgroup[0].size() + (&xg == &xgroup[0] ? 1 : 0) + 1) / 2 + (xgroup[1].size() + (&xg == &xgroup[1] ? 1 : 0) + 1) / 2;
                // (a world with Groups, CSG or meshes runs the KIND_TREE kernels, which hold no transformed-sphere table)
                if (affine && pairs_if <= (size_t)LDS_XPAIRS && L.nodes[0].child_count == 0) { xg.push_back((uint32_t)i); (*quick)[i * QUICK_DW] = kind | LEAF_XPAIR; continue; }
            }
            if (!(kind & LEAF_QUICK)) { qt->others.push_back((uint32_t)i); continue; }
            group[((kind & LEAF_FAST) ? 0 : 2) + ((kind & LEAF_CAST_SHADOW) ? 0 : 1)].push_back((uint32_t)i);
        }
        qt->pairs.assign((size_t)PAIRS_IMAGE_DW, 0u);
        {   // the transformed-sphere table: m0a m0b ... m11a m11b id0 id1 - -; a filler is the unit sphere 1e15 away
            uint32_t n_x = 0, xp[2] = {0, 0};
            for (int gi = 0; gi < 2; ++gi)
                for (size_t k = 0; k < xgroup[gi].size(); k += 2, ++n_x, ++xp[gi]) {
                    uint32_t *rec = qt->pairs.data() + (size_t)LDS_PAIRS * PAIR_DW + (size_t)n_x * XPAIR_DW;
                    for (int h = 0; h < 2; ++h) {
                        float m[12] = {1, 0, 0, 1e15f, 0, 1, 0, 0, 0, 0, 1, 0};
                        uint32_t id = 0xFFFFFFFEu;
                        if (k + h < xgroup[gi].size()) {
                            id = xgroup[gi][k + h];
                            for (int a = 0; a < 12; ++a) m[a] = (float)leaves[id].inv[a];
                        }
                        for (int a = 0; a < 12; ++a) memcpy(rec + 2 * a + h, &m[a], 4);
                        rec[24 + h] = id;
                    }
                }
            qt->xpair_counts = (xp[0] + xp[1]) | (xp[0] << 8) | ((uint32_t)(xgroup[0].size() + xgroup[1].size()) << 16) | ((uint32_t)xgroup[0].size() << 24);
        }
        uint32_t n_pairs = 0, pc[4] = {0, 0, 0, 0};
        for (int gi = 0; gi < 4; ++gi) {
            for (size_t k = 0; k < group[gi].size(); k += 2, ++n_pairs, ++pc[gi]) {
                uint32_t *rec = qt->pairs.data() + (size_t)n_pairs * PAIR_DW;
                for (int h = 0; h < 2; ++h) {
                    float v[4] = {0.0f, 0.0f, 0.0f, gi < 2 ? -1.0f : 0.0f}; // the filler: a sphere of radius^2 -1 / a plane with a zero row
                    uint32_t id = 0xFFFFFFFEu;
                    if (k + h < group[gi].size()) {
                        id = group[gi][k + h];
                        memcpy(v, quick->data() + (size_t)id * QUICK_DW + 4, sizeof v);
                    }
                    for (int a = 0; a < 4; ++a) memcpy(rec + 2 * a + h, &v[a], 4);
                    rec[8 + h] = id;
                }
            }
        }
        // (n_pairs <= LDS_PAIRS: at most LDS_QUICK leaves in four groups)
        qt->pair_counts = (pc[0] + pc[1]) | (pc[0] << 8) | ((pc[2] + pc[3]) << 16) | (pc[2] << 24);
        qt->real_counts = (uint32_t)(group[0].size() + group[1].size()) | ((uint32_t)group[0].size() << 8) | ((uint32_t)(group[2].size() + group[3].size()) << 16) |
                          ((uint32_t)group[2].size() << 24);
    }
    std::vector<DMaterial<R>> mats((size_t)d->n_materials);
    for (int i = 0; i < d->n_materials; ++i) {
        const rtmi_material &s = d->materials[i];
        DMaterial<R> &m = mats[(size_t)i];
        memset(&m, 0, sizeof m);
        cp3(m.color, s.color);
        m.emissive = (R)s.emissive; m.diffuse = (R)s.diffuse; m.specular = (R)s.specular; m.shininess = (R)s.shininess;
        m.reflective = (R)s.reflective; m.transparency = (R)s.transparency; m.ior = (R)s.refractive_index;
        m.pattern = s.pattern;
    }
    std::vector<DPattern<R>> pats((size_t)d->n_patterns);
    for (int i = 0; i < d->n_patterns; ++i) {
        const rtmi_pattern &s = d->patterns[i];
        DPattern<R> &p = pats[(size_t)i];
        memset(&p, 0, sizeof p);
        cp12(p.inv, s.inv); cp3(p.a, s.a); cp3(p.b, s.b);
        p.kind = s.kind;
        static const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        p.pad = memcmp(s.inv, ident, sizeof ident) != 0; // priced as a second transform in the FLOP accounting
    }
    std::vector<DLight<R>> lights((size_t)d->n_lights);
    for (int i = 0; i < d->n_lights; ++i) {
        const rtmi_light &s = d->lights[i];
        DLight<R> &l = lights[(size_t)i];
        memset(&l, 0, sizeof l);
        cp3(l.pos, s.position); cp3(l.intensity, s.intensity);
        l.radius = (R)s.radius;
        cp3(l.normal, s.normal); cp3(l.width, s.width); cp3(l.height, s.height);
        cp3(l.up, s.up); cp3(l.right, s.right); cp3(l.dir, s.direction);
        double u2 = s.up[0] * s.up[0] + s.up[1] * s.up[1] + s.up[2] * s.up[2];
        double r2 = s.right[0] * s.right[0] + s.right[1] * s.right[1] + s.right[2] * s.right[2];
        l.inv_up2 = (R)(u2 > 0 ? 1.0 / u2 : 0.0); l.inv_right2 = (R)(r2 > 0 ? 1.0 / r2 : 0.0);
        l.kind = s.kind;
    }
    // DCsgLeaf records (rtmi_device.hpp): what csg_root's first pass over a run of leaves needs of each, 32 bytes a leaf
    std::vector<DCsgLeaf> cleaf(leaves.size());
    {
        const std::vector<DGuard> gd = L.guards();
        for (size_t i = 0; i < leaves.size(); ++i) {
            DCsgLeaf &c = cleaf[i];
            memset(&c, 0, sizeof c);
            c.kind = leaves[i].kind;
            for (int a = 0; a < 3; ++a) c.c[a] = (leaves[i].kind & LEAF_BOUND) ? (float)leaves[i].fast[a] : 0.0f;
            c.r2 = (leaves[i].kind & LEAF_BOUND) ? (float)leaves[i].fast[3] : -1.0f;
            c.gstart = gd[i].start; c.gskip = gd[i].skip;
        }
        for (const DNode &nd : L.nodes) {
            if (!(nd.flags & NODE_CSGROOT)) continue;
            // run: the leaves from this one to the next guard record (or the end of the root), CSG_RUN at most
            uint32_t run = 0;
            for (uint32_t i = nd.leaf_first + nd.leaf_count; i-- > nd.leaf_first;) {
                run = (cleaf[i].gskip != 0u || run >= (uint32_t)CSG_RUN) ? 1u : run + 1u;
                cleaf[i].run = run;
            }
        }
    }
    int rc;
    if ((rc = upload(&A.csg_leaf, cleaf)) != RTMI_OK) return rc;
    if ((rc = upload(&A.leaves, leaves)) != RTMI_OK) return rc;
    if ((rc = upload(&A.tris, tris)) != RTMI_OK) return rc;
    if ((rc = upload(&A.materials, mats)) != RTMI_OK) return rc;
    if ((rc = upload(&A.patterns, pats)) != RTMI_OK) return rc;
    if ((rc = upload(&A.lights, lights)) != RTMI_OK) return rc;
    return RTMI_OK;
}

template <typename R> static void header(const rtmi_scene *s, const rtmi_scene_desc *d, const Layout &L, const DeviceArrays<R> &A, DScene<R> &h) {
    h.leaves = A.leaves; h.tris = A.tris; h.materials = A.materials; h.patterns = A.patterns; h.lights = A.lights;
    h.nodes = s->nodes; h.csgs = s->csgs; h.csg_leaf = A.csg_leaf; h.csg_chain = s->csg_chain;
    h.photons = s->n_photons ? s->photons + 1 : nullptr; h.n_photons = s->n_photons; h.n_nearest = s->n_photons ? (uint32_t)d->n_nearest : 0u;
    h.no_direct = d->no_direct_illumination ? 1u : 0u; h.pad3 = 0;
    h.photon_radius = d->max_search_radius > 0.0 ? (R)d->max_search_radius : (R)INFINITY; h.leaf_orig = s->leaf_orig;
    h.wtris = sizeof(R) == 4 ? s->wtris : nullptr;
    h.bvh2 = sizeof(R) == 4 ? s->bvh2 : nullptr;
    h.n_bvh2 = sizeof(R) == 4 ? s->n_bvh2 : 0u; h.n_btris = sizeof(R) == 4 ? s->n_btris : 0u;
    h.btris = sizeof(R) == 4 ? s->btris : nullptr;
    h.n_leaves = (uint32_t)L.leaf_orig.size(); h.n_lights = (uint32_t)d->n_lights; h.n_nodes = (uint32_t)L.nodes.size();
    h.max_depth = (uint32_t)d->max_depth;
    h.root_leaf_count = L.nodes[0].leaf_count; h.root_child_first = L.nodes[0].child_first; h.root_child_count = L.nodes[0].child_count;
    h.has_units = 0;
    for (const DNode &n : L.nodes) if (n.flags & NODE_CSGROOT) h.has_units = 1;
    if (sizeof(R) == 4 && s->bvh_has_units) h.has_units = 1;
    h.quick = nullptr; h.n_quick = 0; h.n_lds_materials = 0; h.n_lds_lights = 0;
    h.pairs = nullptr; h.others = nullptr; h.n_others = 0; h.pair_counts = 0; h.real_counts = 0; h.xpair_counts = 0; h.pad5 = 0;
    if (sizeof(R) == 4) {
        h.quick = s->quick; h.n_quick = s->n_quick;
        h.pairs = s->pairs; h.others = s->others; h.n_others = s->n_others; h.pair_counts = s->pair_counts; h.real_counts = s->real_counts; h.xpair_counts = s->xpair_counts;
        h.n_lds_materials = d->n_materials <= LDS_MATS ? (uint32_t)d->n_materials : 0u;
        h.n_lds_lights = d->n_lights <= LDS_LIGHTS ? (uint32_t)d->n_lights : 0u;
    }
}

static int validate(const rtmi_scene_desc *d) {
    if (!d) return fail(RTMI_ERR_INVALID, "null scene descriptor");
    if (d->abi_version != RTMI_ABI_VERSION) return fail(RTMI_ERR_INVALID, "abi_version mismatch");
    if (d->n_leaves < 0 || d->n_nodes < 0 || d->n_children < 0 || d->n_top < 0 || d->n_materials < 0 || d->n_patterns < 0 || d->n_lights < 0)
        return fail(RTMI_ERR_INVALID, "negative count");
    if ((d->n_leaves && !d->leaves) || (d->n_nodes && !d->nodes) || (d->n_children && !d->children) || (d->n_top && !d->top_nodes) ||
        (d->n_materials && !d->materials) || (d->n_patterns && !d->patterns) || (d->n_lights && !d->lights))
        return fail(RTMI_ERR_INVALID, "null array with non-zero count");
    if (d->max_depth < 0 || d->max_depth > MAX_DEPTH)
        return fail(RTMI_ERR_UNSUPPORTED, "max_depth must be in [0, " + std::to_string(MAX_DEPTH) + "]");
    if ((unsigned)d->n_leaves >= PEND_LEAF_MASK) return fail(RTMI_ERR_UNSUPPORTED, "too many leaves (the limit is 2^26 - 2)");
    for (int i = 0; i < d->n_leaves; ++i) {
        const rtmi_leaf &l = d->leaves[i];
        if (l.kind < RTMI_SPHERE || l.kind > RTMI_SMOOTH_TRIANGLE) return fail(RTMI_ERR_INVALID, "leaf " + std::to_string(i) + ": bad kind");
        if (l.material < 0 || l.material >= d->n_materials) return fail(RTMI_ERR_INVALID, "leaf " + std::to_string(i) + ": material out of range");
    }
    for (int i = 0; i < d->n_materials; ++i)
        if (d->materials[i].pattern >= d->n_patterns) return fail(RTMI_ERR_INVALID, "material " + std::to_string(i) + ": pattern out of range");
    for (int i = 0; i < d->n_patterns; ++i)
        if (d->patterns[i].kind < RTMI_PATTERN_STRIPE || d->patterns[i].kind > RTMI_PATTERN_XYZ)
            return fail(RTMI_ERR_INVALID, "pattern " + std::to_string(i) + ": bad kind (closures cannot cross the ABI)");
    if (d->n_photons < 0 || (d->n_photons > 0 && !d->photons)) return fail(RTMI_ERR_INVALID, "photon map: negative count or null array");
    if (d->n_photons >= (1ll << 30)) return fail(RTMI_ERR_UNSUPPORTED, "photon map: the limit is 2^30 - 1 photons");
    if (d->n_photons > 0 && (d->n_nearest < 1 || d->n_nearest > RTMI_MAX_NEAREST))
        return fail(RTMI_ERR_UNSUPPORTED, "photon map: n_nearest must be in [1, " + std::to_string(RTMI_MAX_NEAREST) + "]");
    if (d->n_photons > 0 && std::isnan(d->max_search_radius)) return fail(RTMI_ERR_INVALID, "photon map: max_search_radius is not a number");
    for (int i = 0; i < d->n_lights; ++i)
        if (d->lights[i].kind < RTMI_LIGHT_POINT || d->lights[i].kind > RTMI_LIGHT_BEAM) return fail(RTMI_ERR_INVALID, "light " + std::to_string(i) + ": bad kind");
    return RTMI_OK;
}

static int ensure_device(int *dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    HIP_TRY(hipGetDevice(dev));
    return RTMI_OK;
}

static void host_cache_free(HostFrameCache &c) { // (the scene's device current)
    (void)hipFree(c.d_out); (void)hipFree(c.d_stats);
    if (c.pinned) (void)hipHostFree(c.pinned);
    if (c.stream) (void)hipStreamDestroy(c.stream);
    if (c.e0) (void)hipEventDestroy(c.e0);
    if (c.e1) (void)hipEventDestroy(c.e1);
    for (hipEvent_t &e : c.chunk) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    c.d_out = nullptr; c.d_stats = nullptr; c.pinned = nullptr; c.stream = nullptr; c.e0 = nullptr; c.e1 = nullptr; c.values = 0;
}
static void multi_cache_free(MultiCache &c, int dev0) {
    for (MultiShare &sh : c.shares) {
        if (sh.device >= 0) { (void)hipSetDevice(sh.device); (void)hipFree(sh.rows); (void)hipFree(sh.stats); if (sh.rendered) (void)hipEventDestroy(sh.rendered); }
        (void)hipSetDevice(dev0);
        (void)hipFree(sh.stage);
        if (sh.gather) (void)hipStreamDestroy(sh.gather);
        if (sh.placed) (void)hipEventDestroy(sh.placed);
    }
    for (auto &ds : c.render_streams) { (void)hipSetDevice(ds.first); if (ds.second) (void)hipStreamDestroy(ds.second); }
    c.shares.clear(); c.render_streams.clear(); c.key.clear();
    c.hsize = c.vsize = c.tile_rows = 0;
}

extern "C" void rtmi_scene_destroy(rtmi_scene *s) {
    if (!s) return;
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && s->device >= 0 && prev != s->device) (void)hipSetDevice(s->device);
    (void)hipFree(s->f32.leaves); (void)hipFree(s->f32.tris); (void)hipFree(s->f32.materials); (void)hipFree(s->f32.patterns); (void)hipFree(s->f32.lights); (void)hipFree(s->f32.csg_leaf);
    (void)hipFree(s->f64.leaves); (void)hipFree(s->f64.tris); (void)hipFree(s->f64.materials); (void)hipFree(s->f64.patterns); (void)hipFree(s->f64.lights); (void)hipFree(s->f64.csg_leaf);
    (void)hipFree(s->nodes); (void)hipFree(s->csgs); (void)hipFree(s->csg_chain); (void)hipFree(s->photons); (void)hipFree(s->leaf_orig); (void)hipFree(s->wtris); (void)hipFree(s->quick); (void)hipFree(s->pairs); (void)hipFree(s->others); (void)hipFree(s->bvh2); (void)hipFree(s->btris);
    host_cache_free(s->host_cache);
    multi_cache_free(s->multi_cache, s->device);
    (void)hipSetDevice(s->device);
    if (s->pooled) pool_release(s->device);
    if (prev >= 0 && prev != s->device) (void)hipSetDevice(prev);
    delete s;
}

extern "C" int rtmi_scene_create(const rtmi_scene_desc *desc, rtmi_scene **out) {
    if (!out) return fail(RTMI_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int rc = validate(desc);
    if (rc != RTMI_OK) return rc;
    Layout L;
    L.d = desc;
    if (!L.run()) return fail(L.err_code, L.err);
    int dev = 0;
    if ((rc = ensure_device(&dev)) != RTMI_OK) return rc;
    rtmi_scene *s = new rtmi_scene();
    s->device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) s->num_cu = prop.multiProcessorCount;
    for (int32_t i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].kind != RTMI_LIGHT_AMBIENT) s->emit_power += (desc->lights[i].intensity[0] + desc->lights[i].intensity[1] + desc->lights[i].intensity[2]) / 3.0;
    std::vector<uint32_t> tri_of_leaf;
    std::vector<DWTri> wtris = world_space_triangles(desc, L, tri_of_leaf);
    std::vector<DBvh2> bvh2;
    std::vector<DWTri> btris;
    std::vector<int> bvh_depth;
    build_unit_bvhs(desc, L, wtris, tri_of_leaf, bvh2, btris, bvh_depth);
    for (const DBvh2 &b : bvh2) if ((b.child[0] & ENT_KIND) == ENT_LEAF || (b.child[1] & ENT_KIND) == ENT_LEAF) s->bvh_has_units = true;
    {   // the wave's shared stack (packet_scene / packet_tris, f32 path): siblings that wait while a child is walked, plus
        // one entry per level of a triangle BVH below
        std::function<int(uint32_t)> need = [&](uint32_t slot) -> int {
            const DNode &o = L.nodes[slot];
            if (o.flags & NODE_CSGROOT) return 0;
            if (o.flags & NODE_TRIBVH) return bvh_depth[slot];
            int n = (int)o.child_count;
            for (uint32_t i = 0; i < o.child_count; ++i) n = std::max(n, (int)(o.child_count - 1 - i) + need(o.child_first + i));
            return n;
        };
        if (need(0) > PACKET_STACK) { delete s; return fail(RTMI_ERR_UNSUPPORTED, "scene tree needs a deeper traversal stack than " + std::to_string(PACKET_STACK)); }
    }
    rc = upload(&s->nodes, L.nodes);
    if (rc == RTMI_OK) rc = upload(&s->wtris, wtris);
    s->n_bvh2 = (uint32_t)bvh2.size(); s->n_btris = (uint32_t)btris.size();
    if (rc == RTMI_OK) rc = upload(&s->bvh2, bvh2);
    if (rc == RTMI_OK) rc = upload(&s->btris, btris);
    if (rc == RTMI_OK) rc = upload(&s->csgs, L.csgs);
    if (rc == RTMI_OK) rc = upload(&s->csg_chain, L.chains());
    if (rc == RTMI_OK && desc->n_photons > 0) {
        // (one unused record in front: node 0 then lies 32 bytes into the allocation and the two children of a node - the
        // near leaf and the far leaf that most searches look at one after the other - share one 64-byte sector)
        std::vector<DPhoton> tree = balance_photons(desc->photons, desc->n_photons);
        tree.insert(tree.begin(), tree.front());
        s->n_photons = (uint32_t)desc->n_photons;
        rc = upload(&s->photons, tree);
    }
    if (rc == RTMI_OK) rc = upload(&s->leaf_orig, L.leaf_orig);
    QuickTables qt;
    if (rc == RTMI_OK) rc = upload_typed<float>(desc, L, s->f32, &qt);
    if (rc == RTMI_OK) { s->n_quick = (uint32_t)std::min<size_t>(L.nodes[0].leaf_count, (size_t)LDS_QUICK); rc = upload(&s->quick, qt.quick); }
    if (rc == RTMI_OK) rc = upload(&s->pairs, qt.pairs);
    if (rc == RTMI_OK) { s->n_others = (uint32_t)qt.others.size(); s->pair_counts = qt.pair_counts; s->real_counts = qt.real_counts; s->xpair_counts = qt.xpair_counts; rc = upload(&s->others, qt.others); }
    if (rc == RTMI_OK) rc = upload_typed<double>(desc, L, s->f64);
    if (rc == RTMI_OK && (rc = pool_acquire(dev)) == RTMI_OK) s->pooled = true;
    void *unused_scratch = nullptr;
    if (rc == RTMI_OK) rc = static_scratch(dev, &unused_scratch); // set-up of rtmi_render_device_noalloc, once per device
    if (rc != RTMI_OK) { rtmi_scene_destroy(s); return rc; }
    header(s, desc, L, s->f32, s->h32);
    header(s, desc, L, s->f64, s->h64);
    *out = s;
    return RTMI_OK;
}

// ---------------------------------------------------------------------------------------------
// host: launches
// ---------------------------------------------------------------------------------------------
static int norm_tile_rows(int tile_rows) {
    if (tile_rows <= 0) return 8;
    return ((tile_rows + 7) / 8) * 8;
}

extern "C" int32_t rtmi_rank_rows(int32_t vsize, int32_t rank, int32_t nranks, int32_t tile_rows) {
    if (nranks <= 1) return vsize > 0 ? vsize : 0;
    if (rank < 0 || rank >= nranks || vsize <= 0) return 0;
    int tr = norm_tile_rows(tile_rows);
    int n_tiles = (vsize + tr - 1) / tr, rows = 0;
    for (int t = rank; t < n_tiles; t += nranks) rows += std::min(tr, vsize - t * tr);
    return rows;
}

template <typename R> static DCamera<R> make_camera(const rtmi_camera *c, const rtmi_render_opts *o) {
    DCamera<R> k;
    cp12(k.inv, c->inv);
    k.pixel_size = (R)c->pixel_size; k.half_width = (R)c->half_width; k.half_height = (R)c->half_height;
    k.se2 = (R)(c->allowed_standard_error * c->allowed_standard_error);
    k.focal_distance = (R)c->focal_distance; k.aperture_size = (R)c->aperture_size;
    k.hsize = c->hsize; k.vsize = c->vsize; k.min_samples = c->min_samples; k.max_samples = o ? o->max_samples : 0;
    return k;
}

static int check_camera(const rtmi_camera *c) {
    if (!c) return fail(RTMI_ERR_INVALID, "null camera");
    if (c->hsize <= 0 || c->vsize <= 0 || (long long)c->hsize * c->vsize > 0x7FFFFFFFll) return fail(RTMI_ERR_INVALID, "bad camera size");
    if (!(c->pixel_size > 0)) return fail(RTMI_ERR_INVALID, "bad pixel_size");
    return RTMI_OK;
}

#ifndef RTMI_GRID_PER_CU
#define RTMI_GRID_PER_CU 32 // waves of the first pass per CU (they stride over the visits) when a visit is one sample
#endif
#ifndef RTMI_GRID_PER_CU_MANY
#define RTMI_GRID_PER_CU_MANY 192 // ... and when every pixel takes several samples per visit
#endif
#ifndef RTMI_SPREAD_BELOW
#define RTMI_SPREAD_BELOW 8 // a pixel's first samples go on lanes of their own when the frame has fewer 8x8 tiles than this per resident wave
#endif
#ifndef RTMI_KEEP_LANES
#define RTMI_KEEP_LANES 24 // first pass: lanes of a wave that must still be sampling for them to keep their pixels (fewer: all go to the queues)
#endif
#ifndef RTMI_INLANE_NEED
#define RTMI_INLANE_NEED 4096 // ... a pixel predicted to need more further samples than this goes to the queues at once
#endif
#ifndef RTMI_INLANE_MAX
#define RTMI_INLANE_MAX 4096 // ... and so does one that has taken this many past the first samples
#endif
// Tree scenes keep short limits: their samples are several times as long (a dice sample is 8 slow rays), a lane's samples
// run one after the other, and a lane that runs hundreds of them holds its wave - and the launch - for as long as they take
// (dice frame with the long limits: 287 ms against 248, one call alone 393 against 253; teapot: no difference).  The
// soft-shadow frame, flat: 67 ms with the short limits, 48 ms with the long ones; depth of field 244 / 231 ms.
#ifndef RTMI_INLANE_NEED_TREE
#define RTMI_INLANE_NEED_TREE 192
#endif
#ifndef RTMI_INLANE_MAX_TREE
#define RTMI_INLANE_MAX_TREE 384
#endif
#ifndef RTMI_INLANE_MIN_VISITS
#define RTMI_INLANE_MIN_VISITS 8 // ... all of which applies to frames with at least this many tiles per resident wave
#endif
#ifndef RTMI_RESAMPLE_WAVES
#define RTMI_RESAMPLE_WAVES 3 // waves per SIMD of the second pass's persistent grid
#endif

// RTMI_SPREAD_BELOW in the environment (read at every call) replaces the constant: 0 = never spread, a large number =
// always; for the tests that hold both ways of the first pass against each other
static long spread_below() { return env_long("RTMI_SPREAD_BELOW", (long)RTMI_SPREAD_BELOW); }

static int launch_render(const rtmi_scene *s, const rtmi_camera *c, const rtmi_render_opts *o, float *d_out, uint64_t *d_stats, hipStream_t st,
                         void *static_work = nullptr, size_t static_bytes = 0, hipEvent_t between = nullptr, void *state = nullptr) {
    int rank = 0, nranks = 1, tr = 8;
    if (o && o->nranks > 1) {
        nranks = o->nranks; rank = o->rank; tr = norm_tile_rows(o->tile_rows);
        if (rank < 0 || rank >= nranks) return fail(RTMI_ERR_INVALID, "rank out of range");
    }
    int rows = rtmi_rank_rows(c->vsize, rank, nranks, tr);
    if (rows == 0) return RTMI_OK;
    int cur_dev = -1;
    HIP_TRY(hipGetDevice(&cur_dev));
    if (cur_dev != s->device)
        return fail(RTMI_ERR_DEVICE, "the scene lives on device " + std::to_string(s->device) + " but the current HIP device is " + std::to_string(cur_dev));
    DRender rp;
    uint64_t seed = o ? o->seed : 0;
    rp.seed_lo = (uint32_t)seed; rp.seed_hi = (uint32_t)(seed >> 32);
    rp.rank = rank; rp.nranks = nranks; rp.tile_rows = tr;
    rp.tiles_x = (c->hsize + 7) / 8;
    rp.local_tile_rows8 = (rows + 7) / 8;
    rp.n_tiles = rp.tiles_x * rp.local_tile_rows8;
    // The first pass (render_kernel) takes every pixel's first `first_samples` samples either on one lane, one after the
    // other, or on `first_samples` lanes at once; same sums either way.  The second way makes a wave's visit one sample
    // long instead of first_samples: that is what a small frame needs (its last visits ARE its first pass) and what a
    // large one pays 15 % for (idle lanes when 64 is not a multiple, no gain at the end that 40 visits per wave do not
    // give anyway).  Waves stride over the visits; more waves than the 12 resident per CU lets the hardware dispatcher
    // balance uneven visits, but each extra wave costs its launch (measured on chapter-11a at 3840x2160: 96 per CU beats
    // 32 by 11 % with 5 samples per visit and 192 beats 96 by 3 % (one call alone; nothing with 3 frames in flight), 32 beats 96 by 50 % with one; a wave with a single visit is mostly launch
    // cost: at least 3 visits per wave while that still fills the GPU).
    const int first_samples = std::max(1, (o && o->max_samples > 0 && o->max_samples < c->min_samples) ? o->max_samples : c->min_samples);
    // (Both instantiations round every sample alike - Makefile, -ffp-contract=on - so the choice may look at this rank's
    // share of the frame without the image depending on the sharding: tests/test_gpu_first_pass_modes.py.)
    const bool spread = !state && s->n_photons == 0 && first_samples > 1 && first_samples <= 64 && (long long)rp.n_tiles < (long long)s->num_cu * 12 * spread_below(); // (a progressive pass resumes pixels: one lane each)
    rp.first = (uint32_t)first_samples;
    rp.group = spread ? (uint32_t)first_samples : 1u;
    rp.ppw = 64u / rp.group;
    const unsigned long long visits = ((unsigned long long)rp.n_tiles * 64u + rp.ppw - 1) / rp.ppw;
    if (visits > 0x7FFFFFFFull) return fail(RTMI_ERR_UNSUPPORTED, "frame too large for one call");
    rp.n_visits = (uint32_t)visits;
    const int per_cu = (int)env_long("RTMI_GRID_PER_CU", (!spread && first_samples >= 4) ? RTMI_GRID_PER_CU_MANY : RTMI_GRID_PER_CU);
    int grid = (int)std::min<unsigned long long>(visits, (unsigned long long)s->num_cu * per_cu);
    const long vpw = std::max(1l, env_long("RTMI_VISITS_PER_WAVE", 3));
    grid = std::max(std::min(grid, (int)((visits + (unsigned long long)vpw - 1) / (unsigned long long)vpw)), (int)std::min<unsigned long long>(visits, (unsigned long long)s->num_cu * 12));
    bool f64 = o && o->precision == RTMI_PRECISION_F64;
    if (o && o->precision != RTMI_PRECISION_F32 && o->precision != RTMI_PRECISION_F64) return fail(RTMI_ERR_INVALID, "bad precision");
    const bool tree = s->h32.root_child_count != 0 || s->n_photons != 0; // (a photon map: the tree kind's kernels, which take flat scenes too)
    DCamera<double> cam = make_camera<double>(c, o);
    unsigned long long *ds = (unsigned long long *)d_stats;
    // The open-ended part of Camera::multisample (camera.rs:290-299) runs over work queues - if it can be reached at all.
    // The reference seeds the sum of squares with the first sample un-squared (camera.rs:277), so even one sample has a
    // "variance": mean(c - c^2) <= 1/4 for any colour.  With min_samples <= 1 nothing can follow the first sample only
    // when SE^2 is above that bound (the deterministic set-up of camera.rs:378-389: min 1, SE 1.0); render_interactive's
    // min 1 / SE 0.1 (camera.rs:182) does take more.
    const bool one_sample = cam.min_samples <= 1 && cam.se2 >= 0.2501;
    const bool open_ended = std::isfinite(c->allowed_standard_error) && !one_sample && !(cam.max_samples > 0 && cam.max_samples <= first_samples);
    // How long a lane of the first pass keeps its pixel past the first samples (render_kernel): on a large frame most
    // pixels need a few dozen samples (a flat mid-grey pixel takes sqrt(c - c^2) / SE of them, see above) and are cheapest
    // where they are; on a small frame - or a small share of one - a wave has few visits and a visit of many samples
    // in a row would be the end of the launch, so everything past the first samples goes to the queues.
    // (the environment replaces the constants, read at every call: tuning sweeps, and the tests that hold the policies
    // against each other bit for bit)
    const long env_keep = env_long("RTMI_KEEP_LANES", RTMI_KEEP_LANES), env_need = env_long("RTMI_INLANE_NEED", tree ? RTMI_INLANE_NEED_TREE : RTMI_INLANE_NEED),
               env_max = env_long("RTMI_INLANE_MAX", tree ? RTMI_INLANE_MAX_TREE : RTMI_INLANE_MAX), env_visits = env_long("RTMI_INLANE_MIN_VISITS", RTMI_INLANE_MIN_VISITS);
    const bool inlane = open_ended && !spread && (long long)rp.n_tiles >= (long long)s->num_cu * 12 * env_visits;
    rp.keep_lanes = inlane ? (uint32_t)std::max(1l, std::min(64l, env_keep)) : 65u;
    rp.inlane_need = (uint32_t)std::max(0l, env_need);
    rp.inlane_max = inlane ? (uint32_t)std::max(0l, env_max) : 0u;
    rp.heap_slots = 0; // (launch_render_kernel sets it where the call has a heap workspace)
    void *work = nullptr;
    int slot = -1;
    std::unique_lock<std::mutex> lock(g_pool_mutex, std::defer_lock);
    size_t wbytes = work_bytes();
    if (static_work && open_ended) { // (a call that cannot reach the second pass - one sample, a cap within the first samples - touches no work memory)
        // no queue may overflow here: a frame (or rank share) is accepted only if each queue can hold every pixel of it
        const size_t cap = f64 ? make_work<double>(static_work, static_bytes).cap : make_work<float>(static_work, static_bytes).cap;
        const size_t pixels = (size_t)rows * (size_t)c->hsize;
        if (pixels > cap)
            return fail(RTMI_ERR_UNSUPPORTED, "rtmi_render_device_noalloc: " + std::to_string(pixels) + " pixels in this call, the static work memory (" +
                                                  std::to_string(RTMI_STATIC_SCRATCH_MB) + " MiB) holds " + std::to_string(cap) + " in " + (f64 ? "f64" : "f32") +
                                                  " mode; render in row-tile shares (opts.rank / nranks) or use rtmi_render_device");
    }
    // (a photon scene's f32 kernels keep their candidate heaps in the tail of the call's work memory - photon_estimate - so such
    // a call takes a slot of the pool even when it has no second pass)
    const bool heap_work = !f64 && s->n_photons != 0 && !static_work;
    if (open_ended || heap_work) {
        lock.lock(); // held until both kernels are enqueued: calls sharing a stream must not interleave their launches
        if (static_work) { work = static_work; wbytes = static_bytes; }
        else {
            int rc = pool_slot(s->device, st, &slot);
            if (rc != RTMI_OK) return rc;
            work = g_pool[s->device].buf[slot];
        }
        if (open_ended) HIP_TRY(hipMemsetAsync(work, 0, WORK_HEADER_BYTES, st)); // (the array keeps the last call's contents: the counters must start at zero)
    }
    if (s->n_photons != 0) grid = std::min(grid, (int)HEAP_WS_SLOTS); // (a wave's run in the heap workspace is its block's: photon_estimate)
    int grid2 = open_ended ? s->num_cu * 4 * RTMI_RESAMPLE_WAVES : 0;
    if (s->n_photons != 0) grid2 = std::min(grid2, (int)HEAP_WS_SLOTS);
    if (f64) launch_render_kernel<double>(tree, grid, grid2, st, s->h64, cam, rp, d_out, ds, work, wbytes, between, state);
    else launch_render_kernel<float>(tree, grid, grid2, st, s->h32, cam, rp, d_out, ds, work, wbytes, between, state);
    HIP_TRY(hipGetLastError());
    if (slot >= 0) { // the slot is this stream's until this point of the stream has been reached
        WorkPool &p = g_pool[s->device];
        HIP_TRY(hipEventRecord(p.done[slot], st));
        p.stamp[slot] = ++p.clock;
    }
    return RTMI_OK;
}

extern "C" int rtmi_render_device(const rtmi_scene *scene, const rtmi_camera *camera, const rtmi_render_opts *opts, float *d_rgb_out,
                                  uint64_t *d_stats, void *hip_stream) {
    if (!scene || !d_rgb_out) return fail(RTMI_ERR_INVALID, "null scene or output");
    int rc = check_camera(camera);
    if (rc != RTMI_OK) return rc;
    return launch_render(scene, camera, opts, d_rgb_out, d_stats, (hipStream_t)hip_stream);
}

extern "C" int rtmi_render_device_timed(const rtmi_scene *scene, const rtmi_camera *camera, const rtmi_render_opts *opts, float *d_rgb_out,
                                        uint64_t *d_stats, void *hip_stream, float *ms_first_pass, float *ms_second_pass) {
    if (!scene || !d_rgb_out || !ms_first_pass || !ms_second_pass) return fail(RTMI_ERR_INVALID, "null scene, output or duration pointer");
    int rc = check_camera(camera);
    if (rc != RTMI_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    *ms_first_pass = 0; *ms_second_pass = 0;
    {   // a rank that owns no rows launches nothing and records no event: zero durations, not an error
        int rank = 0, nranks = 1, tr = 8;
        if (opts && opts->nranks > 1) { nranks = opts->nranks; rank = opts->rank; tr = norm_tile_rows(opts->tile_rows); }
        if (rank >= 0 && rank < nranks && rtmi_rank_rows(camera->vsize, rank, nranks, tr) == 0) return RTMI_OK;
    }
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    auto cleanup = [&]() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
#define TRY_EV(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) { cleanup(); return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    for (hipEvent_t &e : ev) TRY_EV(hipEventCreate(&e));
    TRY_EV(hipEventRecord(ev[0], st));
    rc = launch_render(scene, camera, opts, d_rgb_out, d_stats, st, nullptr, 0, ev[1]);
    if (rc != RTMI_OK) { cleanup(); return rc; }
    TRY_EV(hipEventRecord(ev[2], st));
    TRY_EV(hipEventSynchronize(ev[2]));
    TRY_EV(hipEventElapsedTime(ms_first_pass, ev[0], ev[1]));
    TRY_EV(hipEventElapsedTime(ms_second_pass, ev[1], ev[2]));
#undef TRY_EV
    cleanup();
    return RTMI_OK;
}

// ---- one process, several GPUs: row tiles rendered where the scenes live, rows gathered to the first scene's device ----
// Every device renders its shares on a stream of its own; share r's rows leave for scenes[0]'s device as soon as THEY are
// done: a stream per share on that device waits for the share's event, copies the rows over (hipMemcpyPeerAsync: the N - 1
// copies are in flight together, each on its own xGMI link - what a gather is on this point-to-point fabric; nothing is
// reduced, so RCCL would add a communicator and nothing else) into the share's own staging buffer and puts them in
// place with a small kernel while the other copies are still coming.  Nothing is allocated after the first call.
extern "C" int rtmi_render_multi(const rtmi_scene *const *scenes, int32_t n_scenes, const rtmi_camera *camera, const rtmi_render_opts *opts,
                                 int32_t tile_rows, float *d_rgb_out, rtmi_stats *stats_out) {
    if (!scenes || n_scenes <= 0 || !d_rgb_out) return fail(RTMI_ERR_INVALID, "null scenes or output");
    for (int r = 0; r < n_scenes; ++r)
        if (!scenes[r]) return fail(RTMI_ERR_INVALID, "null scene " + std::to_string(r));
    int rc = check_camera(camera);
    if (rc != RTMI_OK) return rc;
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    const int N = n_scenes, tr = norm_tile_rows(tile_rows), dev0 = scenes[0]->device;
    const size_t row_values = (size_t)camera->hsize * 3;
    MultiCache &mc = scenes[0]->multi_cache;
    std::lock_guard<std::mutex> lock(mc.m); // (calls that gather to the same scenes[0] take turns)
#define TRY_MULTI(expr)                                                                                            \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) { (void)hipSetDevice(prev); return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    std::vector<const void *> key((size_t)N);
    for (int r = 0; r < N; ++r) key[(size_t)r] = scenes[r];
    if (mc.key != key || mc.hsize != camera->hsize || mc.vsize != camera->vsize || mc.tile_rows != tr) {
        multi_cache_free(mc, dev0);
        mc.shares.assign((size_t)N, MultiShare());
        for (int r = 0; r < N; ++r) {
            MultiShare &sh = mc.shares[(size_t)r];
            sh.device = scenes[r]->device;
            sh.values = (size_t)rtmi_rank_rows(camera->vsize, r, N, tr) * row_values;
            TRY_MULTI(hipSetDevice(sh.device));
            bool have = false;
            for (auto &ds : mc.render_streams) have = have || ds.first == sh.device;
            if (!have) { hipStream_t st = nullptr; TRY_MULTI(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); mc.render_streams.emplace_back(sh.device, st); }
            TRY_MULTI(hipMalloc((void **)&sh.stats, RTMI_N_STATS * sizeof(uint64_t)));
            TRY_MULTI(hipMalloc((void **)&sh.rows, std::max<size_t>(sh.values, 1) * sizeof(float)));
            TRY_MULTI(hipEventCreateWithFlags(&sh.rendered, hipEventDisableTiming));
            TRY_MULTI(hipSetDevice(dev0));
            if (sh.device != dev0) TRY_MULTI(hipMalloc((void **)&sh.stage, std::max<size_t>(sh.values, 1) * sizeof(float)));
            TRY_MULTI(hipStreamCreateWithFlags(&sh.gather, hipStreamNonBlocking));
            TRY_MULTI(hipEventCreateWithFlags(&sh.placed, hipEventDisableTiming));
        }
        mc.key = key; mc.hsize = camera->hsize; mc.vsize = camera->vsize; mc.tile_rows = tr;
    }
    rtmi_render_opts o;
    memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    o.nranks = N; o.tile_rows = tr;
    const auto t0 = std::chrono::steady_clock::now();
    // every share's kernels are enqueued before any is waited for: the devices render at the same time
    for (int r = 0; r < N; ++r) {
        MultiShare &sh = mc.shares[(size_t)r];
        hipStream_t rs = nullptr;
        for (auto &ds : mc.render_streams) if (ds.first == sh.device) rs = ds.second;
        TRY_MULTI(hipSetDevice(sh.device));
        TRY_MULTI(hipMemsetAsync(sh.stats, 0, RTMI_N_STATS * sizeof(uint64_t), rs));
        if (sh.values) {
            o.rank = r;
            rc = launch_render(scenes[r], camera, &o, sh.rows, sh.stats, rs);
            if (rc != RTMI_OK) { (void)hipSetDevice(prev); return rc; }
        }
        TRY_MULTI(hipEventRecord(sh.rendered, rs));
    }
    TRY_MULTI(hipSetDevice(dev0));
    for (int r = 0; r < N; ++r) {
        MultiShare &sh = mc.shares[(size_t)r];
        TRY_MULTI(hipStreamWaitEvent(sh.gather, sh.rendered, 0));
        if (sh.values) {
            const float *src = sh.rows;
            if (sh.device != dev0) { // over xGMI into this share's staging buffer (the copy engine; no kernel reads remote memory)
                TRY_MULTI(hipMemcpyPeerAsync(sh.stage, dev0, sh.rows, sh.device, sh.values * sizeof(float), sh.gather));
                src = sh.stage;
            }
            const int grid = (int)std::min<size_t>((sh.values + 255) / 256, 65535);
            hipLaunchKernelGGL(deinterleave_kernel, dim3(grid), dim3(256), 0, sh.gather, src, d_rgb_out, (long long)sh.values, (int)row_values, tr, r, N);
            TRY_MULTI(hipGetLastError());
        }
        TRY_MULTI(hipEventRecord(sh.placed, sh.gather));
    }
    rtmi_stats total;
    memset(&total, 0, sizeof total);
    for (int r = 0; r < N; ++r) {
        MultiShare &sh = mc.shares[(size_t)r];
        TRY_MULTI(hipEventSynchronize(sh.placed)); // (after sh.rendered by stream order: the share's counters are final)
        uint64_t c[RTMI_N_STATS];
        TRY_MULTI(hipSetDevice(sh.device));
        TRY_MULTI(hipMemcpy(c, sh.stats, sizeof c, hipMemcpyDeviceToHost));
        for (int k = 0; k < RTMI_N_STATS; ++k) total.counter[k] += c[k];
    }
    total.kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); // the wall span of the call, not kernel time
    if (stats_out) *stats_out = total;
#undef TRY_MULTI
    (void)hipSetDevice(prev);
    return RTMI_OK;
}

// ---- progressive rendering: the frame pass by pass, every pass going on from the sums the pass before left ----
struct rtmi_progressive {
    const rtmi_scene *scene = nullptr;
    rtmi_camera camera;
    rtmi_render_opts opts;
    float *d_out = nullptr;
    void *d_state = nullptr;
    uint64_t *d_stats = nullptr;
    int passes = 0;
};

extern "C" void rtmi_progressive_end(rtmi_progressive *p) {
    if (!p) return;
    (void)hipFree(p->d_out); (void)hipFree(p->d_state); (void)hipFree(p->d_stats);
    delete p;
}

extern "C" int rtmi_progressive_begin(const rtmi_scene *scene, const rtmi_camera *camera, const rtmi_render_opts *opts, rtmi_progressive **out) {
    if (!out) return fail(RTMI_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!scene) return fail(RTMI_ERR_INVALID, "null scene");
    int rc = check_camera(camera);
    if (rc != RTMI_OK) return rc;
    rtmi_progressive *p = new rtmi_progressive();
    p->scene = scene; p->camera = *camera;
    memset(&p->opts, 0, sizeof p->opts);
    if (opts) p->opts = *opts;
    p->opts.rank = 0; p->opts.nranks = 1;
    if (p->opts.precision != RTMI_PRECISION_F32 && p->opts.precision != RTMI_PRECISION_F64) { delete p; return fail(RTMI_ERR_INVALID, "bad precision"); }
    const size_t pixels = (size_t)camera->hsize * camera->vsize;
    const size_t state_bytes = pixels * (p->opts.precision == RTMI_PRECISION_F64 ? sizeof(DPixelState<double>) : sizeof(DPixelState<float>));
    hipError_t e = hipMalloc((void **)&p->d_out, pixels * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&p->d_state, state_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_stats, RTMI_N_STATS * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemset(p->d_state, 0, state_bytes); // cnt = 0: nothing taken yet
    if (e == hipSuccess) e = hipMemset(p->d_out, 0, pixels * 3 * sizeof(float));
    if (e != hipSuccess) { rtmi_progressive_end(p); return fail(RTMI_ERR_NOMEM, std::string("rtmi_progressive_begin: ") + hipGetErrorString(e)); }
    *out = p;
    return RTMI_OK;
}

extern "C" int rtmi_progressive_pass(rtmi_progressive *p, int32_t max_samples, float *rgb_out, rtmi_stats *stats_out) {
    if (!p || !rgb_out) return fail(RTMI_ERR_INVALID, "null progressive handle or output");
    if (max_samples < 0) return fail(RTMI_ERR_INVALID, "negative sample cap");
    rtmi_render_opts o = p->opts;
    o.max_samples = max_samples;
    HIP_TRY(hipMemset(p->d_stats, 0, RTMI_N_STATS * sizeof(uint64_t)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    hipError_t err = hipEventCreate(&e1);
    if (err != hipSuccess) { (void)hipEventDestroy(e0); return fail(RTMI_ERR_DEVICE, hipGetErrorString(err)); }
    (void)hipEventRecord(e0, 0);
    int rc = launch_render(p->scene, &p->camera, &o, p->d_out, p->d_stats, 0, nullptr, 0, nullptr, p->d_state);
    (void)hipEventRecord(e1, 0);
    if (rc == RTMI_OK) {
        err = hipEventSynchronize(e1);
        if (err == hipSuccess) err = hipMemcpy(rgb_out, p->d_out, (size_t)p->camera.hsize * p->camera.vsize * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (err == hipSuccess && stats_out) {
            memset(stats_out, 0, sizeof *stats_out);
            err = hipMemcpy(stats_out->counter, p->d_stats, RTMI_N_STATS * sizeof(uint64_t), hipMemcpyDeviceToHost);
            float ms = 0;
            if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
            stats_out->kernel_ms = ms;
        }
        if (err != hipSuccess) rc = fail(RTMI_ERR_DEVICE, std::string("rtmi_progressive_pass: ") + hipGetErrorString(err));
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc == RTMI_OK) ++p->passes;
    return rc;
}

extern "C" int rtmi_render_device_noalloc(const rtmi_scene *scene, const rtmi_camera *camera, const rtmi_render_opts *opts, float *d_rgb_out,
                                          uint64_t *d_stats, void *hip_stream) {
    if (!scene || !d_rgb_out) return fail(RTMI_ERR_INVALID, "null scene or output");
    int rc = check_camera(camera);
    if (rc != RTMI_OK) return rc;
    void *work = nullptr;
    if ((rc = static_scratch(scene->device, &work)) != RTMI_OK) return rc;
    return launch_render(scene, camera, opts, d_rgb_out, d_stats, (hipStream_t)hip_stream, work, STATIC_SCRATCH_BYTES);
}

// Camera::render with a host Canvas out (camera.rs:155-161).  The frame is rendered into a device buffer the scene keeps,
// then crosses PCIe in HOST_CHUNKS pieces into a pinned staging buffer (asynchronous copies, one event per piece); as each
// piece lands, host threads copy it on to the caller's (pageable) memory while the next piece is still in flight - the
// copy engine and the host's memory bandwidth work at the same time instead of one hipMemcpy staging the whole frame
// through the runtime's own bounce buffers.  Buffers, stream and events are made at the first call for a frame size.
// A call that finds the scene's buffers in use by another thread takes buffers of its own for the call.
static int render_to_host(const rtmi_scene *scene, const rtmi_camera *camera, const rtmi_render_opts *o, float *rgb_out, rtmi_stats *stats_out,
                          HostFrameCache &c) {
    const size_t n = (size_t)camera->hsize * camera->vsize * 3;
#define TRY_HOST(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    if (c.values != n) {
        host_cache_free(c);
        TRY_HOST(hipMalloc((void **)&c.d_out, n * sizeof(float)));
        TRY_HOST(hipMalloc((void **)&c.d_stats, RTMI_N_STATS * sizeof(uint64_t)));
        TRY_HOST(hipHostMalloc((void **)&c.pinned, n * sizeof(float), hipHostMallocDefault));
        TRY_HOST(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        TRY_HOST(hipEventCreate(&c.e0));
        TRY_HOST(hipEventCreate(&c.e1));
        for (hipEvent_t &e : c.chunk) TRY_HOST(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c.values = n;
    }
    TRY_HOST(hipMemsetAsync(c.d_stats, 0, RTMI_N_STATS * sizeof(uint64_t), c.stream));
    TRY_HOST(hipEventRecord(c.e0, c.stream));
    int rc = launch_render(scene, camera, o, c.d_out, c.d_stats, c.stream);
    if (rc != RTMI_OK) { (void)hipStreamSynchronize(c.stream); return rc; }
    TRY_HOST(hipEventRecord(c.e1, c.stream));
    // whole rows per piece
    const size_t row_values = (size_t)camera->hsize * 3, rows = (size_t)camera->vsize;
    size_t first[HOST_CHUNKS + 1];
    for (int k = 0; k <= HOST_CHUNKS; ++k) first[k] = rows * (size_t)k / HOST_CHUNKS * row_values;
    for (int k = 0; k < HOST_CHUNKS; ++k) {
        const size_t len = first[k + 1] - first[k];
        if (len) TRY_HOST(hipMemcpyAsync(c.pinned + first[k], c.d_out + first[k], len * sizeof(float), hipMemcpyDeviceToHost, c.stream));
        TRY_HOST(hipEventRecord(c.chunk[k], c.stream));
    }
    // pinned -> the caller's memory, piece by piece as the events complete, each piece split over a few threads
    const int n_threads = (int)std::max(1l, std::min(8l, env_long("RTMI_HOST_COPY_THREADS", 4)));
    hipError_t werr = hipSuccess;
    for (int k = 0; k < HOST_CHUNKS; ++k) {
        const hipError_t e = hipEventSynchronize(c.chunk[k]);
        if (e != hipSuccess) { werr = e; break; }
        const size_t len = first[k + 1] - first[k];
        if (len < (size_t)1 << 18 || n_threads == 1) { memcpy(rgb_out + first[k], c.pinned + first[k], len * sizeof(float)); continue; }
        std::vector<std::thread> pool;
        for (int t = 1; t < n_threads; ++t) {
            const size_t a = first[k] + len * (size_t)t / n_threads, b = first[k] + len * (size_t)(t + 1) / n_threads;
            pool.emplace_back([=, &c]() { memcpy(rgb_out + a, c.pinned + a, (b - a) * sizeof(float)); });
        }
        memcpy(rgb_out + first[k], c.pinned + first[k], (len / n_threads) * sizeof(float));
        for (auto &th : pool) th.join();
    }
    if (werr != hipSuccess) { (void)hipStreamSynchronize(c.stream); return fail(RTMI_ERR_DEVICE, std::string("rtmi_render: ") + hipGetErrorString(werr)); }
    if (stats_out) {
        memset(stats_out, 0, sizeof *stats_out);
        TRY_HOST(hipMemcpy(stats_out->counter, c.d_stats, RTMI_N_STATS * sizeof(uint64_t), hipMemcpyDeviceToHost));
        float ms = 0;
        TRY_HOST(hipEventElapsedTime(&ms, c.e0, c.e1));
        stats_out->kernel_ms = ms;
    }
#undef TRY_HOST
    return RTMI_OK;
}

extern "C" int rtmi_render(const rtmi_scene *scene, const rtmi_camera *camera, const rtmi_render_opts *opts, float *rgb_out, rtmi_stats *stats_out) {
    if (!scene || !rgb_out) return fail(RTMI_ERR_INVALID, "null scene or output");
    int rc = check_camera(camera);
    if (rc != RTMI_OK) return rc;
    rtmi_render_opts o;
    memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    o.rank = 0; o.nranks = 1;
    int cur_dev = -1;
    HIP_TRY(hipGetDevice(&cur_dev));
    if (cur_dev != scene->device)
        return fail(RTMI_ERR_DEVICE, "the scene lives on device " + std::to_string(scene->device) + " but the current HIP device is " + std::to_string(cur_dev));
    std::unique_lock<std::mutex> lock(scene->host_cache.m, std::try_to_lock);
    if (lock.owns_lock()) return render_to_host(scene, camera, &o, rgb_out, stats_out, scene->host_cache);
    HostFrameCache own; // another thread is rendering this scene to the host right now: buffers of this call's own
    rc = render_to_host(scene, camera, &o, rgb_out, stats_out, own);
    host_cache_free(own);
    return rc;
}

// (host entry points below: on a HIP error free what the call allocated - `cleanup` of the enclosing function - and fail)
#define TRY_OR_CLEAN(expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) { cleanup(); return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

extern "C" int rtmi_tonemap_u8_device(const float *d_rgb, uint8_t *d_rgb8, int64_t n_values, void *hip_stream) {
    if (!d_rgb || !d_rgb8 || n_values < 0) return fail(RTMI_ERR_INVALID, "bad tonemap arguments");
    if (n_values == 0) return RTMI_OK;
    long long threads = (n_values + 3) / 4;
    int grid = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(tonemap_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, d_rgb, d_rgb8, (long long)n_values);
    HIP_TRY(hipGetLastError());
    return RTMI_OK;
}

extern "C" int rtmi_trace_rays(const rtmi_scene *scene, const rtmi_render_opts *opts, int64_t n, const double *origins, const double *directions,
                               const int32_t *remaining, double *rgb_out, rtmi_hit_record *hits_out) {
    if (!scene || !origins || !directions || n < 0) return fail(RTMI_ERR_INVALID, "bad trace arguments");
    if (n == 0) return RTMI_OK;
    if (n > ((int64_t)1 << 30)) return fail(RTMI_ERR_UNSUPPORTED, "more than 2^30 rays in one rtmi_trace_rays call");
    int dev = -1, rc = ensure_device(&dev);
    if (rc != RTMI_OK) return rc;
    if (dev != scene->device)
        return fail(RTMI_ERR_DEVICE, "the scene lives on device " + std::to_string(scene->device) + " but the current HIP device is " + std::to_string(dev));
    if (remaining) // the pending-ray stack holds MAX_DEPTH entries and packs `remaining` into 5 bits
        for (int64_t i = 0; i < n; ++i)
            if (remaining[i] < 0 || remaining[i] > MAX_DEPTH)
                return fail(RTMI_ERR_UNSUPPORTED, "remaining[" + std::to_string(i) + "] = " + std::to_string(remaining[i]) + " is outside [0, " + std::to_string(MAX_DEPTH) + "]");
    double *d_o = nullptr, *d_d = nullptr, *d_rgb = nullptr;
    int *d_rem = nullptr;
    rtmi_hit_record *d_rec = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_rgb); (void)hipFree(d_rem); (void)hipFree(d_rec); };
    size_t b3 = (size_t)n * 3 * sizeof(double);
    TRY_OR_CLEAN(hipMalloc((void **)&d_o, b3));
    TRY_OR_CLEAN(hipMalloc((void **)&d_d, b3));
    TRY_OR_CLEAN(hipMemcpy(d_o, origins, b3, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(d_d, directions, b3, hipMemcpyHostToDevice));
    if (remaining) {
        TRY_OR_CLEAN(hipMalloc((void **)&d_rem, (size_t)n * sizeof(int)));
        TRY_OR_CLEAN(hipMemcpy(d_rem, remaining, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    }
    if (rgb_out) TRY_OR_CLEAN(hipMalloc((void **)&d_rgb, b3));
    if (hits_out) TRY_OR_CLEAN(hipMalloc((void **)&d_rec, (size_t)n * sizeof(rtmi_hit_record)));
    uint64_t seed = opts ? opts->seed : 0;
    bool f64 = opts && opts->precision == RTMI_PRECISION_F64;
    int grid = (int)((n + 63) / 64);
    if (f64) launch_trace_kernel<double>(grid, scene->h64, (uint32_t)seed, (uint32_t)(seed >> 32), (long long)n, d_o, d_d, d_rem, d_rgb, d_rec);
    else launch_trace_kernel<float>(grid, scene->h32, (uint32_t)seed, (uint32_t)(seed >> 32), (long long)n, d_o, d_d, d_rem, d_rgb, d_rec);
    TRY_OR_CLEAN(hipGetLastError());
    TRY_OR_CLEAN(hipDeviceSynchronize());
    if (rgb_out) TRY_OR_CLEAN(hipMemcpy(rgb_out, d_rgb, b3, hipMemcpyDeviceToHost));
    if (hits_out) TRY_OR_CLEAN(hipMemcpy(hits_out, d_rec, (size_t)n * sizeof(rtmi_hit_record), hipMemcpyDeviceToHost));
    cleanup();
    return RTMI_OK;
}

extern "C" int rtmi_trace_photons(const rtmi_scene *scene, const rtmi_photon_opts *opts, rtmi_photon *out, int64_t capacity, rtmi_photon_result *result) {
    if (!scene || !opts || !result || capacity < 0 || (capacity > 0 && !out)) return fail(RTMI_ERR_INVALID, "bad trace_photons arguments");
    memset(result, 0, sizeof *result);
    if (opts->n_emit < 1 || opts->n_emit > 0x7FFFFFFFll) return fail(RTMI_ERR_INVALID, "n_emit = " + std::to_string(opts->n_emit) + " is outside [1, 2^31 - 1]");
    if (opts->max_bounces < 0 || opts->max_bounces > 65535) return fail(RTMI_ERR_INVALID, "max_bounces = " + std::to_string(opts->max_bounces) + " is outside [0, 65535]");
    if (opts->precision != RTMI_PRECISION_F32 && opts->precision != RTMI_PRECISION_F64) return fail(RTMI_ERR_INVALID, "bad precision");
    if (!(scene->emit_power > 0.0)) return fail(RTMI_ERR_INVALID, "no light of the scene has power to emit photons (world.rs:212)");
    int dev = -1, rc = ensure_device(&dev);
    if (rc != RTMI_OK) return rc;
    if (dev != scene->device)
        return fail(RTMI_ERR_DEVICE, "the scene lives on device " + std::to_string(scene->device) + " but the current HIP device is " + std::to_string(dev));
    DPhotonTrace pt;
    pt.seed_lo = (uint32_t)opts->seed; pt.seed_hi = (uint32_t)(opts->seed >> 32) ^ 0x50484F54u;
    if (opts->first < 0 || opts->first >= opts->n_emit || opts->count < 0 || opts->count > opts->n_emit - opts->first)
        return fail(RTMI_ERR_INVALID, "first = " + std::to_string(opts->first) + ", count = " + std::to_string(opts->count) + " is not a share of n_emit = " + std::to_string(opts->n_emit));
    const long long share = opts->count > 0 ? opts->count : opts->n_emit - opts->first;
    pt.first = (uint32_t)opts->first; pt.end = (uint32_t)(opts->first + share); pt.pad = 0;
    pt.max_bounces = (uint32_t)(opts->max_bounces > 0 ? opts->max_bounces : RTMI_PHOTON_BOUNCES);
    pt.enable_direct = opts->enable_direct != 0; pt.enable_diffuse = opts->enable_diffuse != 0; pt.enable_caustic = opts->enable_caustic != 0;
    pt.direct_only = pt.enable_direct && !pt.enable_caustic && !pt.enable_diffuse;
    pt.inv_n = (float)(1.0 / (double)opts->n_emit);
    pt.cap = 0;
    const bool f64 = opts->precision == RTMI_PRECISION_F64;
    // lanes: enough waves to fill the GPU a few times over, never more lanes than photons
    const long long waves_wanted = (long long)scene->num_cu * 16;
    const int grid = (int)std::max<long long>(1, std::min<long long>(waves_wanted, (share + 63) / 64));
    const uint32_t n = (uint32_t)share, nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    rtmi_photon *d_out = nullptr;
    uint32_t *d_counts = nullptr, *d_offsets = nullptr;
    unsigned long long *d_sums = nullptr, *d_cnt = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_out); (void)hipFree(d_counts); (void)hipFree(d_offsets); (void)hipFree(d_sums); (void)hipFree(d_cnt);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        d_out = nullptr; d_counts = d_offsets = nullptr; d_sums = d_cnt = nullptr; e0 = e1 = nullptr;
    };
    TRY_OR_CLEAN(hipMalloc((void **)&d_cnt, PT_COUNTERS * sizeof(unsigned long long)));
    TRY_OR_CLEAN(hipMalloc((void **)&d_counts, (size_t)n * sizeof(uint32_t)));
    TRY_OR_CLEAN(hipMalloc((void **)&d_offsets, (size_t)n * sizeof(uint32_t)));
    TRY_OR_CLEAN(hipMalloc((void **)&d_sums, ((size_t)nb + 1) * sizeof(unsigned long long)));
    TRY_OR_CLEAN(hipEventCreate(&e0));
    TRY_OR_CLEAN(hipEventCreate(&e1));
    TRY_OR_CLEAN(hipMemset(d_cnt, 0, PT_COUNTERS * sizeof(unsigned long long)));
    // pass 1: every photon's path, counting the hits it keeps; then where each photon's hits go
    TRY_OR_CLEAN(hipEventRecord(e0, nullptr));
    if (f64) launch_photon_trace_kernel<double>(false, grid, nullptr, scene->h64, pt, nullptr, d_counts, nullptr, d_cnt);
    else launch_photon_trace_kernel<float>(false, grid, nullptr, scene->h32, pt, nullptr, d_counts, nullptr, d_cnt);
    TRY_OR_CLEAN(hipGetLastError());
    scan_counts(d_counts, n, d_sums, d_offsets, nullptr);
    TRY_OR_CLEAN(hipGetLastError());
    TRY_OR_CLEAN(hipEventRecord(e1, nullptr));
    TRY_OR_CLEAN(hipDeviceSynchronize());
    unsigned long long cnt[PT_COUNTERS] = {0, 0, 0, 0}, total = 0;
    TRY_OR_CLEAN(hipMemcpy(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost));
    TRY_OR_CLEAN(hipMemcpy(&total, d_sums + nb, sizeof total, hipMemcpyDeviceToHost));
    float ms = 0, ms2 = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (cnt[PT_BAD_LEAF]) {
        cleanup();
        return fail(RTMI_ERR_INVALID, "the material of leaf " + std::to_string(cnt[PT_BAD_LEAF] - 1) + " reflects and transmits more than it receives: diffuse * colour + max(specular, reflective) + transparency > 1 (materials.rs:443-445)");
    }
    if (total > 0xFFFFFFF0ull) { cleanup(); return fail(RTMI_ERR_UNSUPPORTED, "the call stores " + std::to_string(total) + " photons: more than one call holds (2^32 - 16)"); }
    result->n_stored = (int64_t)total; result->n_rays = (int64_t)cnt[PT_RAYS]; result->n_truncated = (int64_t)cnt[PT_TRUNCATED];
    if (capacity > 0 && total > 0) {
        // pass 2: the same paths again, each hit written to its place
        pt.cap = (uint32_t)total;
        TRY_OR_CLEAN(hipMalloc((void **)&d_out, (size_t)total * sizeof(rtmi_photon)));
        TRY_OR_CLEAN(hipEventRecord(e0, nullptr));
        if (f64) launch_photon_trace_kernel<double>(true, grid, nullptr, scene->h64, pt, d_out, nullptr, d_offsets, d_cnt);
        else launch_photon_trace_kernel<float>(true, grid, nullptr, scene->h32, pt, d_out, nullptr, d_offsets, d_cnt);
        TRY_OR_CLEAN(hipGetLastError());
        TRY_OR_CLEAN(hipEventRecord(e1, nullptr));
        TRY_OR_CLEAN(hipDeviceSynchronize());
        (void)hipEventElapsedTime(&ms2, e0, e1);
        TRY_OR_CLEAN(hipMemcpy(out, d_out, (size_t)std::min<unsigned long long>(total, (unsigned long long)capacity) * sizeof(rtmi_photon), hipMemcpyDeviceToHost));
    }
    result->kernel_ms = (double)ms + (double)ms2;
    cleanup();
    return RTMI_OK;
}

extern "C" int rtmi_camera_rays(const rtmi_camera *camera, const rtmi_render_opts *opts, int64_t n, const int32_t *px, const int32_t *py,
                                const int32_t *sample, double *origins_out, double *directions_out) {
    if (!px || !py || !origins_out || !directions_out || n < 0) return fail(RTMI_ERR_INVALID, "bad camera_rays arguments");
    int rc = check_camera(camera);
    if (rc != RTMI_OK) return rc;
    if (n == 0) return RTMI_OK;
    int dev;
    if ((rc = ensure_device(&dev)) != RTMI_OK) return rc;
    int *d_px = nullptr, *d_py = nullptr, *d_s = nullptr;
    double *d_o = nullptr, *d_d = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_px); (void)hipFree(d_py); (void)hipFree(d_s); (void)hipFree(d_o); (void)hipFree(d_d); };
    size_t bi = (size_t)n * sizeof(int), b3 = (size_t)n * 3 * sizeof(double);
    TRY_OR_CLEAN(hipMalloc((void **)&d_px, bi));
    TRY_OR_CLEAN(hipMalloc((void **)&d_py, bi));
    TRY_OR_CLEAN(hipMemcpy(d_px, px, bi, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(d_py, py, bi, hipMemcpyHostToDevice));
    if (sample) {
        TRY_OR_CLEAN(hipMalloc((void **)&d_s, bi));
        TRY_OR_CLEAN(hipMemcpy(d_s, sample, bi, hipMemcpyHostToDevice));
    }
    TRY_OR_CLEAN(hipMalloc((void **)&d_o, b3));
    TRY_OR_CLEAN(hipMalloc((void **)&d_d, b3));
    uint64_t seed = opts ? opts->seed : 0;
    bool f64 = opts && opts->precision == RTMI_PRECISION_F64;
    int grid = (int)((n + 255) / 256);
    if (f64) launch_camera_rays_kernel<double>(grid, make_camera<double>(camera, opts), (uint32_t)seed, (uint32_t)(seed >> 32), (long long)n, d_px, d_py, d_s, d_o, d_d);
    else launch_camera_rays_kernel<float>(grid, make_camera<double>(camera, opts), (uint32_t)seed, (uint32_t)(seed >> 32), (long long)n, d_px, d_py, d_s, d_o, d_d);
    TRY_OR_CLEAN(hipGetLastError());
    TRY_OR_CLEAN(hipDeviceSynchronize());
    TRY_OR_CLEAN(hipMemcpy(origins_out, d_o, b3, hipMemcpyDeviceToHost));
    TRY_OR_CLEAN(hipMemcpy(directions_out, d_d, b3, hipMemcpyDeviceToHost));
    cleanup();
    return RTMI_OK;
}

extern "C" int rtmi_describe(char *buf, int32_t buf_len) {
    if (!buf || buf_len <= 0) return fail(RTMI_ERR_INVALID, "bad buffer");
    int n = 0;
    std::string dev = "no device";
    if (hipGetDeviceCount(&n) == hipSuccess && n > 0) {
        int d = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess)
            dev = std::string(p.name) + " " + p.gcnArchName + " CUs=" + std::to_string(p.multiProcessorCount) + " clock_khz=" + std::to_string(p.clockRate);
    }
    snprintf(buf, (size_t)buf_len, "rtmi abi=%d target=gfx950 devices=%d [%s]", RTMI_ABI_VERSION, n, dev.c_str());
    return RTMI_OK;
}
