// rtmi_kernels.hip — the device kernels of the path, compiled once per arithmetic type (-DRTMI_REAL=float|double).
// The f64 build is compiled with -ffp-contract=off so that the validation mode rounds operation by operation like
// the reference's (and the oracle's) IEEE f64 arithmetic; the f32 product build lets the compiler fuse multiply-adds.
#include "rtmi_launch.hpp"

using namespace rtmi_dev;

#ifndef RTMI_REAL
#error "compile with -DRTMI_REAL=float or -DRTMI_REAL=double"
#endif
#ifndef RTMI_WAVES
#define RTMI_WAVES 3 // minimum waves per SIMD the flat-scene render kernels are register-allocated for (168 VGPRs)
#endif
#ifndef RTMI_WAVES_PHOTON
#define RTMI_WAVES_PHOTON 3 // ... the kernels that gather from a photon map
#endif
#ifndef RTMI_PHOTON_STRIP
#define RTMI_PHOTON_STRIP 32 // photon scenes: the first pass visits the frame's tiles in vertical strips this many tiles wide
#endif
#ifndef RTMI_WAVES_TREE
#define RTMI_WAVES_TREE 3 // ... and the tree-scene kernels
#endif
// rtmi_trace_rays on a flat scene through the flat-scene code of the render kernels (trace_kernel<R, PHOTON, false>) instead
// of the tree kind's.  On in the host emulation (tests/emul: every known-answer ray then goes through quick_pairs under
// ASan / UBSan); OFF in the GPU build: the first GPU run of round 3 faulted in the one test that launched it with full
// waves, the cause could not be found on the CPU (the emulation of the same 10,000 rays is clean) and GPU access ended
// with that run - so the GPU build launches the instantiation round 2 shipped (same instruction stream, checked).
#ifndef RTMI_TRACE_FLAT
#ifdef RTMI_HOST_EMUL
#define RTMI_TRACE_FLAT 1
#else
#define RTMI_TRACE_FLAT 0
#endif
#endif

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
#ifndef RTMI_HOST_EMUL // the wave-cooperative kernels need a real wave (tests/emul covers the lane-independent ones)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// (second: the kernel belongs to the second pass, its FLOPs are reported in RTMI_STAT_FLOPS_SECOND as well)
__device__ __forceinline__ void flush_stats(const Stats &st, unsigned long long *__restrict__ stats, bool second = false) {
    if (!stats) return;
    if (RTMI_SKIP & 4) return; // (diagnostic build: with nobody reading them, the compiler drops the counters' arithmetic)
#ifdef RTMI_CYCLES
    {
        const int slot[4] = {RTMI_STAT_AABB_TESTS, RTMI_STAT_LEAF_TESTS, RTMI_STAT_NODE_VISITS, RTMI_STAT_FLOPS};
        for (int i = 0; i < 4; ++i) {
            unsigned long long m = st.cyc[i] + (i == 3 ? __builtin_readcyclecounter() : 0ull);
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
            if ((threadIdx.x & 63) == 0) atomicAdd(&stats[slot[i]], m);
        }
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < RTMI_N_STATS; ++i) {
        if (i == RTMI_STAT_FLOPS || i == RTMI_STAT_FLOPS_SECOND) continue;
        uint32_t s = wave_sum(st.get(i));
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&stats[i], (unsigned long long)s);
    }
    unsigned long long f = st.flops;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) f += __shfl_xor(f, off, 64);
    if ((threadIdx.x & 63) == 0 && f) {
        atomicAdd(&stats[RTMI_STAT_FLOPS], f);
        if (second) atomicAdd(&stats[RTMI_STAT_FLOPS_SECOND], f);
    }
}

// Camera::multisample's loop condition (camera.rs:281,290-291 with color_variance_of_mean, 305-310):
// true while the pixel that has taken `cnt` samples (n == cnt as a float) must take another one.
template <typename R> __device__ __forceinline__ bool need_more(const DCamera<double> &cam, V3<R> sum, V3<R> sos, R n, uint32_t cnt) {
#pragma clang fp contract(off) // the same rounding wherever this is inlined: the stop decision must not depend on the kernel that takes it
    if (cnt == 0) return true;
    if (cam.max_samples > 0 && cnt >= (uint32_t)cam.max_samples) return false;
    if (n < (R)cam.min_samples) return true;
    const R inv_n = rcp_(n);
    const R mx = sum.x * inv_n, my = sum.y * inv_n, mz = sum.z * inv_n;
    const R qx = sos.x * inv_n, qy = sos.y * inv_n, qz = sos.z * inv_n;
    const R sx = mx * mx, sy = my * my, sz = mz * mz;
    const R vx = qx - sx, vy = qy - sy, vz = qz - sz;
    const R variance = div_((vx + vy) + vz, (R)3);
    return div_(variance, n) > (R)cam.se2;
}

// c * c per channel, never fused into the sums it is added to (see need_more)
template <typename R> __device__ __forceinline__ V3<R> squares(V3<R> c) {
#pragma clang fp contract(off)
    V3<R> q;
    q.x = c.x * c.x; q.y = c.y * c.y; q.z = c.z * c.z;
    return q;
}

template <typename R> __device__ __forceinline__ V3<R> shfl3(V3<R> v, int src) {
    return {__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64)};
}

__device__ __forceinline__ void stats_zero(Stats &s) {
    s.zero();
#ifdef RTMI_CYCLES
    s.cyc[3] = 0ull - __builtin_readcyclecounter(); // + the clock at flush_stats = the cycles between
#endif
}
// The counters of a sampling loop: in the wave's LDS block for the f32 flat-scene kernels, else in registers.  (Measured, A/B
// on whole frames: flat scenes +7 ... +8.5 % with the block - chapter-11a 79.8 -> 86.4 Grays/s, the spills gone; tree scenes
// -13 ... -20 % - a tree walk counts a node visit, box tests and leaf tests per step, and that many LDS atomics cost
// more than the registers they free.  The photon-map kernels' LDS is full of candidate heaps.)
template <typename R, int TREE, bool PHOTON> __device__ __forceinline__ void stats_begin(Stats &s, int lane) {
#ifndef RTMI_TREE_STATS_LDS
#define RTMI_TREE_STATS_LDS 0
#endif
    if constexpr (sizeof(R) == 4 && !PHOTON && (TREE != KIND_TREE || RTMI_TREE_STATS_LDS)) s.zero_in_lds(stats_block<TREE>(), TREE == KIND_TREE ? STAT_SLOTS_TREE : STAT_SLOTS_FLAT, lane);
    else s.zero();
#ifdef RTMI_CYCLES
    s.cyc[3] = 0ull - __builtin_readcyclecounter();
#endif
}
// b's counters (a sample's, possibly in LDS) added to a's (registers)
__device__ __forceinline__ void stats_add(Stats &a, const Stats &b) {
#pragma unroll
    for (int i = 0; i < RTMI_N_STATS; ++i)
        if (i != RTMI_STAT_FLOPS && i != RTMI_STAT_FLOPS_SECOND) a.c[i] += b.get(i);
    a.flops += b.flops;
#ifdef RTMI_CYCLES
    a.cyc[0] += b.cyc[0]; a.cyc[1] += b.cyc[1]; a.cyc[2] += b.cyc[2];
#endif
}

// This wave's run in the photon heap workspace (photon_estimate), or null.
template <typename R, bool PHOTON> __device__ __forceinline__ HeapEntry<R> *heap_run(HeapEntry<R> *ws, const DRender &rp) {
    if constexpr (!PHOTON) return nullptr;
    if (!ws || blockIdx.x >= rp.heap_slots) return nullptr; // (wave-uniform; the host caps the grid, so the second test never fails)
    return ws + ((size_t)blockIdx.x * 64u + (threadIdx.x & 63u)) * (size_t)HEAP_WS_ENTRIES;
}

// One sample of Camera::multisample's loop body (camera.rs:274-276,283-285,292-296): sample 0 is the centred ray.
template <typename R, int TREE, bool PHOTON = false>
__device__ __forceinline__ V3<R> take_sample(const DScene<R> &sc, const DCamera<double> &cam, Rng &rng, Stats &st, int px, int py, uint32_t s, HeapEntry<R> *heap_ws = nullptr) {
    rng_begin(rng, (uint32_t)(py * cam.hsize + px), s);
    st.inc(RTMI_STAT_SAMPLES);
    st.fl += camera_flops(cam);
    CYC3_BEGIN(c_cam);
    Ray<R> ray = camera_ray<R>(cam, px, py, s != 0, rng);
    CYC3_END(st, 0, c_cam);
    const V3<R> c = color_at<R, TREE, PHOTON>(sc, ray, (int)sc.max_depth, rng, st, nullptr, nullptr, heap_ws);
    stats_sample_end(st);
    return c;
}

// Further samples the stopping rule is expected to ask of a pixel with these running sums.  The reference's sum of squares
// starts from the first sample un-squared (camera.rs:277), so what it compares with SE^2 is (V + A / n) / n with
// V = the samples' variance and A = mean(c0 - c0^2) (c0 ~ the mean): the loop stops at the n that solves
// SE^2 n^2 - V n - A = 0.  Only how work is scheduled depends on this; no pixel's result does.
template <typename R> __device__ __forceinline__ R predict_want(const DCamera<double> &cam, V3<R> sum, V3<R> sos, uint32_t cnt) {
    const R inv_n = rcp_((R)cnt), third = (R)(1.0 / 3.0);
    const V3<R> mean = sum * inv_n, var = sos * inv_n - had(mean, mean), a3 = mean - had(mean, mean);
    const R q = (var.x + var.y + var.z) * third;
    const R a = max_((a3.x + a3.y + a3.z) * third, (R)0);
    const R v = max_(q - a * inv_n, (R)0);
    const R se2 = (R)cam.se2;
    const R n = div_(v + sqrt_(v * v + (R)4 * se2 * a), (R)2 * se2);
    return n - (R)cnt;
}

// The 64-sample units a pixel with these running sums would be given in a pass: what predict_want says it still
// needs, within RTMI_SPEC_MAX and the camera's sample cap.
template <typename R> __device__ __forceinline__ uint32_t predict_units(const DCamera<double> &cam, V3<R> sum, V3<R> sos, uint32_t cnt) {
    const R want = predict_want(cam, sum, sos, cnt);
    uint32_t lim = (uint32_t)RTMI_SPEC_MAX;
    if (cam.max_samples > 0) lim = min(lim, ((uint32_t)cam.max_samples - cnt + 63u) >> 6);
    lim = max(lim, 1u);
    const R units = (R)ceil(want * (R)(RTMI_SPEC_MARGIN / 64.0));
    return units >= (R)lim ? lim : (units >= (R)1 ? (uint32_t)units : 1u); // (a NaN takes one unit)
}

// Heavy pixel `p` (its record holds the sums) gets `n` units of pass `pass` (1 .. RTMI_SPEC_PASSES), or - after the last
// pass, or when the pass's unit list is full - goes to the last launch's sequential loop.  Called by one lane per pixel.
template <typename R>
__device__ __forceinline__ void push_heavy(uint32_t *__restrict__ hdr, DWorkItem<R> *__restrict__ heavy, uint32_t *__restrict__ final_idx,
                                           uint32_t *__restrict__ unit_pixel, uint32_t unit_cap, uint32_t p, uint32_t n, int pass) {
    if (pass <= RTMI_SPEC_PASSES) {
        const uint32_t base = atomicAdd(&hdr[WK_UNIT_COUNT + pass], n);
        uint32_t *up = unit_pixel + (size_t)(pass & 1) * unit_cap;
        if (base <= unit_cap && n <= unit_cap - base) {
            heavy[p].unit_base = base; heavy[p].n_chunks = n;
#pragma unroll 1
            for (uint32_t k = 0; k < n; ++k) up[base + k] = p;
            return;
        }
#pragma unroll 1
        for (uint32_t k = 0; k < n && base < unit_cap && k < unit_cap - base; ++k) up[base + k] = 0xFFFFFFFFu; // claimed, unusable
    }
    const uint32_t f = atomicAdd(&hdr[WK_FINAL_COUNT], 1u);
    final_idx[f] = p; // f < cap: a heavy pixel comes here at most once
}

// Popping work.  One counter for the whole device serialises its pops (about 12 ns each: at one unit per pop the
// soft-shadow scene's cheap units spent 3/4 of their time waiting for it).  So every list is cut into POP_LISTS
// sublists, each with its own counter in its own cache line; a wave starts at the sublist its index names and moves
// on when one is empty; and a pop takes several items while the sublist is long and single ones near its end, so that
// the launch ends evenly.  Counters only grow and a wave leaves a sublist for good: every wave runs out of lists.
constexpr uint32_t POP_LISTS = 16;
enum { POP_LIGHT = 0, POP_HEAVY = 1, POP_FINAL = 2, POP_UNITS = 3 /* + pass */, POP_QUEUES = 3 + 8 };
static_assert((64 + POP_QUEUES * POP_LISTS * 32) * 4 <= WORK_HEADER_BYTES, "pop counters fit the header");
struct Popper {
    uint32_t list, done, seen; // the sublist this wave is at, how many it has left behind, the last head it saw there
};
__device__ __forceinline__ void pop_begin(Popper &pp) { pp.list = blockIdx.x % POP_LISTS; pp.done = 0; pp.seen = 0; }
// Takes up to `most` granules of `granule` items from list `queue` of `n` items: true with [first, first + count).
__device__ __forceinline__ bool pop(uint32_t *__restrict__ hdr, int queue, uint32_t n, uint32_t granule, uint32_t most, Popper &pp, uint32_t &first,
                                    uint32_t &count, int lane) {
    while (pp.done < POP_LISTS) {
        const uint32_t lo = (uint32_t)(((unsigned long long)n * pp.list) / POP_LISTS), hi = (uint32_t)(((unsigned long long)n * (pp.list + 1u)) / POP_LISTS);
        const uint32_t len = hi - lo, left = len > pp.seen ? len - pp.seen : 0u;
        const uint32_t take = granule * min(most, max(1u, left / (granule * (gridDim.x / POP_LISTS * 8u + 1u))));
        uint32_t b = len;
        if (len) {
            if (lane == 0) b = atomicAdd(hdr + 64 + ((uint32_t)queue * POP_LISTS + pp.list) * 32u, take);
            b = __shfl(b, 0, 64);
        }
        if (b < len) {
            pp.seen = b + take;
            first = lo + b; count = min(take, len - b);
            return true;
        }
        ++pp.done; pp.list = (pp.list + 1u) % POP_LISTS; pp.seen = 0;
    }
    return false;
}

// a finished (for this pass) pixel's sums go back to the progressive state, if there is one
template <typename R>
__device__ __forceinline__ void keep_state(DPixelState<R> *__restrict__ state, size_t pixel, V3<R> sum, V3<R> sos, uint32_t cnt) {
    if (!state) return;
    DPixelState<R> &s = state[pixel];
    s.sum[0] = sum.x; s.sum[1] = sum.y; s.sum[2] = sum.z;
    s.sos[0] = sos.x; s.sos[1] = sos.y; s.sos[2] = sos.z;
    s.cnt = cnt;
}

// Camera::render / trace_pixels / multisample (camera.rs:155-161,250-310), first pass: one wave64 per 8x8 pixel tile,
// waves stride over the tiles this rank owns; every pixel takes its first min_samples samples on its own lane (all 64
// lanes busy, neighbouring rays coherent).  Pixels the reference's loop would stop at (camera.rs:290-291) are written
// out.  The others - a few per cent of the frame, needing anything from one to thousands of further samples - are
// appended with their running sums to one of two work queues, by predicted need (variance / SE^2 - n): resample_kernel
// then spreads them over the whole GPU instead of leaving each to the wave that owns its tile.
template <typename R, int TREE, bool SPREAD, bool PHOTON = false, bool DRAW = false>
__global__ void __launch_bounds__(64, PHOTON ? RTMI_WAVES_PHOTON : TREE == KIND_TREE ? RTMI_WAVES_TREE : RTMI_WAVES) render_kernel(DScene<R> sc, DCamera<double> cam, DRender rp, float *__restrict__ out,
                                                    unsigned long long *__restrict__ stats, uint32_t *__restrict__ wk_hdr,
                                                    DWorkItem<R> *__restrict__ wk_light, DWorkItem<R> *__restrict__ wk_heavy,
                                                    uint32_t wk_cap, DPixelState<R> *__restrict__ state, HeapEntry<R> *__restrict__ heap_ws) {
    // (the work pointers are separate __restrict__ arguments so that the compiler can prove the scene records are never
    // written in this kernel and fetch the wave-uniform ones with scalar loads)
    stage_scene<R, TREE>(sc);
    HeapEntry<R> *const heap = heap_run<R, PHOTON>(heap_ws, rp);
    const int lane = threadIdx.x & 63;
    Stats st;
    stats_begin<R, TREE, PHOTON>(st, lane);
    Rng rng;
    rng.k0 = rp.seed_lo; rng.k1 = rp.seed_hi;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int sub = rp.tile_rows >> 3; // 8-row rows per ownership tile
    // A wave visit takes the first rp.first samples of rp.ppw pixels, in one of two ways that add the samples up in the
    // same order (the host chooses by the size of the whole frame, rtmi.hip launch_render):
    //   group == 1: one pixel per lane, its samples one after the other.  All 64 lanes busy; but a visit of an 8x8 tile
    //               on the glass spheres is five deep samples in a row - 1 ms - and that is the end of the launch.
    //   group == first (<= 64): lane = pixel slot * group + sample, one sample per lane, added up in sample order by
    //               the pixel's first lane.  64 / group * group lanes busy (60 of 64 at 5 samples), a visit is one sample
    //               long: small frames finish in half the time (teapot at 960x540: 11.2 -> 5.7 ms, chapter-11a at
    //               640x360: 3.5 -> 2.7 ms), large ones lose 15 %.
    // Pixels are numbered tile by tile, row-major inside the 8x8 tile.
    // (SPREAD = the second way; a template parameter because one kernel for both ways spilled 10 more registers in the first)
    const uint32_t g = SPREAD ? rp.group : 1u, slot_k = (uint32_t)lane / g, sidx = (uint32_t)lane - slot_k * g;
    const uint32_t first = SPREAD ? g : rp.first; // samples in a pixel's sums when its fate is first decided
    // DRAW: visits are taken from a counter (in the order of the frame's tiles) instead of strided over: a wave that drew
    // cheap tiles goes on to the next ones instead of ending while others still have theirs before them.  (A kernel of its
    // own: the same loop with a run-time switch cost the tree kernel 20 % at one sample per pixel, where nothing is drawn.)
    static_assert(!(DRAW && SPREAD), "a spread visit is one sample long: the counter would be its cost");
    for (uint32_t visit = blockIdx.x;; ) {
        if constexpr (DRAW) {
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&wk_hdr[WK_VISIT_HEAD], 1u);
            visit = __builtin_amdgcn_readfirstlane(v); // (every lane is here: lane 0 is the first)
        }
        if (visit >= rp.n_visits) break;
        const unsigned long long q = (unsigned long long)visit * rp.ppw + slot_k;
        const int tile = (int)(q >> 6), in_tile = (int)(q & 63u);
        int ltr = tile / rp.tiles_x, tx = tile - ltr * rp.tiles_x;
        if constexpr (PHOTON) {
            // The gather is bound by the memory system (each estimate reads some thousand photon records, every lane its
            // own), and the waves in flight are ~2,000 consecutive tiles: in row-major order a band four tiles high across
            // the whole frame.  Tiles in strips RTMI_PHOTON_STRIP wide instead - a compact block of the picture, whose
            // searches meet in a smaller part of the map that the L2s and the MALL hold on to.
            const int W = RTMI_PHOTON_STRIP, rows = rp.n_tiles / rp.tiles_x, whole = rp.tiles_x / W, per_strip = W * rows;
            if (tile < whole * per_strip) {
                const int strip = tile / per_strip, t = tile - strip * per_strip;
                ltr = t / W; tx = strip * W + (t - ltr * W);
            } else if (tile < rp.n_tiles) {
                const int t = tile - whole * per_strip, rem = rp.tiles_x - whole * W;
                ltr = t / rem; tx = whole * W + (t - ltr * rem);
            }
        }
        int lt = ltr / sub, within = ltr - lt * sub;
        int gt = lt * rp.nranks + rp.rank;
        const int x = tx * 8 + (in_tile & 7), y = (gt * sub + within) * 8 + (in_tile >> 3);
        const int ly = ltr * 8 + (in_tile >> 3);
        const bool valid = slot_k < rp.ppw && tile < rp.n_tiles && x < cam.hsize && y < cam.vsize;
        const bool owner = valid && sidx == 0; // the lane that keeps the pixel's sums
        V3<R> sum = mk<R>(0, 0, 0), sos = mk<R>(0, 0, 0);
        uint32_t cnt = 0;
        if (!SPREAD && state && valid) { // a progressive pass goes on from where the pass before left the pixel (never SPREAD)
            const DPixelState<R> &ps = state[(size_t)ly * cam.hsize + x];
            sum = mk<R>(ps.sum[0], ps.sum[1], ps.sum[2]); sos = mk<R>(ps.sos[0], ps.sos[1], ps.sos[2]);
            cnt = ps.cnt;
        }
        // (need_more is true at cnt == 0; a pixel that an earlier pass brought to its end takes nothing and is written again)
        bool go = SPREAD ? true : (valid && need_more(cam, sum, sos, (R)cnt, cnt)), queued = false, full = false, grouped = SPREAD; // (SPREAD: every lane comes along for the shuffles)
        while (go) { // wave-uniform until the pixels' sums hold `first` samples
            V3<R> c = mk<R>(0, 0, 0);
            if (!SPREAD || valid) c = take_sample<R, TREE, PHOTON>(sc, cam, rng, st, x, y, grouped ? sidx : cnt, heap);
            CYC4_BEGIN(c_book);
            // camera.rs:277-278: BOTH sums start from the first sample itself - the sum of squares from c, not from c * c -
            // and only the later samples enter it squared (camera.rs:286,297)
            if (grouped) { // the owner adds its group's samples up in sample order, like the reference's loop
                grouped = false;
                sum = c; sos = c;
                for (uint32_t k = 1; k < g; ++k) {
                    const V3<R> ck = {__shfl_down(c.x, k, 64), __shfl_down(c.y, k, 64), __shfl_down(c.z, k, 64)};
                    sum = sum + ck; sos = sos + squares(ck);
                }
                cnt = g;
                if (!owner) break;
            } else {
                sum = sum + c; sos = sos + (cnt == 0 ? c : squares(c));
                ++cnt;
                if (cnt < first) continue;
            }
            bool more = need_more(cam, sum, sos, (R)cnt, cnt);
            if (wk_hdr && !full) {
                // Who goes on here, who goes to the queues.  A lane keeps its pixel while the wave stays well filled: no
                // queue traffic, no samples taken ahead of the rule, neighbouring pixels on neighbouring lanes.  It hands
                // the pixel over - sums and count - when the pixel is predicted to need much more (its samples are then
                // better spread over a whole wave), when it has been here long enough, or when so few lanes of the wave
                // are left that the others would idle.  (The sums are the reference's sequential ones on either side,
                // so the frame does not depend on any of this.)
                const R want = more ? predict_want(cam, sum, sos, cnt) : (R)0;
                bool leave = more && (SPREAD || !(want <= (R)rp.inlane_need) || cnt >= first + rp.inlane_max);
                if (!SPREAD && (uint32_t)__popcll(__ballot(more && !leave)) < rp.keep_lanes) leave = more;
                if (leave) {
                    const bool heavy = want > (R)RTMI_LIGHT_NEED;
                    const unsigned long long mh = __ballot(heavy), ml = __ballot(!heavy);
                    const unsigned long long mine = heavy ? mh : ml;
                    const int leader = __ffsll((long long)mine) - 1;
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(&wk_hdr[heavy ? WK_HEAVY_COUNT : WK_LIGHT_COUNT], (uint32_t)__popcll(mine));
                    base = __shfl(base, leader, 64);
                    const uint32_t slot = base + (uint32_t)__popcll(mine & lt_mask);
                    if (slot < wk_cap) {
                        DWorkItem<R> *q = (heavy ? wk_heavy : wk_light) + slot;
                        q->px = (uint32_t)x; q->py = (uint32_t)y; q->out_row = (uint32_t)ly; q->cnt = cnt;
                        q->sum[0] = sum.x; q->sum[1] = sum.y; q->sum[2] = sum.z;
                        q->sos[0] = sos.x; q->sos[1] = sos.y; q->sos[2] = sos.z;
                        queued = true;
                        more = false;
                    } else { // the queue is full: this lane finishes its pixel alone (slow, still exact) and asks no more
                        full = true;
                    }
                }
            }
            go = more;
            CYC4_END(st, 2, c_book);
        }
        if (owner && !queued) {
            R inv = (R)1 / (R)cnt;
            size_t o = ((size_t)ly * cam.hsize + x) * 3;
            out[o + 0] = (float)(sum.x * inv); out[o + 1] = (float)(sum.y * inv); out[o + 2] = (float)(sum.z * inv);
            keep_state<R>(state, (size_t)ly * cam.hsize + x, sum, sos, cnt);
        }
        if constexpr (!DRAW) visit += gridDim.x;
    }
    flush_stats(st, stats);
}

// lane l takes lane l - 1's value, lane 0 takes `fill` (v_mov_b32 with the wave_shr:1 DPP control: one full-rate
// instruction per dword, no LDS).  Every lane of the wave must be active.
__device__ __forceinline__ float wave_shr1(float fill, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_shr1(double fill, double v) {
    const unsigned long long f = __builtin_bit_cast(unsigned long long, fill), x = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)f, (int)(uint32_t)x, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(f >> 32), (int)(uint32_t)(x >> 32), 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Camera::multisample adds a pixel's samples to its sums one after the other (camera.rs:285-286,296-297).  Here a group
// of G lanes holds G consecutive samples of one pixel (lane j of the group: sample j) whose sums so far are sum0 / sos0:
// sj / qj become the sums after samples 0 .. j in exactly the reference's association, ((sum0 + c_0) + c_1) + ... + c_j,
// so that neither the image nor a stop decision depends on how many lanes, waves or launches took the samples.  A chain
// of G dependent additions has no parallel form that rounds alike; it runs as a systolic pass instead: in every step
// each lane adds its own sample to its left neighbour's partial sum; lane j holds its final value from step j on
// (recomputing it from a final left neighbour gives the same bits).  G - 1 steps of 6 shifted additions.
// lane l takes lane l - 1's value and the first lane of the wave (CTRL 0x138, wave_shr:1) or of every row of 16 lanes
// (CTRL 0x111, row_shr:1) takes 0: with bound_ctrl the shift folds into the addition that uses it (v_add_f32_dpp).
template <int CTRL> __device__ __forceinline__ float shr1_zero(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double shr1_zero(double v) {
    const unsigned long long x = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// seq_scan for groups that are the whole wave or its rows of 16 lanes.  The head lane's addend is sum0 + c_0 and what
// comes in from its left is 0 (0 + x = x), so every lane runs the same shifted addition in every step, one instruction
// per component - a third of the general form's.
template <typename R, int CTRL, int G>
__device__ __forceinline__ void seq_scan_rows(V3<R> sum0, V3<R> sos0, V3<R> c, bool head, V3<R> &sj, V3<R> &qj) {
    V3<R> q = squares(c);
    if (head) { c = sum0 + c; q = sos0 + q; }
    sj = c; qj = q;
    if constexpr (sizeof(R) == 4) {
        // f32: the shift and the addition as ONE instruction per component (the compiler pairs the additions into
        // v_pk_add_f32 and keeps six separate shifts otherwise: 9 instructions per step instead of 6).  A DPP operand must
        // not be read within two instructions of being written: each of the six chains reads what it wrote six
        // instructions earlier, and the s_nop covers the way into the loop.
        float a0 = sj.x, a1 = sj.y, a2 = sj.z, a3 = qj.x, a4 = qj.y, a5 = qj.z;
        asm volatile("s_nop 1");
#pragma unroll 1
        for (int step = 1; step < G; ++step) {
            if constexpr (CTRL == 0x138) {
                asm volatile("v_add_f32_dpp %0, %0, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %1, %1, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %2, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %3, %3, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %4, %4, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %5, %5, %11 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5)
                             : "v"(c.x), "v"(c.y), "v"(c.z), "v"(q.x), "v"(q.y), "v"(q.z));
            } else {
                asm volatile("v_add_f32_dpp %0, %0, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %1, %1, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %3, %3, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %4, %4, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %5, %5, %11 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5)
                             : "v"(c.x), "v"(c.y), "v"(c.z), "v"(q.x), "v"(q.y), "v"(q.z));
            }
        }
        sj = mk<R>((R)a0, (R)a1, (R)a2); qj = mk<R>((R)a3, (R)a4, (R)a5);
    } else {
#pragma unroll 1
        for (int step = 1; step < G; ++step) {
            sj = mk<R>(shr1_zero<CTRL>(sj.x) + c.x, shr1_zero<CTRL>(sj.y) + c.y, shr1_zero<CTRL>(sj.z) + c.z);
            qj = mk<R>(shr1_zero<CTRL>(qj.x) + q.x, shr1_zero<CTRL>(qj.y) + q.y, shr1_zero<CTRL>(qj.z) + q.z);
        }
    }
}
template <typename R>
__device__ __forceinline__ void seq_scan(V3<R> sum0, V3<R> sos0, V3<R> c, int G, int j, V3<R> &sj, V3<R> &qj) {
    const bool head = j == 0;
    if (G == 64) { seq_scan_rows<R, 0x138, 64>(sum0, sos0, c, head, sj, qj); return; }
    if (G == 16) { seq_scan_rows<R, 0x111, 16>(sum0, sos0, c, head, sj, qj); return; }
    const V3<R> q = squares(c);
    sj = sum0 + c; qj = sos0 + q;
#pragma unroll 1
    for (int step = 1; step < G; ++step) {
        V3<R> ts = {wave_shr1(sum0.x, sj.x), wave_shr1(sum0.y, sj.y), wave_shr1(sum0.z, sj.z)};
        V3<R> tq = {wave_shr1(sos0.x, qj.x), wave_shr1(sos0.y, qj.y), wave_shr1(sos0.z, qj.z)};
        if (head) { ts = sum0; tq = sos0; }
        sj = ts + c; qj = tq + q;
    }
}

// One wave walks the units heavy pixel `p` was given in this pass in sample order, exactly as the sequential loop's
// 64-sample rounds would have: same prefix sums, same stopping test at every sample count.
// Stops the pixel (writes it out), or stores its sums and gives it units of the next pass.  Samples past the stop are
// discarded, their rays not counted.
template <typename R>
__device__ __forceinline__ void sum_up_units(const DCamera<double> &cam, float *__restrict__ out, uint32_t *__restrict__ hdr,
                                             DWorkItem<R> *__restrict__ heavy, uint32_t *__restrict__ final_idx, uint32_t *__restrict__ unit_pixel,
                                             const DChunk<R> *chunks, uint32_t unit_cap, uint32_t p, int pass, Stats &st, int lane,
                                             DPixelState<R> *__restrict__ state) {
    DWorkItem<R> *it = heavy + p;
    V3<R> sum = mk<R>(it->sum[0], it->sum[1], it->sum[2]), sos = mk<R>(it->sos[0], it->sos[1], it->sos[2]);
    uint32_t cnt = it->cnt;
    const uint32_t n = it->n_chunks, base = it->unit_base;
    bool stopped = false;
    uint32_t c = 0;
    // (the next unit's record is fetched while this one is summed: the walk is a chain of short dependent steps)
    R ncol[3];
    uint32_t nst[CHUNK_STAT_WORDS];
    {
        const DChunk<R> *ch = chunks + base;
#pragma unroll
        for (int k = 0; k < 3; ++k) ncol[k] = ch->col[k][lane];
#pragma unroll
        for (int k = 0; k < CHUNK_STAT_WORDS; ++k) nst[k] = ch->st[k][lane];
    }
#pragma unroll 1
    for (; c < n && !stopped; ++c) {
        V3<R> ps = mk<R>(ncol[0], ncol[1], ncol[2]);
        uint32_t w[CHUNK_STAT_WORDS];
#pragma unroll
        for (int k = 0; k < CHUNK_STAT_WORDS; ++k) w[k] = nst[k];
        if (c + 1 < n) {
            const DChunk<R> *ch = chunks + (base + c + 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) ncol[k] = ch->col[k][lane];
#pragma unroll
            for (int k = 0; k < CHUNK_STAT_WORDS; ++k) nst[k] = ch->st[k][lane];
        }
        V3<R> sj, qj;
        seq_scan(sum, sos, ps, 64, lane, sj, qj);
        const uint32_t cntj = cnt + (uint32_t)lane + 1u;
        const unsigned long long sb = __ballot(!need_more(cam, sj, qj, (R)cntj, cntj));
        const int k = sb ? __ffsll((long long)sb) : 64;
        if (lane < k) {
            st.add(RTMI_STAT_PRIMARY, w[0] & 0xFFu); st.add(RTMI_STAT_SAMPLES, (w[0] >> 8) & 0xFFu); st.add(RTMI_STAT_OVERFLOW, w[0] >> 16);
            st.add(RTMI_STAT_REFLECTION, w[1]); st.add(RTMI_STAT_REFRACTION, w[2]);
            st.add(RTMI_STAT_SHADOW, w[3]); st.add(RTMI_STAT_LEAF_TESTS, w[4]);
            st.add(RTMI_STAT_NODE_VISITS, w[5]); st.add(RTMI_STAT_AABB_TESTS, w[6]);
            st.flops += w[7];
        } else st.add(RTMI_STAT_DISCARDED, (w[0] >> 8) & 0xFFu);
        sum = shfl3(sj, k - 1); sos = shfl3(qj, k - 1);
        cnt += (uint32_t)k;
        stopped = sb != 0;
    }
    if (c < n) st.add(RTMI_STAT_DISCARDED, (nst[0] >> 8) & 0xFFu); // (already fetched)
#pragma unroll 1
    for (++c; c < n; ++c) st.add(RTMI_STAT_DISCARDED, (chunks[base + c].st[0][lane] >> 8) & 0xFFu);
    if (lane != 0) return;
    if (stopped) {
        const R inv = (R)1 / (R)cnt;
        const size_t o = ((size_t)it->out_row * cam.hsize + it->px) * 3;
        out[o + 0] = (float)(sum.x * inv); out[o + 1] = (float)(sum.y * inv); out[o + 2] = (float)(sum.z * inv);
        keep_state<R>(state, (size_t)it->out_row * cam.hsize + it->px, sum, sos, cnt);
    } else {
        it->cnt = cnt;
        it->sum[0] = sum.x; it->sum[1] = sum.y; it->sum[2] = sum.z;
        it->sos[0] = sos.x; it->sos[1] = sos.y; it->sos[2] = sos.z;
        push_heavy<R>(hdr, heavy, final_idx, unit_pixel, unit_cap, p, predict_units(cam, sum, sos, cnt), pass + 1);
    }
}

// After each unit pass: one wave per heavy pixel of the pass sums the pixel's units up.  A wave looks at 64 units at a
// time and takes the pixels whose first unit is among them.  No scene access.
template <typename R>
__global__ void __launch_bounds__(64) sum_up_kernel(DCamera<double> cam, float *__restrict__ out, unsigned long long *__restrict__ stats,
                                                    uint32_t *__restrict__ wk_hdr, DWorkItem<R> *__restrict__ wk_heavy, uint32_t *__restrict__ wk_final,
                                                    uint32_t *__restrict__ wk_unit_pixel, const DChunk<R> *__restrict__ wk_chunks, uint32_t wk_cap,
                                                    uint32_t wk_unit_cap, int pass, DPixelState<R> *__restrict__ state) {
    Stats st;
    stats_zero(st);
    const int lane = threadIdx.x & 63;
    const uint32_t n_units = min(wk_hdr[WK_UNIT_COUNT + pass], wk_unit_cap);
    const uint32_t *units = wk_unit_pixel + (size_t)(pass & 1) * wk_unit_cap;
    for (uint32_t u0 = blockIdx.x * 64u; u0 < n_units; u0 += gridDim.x * 64u) { // (n_units <= 2^30: no wrap)
        const uint32_t u = u0 + (uint32_t)lane;
        const uint32_t p = u < n_units ? units[u] : 0xFFFFFFFFu;
        const uint32_t prev = (u > 0 && u < n_units) ? units[u - 1] : 0xFFFFFFFFu;
        unsigned long long firsts = __ballot(p < wk_cap && (u == 0 || prev != p)); // (p >= cap: nobody's unit)
        while (firsts) {
            const int i = __ffsll((long long)firsts) - 1;
            firsts &= firsts - 1ull;
            sum_up_units<R>(cam, out, wk_hdr, wk_heavy, wk_final, wk_unit_pixel, wk_chunks, wk_unit_cap, __shfl(p, i, 64), pass, st, lane, state);
        }
    }
    flush_stats(st, stats, true);
}

// Camera::multisample's open-ended loop (camera.rs:290-299) for the pixels render_kernel queued.  Three kernels share
// it, launched on the stream as  resample(pass 0), { units(pass), sum_up(pass) } for pass 1 .. RTMI_SPEC_PASSES,
// resample(last).  Every launch is a grid of persistent waves that pop work until their lists - all fixed before the
// launch - are empty, so every wave reaches the exit.
//   resample_kernel, pass 0: the sequential loop.  First the heavy queue, one pixel per wave, 64 samples per round, then
//     the light queue, RTMI_LIGHT_BATCH pixels per wave: a group of G lanes computes a pixel's next G samples at once
//     (counter RNG: sample s of a pixel does not depend on who computes it); a prefix sum over the group in sample
//     order then finds the first sample count at which the reference's loop would have stopped.  A pixel that has
//     shown (with 64 or more samples behind its variance) that it needs RTMI_SPEC_MIN_UNITS more rounds or more is
//     PROMOTED instead: its sums go to the heavy table and the rounds it is predicted to need become units of pass 1.
//   units_kernel: a unit is 64 consecutive samples of one promoted pixel, one per lane, left in a DChunk for
//     sum_up_kernel, which hands pixels that are still open to the next pass.  A pixel's thousands of samples thus
//     spread over the whole GPU instead of running 64 at a time on one wave.
//   resample_kernel, last launch (pass < 0): the sequential loop for the promoted pixels still open after the last pass.
// A pixel's samples are always summed in rounds of 64 (light: RTMI_LIGHT_ROUNDS rounds of 64 / RTMI_LIGHT_BATCH first)
// counted from its own first-pass count, whoever computes them and however many were computed ahead: the image does
// not depend on queue order, sharding, promotion or the prediction, bit for bit.
template <typename R, int TREE, bool PHOTON = false>
__global__ void __launch_bounds__(64, PHOTON ? RTMI_WAVES_PHOTON : TREE == KIND_TREE ? RTMI_WAVES_TREE : RTMI_WAVES) resample_kernel(DScene<R> sc, DCamera<double> cam, DRender rp, float *__restrict__ out,
                                                      unsigned long long *__restrict__ stats, uint32_t *__restrict__ wk_hdr,
                                                      const DWorkItem<R> *__restrict__ wk_light, DWorkItem<R> *__restrict__ wk_heavy,
                                                      uint32_t *__restrict__ wk_final, uint32_t *__restrict__ wk_unit_pixel, uint32_t wk_cap,
                                                      uint32_t wk_unit_cap, int pass, DPixelState<R> *__restrict__ state, HeapEntry<R> *__restrict__ heap_ws) {
    Stats st;
    stats_zero(st);
    Rng rng;
    rng.k0 = rp.seed_lo; rng.k1 = rp.seed_hi;
    const int lane = threadIdx.x & 63;
    HeapEntry<R> *const heap = heap_run<R, PHOTON>(heap_ws, rp);
    const uint32_t n_heavy = min(wk_hdr[WK_HEAVY_COUNT], wk_cap); // (fixed by render_kernel)
    const uint32_t n_light = pass == 0 ? min(wk_hdr[WK_LIGHT_COUNT], wk_cap) : 0u;
    // the heavy pixels of the sequential loop: the first pass's in pass 0, the leftovers in the last launch
    const uint32_t n_seq = pass == 0 ? n_heavy : min(wk_hdr[WK_FINAL_COUNT], wk_cap);
    const bool promote = pass == 0 && RTMI_SPEC_PASSES > 0;
    // Promotion shortens the end of this launch (no pixel runs its thousands of samples alone) at the price of the unit
    // passes' own launches and ends: worth it only for pixels that would outlast the queue, so the bar rises with the
    // jobs each wave has to do anyway (headline frame: 55 jobs per wave, nothing is promoted; its 1/8 shard: 7,
    // every pixel predicted to need RTMI_SPEC_MIN_UNITS rounds is).  The image does not depend on it.
    const uint32_t min_units = max((uint32_t)RTMI_SPEC_MIN_UNITS, (n_heavy + n_light / (uint32_t)RTMI_LIGHT_BATCH) / (gridDim.x * (uint32_t)RTMI_SPEC_LOAD_DIV));
    const uint32_t none = 0xFFFFFFFFu;
    // (what the wave knows about its pops lives in LDS: nothing of it is wanted in registers across a sample)
    __shared__ Popper seq_pop, light_pop;
    __shared__ uint32_t popped[4]; // first item and number of items popped and not begun yet, 1 if they are heavy, 1 while the heavy list lasts
    stage_scene<R, TREE>(sc);
    pop_begin(seq_pop); pop_begin(light_pop);
    popped[0] = 0; popped[1] = 0; popped[2] = 0; popped[3] = n_seq > 0;
    __syncthreads();
    for (;;) {
        // ---- take a batch (the longest kind first) ----------------------------------------------------------------
        uint32_t got = popped[0], got_n = popped[1];
        bool got_heavy = popped[2] != 0;
        if (got_n == 0) {
            if (popped[3] && pop(wk_hdr, pass == 0 ? POP_HEAVY : POP_FINAL, n_seq, 1u, 1u, seq_pop, got, got_n, lane)) got_heavy = true;
            else {
                popped[3] = 0;
                if (!pop(wk_hdr, POP_LIGHT, n_light, (uint32_t)RTMI_LIGHT_BATCH, (uint32_t)RTMI_LIGHT_POP, light_pop, got, got_n, lane)) break;
                got_heavy = false;
            }
        }
        const DWorkItem<R> *items = nullptr;
        int P = 0;
        const bool heavy_batch = got_heavy;
        uint32_t slot = none; // the pixel's place in the heavy table, if it has one
        if (heavy_batch) {
            const uint32_t p = pass == 0 ? got : wk_final[got];
            items = wk_heavy + p; P = 1;
            if (lane == 0) slot = p;
        } else {
            items = wk_light + got;
            P = (int)min((uint32_t)RTMI_LIGHT_BATCH, got_n);
        }
        popped[0] = got + (uint32_t)P; popped[1] = got_n - (uint32_t)P; popped[2] = got_heavy;
        bool valid = lane < P;
        V3<R> sum = mk<R>(0, 0, 0), sos = mk<R>(0, 0, 0);
        uint32_t cnt = 0;
        int px = 0, py = 0, out_row = 0;
        if (valid) {
            const DWorkItem<R> &q = items[lane];
            px = (int)q.px; py = (int)q.py; out_row = (int)q.out_row; cnt = q.cnt;
            sum = mk<R>(q.sum[0], q.sum[1], q.sum[2]);
            sos = mk<R>(q.sos[0], q.sos[1], q.sos[2]);
        }
        // ---- rounds -----------------------------------------------------------------------------------------
        // The chunking of a pixel's samples depends on that pixel alone (never on its neighbours in the batch or on
        // the order of the queue): heavy pixels take 64 samples per round; light pixels RTMI_LIGHT_ROUNDS rounds of
        // 64 / RTMI_LIGHT_BATCH, then 64 per round, one pixel after the other.
        for (int round = 0;; ++round) {
            bool act = valid && need_more(cam, sum, sos, (R)cnt, cnt);
            if (promote && act && round >= (heavy_batch ? 1 : RTMI_LIGHT_ROUNDS)) {
                // (from here on the pixel's rounds are 64 samples wide, as its units will be)
                const uint32_t n = predict_units(cam, sum, sos, cnt);
                if (n >= min_units) {
                    if (slot == none) { // a light pixel: heavy slots from the top of the table
                        const uint32_t k = atomicAdd(&wk_hdr[WK_PROMO_COUNT], 1u);
                        if (k < wk_cap - n_heavy) slot = wk_cap - 1u - k;
                    }
                    if (slot != none) {
                        DWorkItem<R> *q = wk_heavy + slot;
                        q->px = (uint32_t)px; q->py = (uint32_t)py; q->out_row = (uint32_t)out_row; q->cnt = cnt;
                        q->sum[0] = sum.x; q->sum[1] = sum.y; q->sum[2] = sum.z;
                        q->sos[0] = sos.x; q->sos[1] = sos.y; q->sos[2] = sos.z;
                        push_heavy<R>(wk_hdr, wk_heavy, wk_final, wk_unit_pixel, wk_unit_cap, slot, n, 1);
                        valid = false; act = false;
                    } // else: the table is full and the pixel stays here (slow, still exact)
                }
            }
            const unsigned long long mask = __ballot(act);
            if (mask == 0) break;
            const bool wide = heavy_batch || round >= RTMI_LIGHT_ROUNDS;
            const int G = wide ? 64 : 64 / RTMI_LIGHT_BATCH;
            const int r = wide ? 0 : lane / G, j = wide ? lane : lane & (G - 1);
            const int owner = wide ? __ffsll((long long)mask) - 1 : (((mask >> r) & 1ull) ? r : -1);
            const bool has = owner >= 0;
            const int src = has ? owner : lane;
            const uint32_t cnt0 = __shfl(cnt, src, 64);
            const V3<R> sum0 = shfl3(sum, src), sos0 = shfl3(sos, src);
            const int ox = __shfl(px, src, 64), oy = __shfl(py, src, 64);
            const uint32_t s = cnt0 + (uint32_t)j;
            V3<R> c = mk<R>(0, 0, 0);
            Stats ls;
            stats_begin<R, TREE, PHOTON>(ls, lane);
            if (has && !(cam.max_samples > 0 && s >= (uint32_t)cam.max_samples)) c = take_sample<R, TREE, PHOTON>(sc, cam, rng, ls, ox, oy, s, heap);
            // the pixel's sums after each of the group's samples, added up one after the other like the reference's loop
            V3<R> sj, qj;
            seq_scan(sum0, sos0, c, G, j, sj, qj);
            const uint32_t cntj = cnt0 + (uint32_t)j + 1u;
            const bool stop = has && !need_more(cam, sj, qj, (R)cntj, cntj);
            const unsigned long long sb = __ballot(stop);
            const unsigned long long gb = wide ? sb : (sb >> (r * G)) & ((1ull << G) - 1ull);
            const int k_used = gb ? __ffsll((long long)gb) : G; // samples of this group that the reference would have taken
            if (has && j < k_used) stats_add(st, ls);
            else st.add(RTMI_STAT_DISCARDED, ls.get(RTMI_STAT_SAMPLES));
            // hand the result back to the pixel's own lane
            const bool upd = act && (!wide || lane == owner);
            const int gbase = upd ? (wide ? 0 : lane * G) : lane;
            const int kk = __shfl(k_used, gbase, 64);
            const int fin = upd ? gbase + kk - 1 : lane;
            const V3<R> fs = shfl3(sj, fin), fq = shfl3(qj, fin);
            if (upd) { sum = fs; sos = fq; cnt += (uint32_t)kk; }
        }
        if (valid) {
            R inv = (R)1 / (R)cnt;
            size_t o = ((size_t)out_row * cam.hsize + px) * 3;
            out[o + 0] = (float)(sum.x * inv); out[o + 1] = (float)(sum.y * inv); out[o + 2] = (float)(sum.z * inv);
            keep_state<R>(state, (size_t)out_row * cam.hsize + px, sum, sos, cnt);
        }
    }
    flush_stats(st, stats, true);
}

// The units of pass `pass` (see resample_kernel): each wave pops batches of units, takes every unit's 64 samples and
// leaves colours and counters in the unit's DChunk.
constexpr int UNIT_BATCH_MAX = 16;
template <typename R, int TREE, bool PHOTON = false>
__global__ void __launch_bounds__(64, PHOTON ? RTMI_WAVES_PHOTON : TREE == KIND_TREE ? RTMI_WAVES_TREE : RTMI_WAVES) units_kernel(DScene<R> sc, DCamera<double> cam, DRender rp, uint32_t *__restrict__ wk_hdr,
                                                   const DWorkItem<R> *__restrict__ wk_heavy, const uint32_t *__restrict__ wk_unit_pixel,
                                                   DChunk<R> *__restrict__ wk_chunks, uint32_t wk_cap, uint32_t wk_unit_cap, int pass, HeapEntry<R> *__restrict__ heap_ws) {
    stage_scene<R, TREE>(sc);
    HeapEntry<R> *const heap = heap_run<R, PHOTON>(heap_ws, rp);
    Rng rng;
    rng.k0 = rp.seed_lo; rng.k1 = rp.seed_hi;
    const int lane = threadIdx.x & 63;
    const uint32_t n_units = min(wk_hdr[WK_UNIT_COUNT + pass], wk_unit_cap);
    const uint32_t *units = wk_unit_pixel + (size_t)(pass & 1) * wk_unit_cap;
    // The batch's pixels are fetched by one lane each and handed out by lane reads.  (No LDS, no barrier: a fence in
    // this loop would stop the compiler from proving that the scene is never written here, and with that proof go the
    // scalar loads of the scene records.)
    Popper pp;
    pop_begin(pp);
    for (;;) {
        uint32_t u0 = 0, nb = 0;
        if (!pop(wk_hdr, POP_UNITS + pass, n_units, 1u, (uint32_t)UNIT_BATCH_MAX, pp, u0, nb, lane)) break;
        uint32_t bp = 0xFFFFFFFFu, bx = 0, by = 0, bc = 0;
        if ((uint32_t)lane < nb) {
            bp = units[u0 + (uint32_t)lane];
            if (bp < wk_cap) {
                const DWorkItem<R> *q = wk_heavy + bp;
                bx = q->px; by = q->py;
                bc = q->cnt + (u0 + (uint32_t)lane - q->unit_base) * 64u; // first sample of the unit
            }
        }
#pragma unroll 1
        for (uint32_t i = 0; i < nb; ++i) {
            if (__shfl(bp, (int)i, 64) >= wk_cap) continue; // a unit nobody owns (the list overflowed)
            const int px = (int)__shfl(bx, (int)i, 64), py = (int)__shfl(by, (int)i, 64);
            const uint32_t s = __shfl(bc, (int)i, 64) + (uint32_t)lane;
            V3<R> c = mk<R>(0, 0, 0);
            Stats ls;
            stats_begin<R, TREE, PHOTON>(ls, lane);
            if (!(cam.max_samples > 0 && s >= (uint32_t)cam.max_samples)) c = take_sample<R, TREE, PHOTON>(sc, cam, rng, ls, px, py, s, heap);
            DChunk<R> *ch = wk_chunks + (u0 + i);
            ch->col[0][lane] = c.x; ch->col[1][lane] = c.y; ch->col[2][lane] = c.z;
            ch->st[0][lane] = (ls.get(RTMI_STAT_PRIMARY) & 0xFFu) | ((ls.get(RTMI_STAT_SAMPLES) & 0xFFu) << 8) | (min(ls.get(RTMI_STAT_OVERFLOW), 0xFFFFu) << 16);
            ch->st[1][lane] = ls.get(RTMI_STAT_REFLECTION); ch->st[2][lane] = ls.get(RTMI_STAT_REFRACTION);
            ch->st[3][lane] = ls.get(RTMI_STAT_SHADOW); ch->st[4][lane] = ls.get(RTMI_STAT_LEAF_TESTS);
            ch->st[5][lane] = ls.get(RTMI_STAT_NODE_VISITS); ch->st[6][lane] = ls.get(RTMI_STAT_AABB_TESTS);
            ch->st[7][lane] = (uint32_t)ls.flops;
        }
    }
}

#endif // RTMI_HOST_EMUL

// World::color_at for arbitrary rays + the IntersectionState of the first hit (rtmi_trace_rays).
// TREE = false: the flat-scene code of the render kernels (LDS scene block, pair tables), for worlds whose objects are all
// primitives - so that the known-answer tests and the host emulation go through the very loops the frames are rendered with.
template <typename R, bool PHOTON = false, int TREE = KIND_TREE>
__global__ void __launch_bounds__(64) trace_kernel(DScene<R> sc, uint32_t seed_lo, uint32_t seed_hi, long long n, const double *origins,
                                                   const double *dirs, const int *remaining, double *rgb, rtmi_hit_record *recs) {
    stage_scene<R, TREE>(sc); // (every lane of the wave, before any leaves)
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Stats st;
    st.zero();
    Rng rng;
    rng.k0 = seed_lo; rng.k1 = seed_hi;
    rng_begin(rng, (uint32_t)i, 0);
    Ray<R> ray = {mk<R>((R)origins[3 * i], (R)origins[3 * i + 1], (R)origins[3 * i + 2]),
                  mk<R>((R)dirs[3 * i], (R)dirs[3 * i + 1], (R)dirs[3 * i + 2])};
    Hit<R> h;
    Comps<R> c;
    c.point = mk<R>(0, 0, 0); c.normalv = c.point; c.eyev = c.point; c.inside = false; c.side_plus = false; c.n1 = 1; c.n2 = 1;
    int rem = remaining ? remaining[i] : (int)sc.max_depth;
    V3<R> col = color_at<R, TREE, PHOTON>(sc, ray, rem, rng, st, &h, &c);
    if (rgb) { rgb[3 * i] = (double)col.x; rgb[3 * i + 1] = (double)col.y; rgb[3 * i + 2] = (double)col.z; }
    if (recs) {
        rtmi_hit_record r;
        memset(&r, 0, sizeof r);
        r.shadowed = -1.0;
        if (h.leaf != NONE) {
            r.hit = 1;
            r.leaf = (int32_t)sc.leaf_orig[h.leaf];
            r.inside = c.inside ? 1 : 0;
            r.t = (double)h.t;
            r.point[0] = (double)c.point.x; r.point[1] = (double)c.point.y; r.point[2] = (double)c.point.z;
            r.normal[0] = (double)c.normalv.x; r.normal[1] = (double)c.normalv.y; r.normal[2] = (double)c.normalv.z;
            r.n1 = (double)c.n1; r.n2 = (double)c.n2;
            if (sc.n_lights > 0) {
                int k = sc.lights[0].kind;
                if (k == RTMI_LIGHT_POINT || k == RTMI_LIGHT_REALISTIC_POINT || k == RTMI_LIGHT_BEAM) {
                    V3<R> over_point = c.point + c.normalv * (R)RTMI_EPSILON;
                    Incoming<R> in = incoming_at(sc.lights[0], over_point, rng);
                    if (in.kind == IL_RAY) r.shadowed = is_shadowed<R, TREE>(sc, in, over_point, h.leaf, c.side_plus ? SIDE_PLUS : 0u, st) ? 1.0 : 0.0;
                }
            }
        }
        recs[i] = r;
    }
}

// World::compute_photon_map's tracing half (rtmi_trace_photons): lane l of the grid traces photons l, l + lanes, ...
template <typename R, bool STORE>
__global__ void __launch_bounds__(64) photon_trace_kernel(DScene<R> sc, DPhotonTrace pt, rtmi_photon *out, uint32_t *counts, const uint32_t *offsets,
                                                          unsigned long long *counters) {
    Stats st;
    st.zero();
    trace_photons<R, STORE>(sc, pt, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, out, counts, offsets, counters, st);
}

// Camera::ray_for_pixel for a batch of (px, py, sample) (rtmi_camera_rays)
template <typename R>
__global__ void camera_rays_kernel(DCamera<double> cam, uint32_t seed_lo, uint32_t seed_hi, long long n, const int *px, const int *py,
                                   const int *sample, double *origins, double *dirs) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng;
    rng.k0 = seed_lo; rng.k1 = seed_hi;
    int s = sample ? sample[i] : 0;
    rng_begin(rng, (uint32_t)(py[i] * cam.hsize + px[i]), (uint32_t)s);
    Ray<R> r = camera_ray<R>(cam, px[i], py[i], s != 0, rng);
    origins[3 * i] = (double)r.o.x; origins[3 * i + 1] = (double)r.o.y; origins[3 * i + 2] = (double)r.o.z;
    dirs[3 * i] = (double)r.d.x; dirs[3 * i + 1] = (double)r.d.y; dirs[3 * i + 2] = (double)r.d.z;
}

// ---------------------------------------------------------------------------------------------
// launchers (declared in rtmi_launch.hpp, one specialisation per build of this file)
// ---------------------------------------------------------------------------------------------
namespace rtmi_dev {
#ifndef RTMI_HOST_EMUL
template <>
void launch_render_kernel<RTMI_REAL>(bool tree, int grid, int grid_resample, hipStream_t st, const DScene<RTMI_REAL> &sc,
                                     const DCamera<double> &cam, const DRender &rp_, float *out, unsigned long long *stats, void *work,
                                     size_t work_bytes, hipEvent_t between, void *state_) {
    const bool photon = sc.n_photons != 0; // (kernels of their own, tree kind, one lane per pixel: the host chooses tree = true, no spread)
    // A photon scene in f32 mode: the tail of the work memory is the heap workspace (photon_estimate), if the memory is the
    // pool's (the 96 MiB of the entry point that allocates nothing are too few: per-lane scratch there, as in f64 mode)
    HeapEntry<RTMI_REAL> *heap_ws = nullptr;
    DRender rp = rp_;
    rp.heap_slots = 0;
    if (sizeof(RTMI_REAL) == 4 && photon && work) {
        const size_t heap_bytes = ((size_t)HEAP_WS_SLOTS * 64u * (size_t)HEAP_WS_ENTRIES * sizeof(HeapEntry<RTMI_REAL>) + 255u) & ~(size_t)255u;
        if (work_bytes >= heap_bytes + ((size_t)512 << 20)) {
            work_bytes -= heap_bytes;
            heap_ws = (HeapEntry<RTMI_REAL> *)((char *)work + work_bytes);
            rp.heap_slots = HEAP_WS_SLOTS;
        }
    }
    DWork<RTMI_REAL> wk = make_work<RTMI_REAL>(work, work_bytes);
    DPixelState<RTMI_REAL> *state = (DPixelState<RTMI_REAL> *)state_;
    const bool spread = rp.group > 1;
    const bool draw = RTMI_DRAW_VISITS && !spread && wk.hdr != nullptr && grid_resample > 0; // (an open-ended call: its visits are long and uneven)
    bool flatx = false; // a flat world with a transformed-sphere table: the KIND_FLATX kernels (f32 only)
    if constexpr (sizeof(RTMI_REAL) == 4) flatx = !tree && !photon && (sc.xpair_counts & 255u) != 0u;
#define RTMI_LAUNCH_RENDER(...) hipLaunchKernelGGL((render_kernel<RTMI_REAL, __VA_ARGS__>), dim3(grid), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.cap, state, heap_ws)
    if (photon && draw) RTMI_LAUNCH_RENDER(true, false, true, true);
    else if (photon) RTMI_LAUNCH_RENDER(true, false, true, false);
    else if (tree && spread) RTMI_LAUNCH_RENDER(true, true, false, false);
    else if (tree && draw) RTMI_LAUNCH_RENDER(true, false, false, true);
    else if (tree) RTMI_LAUNCH_RENDER(true, false, false, false);
    else if (flatx) {
        if constexpr (sizeof(RTMI_REAL) == 4) {
            if (spread) RTMI_LAUNCH_RENDER(KIND_FLATX, true, false, false);
            else if (draw) RTMI_LAUNCH_RENDER(KIND_FLATX, false, false, true);
            else RTMI_LAUNCH_RENDER(KIND_FLATX, false, false, false);
        }
    }
    else if (spread) RTMI_LAUNCH_RENDER(false, true, false, false);
    else if (draw) RTMI_LAUNCH_RENDER(false, false, false, true);
    else RTMI_LAUNCH_RENDER(false, false, false, false);
#undef RTMI_LAUNCH_RENDER
    if (between) (void)hipEventRecord(between, st); // (rtmi_render_device_timed)
    if (!wk.hdr || grid_resample <= 0) return;
    for (int l = 0; l <= RTMI_SPEC_PASSES + 1; ++l) {
        const int pass = l <= RTMI_SPEC_PASSES ? l : -1;
        if (pass < 0 && RTMI_SPEC_PASSES == 0) break;
        if (pass > 0) {
            if (photon) hipLaunchKernelGGL((units_kernel<RTMI_REAL, true, true>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, wk.hdr, wk.heavy, wk.unit_pixel, wk.chunks, wk.cap, wk.unit_cap, pass, heap_ws);
            else if (tree) hipLaunchKernelGGL((units_kernel<RTMI_REAL, true>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, wk.hdr, wk.heavy, wk.unit_pixel, wk.chunks, wk.cap, wk.unit_cap, pass, heap_ws);
            else if (flatx) { if constexpr (sizeof(RTMI_REAL) == 4) hipLaunchKernelGGL((units_kernel<RTMI_REAL, KIND_FLATX>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, wk.hdr, wk.heavy, wk.unit_pixel, wk.chunks, wk.cap, wk.unit_cap, pass, heap_ws); }
            else hipLaunchKernelGGL((units_kernel<RTMI_REAL, false>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, wk.hdr, wk.heavy, wk.unit_pixel, wk.chunks, wk.cap, wk.unit_cap, pass, heap_ws);
            hipLaunchKernelGGL(sum_up_kernel<RTMI_REAL>, dim3(grid_resample), dim3(64), 0, st, cam, out, stats, wk.hdr, wk.heavy, wk.final_idx, wk.unit_pixel, wk.chunks, wk.cap, wk.unit_cap, pass, state);
        } else {
            if (photon) hipLaunchKernelGGL((resample_kernel<RTMI_REAL, true, true>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.final_idx, wk.unit_pixel, wk.cap, wk.unit_cap, pass, state, heap_ws);
            else if (tree) hipLaunchKernelGGL((resample_kernel<RTMI_REAL, true>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.final_idx, wk.unit_pixel, wk.cap, wk.unit_cap, pass, state, heap_ws);
            else if (flatx) { if constexpr (sizeof(RTMI_REAL) == 4) hipLaunchKernelGGL((resample_kernel<RTMI_REAL, KIND_FLATX>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.final_idx, wk.unit_pixel, wk.cap, wk.unit_cap, pass, state, heap_ws); }
            else hipLaunchKernelGGL((resample_kernel<RTMI_REAL, false>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.final_idx, wk.unit_pixel, wk.cap, wk.unit_cap, pass, state, heap_ws);
        }
    }
}
#else
template <>
void launch_render_kernel<RTMI_REAL>(bool, int, int, hipStream_t, const DScene<RTMI_REAL> &, const DCamera<double> &, const DRender &, float *,
                                     unsigned long long *, void *, size_t, hipEvent_t, void *) {}
#endif
template <>
void launch_trace_kernel<RTMI_REAL>(int grid, const DScene<RTMI_REAL> &sc, uint32_t seed_lo, uint32_t seed_hi, long long n,
                                    const double *origins, const double *dirs, const int *remaining, double *rgb, rtmi_hit_record *recs) {
    if (sc.n_photons) hipLaunchKernelGGL((trace_kernel<RTMI_REAL, true>), dim3(grid), dim3(64), 0, 0, sc, seed_lo, seed_hi, n, origins, dirs, remaining, rgb, recs);
#if RTMI_TRACE_FLAT
    else if (staged_v<RTMI_REAL, KIND_FLATX> && sc.root_child_count == 0) hipLaunchKernelGGL((trace_kernel<RTMI_REAL, false, KIND_FLATX>), dim3(grid), dim3(64), 0, 0, sc, seed_lo, seed_hi, n, origins, dirs, remaining, rgb, recs);
#endif
    else hipLaunchKernelGGL((trace_kernel<RTMI_REAL, false>), dim3(grid), dim3(64), 0, 0, sc, seed_lo, seed_hi, n, origins, dirs, remaining, rgb, recs);
}
template <>
void launch_photon_trace_kernel<RTMI_REAL>(bool store, int grid, hipStream_t st, const DScene<RTMI_REAL> &sc, const DPhotonTrace &pt, rtmi_photon *out,
                                           uint32_t *counts, const uint32_t *offsets, unsigned long long *counters) {
    if (store) hipLaunchKernelGGL((photon_trace_kernel<RTMI_REAL, true>), dim3(grid), dim3(64), 0, st, sc, pt, out, counts, offsets, counters);
    else hipLaunchKernelGGL((photon_trace_kernel<RTMI_REAL, false>), dim3(grid), dim3(64), 0, st, sc, pt, out, counts, offsets, counters);
}
template <>
void launch_camera_rays_kernel<RTMI_REAL>(int grid, const DCamera<double> &cam, uint32_t seed_lo, uint32_t seed_hi, long long n,
                                          const int *px, const int *py, const int *sample, double *origins, double *dirs) {
    hipLaunchKernelGGL(camera_rays_kernel<RTMI_REAL>, dim3(grid), dim3(256), 0, 0, cam, seed_lo, seed_hi, n, px, py, sample, origins, dirs);
}
} // namespace rtmi_dev
