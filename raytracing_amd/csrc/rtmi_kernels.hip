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
#ifndef RTMI_WAVES_TREE
#define RTMI_WAVES_TREE 3 // ... and the tree-scene kernels
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

__device__ __forceinline__ void flush_stats(const Stats &st, unsigned long long *__restrict__ stats) {
    if (!stats) return;
#pragma unroll
    for (int i = 0; i < RTMI_N_STATS; ++i) {
        if (i == RTMI_STAT_FLOPS) continue;
        uint32_t s = wave_sum(st.c[i]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&stats[i], (unsigned long long)s);
    }
    unsigned long long f = st.flops;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) f += __shfl_xor(f, off, 64);
    if ((threadIdx.x & 63) == 0 && f) atomicAdd(&stats[RTMI_STAT_FLOPS], f);
}

// Camera::multisample's loop condition (camera.rs:281,290-291 with color_variance_of_mean, 305-310):
// true while the pixel that has taken `cnt` samples (n == cnt as a float) must take another one.
template <typename R> __device__ __forceinline__ bool need_more(const DCamera<double> &cam, V3<R> sum, V3<R> sos, R n, uint32_t cnt) {
    if (cnt == 0) return true;
    if (cam.max_samples > 0 && cnt >= (uint32_t)cam.max_samples) return false;
    if (n < (R)cam.min_samples) return true;
    R inv_n = rcp_(n);
    V3<R> mean = sum * inv_n;
    V3<R> var = sos * inv_n - had(mean, mean);
    R variance = div_(var.x + var.y + var.z, (R)3);
    return div_(variance, n) > (R)cam.se2;
}

template <typename R> __device__ __forceinline__ V3<R> shfl3(V3<R> v, int src) {
    return {__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64)};
}

__device__ __forceinline__ void stats_zero(Stats &s) {
#pragma unroll
    for (int i = 0; i < RTMI_N_STATS; ++i) s.c[i] = 0;
    s.flops = 0;
}
__device__ __forceinline__ void stats_add(Stats &a, const Stats &b) {
#pragma unroll
    for (int i = 0; i < RTMI_N_STATS; ++i) a.c[i] += b.c[i];
    a.flops += b.flops;
}

// One sample of Camera::multisample's loop body (camera.rs:274-276,283-285,292-296): sample 0 is the centred ray.
template <typename R, bool TREE>
__device__ __forceinline__ V3<R> take_sample(const DScene<R> &sc, const DCamera<double> &cam, Rng &rng, Stats &st, int px, int py, uint32_t s) {
    rng_begin(rng, (uint32_t)(py * cam.hsize + px), s);
    st.c[RTMI_STAT_SAMPLES]++;
    st.flops += camera_flops(cam);
    Ray<R> ray = camera_ray<R>(cam, px, py, s != 0, rng);
    return color_at<R, TREE>(sc, ray, (int)sc.max_depth, rng, st, nullptr, nullptr);
}

// Camera::render / trace_pixels / multisample (camera.rs:155-161,250-310), first pass: one wave64 per 8x8 pixel tile,
// waves stride over the tiles this rank owns; every pixel takes its first min_samples samples on its own lane (all 64
// lanes busy, neighbouring rays coherent).  Pixels the reference's loop would stop at (camera.rs:290-291) are written
// out.  The others - a few per cent of the frame, needing anything from one to thousands of further samples - are
// appended with their running sums to one of two work queues, by predicted need (variance / SE^2 - n): resample_kernel
// then spreads them over the whole GPU instead of leaving each to the wave that owns its tile.
template <typename R, bool TREE>
__global__ void __launch_bounds__(64, TREE ? RTMI_WAVES_TREE : RTMI_WAVES) render_kernel(DScene<R> sc, DCamera<double> cam, DRender rp, float *__restrict__ out,
                                                    unsigned long long *__restrict__ stats, uint32_t *__restrict__ wk_hdr,
                                                    DWorkItem<R> *__restrict__ wk_light, DWorkItem<R> *__restrict__ wk_heavy, uint32_t wk_cap) {
    // (the queue pointers are separate __restrict__ arguments so that the compiler can prove the scene records are never
    // written in this kernel and fetch the wave-uniform ones with scalar loads)
    DWork<R> wk;
    wk.hdr = wk_hdr; wk.light = wk_light; wk.heavy = wk_heavy; wk.cap = wk_cap;
    Stats st;
    stats_zero(st);
    Rng rng;
    rng.k0 = rp.seed_lo; rng.k1 = rp.seed_hi;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int sub = rp.tile_rows >> 3; // 8-row rows per ownership tile
    for (int tile = blockIdx.x; tile < rp.n_tiles; tile += gridDim.x) {
        int ltr = tile / rp.tiles_x, tx = tile - ltr * rp.tiles_x;
        int lt = ltr / sub, within = ltr - lt * sub;
        int gt = lt * rp.nranks + rp.rank;
        const int x = tx * 8 + (lane & 7), y = (gt * sub + within) * 8 + (lane >> 3);
        const int ly = ltr * 8 + (lane >> 3);
        const bool valid = x < cam.hsize && y < cam.vsize;
        V3<R> sum = mk<R>(0, 0, 0), sos = mk<R>(0, 0, 0);
        uint32_t cnt = 0;
        const uint32_t first = cam.max_samples > 0 && cam.max_samples < cam.min_samples ? (uint32_t)cam.max_samples : (uint32_t)cam.min_samples;
        bool go = valid, queued = false;
        while (go) { // wave-uniform for the first `first` iterations; afterwards only lanes without a queue slot stay
            V3<R> c = take_sample<R, TREE>(sc, cam, rng, st, x, y, cnt);
            sum = sum + c; sos = sos + had(c, c);
            ++cnt;
            if (cnt < first) continue;
            bool more = need_more(cam, sum, sos, (R)cnt, cnt);
            if (more && cnt == first && wk.hdr) { // decided once per pixel, by all such lanes of the tile together
                // predicted samples still to take: the loop stops when variance / n <= SE^2
                const R inv_n = rcp_((R)cnt);
                const V3<R> mean = sum * inv_n, var = sos * inv_n - had(mean, mean);
                const R want = div_(var.x + var.y + var.z, (R)3 * (R)cam.se2) - (R)cnt;
                const bool heavy = want > (R)RTMI_LIGHT_NEED;
                const unsigned long long mh = __ballot(heavy), ml = __ballot(!heavy);
                const unsigned long long mine = heavy ? mh : ml;
                const int leader = __ffsll((long long)mine) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&wk.hdr[heavy ? WK_HEAVY_COUNT : WK_LIGHT_COUNT], (uint32_t)__popcll(mine));
                base = __shfl(base, leader, 64);
                const uint32_t slot = base + (uint32_t)__popcll(mine & lt_mask);
                if (slot < wk.cap) {
                    DWorkItem<R> *q = (heavy ? wk.heavy : wk.light) + slot;
                    q->px = (uint32_t)x; q->py = (uint32_t)y; q->out_row = (uint32_t)ly; q->cnt = cnt;
                    q->sum[0] = sum.x; q->sum[1] = sum.y; q->sum[2] = sum.z;
                    q->sos[0] = sos.x; q->sos[1] = sos.y; q->sos[2] = sos.z;
                    queued = true;
                    more = false;
                } // else: the queue is full and this lane finishes its pixel alone (slow, still exact)
            }
            go = more;
        }
        if (valid && !queued) {
            R inv = (R)1 / (R)cnt;
            size_t o = ((size_t)ly * cam.hsize + x) * 3;
            out[o + 0] = (float)(sum.x * inv); out[o + 1] = (float)(sum.y * inv); out[o + 2] = (float)(sum.z * inv);
        }
    }
    flush_stats(st, stats);
}

// Camera::multisample's open-ended loop (camera.rs:290-299) for the pixels render_kernel queued.  Waves pop work until
// both queues are empty: first the heavy queue, one pixel per wave (longest jobs first), then the light queue,
// RTMI_LIGHT_BATCH pixels per wave.  A group of G lanes computes a pixel's next G samples at once (counter RNG: sample s of a pixel
// does not depend on who computes it); a prefix sum over the group in sample order then finds the first sample count
// at which the reference's sequential loop would have stopped.  Samples past it are discarded, their rays not counted.
// Every wave reaches the exit: the pop counters only grow and the queue lengths are fixed before the launch.
template <typename R, bool TREE>
__global__ void __launch_bounds__(64, TREE ? RTMI_WAVES_TREE : RTMI_WAVES) resample_kernel(DScene<R> sc, DCamera<double> cam, DRender rp, float *__restrict__ out,
                                                      unsigned long long *__restrict__ stats, uint32_t *__restrict__ wk_hdr,
                                                      const DWorkItem<R> *__restrict__ wk_light, const DWorkItem<R> *__restrict__ wk_heavy,
                                                      uint32_t wk_cap) {
    struct { uint32_t *hdr; const DWorkItem<R> *light, *heavy; uint32_t cap; } wk = {wk_hdr, wk_light, wk_heavy, wk_cap};
    Stats st;
    stats_zero(st);
    Rng rng;
    rng.k0 = rp.seed_lo; rng.k1 = rp.seed_hi;
    const int lane = threadIdx.x & 63;
    const uint32_t n_heavy = min(wk.hdr[WK_HEAVY_COUNT], wk.cap), n_light = min(wk.hdr[WK_LIGHT_COUNT], wk.cap);
    bool heavy_left = n_heavy > 0;
    for (;;) {
        // ---- pop a batch ------------------------------------------------------------------------------------
        const DWorkItem<R> *items = nullptr;
        int P = 0;
        bool heavy_batch = false;
        if (heavy_left) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&wk.hdr[WK_HEAVY_HEAD], 1u);
            b = __shfl(b, 0, 64);
            if (b < n_heavy) { items = wk.heavy + b; P = 1; heavy_batch = true; }
            else heavy_left = false;
        }
        if (!P) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&wk.hdr[WK_LIGHT_HEAD], (uint32_t)RTMI_LIGHT_BATCH);
            b = __shfl(b, 0, 64);
            if (b >= n_light) break;
            items = wk.light + b;
            P = (int)min((uint32_t)RTMI_LIGHT_BATCH, n_light - b);
        }
        const bool valid = lane < P;
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
        // the order of the queue), so its sums - and with them the image - are reproducible bit for bit:
        // heavy pixels take 64 samples per round; light pixels RTMI_LIGHT_ROUNDS rounds of 64 / RTMI_LIGHT_BATCH, then 64
        // per round, one pixel after the other.
        for (int round = 0;; ++round) {
            const bool act = valid && need_more(cam, sum, sos, (R)cnt, cnt);
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
            stats_zero(ls);
            if (has && !(cam.max_samples > 0 && s >= (uint32_t)cam.max_samples)) c = take_sample<R, TREE>(sc, cam, rng, ls, ox, oy, s);
            // inclusive prefix sums over the group, in sample order (the association depends on j only)
            V3<R> ps = c, pq = had(c, c);
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                if (off < G) {
                    V3<R> a = {__shfl_up(ps.x, off, 64), __shfl_up(ps.y, off, 64), __shfl_up(ps.z, off, 64)};
                    V3<R> b = {__shfl_up(pq.x, off, 64), __shfl_up(pq.y, off, 64), __shfl_up(pq.z, off, 64)};
                    if (j >= off) { ps = ps + a; pq = pq + b; }
                }
            }
            const V3<R> sj = sum0 + ps, qj = sos0 + pq;
            const uint32_t cntj = cnt0 + (uint32_t)j + 1u;
            const bool stop = has && !need_more(cam, sj, qj, (R)cntj, cntj);
            const unsigned long long sb = __ballot(stop);
            const unsigned long long gb = wide ? sb : (sb >> (r * G)) & ((1ull << G) - 1ull);
            const int k_used = gb ? __ffsll((long long)gb) : G; // samples of this group that the reference would have taken
            if (has && j < k_used) stats_add(st, ls);
            else st.c[RTMI_STAT_DISCARDED] += ls.c[RTMI_STAT_SAMPLES];
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
        }
    }
    flush_stats(st, stats);
}

#endif // RTMI_HOST_EMUL

// World::color_at for arbitrary rays + the IntersectionState of the first hit (rtmi_trace_rays)
template <typename R>
__global__ void __launch_bounds__(64) trace_kernel(DScene<R> sc, uint32_t seed_lo, uint32_t seed_hi, long long n, const double *origins,
                                                   const double *dirs, const int *remaining, double *rgb, rtmi_hit_record *recs) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Stats st;
#pragma unroll
    for (int k = 0; k < RTMI_N_STATS; ++k) st.c[k] = 0;
    st.flops = 0;
    Rng rng;
    rng.k0 = seed_lo; rng.k1 = seed_hi;
    rng_begin(rng, (uint32_t)i, 0);
    Ray<R> ray = {mk<R>((R)origins[3 * i], (R)origins[3 * i + 1], (R)origins[3 * i + 2]),
                  mk<R>((R)dirs[3 * i], (R)dirs[3 * i + 1], (R)dirs[3 * i + 2])};
    Hit<R> h;
    Comps<R> c;
    c.point = mk<R>(0, 0, 0); c.normalv = c.point; c.eyev = c.point; c.inside = false; c.side_plus = false; c.n1 = 1; c.n2 = 1;
    int rem = remaining ? remaining[i] : (int)sc.max_depth;
    V3<R> col = color_at<R, true>(sc, ray, rem, rng, st, &h, &c);
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
                    if (in.kind == IL_RAY) r.shadowed = is_shadowed<R, true>(sc, in, over_point, h.leaf, c.side_plus ? SIDE_PLUS : 0u, st) ? 1.0 : 0.0;
                }
            }
        }
        recs[i] = r;
    }
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
                                     const DCamera<double> &cam, const DRender &rp, float *out, unsigned long long *stats, void *work,
                                     size_t work_bytes) {
    DWork<RTMI_REAL> wk = make_work<RTMI_REAL>(work, work_bytes);
    if (tree) hipLaunchKernelGGL((render_kernel<RTMI_REAL, true>), dim3(grid), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.cap);
    else hipLaunchKernelGGL((render_kernel<RTMI_REAL, false>), dim3(grid), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.cap);
    if (!wk.hdr || grid_resample <= 0) return;
    if (tree) hipLaunchKernelGGL((resample_kernel<RTMI_REAL, true>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.cap);
    else hipLaunchKernelGGL((resample_kernel<RTMI_REAL, false>), dim3(grid_resample), dim3(64), 0, st, sc, cam, rp, out, stats, wk.hdr, wk.light, wk.heavy, wk.cap);
}
#else
template <>
void launch_render_kernel<RTMI_REAL>(bool, int, int, hipStream_t, const DScene<RTMI_REAL> &, const DCamera<double> &, const DRender &, float *,
                                     unsigned long long *, void *, size_t) {}
#endif
template <>
void launch_trace_kernel<RTMI_REAL>(int grid, const DScene<RTMI_REAL> &sc, uint32_t seed_lo, uint32_t seed_hi, long long n,
                                    const double *origins, const double *dirs, const int *remaining, double *rgb, rtmi_hit_record *recs) {
    hipLaunchKernelGGL(trace_kernel<RTMI_REAL>, dim3(grid), dim3(64), 0, 0, sc, seed_lo, seed_hi, n, origins, dirs, remaining, rgb, recs);
}
template <>
void launch_camera_rays_kernel<RTMI_REAL>(int grid, const DCamera<double> &cam, uint32_t seed_lo, uint32_t seed_hi, long long n,
                                          const int *px, const int *py, const int *sample, double *origins, double *dirs) {
    hipLaunchKernelGGL(camera_rays_kernel<RTMI_REAL>, dim3(grid), dim3(256), 0, 0, cam, seed_lo, seed_hi, n, px, py, sample, origins, dirs);
}
} // namespace rtmi_dev
