// rtmi_kernels.hip — the device kernels of the path, compiled once per arithmetic type (-DRTMI_REAL=float|double).
// The f64 build is compiled with -ffp-contract=off so that the validation mode rounds operation by operation like
// the reference's (and the oracle's) IEEE f64 arithmetic; the f32 product build lets the compiler fuse multiply-adds.
#include "rtmi_launch.hpp"

using namespace rtmi_dev;

#ifndef RTMI_REAL
#error "compile with -DRTMI_REAL=float or -DRTMI_REAL=double"
#endif
#ifndef RTMI_WAVES
#define RTMI_WAVES 3 // minimum waves per SIMD the render kernels are register-allocated for
#endif

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void flush_stats(const Stats &st, unsigned long long *stats) {
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

// Camera::render / trace_pixels / multisample (camera.rs:155-161,250-310): one wave64 per 8x8 pixel tile,
// waves stride over the tiles this rank owns.
//
// Adaptive multisampling without the divergence of "one lane loops until its pixel converges": every round the
// wave counts the pixels of its tile that still need samples (u).  While u > 32 each such pixel's own lane takes
// its next sample.  Once u <= 32 the 64 lanes are dealt out evenly, G = 64 / pow2ceil(u) lanes per unconverged
// pixel, and compute that pixel's next G samples (counter RNG: sample s of a pixel is independent of who computes
// it); a prefix sum over the group in sample order then finds the first sample count at which the reference's
// loop would have stopped, and samples past it are discarded - their rays are not counted either.
template <typename R, bool TREE>
__global__ void __launch_bounds__(64, RTMI_WAVES) render_kernel(DScene<R> sc, DCamera<double> cam, DRender rp, float *__restrict__ out,
                                                    unsigned long long *stats) {
    Stats st;
    stats_zero(st);
    Rng rng;
    rng.k0 = rp.seed_lo; rng.k1 = rp.seed_hi;
    const int lane = threadIdx.x & 63;
    const int sub = rp.tile_rows >> 3; // 8-row rows per ownership tile
    for (int tile = blockIdx.x; tile < rp.n_tiles; tile += gridDim.x) {
        int ltr = tile / rp.tiles_x, tx = tile - ltr * rp.tiles_x;
        int lt = ltr / sub, within = ltr - lt * sub;
        int gt = lt * rp.nranks + rp.rank;
        const int x0 = tx * 8, y0 = (gt * sub + within) * 8;
        const int x = x0 + (lane & 7), y = y0 + (lane >> 3);
        const int ly = ltr * 8 + (lane >> 3);
        const bool valid = x < cam.hsize && y < cam.vsize;
        V3<R> sum = mk<R>(0, 0, 0), sos = mk<R>(0, 0, 0);
        R n = 0;
        uint32_t cnt = 0;
        for (;;) {
            const bool act = valid && need_more(cam, sum, sos, n, cnt);
            const unsigned long long mask = __ballot(act);
            if (mask == 0) break;
            const int u = __popcll(mask);
            // lanes per pixel this round
            int p2 = 64;
            if (u <= 32) { p2 = 1; while (p2 < u) p2 <<= 1; }
            const int G = 64 / p2;
            const int r = lane / G, j = lane & (G - 1);
            int owner = act ? lane : -1;
            if (G > 1) { // the r-th unconverged pixel of the tile belongs to group r
                owner = -1;
                unsigned long long m = mask;
                for (int i = 0; i < p2 && m; ++i) {
                    int o = __ffsll((long long)m) - 1;
                    if (i == r) owner = o;
                    m &= m - 1;
                }
            }
            const bool has = owner >= 0;
            const int src = has ? owner : lane;
            const R n0 = __shfl(n, src, 64);
            const uint32_t cnt0 = __shfl(cnt, src, 64);
            const V3<R> sum0 = shfl3(sum, src), sos0 = shfl3(sos, src);
            const uint32_t s = cnt0 + (uint32_t)j;
            V3<R> c = mk<R>(0, 0, 0);
            Stats ls;
            stats_zero(ls);
            if (has && !(cam.max_samples > 0 && s >= (uint32_t)cam.max_samples)) {
                const int ox = x0 + (src & 7), oy = y0 + (src >> 3);
                rng_begin(rng, (uint32_t)(oy * cam.hsize + ox), s);
                ls.c[RTMI_STAT_SAMPLES] = 1;
                ls.flops = camera_flops(cam);
                Ray<R> ray = camera_ray<R>(cam, ox, oy, s != 0, rng);
                c = color_at<R, TREE>(sc, ray, (int)sc.max_depth, rng, ls, nullptr, nullptr);
            }
            // inclusive prefix sums over the group, in sample order
            V3<R> ps = c, pq = had(c, c);
            for (int off = 1; off < G; off <<= 1) {
                V3<R> a = {__shfl_up(ps.x, off, G), __shfl_up(ps.y, off, G), __shfl_up(ps.z, off, G)};
                V3<R> b = {__shfl_up(pq.x, off, G), __shfl_up(pq.y, off, G), __shfl_up(pq.z, off, G)};
                if (j >= off) { ps = ps + a; pq = pq + b; }
            }
            const V3<R> sj = sum0 + ps, qj = sos0 + pq;
            const R nj = n0 + (R)(j + 1);
            const uint32_t cntj = cnt0 + (uint32_t)j + 1u;
            const bool stop = has && !need_more(cam, sj, qj, nj, cntj);
            const unsigned long long sb = __ballot(stop);
            const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
            const unsigned long long gb = (sb >> (r * G)) & gmask;
            const int k_used = gb ? __ffsll((long long)gb) : G; // samples of this group that the reference would have taken
            if (has && j < k_used) stats_add(st, ls);
            // hand the result back to the pixel's own lane
            const int rk = G > 1 ? __popcll(mask & ((1ull << lane) - 1ull)) : lane;
            const int gbase = act ? rk * G : lane;
            const int kk = __shfl(k_used, gbase, 64);
            const int fin = act ? gbase + kk - 1 : lane;
            const V3<R> fs = shfl3(sj, fin), fq = shfl3(qj, fin);
            const R fn = __shfl(nj, fin, 64);
            if (act) { sum = fs; sos = fq; n = fn; cnt += (uint32_t)kk; }
        }
        if (valid) {
            R inv_n = (R)1 / n;
            size_t o = ((size_t)ly * cam.hsize + x) * 3;
            out[o + 0] = (float)(sum.x * inv_n);
            out[o + 1] = (float)(sum.y * inv_n);
            out[o + 2] = (float)(sum.z * inv_n);
        }
    }
    flush_stats(st, stats);
}

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
template <>
void launch_render_kernel<RTMI_REAL>(bool tree, int grid, hipStream_t st, const DScene<RTMI_REAL> &sc, const DCamera<double> &cam,
                                     const DRender &rp, float *out, unsigned long long *stats) {
    if (tree) hipLaunchKernelGGL((render_kernel<RTMI_REAL, true>), dim3(grid), dim3(64), 0, st, sc, cam, rp, out, stats);
    else hipLaunchKernelGGL((render_kernel<RTMI_REAL, false>), dim3(grid), dim3(64), 0, st, sc, cam, rp, out, stats);
}
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
