// rtmi_launch.hpp — the seam between the host translation unit (rtmi.hip) and the two builds of rtmi_kernels.hip.
#pragma once
#include "rtmi_device.hpp"

namespace rtmi_dev {
// Work memory of the adaptive second pass (resample_kernel, sum_up_kernel).  `work` is WORK_HEADER_BYTES of counters
// (zero when render_kernel starts) followed by
//   light[cap], heavy[cap]   queued pixels with their running sums, by predicted need (render_kernel fills both from
//                            the bottom; light pixels promoted later take heavy slots from the top)
//   final_idx[cap]           promoted pixels left to the sequential loop after the last pass
//   unit_pixel[2][unit_cap]  per pass (ping-pong): the heavy pixel each 64-sample unit belongs to
//   chunks[unit_cap]         the samples a unit took, one record per unit, consumed by the pass's sum_up_kernel
// work == nullptr: no second pass, every lane finishes its own pixel.
constexpr size_t WORK_HEADER_BYTES = 32768; // 64 counters, then the unit passes' pop counters, one per cache line
#ifndef RTMI_SPEC_PASSES
#define RTMI_SPEC_PASSES 3 // launches that take the promoted pixels' samples as units, before the sequential loop gets what is left (0..7)
#endif
#ifndef RTMI_SPEC_MIN_UNITS
#define RTMI_SPEC_MIN_UNITS 4 // a pixel in the sequential loop is promoted when it is predicted to need at least this many more rounds
#endif
#ifndef RTMI_SPEC_LOAD_DIV
#define RTMI_SPEC_LOAD_DIV 1 // ... and at least 1 / this of the jobs (heavy pixels + light batches) the pass has per wave
#endif
#ifndef RTMI_SPEC_MARGIN
#define RTMI_SPEC_MARGIN 1.0 // units are given for this multiple of the predicted need
#endif
#ifndef RTMI_SPEC_MAX
#define RTMI_SPEC_MAX 64 // most units a pixel is given in one pass
#endif
#ifndef RTMI_DRAW_VISITS
#define RTMI_DRAW_VISITS 1 // the first pass's waves draw their visits from a counter instead of striding over them
#endif
constexpr int WK_MAX_LAUNCHES = 9; // RTMI_SPEC_PASSES + 2
enum {
    WK_LIGHT_COUNT = 0, WK_HEAVY_COUNT = 1, WK_FINAL_COUNT = 2, WK_FINAL_HEAD = 3, WK_HEAVY_HEAD = 4, WK_PROMO_COUNT = 5, WK_LIGHT_HEAD = 6, WK_VISIT_HEAD = 7,
    WK_UNIT_COUNT = 8   // + pass (1 .. RTMI_SPEC_PASSES)
};
static_assert(RTMI_SPEC_PASSES >= 0 && RTMI_SPEC_PASSES <= 7, "header layout");
#ifndef RTMI_LIGHT_NEED
#define RTMI_LIGHT_NEED 128 // predicted further samples up to which a pixel goes to the light queue
#endif
#ifndef RTMI_LIGHT_BATCH
#define RTMI_LIGHT_BATCH 4  // light pixels per wave (a power of two): 16 lanes per pixel (round 2, with most pixels past min_samples: 4 beats 2 by 2 - 3 %)
#endif
#ifndef RTMI_LIGHT_POP
#define RTMI_LIGHT_POP 1 // most light batches a wave pops at once (neighbours in the queue are neighbours in the image and cost
// alike: 8 at once left the waves that met the glass spheres 0.7 ms behind the others on the headline frame)
#endif
#ifndef RTMI_LIGHT_ROUNDS
#define RTMI_LIGHT_ROUNDS 4 // rounds a light batch runs 32 lanes per pixel before its leftovers get the whole wave (or are promoted)
// (measured on chapter-11a at 3840x2160.  Before promotion existed: 2 / 192 / 6 rendered the frame in 9.8 ms, 8 / 24 / 4 in
// 10.8 ms, everything heavy in 12.3 ms: rays of one pixel are coherent, rays of eight different edge pixels are not.  With it,
// one call alone / 3 frames in flight, whole frame and 1/8 shard: 2 / 192 / 6 10.17 / 8.19 and 2.94 / 1.37 ms, 2 / 128 / 4 10.04 /
// 8.16 and 2.43 / 1.28 ms, 2 / 256 / 8 10.2 / 8.19 and 3.04 / 1.40 ms; 96 / 3, 192 / 3, 128 / 2, 64 / 2, 64 / 4 like 128 / 4: a light
// batch is the longest job of its launch, and long jobs are what a small frame cannot afford.  4 pixels per wave with
// need <= 48 and 3 narrow rounds: 10.2 / 7.99 and 2.48 / 1.26 ms - less work in all (fewer discarded samples), a longer end of the
// launch when a call runs alone - and a loss on the other scenes (as shipped, 3 frames in flight, 2 / 128 / 4 -> 4 / 48 / 3:
// chapter-07 2.29 -> 2.61 ms, chapter-11b 3.63 -> 3.93 ms, teapot 16.3 -> 16.6 ms against chapter-11a 8.33 -> 8.20 ms).  A third queue (tiny: 4 per wave, need <= 48, 3 narrow rounds, beside light and heavy as they
// are) was built and measured too: 10.4 - 10.6 / 8.00 - 8.03 and 2.48 / 1.27 ms in either pop order - the same trade, not both gains)
#endif
template <typename R> struct DWorkItem {
    uint32_t px, py, out_row, cnt; // global pixel, row in this rank's compact buffer, samples taken so far
    R sum[3], sos[3];              // running sum and sum of squares (camera.rs:277-279)
    uint32_t unit_base, n_chunks;  // heavy pixels: this pass's units are [unit_base, unit_base + n_chunks)
};
// what one unit (64 consecutive samples of one pixel, one per lane) leaves for the wave that sums the pixel up
constexpr int CHUNK_STAT_WORDS = 8;
template <typename R> struct DChunk {
    R col[3][64];
    uint32_t st[CHUNK_STAT_WORDS][64]; // primary | samples << 8 | overflow << 16, reflection, refraction, shadow, leaf tests, node visits, slab tests, flops
};
// What a pixel's sampling has come to, kept between the passes of a progressive render (rtmi_progressive_*): the next
// pass goes on from these sums exactly where Camera::multisample's loop (camera.rs:281-299) would.
template <typename R> struct DPixelState {
    R sum[3], sos[3];
    uint32_t cnt, pad;
};
template <typename R> struct DWork {
    uint32_t *hdr;
    DWorkItem<R> *light, *heavy;
    uint32_t *final_idx, *unit_pixel;
    DChunk<R> *chunks;
    uint32_t cap;      // items per queue
    uint32_t unit_cap; // units per pass
};
template <typename R> inline DWork<R> make_work(void *work, size_t bytes) {
    DWork<R> w = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (!work || bytes < (1u << 20)) return w;
    // half of the memory for the two queues, the rest (less final_idx) for the units
    size_t cap = bytes / 2 / (2 * sizeof(DWorkItem<R>));
    if (cap > 0x7FFFFFFFu) cap = 0x7FFFFFFFu;
    char *p = (char *)work;
    w.hdr = (uint32_t *)p; p += WORK_HEADER_BYTES;
    w.light = (DWorkItem<R> *)p; p += cap * sizeof(DWorkItem<R>);
    w.heavy = (DWorkItem<R> *)p; p += cap * sizeof(DWorkItem<R>);
    w.final_idx = (uint32_t *)p; p += cap * sizeof(uint32_t);
    p = (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    size_t left = (size_t)((char *)work + bytes - p);
    size_t unit_cap = left / (sizeof(DChunk<R>) + 2 * sizeof(uint32_t) + 1); // (+1: room for the alignment below)
    if (unit_cap > 0x3FFFFFFFu) unit_cap = 0x3FFFFFFFu;
    w.chunks = (DChunk<R> *)p; p += unit_cap * sizeof(DChunk<R>);
    w.unit_pixel = (uint32_t *)p;
    w.cap = (uint32_t)cap; w.unit_cap = (uint32_t)unit_cap;
    return w;
}

template <typename R>
void launch_render_kernel(bool tree, int grid, int grid_resample, hipStream_t st, const DScene<R> &sc, const DCamera<double> &cam,
                          const DRender &rp, float *out, unsigned long long *stats, void *work, size_t work_bytes, hipEvent_t between, void *state);
template <typename R>
void launch_trace_kernel(int grid, const DScene<R> &sc, uint32_t seed_lo, uint32_t seed_hi, long long n, const double *origins,
                         const double *dirs, const int *remaining, double *rgb, rtmi_hit_record *recs);
template <typename R>
void launch_camera_rays_kernel(int grid, const DCamera<double> &cam, uint32_t seed_lo, uint32_t seed_hi, long long n, const int *px,
                               const int *py, const int *sample, double *origins, double *dirs);
template <typename R>
void launch_photon_trace_kernel(bool store, int grid, hipStream_t st, const DScene<R> &sc, const DPhotonTrace &pt, rtmi_photon *out, uint32_t *counts,
                                const uint32_t *offsets, unsigned long long *counters);
template <> void launch_photon_trace_kernel<float>(bool, int, hipStream_t, const DScene<float> &, const DPhotonTrace &, rtmi_photon *, uint32_t *, const uint32_t *, unsigned long long *);
template <> void launch_photon_trace_kernel<double>(bool, int, hipStream_t, const DScene<double> &, const DPhotonTrace &, rtmi_photon *, uint32_t *, const uint32_t *, unsigned long long *);
template <> void launch_render_kernel<float>(bool, int, int, hipStream_t, const DScene<float> &, const DCamera<double> &, const DRender &, float *, unsigned long long *, void *, size_t, hipEvent_t, void *);
template <> void launch_render_kernel<double>(bool, int, int, hipStream_t, const DScene<double> &, const DCamera<double> &, const DRender &, float *, unsigned long long *, void *, size_t, hipEvent_t, void *);
template <> void launch_trace_kernel<float>(int, const DScene<float> &, uint32_t, uint32_t, long long, const double *, const double *, const int *, double *, rtmi_hit_record *);
template <> void launch_trace_kernel<double>(int, const DScene<double> &, uint32_t, uint32_t, long long, const double *, const double *, const int *, double *, rtmi_hit_record *);
template <> void launch_camera_rays_kernel<float>(int, const DCamera<double> &, uint32_t, uint32_t, long long, const int *, const int *, const int *, double *, double *);
template <> void launch_camera_rays_kernel<double>(int, const DCamera<double> &, uint32_t, uint32_t, long long, const int *, const int *, const int *, double *, double *);
} // namespace rtmi_dev
