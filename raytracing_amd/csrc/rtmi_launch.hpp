// rtmi_launch.hpp — the seam between the host translation unit (rtmi.hip) and the two builds of rtmi_kernels.hip.
#pragma once
#include "rtmi_device.hpp"

namespace rtmi_dev {
// Work queues of the adaptive second pass (resample_kernel).  `work` is WORK_HEADER_BYTES of counters followed by two
// equal arrays of DWorkItem<R>; the header must be zero when render_kernel starts.  work == nullptr: no second pass,
// every lane finishes its own pixel.
constexpr size_t WORK_HEADER_BYTES = 256;
enum { WK_LIGHT_COUNT = 0, WK_HEAVY_COUNT = 1, WK_LIGHT_HEAD = 2, WK_HEAVY_HEAD = 3 };
#ifndef RTMI_LIGHT_NEED
#define RTMI_LIGHT_NEED 192 // predicted further samples up to which a pixel goes to the light queue
#endif
#ifndef RTMI_LIGHT_BATCH
#define RTMI_LIGHT_BATCH 2  // light pixels per wave (a power of two): 32 lanes per pixel
#endif
#ifndef RTMI_LIGHT_ROUNDS
#define RTMI_LIGHT_ROUNDS 6 // rounds a light batch runs 32 lanes per pixel before its leftovers get the whole wave
// (measured on chapter-11a at 3840x2160: 2 / 192 / 6 renders the frame in 9.8 ms, 8 / 24 / 4 in 10.8 ms, everything heavy in
// 12.3 ms: rays of one pixel are coherent, rays of eight different edge pixels are not)
#endif
template <typename R> struct DWorkItem {
    uint32_t px, py, out_row, cnt; // global pixel, row in this rank's compact buffer, samples taken so far
    R sum[3], sos[3];              // running sum and sum of squares (camera.rs:277-279)
};
template <typename R> struct DWork {
    uint32_t *hdr;
    DWorkItem<R> *light, *heavy;
    uint32_t cap; // items per queue
};
template <typename R> inline DWork<R> make_work(void *work, size_t bytes) {
    DWork<R> w = {nullptr, nullptr, nullptr, 0};
    if (!work || bytes < WORK_HEADER_BYTES + 2 * sizeof(DWorkItem<R>)) return w;
    w.hdr = (uint32_t *)work;
    w.cap = (uint32_t)((bytes - WORK_HEADER_BYTES) / (2 * sizeof(DWorkItem<R>)));
    w.light = (DWorkItem<R> *)((char *)work + WORK_HEADER_BYTES);
    w.heavy = w.light + w.cap;
    return w;
}

template <typename R>
void launch_render_kernel(bool tree, int grid, int grid_resample, hipStream_t st, const DScene<R> &sc, const DCamera<double> &cam,
                          const DRender &rp, float *out, unsigned long long *stats, void *work, size_t work_bytes);
template <typename R>
void launch_trace_kernel(int grid, const DScene<R> &sc, uint32_t seed_lo, uint32_t seed_hi, long long n, const double *origins,
                         const double *dirs, const int *remaining, double *rgb, rtmi_hit_record *recs);
template <typename R>
void launch_camera_rays_kernel(int grid, const DCamera<double> &cam, uint32_t seed_lo, uint32_t seed_hi, long long n, const int *px,
                               const int *py, const int *sample, double *origins, double *dirs);
template <> void launch_render_kernel<float>(bool, int, int, hipStream_t, const DScene<float> &, const DCamera<double> &, const DRender &, float *, unsigned long long *, void *, size_t);
template <> void launch_render_kernel<double>(bool, int, int, hipStream_t, const DScene<double> &, const DCamera<double> &, const DRender &, float *, unsigned long long *, void *, size_t);
template <> void launch_trace_kernel<float>(int, const DScene<float> &, uint32_t, uint32_t, long long, const double *, const double *, const int *, double *, rtmi_hit_record *);
template <> void launch_trace_kernel<double>(int, const DScene<double> &, uint32_t, uint32_t, long long, const double *, const double *, const int *, double *, rtmi_hit_record *);
template <> void launch_camera_rays_kernel<float>(int, const DCamera<double> &, uint32_t, uint32_t, long long, const int *, const int *, const int *, double *, double *);
template <> void launch_camera_rays_kernel<double>(int, const DCamera<double> &, uint32_t, uint32_t, long long, const int *, const int *, const int *, double *, double *);
} // namespace rtmi_dev
