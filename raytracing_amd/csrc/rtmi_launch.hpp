// rtmi_launch.hpp — the seam between the host translation unit (rtmi.hip) and the two builds of rtmi_kernels.hip.
#pragma once
#include "rtmi_device.hpp"

namespace rtmi_dev {
template <typename R>
void launch_render_kernel(bool tree, int grid, hipStream_t st, const DScene<R> &sc, const DCamera<double> &cam, const DRender &rp,
                          float *out, unsigned long long *stats);
template <typename R>
void launch_trace_kernel(int grid, const DScene<R> &sc, uint32_t seed_lo, uint32_t seed_hi, long long n, const double *origins,
                         const double *dirs, const int *remaining, double *rgb, rtmi_hit_record *recs);
template <typename R>
void launch_camera_rays_kernel(int grid, const DCamera<double> &cam, uint32_t seed_lo, uint32_t seed_hi, long long n, const int *px,
                               const int *py, const int *sample, double *origins, double *dirs);
template <> void launch_render_kernel<float>(bool, int, hipStream_t, const DScene<float> &, const DCamera<double> &, const DRender &, float *, unsigned long long *);
template <> void launch_render_kernel<double>(bool, int, hipStream_t, const DScene<double> &, const DCamera<double> &, const DRender &, float *, unsigned long long *);
template <> void launch_trace_kernel<float>(int, const DScene<float> &, uint32_t, uint32_t, long long, const double *, const double *, const int *, double *, rtmi_hit_record *);
template <> void launch_trace_kernel<double>(int, const DScene<double> &, uint32_t, uint32_t, long long, const double *, const double *, const int *, double *, rtmi_hit_record *);
template <> void launch_camera_rays_kernel<float>(int, const DCamera<double> &, uint32_t, uint32_t, long long, const int *, const int *, const int *, double *, double *);
template <> void launch_camera_rays_kernel<double>(int, const DCamera<double> &, uint32_t, uint32_t, long long, const int *, const int *, const int *, double *, double *);
} // namespace rtmi_dev
