/*
 * ftn_kernels.h -- launch parameter blocks shared by the host driver (ftn_host.cpp) and the gfx950 kernels
 * (ftn_kernels.hip, ftn_wavefront.hip).  Plain data only.
 */
#ifndef FTN_KERNELS_H
#define FTN_KERNELS_H

#include "ftn_device.h"

namespace ftn {

struct DevStats {
    unsigned long long rays_closest, rays_any, nodes_visited, prims_tested, camera_samples, spill_samples, nodes_any, prims_any;
    unsigned long long bc_writes;   /* samples added into the spill accumulators accB / accC (0: both are still all zero) */
    int error;          /* 0 or an ftn_status (NaN radiance, unsupported material) */
    int _pad;
    unsigned long long quad_records, quad_records_any;   /* four-box records fetched by the counting builds of k_wf_trace4 / k_wf_trace4_any */
    /* lane occupancy of the counting builds of k_wf_trace4 ([0..6]) / k_wf_trace4_any ([7..13]), wave-level tallies: control rounds, record steps
     * executed, lanes active in them, leaf steps, lanes active in them, refills, lanes re-armed (FTN_WF_DEBUG=1 prints them) */
    unsigned long long t4_occ[14];
};

struct DTile { int x0, y0, x1, y1; unsigned long long tile_id; uint32_t valid_off, _pad; };   /* sample-space tile, its sampler seed, exclusive prefix sum of pixel counts */

struct RenderParams {
    DScene S; DCamera C;
    int crop[4];                      /* cropped_pixel_bounds x0,y0,x1,y1 */
    float radius[2], inv_radius[2];   /* BoxFilter */
    uint32_t sampler_kind, spp, first_sample, last_sample;   /* render 0-based samples [first, last) */
    unsigned long long seed;
    uint32_t integrator_kind, max_depth; float rr_threshold;
    const DTile* tiles; uint32_t n_tiles;
    float4 *accA, *accB, *accC;       /* per crop pixel: rgb sum + weight: own samples / in-tile spill / cross-tile spill */
    DevStats* stats;
    uint32_t stack_entries;           /* LDS stack depth per lane */
};

/* kernels (defined in ftn_kernels.hip) -- host-callable launchers */
void launch_render_mega(const RenderParams& p, bool count, hipStream_t stream);
void launch_film_resolve(const RenderParams& p, ftn_pixel* device_pixels, hipStream_t stream);   /* reads p.stats->bc_writes on the device */
void launch_trace_batch(const DScene& S, const float* rays, size_t n, int mode /*0 closest,1 any,2 full*/, float* t_hit, int* prim,
                        float* bary, unsigned char* occluded, float* out24, DevStats* stats, uint32_t stack_entries, bool count, hipStream_t stream);

void launch_spectrum_buffer(const ftn_pixel* device_pixels, size_t n, float* device_rgb, hipStream_t stream);
void launch_test_texture_eval(const DScene& S, int texture, const float* uv_diffs6, size_t n, float* rgb_out, hipStream_t stream);
void launch_test_math(int which, const float* x, const float* y, size_t n, float* out, hipStream_t stream);

}  // namespace ftn
#endif
