/*
 * ftn_wavefront.h -- host interface of the wavefront pipeline (ftn_wavefront.hip): SoA path state and ray queues in HBM,
 * separate generate / trace / shade kernels per bounce (BASELINE configs 3-5).
 */
#ifndef FTN_WAVEFRONT_H
#define FTN_WAVEFRONT_H
#include "ftn_kernels.h"
#include <vector>

namespace ftn {
struct WavefrontState;
struct WavefrontTimes { double trace_ms; unsigned long long trace_launches;
                        double any_ms; unsigned long long any_launches; double shade_ms; unsigned long long shade_launches; double sort_ms;   /* the other kernel groups of a bounce (HIP events on their stream) */
                        unsigned long long mis_any_rays; /* Scene::intersect calls of estimate_direct answered by the any-hit kernel (infinite lights: only hit / miss matters) */ };
int wavefront_render(WavefrontState** state, const RenderParams& P, const std::vector<DTile>& tiles, bool count, hipStream_t stream, WavefrontTimes* times,
                     bool count_production = false);
/* Scene::intersect (mode 0: t / prim / barycentrics, mode 2: + full interaction in out24) and intersect_test (mode 1) for n rays of
 * 8 floats {o, d, t_max, time} in DEVICE memory; outputs are device pointers (any may be NULL) */
int wavefront_trace_batch(WavefrontState** state, const DScene& S, uint32_t stack_entries, const float* d_rays8, size_t n, int mode, bool count,
                          float* t_hit, int* prim, float* bary, unsigned char* occluded, float* out24, DevStats* stats, hipStream_t stream);
void wavefront_destroy(WavefrontState* state);
const char* wavefront_error();
}  // namespace ftn
#endif
