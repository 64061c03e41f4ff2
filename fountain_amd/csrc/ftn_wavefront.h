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
struct WavefrontTimes { double trace_ms; unsigned long long trace_launches; };
int wavefront_render(WavefrontState** state, const RenderParams& P, const std::vector<DTile>& tiles, bool count, hipStream_t stream, WavefrontTimes* times);
void wavefront_destroy(WavefrontState* state);
const char* wavefront_error();
}  // namespace ftn
#endif
