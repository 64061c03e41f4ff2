#include "ftn_wavefront.h"
namespace ftn {
struct WavefrontState { int unused; };
int wavefront_render(WavefrontState**, const RenderParams&, const std::vector<DTile>&, bool, hipStream_t, WavefrontTimes*) { return FTN_ERR_UNSUPPORTED; }
void wavefront_destroy(WavefrontState* s) { delete s; }
const char* wavefront_error() { return "wavefront pipeline not built yet"; }
}
