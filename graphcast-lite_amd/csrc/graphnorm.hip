// PyG LayerNorm(mode="graph") - statistics over all n*F elements of a sample.
// No BASELINE.json configuration uses graph mode (every config sets layer_norm_mode="node"), so the
// entry points are declared and report GCL_EUNSUPPORTED until the row in SURVEY.md §8f is reached.
#include "common.h"

extern "C" size_t gcl_graphnorm_ws_bytes(int32_t, int32_t, int32_t) { return 0; }

extern "C" int gcl_graphnorm_fwd(const float*, int64_t, int64_t, const float*, const float*, float, float*, int64_t,
                                 int64_t, float*, int32_t, int32_t, int32_t, void*, size_t, gcl_stream_t) {
  gcl::set_error("graphnorm_fwd: LayerNorm(mode=\"graph\") is not implemented on the HIP path yet");
  return GCL_EUNSUPPORTED;
}

extern "C" int gcl_graphnorm_bwd(const float*, int64_t, int64_t, const float*, int64_t, int64_t, const float*,
                                 const float*, float, float*, int64_t, int64_t, float*, float*, int32_t, int32_t,
                                 int32_t, int32_t, void*, size_t, gcl_stream_t) {
  gcl::set_error("graphnorm_bwd: LayerNorm(mode=\"graph\") is not implemented on the HIP path yet");
  return GCL_EUNSUPPORTED;
}
