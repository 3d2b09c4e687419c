// mlp_bf16.hip -- bf16-MFMA variant of the fused MLP (BASELINE config #5).  Not built yet in this
// revision: the entry points fail loudly (no silent fp32 substitution).
#include "common.h"
#include "layout.h"

namespace minerf {

size_t packed_bytes_bf16(const mi_nerf_net*) { return 0; }

int pack_bf16(const mi_nerf_net*, const mi_nerf_params*, void*, size_t) {
    set_error("bf16 MFMA variant is not implemented in this build");
    return MI_NERF_EINVAL;
}

int mlp_rays_bf16(const mi_nerf_net*, const void*, const float*, const float*, int64_t, int, float*, hipStream_t) {
    set_error("bf16 MFMA variant is not implemented in this build");
    return MI_NERF_EINVAL;
}

}  // namespace minerf
