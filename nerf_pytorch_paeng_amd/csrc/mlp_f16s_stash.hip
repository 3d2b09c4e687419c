// mlp_f16s_stash.hip -- the TRAINING forward of the split-precision variant: mlp_f16s_kernel<STASH = true> (its own translation unit: the
// instantiation compiles for a minute).  Design: mlp_f16s.hip.
#include "mlp_f16s_core.h"

namespace minerf {

// training forward in split precision: the same outputs plus the activation stash of mlp_rays_fp32_stash (same tensors, same layouts:
// the backward pass does not know which forward ran)
int mlp_rays_f16s_stash(const mi_nerf_net* net, const void* packed_dev, const float* rays_dev, const float* z_dev, int64_t n_rays, int S,
                        float* raw_dev, float* stash_h, float* stash_f, float* stash_g, unsigned* mask_h, unsigned* mask_g, hipStream_t st) {
    MN_CHECK_ARG(stash_h && stash_f && stash_g && mask_h && mask_g, "NULL stash pointer");
    const StashF16s sp{stash_h, stash_f, stash_g, mask_h, mask_g};
    return launch_f16s<true>(net, packed_dev, rays_dev, z_dev, n_rays, S, raw_dev, &sp, st);
}


}  // namespace minerf
