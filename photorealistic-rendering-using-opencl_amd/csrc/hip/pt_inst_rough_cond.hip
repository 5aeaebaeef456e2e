// pt_inst_rough_cond.hip -- render_kernel compiled for the material set LIGHT|DIFF|ROUGH_COND (config 3a), without a medium
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_rough_cond) {
    constexpr unsigned M = PRT_MAT_LIGHT | PRT_MAT_DIFF | PRT_MAT_ROUGH_COND;
    if (medium) return launch_set_generic(true, sc, cam, S, fa, fb, stream, lo);      // (not compiled with a medium: no BASELINE config has both)
    // the set once more for scenes whose microfacet lobes are all GGX (the BASELINE config's): PT_MATS_DISTS, pt_device.h
    if (!lo.any_dist && sc.dist_mask == (unsigned)PRT_DIST_GGX)
        return launch_variant<M | ((unsigned)PRT_DIST_GGX << PT_MATS_DIST_SHIFT), false>("render_kernel<LIGHT|DIFF|ROUGH_COND; GGX>", sc, cam, S, fa, fb, stream, lo);
    return launch_variant<M, false>("render_kernel<LIGHT|DIFF|ROUGH_COND>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
