// pt_inst_coat.hip -- render_kernel compiled for the material set LIGHT|DIFF|COAT (scenes/cornell.json as shipped (config 1)), without a medium
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_coat) {
    constexpr unsigned M = PRT_MAT_LIGHT | PRT_MAT_DIFF | PRT_MAT_COAT;
    if (medium) return launch_set_generic(true, sc, cam, S, fa, fb, stream, lo);      // (not compiled with a medium: no BASELINE config has both)
    // the set once more for scenes whose microfacet lobes are all Beckmann (the BASELINE config's): PT_MATS_DISTS, pt_device.h
    if (!lo.any_dist && sc.dist_mask == (unsigned)PRT_DIST_BECKMANN)
        return launch_variant<M | ((unsigned)PRT_DIST_BECKMANN << PT_MATS_DIST_SHIFT), false>("render_kernel<LIGHT|DIFF|COAT; Beckmann>", sc, cam, S, fa, fb, stream, lo);
    return launch_variant<M, false>("render_kernel<LIGHT|DIFF|COAT>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
