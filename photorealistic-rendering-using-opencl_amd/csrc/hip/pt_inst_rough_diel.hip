// pt_inst_rough_diel.hip -- render_kernel compiled for the material set LIGHT|DIFF|DIEL|ROUGH_DIEL (config 3b), without a medium
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_rough_diel) {
    constexpr unsigned M = PRT_MAT_LIGHT | PRT_MAT_DIFF | PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL;
    if (medium) return launch_set_generic(true, sc, cam, S, fa, fb, stream, lo);      // (not compiled with a medium: no BASELINE config has both)
    return launch_variant<M, false>("render_kernel<LIGHT|DIFF|DIEL|ROUGH_DIEL>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
