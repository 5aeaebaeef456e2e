// pt_inst_light_diff.hip -- render_kernel compiled for the material set LIGHT|DIFF (configs 2, 4, 5: Lambert + light), medium off / on
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_light_diff) {
    constexpr unsigned M = PRT_MAT_LIGHT | PRT_MAT_DIFF;
    if (medium) return launch_variant<M, true>("render_kernel<LIGHT|DIFF,medium>", sc, cam, S, fa, fb, stream, lo);
    return launch_variant<M, false>("render_kernel<LIGHT|DIFF>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
