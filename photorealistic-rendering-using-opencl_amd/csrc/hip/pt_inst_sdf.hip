// pt_inst_sdf.hip -- render_kernel compiled for the material set generic,sdf (H_SDF scenes: the generic set with the raymarcher), medium off / on
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_sdf) {
    constexpr unsigned M = PT_MATS_SDF;
    if (medium) return launch_variant<M, true>("render_kernel<generic,sdf,medium>", sc, cam, S, fa, fb, stream, lo);
    return launch_variant<M, false>("render_kernel<generic,sdf>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
