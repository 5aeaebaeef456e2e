// pt_inst_pick.hip -- render_kernel with PICK_RANDOM_LIGHT (kernels/integrators/base.cl:9; prt_config::pick_random_light): the
// run-time-dispatched material set, medium off / on
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_pick) {
    constexpr unsigned M = PT_MATS_PICK;
    if (medium) return launch_variant<M, true>("render_kernel<generic,pick_random_light,medium>", sc, cam, S, fa, fb, stream, lo);
    return launch_variant<M, false>("render_kernel<generic,pick_random_light>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
