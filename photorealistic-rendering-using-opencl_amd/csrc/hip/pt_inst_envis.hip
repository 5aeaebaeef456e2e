// pt_inst_envis.hip -- render_kernel with environment-map importance sampling (prt_config::env_importance_sampling; not in the
// reference): the run-time-dispatched material set, surfaces only
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_envis) {
    (void)medium;                                        // pack_scene refuses the combination
    return launch_variant<PT_MATS_ENVIS, false>("render_kernel<generic,env_importance_sampling>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
