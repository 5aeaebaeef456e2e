// pt_inst_generic.hip -- render_kernel compiled for the material set generic (any other ACTIVE_MATS: the materials dispatched at run time), medium off / on
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_generic) {
    constexpr unsigned M = 0u;
    if (medium) return launch_variant<M, true>("render_kernel<generic,medium>", sc, cam, S, fa, fb, stream, lo);
    return launch_variant<M, false>("render_kernel<generic>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
