// pt_inst_view.hip -- render_kernel compiled for the material set generic,view (the debug views VIEW_NORMAL / VIEW_BVH_HIT), medium off / on
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_view) {
    constexpr unsigned M = PT_MATS_VIEW;
    if (medium) return launch_variant<M, true>("render_kernel<generic,view,medium>", sc, cam, S, fa, fb, stream, lo);
    return launch_variant<M, false>("render_kernel<generic,view>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
