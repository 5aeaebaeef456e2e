// pt_inst_view_sdf.hip -- render_kernel compiled for the material set generic,sdf,view (the debug views of H_SDF scenes), medium off / on
#include "pt_render.h"

namespace prt {

PT_DECLARE_SET(launch_set_view_sdf) {
    constexpr unsigned M = PT_MATS_VIEW | PT_MATS_SDF;
    if (medium) return launch_variant<M, true>("render_kernel<generic,sdf,view,medium>", sc, cam, S, fa, fb, stream, lo);
    return launch_variant<M, false>("render_kernel<generic,sdf,view>", sc, cam, S, fa, fb, stream, lo);
}

}  // namespace prt
