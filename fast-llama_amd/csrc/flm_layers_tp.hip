// flm_layers_tp.hip -- host side of the RANK-SPANNING k_layers (flm_layer.h: k_layers<.., TP>; round 6): all layers of a SHARDED token in one launch per rank.
// A translation unit of its own: the instantiations compile beside flm_layers.hip's (same FLM_OPAQUE_TID build, see there).
#define FLM_OPAQUE_TID 1
#include "flm_host.h"

namespace fh {

// the TP instantiations: all-to-all hand-offs (R5 = 0), no tail; [QT][XR2 = 1 | 3][SPLIT][granules | flag rounds]
int launch_layers_tp(flm_ctx* c, hipStream_t st, const LayerArgs* LA, const BackArgs& p, int grid, int r2, int l0, int l1, int G) {
    {
        static std::mutex mu; static bool done[64] = {false};
        std::lock_guard<std::mutex> lk(mu);
        if (c->device >= 0 && c->device < 64 && !done[c->device]) {
            const void* fns[] = {(const void*)&k_layers<QT_INT8, 1, false, 0, false, true>, (const void*)&k_layers<QT_INT8, 3, false, 0, false, true>, (const void*)&k_layers<QT_INT16, 1, false, 0, false, true>, (const void*)&k_layers<QT_INT16, 3, false, 0, false, true>,
                                 (const void*)&k_layers<QT_INT8, 1, true, 0, false, true>, (const void*)&k_layers<QT_INT8, 3, true, 0, false, true>, (const void*)&k_layers<QT_INT16, 1, true, 0, false, true>, (const void*)&k_layers<QT_INT16, 3, true, 0, false, true>,
                                 (const void*)&k_layers<QT_INT8, 1, false, 0, false, true, true>, (const void*)&k_layers<QT_INT8, 3, false, 0, false, true, true>, (const void*)&k_layers<QT_INT16, 1, false, 0, false, true, true>, (const void*)&k_layers<QT_INT16, 3, false, 0, false, true, true>,
                                 (const void*)&k_layers<QT_INT8, 1, true, 0, false, true, true>, (const void*)&k_layers<QT_INT8, 3, true, 0, false, true, true>, (const void*)&k_layers<QT_INT16, 1, true, 0, false, true, true>, (const void*)&k_layers<QT_INT16, 3, true, 0, false, true, true>};
            for (const void* f : fns) HIPC(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
            done[c->device] = true;
        }
    }
    const dim3 g3(grid), b3(kGemvBlock);
    const bool i8 = c->d.quant_type == FLM_QT_INT8, one = r2 <= 1;
    // (granules or flag rounds: two instantiations -- both forms in one rank-spanning kernel spill)
#define FLM_LAUNCH_TP(QT, XR2, SP) do { if (p.gr) hipLaunchKernelGGL((k_layers<QT, XR2, SP, 0, false, true, true>), g3, b3, kLdsMax, st, LA, p, l0, l1, (const TailArgs*)nullptr); \
                                        else hipLaunchKernelGGL((k_layers<QT, XR2, SP, 0, false, true, false>), g3, b3, kLdsMax, st, LA, p, l0, l1, (const TailArgs*)nullptr); } while (0)
    if (G > 1) { if (i8) { if (one) FLM_LAUNCH_TP(QT_INT8, 1, true); else FLM_LAUNCH_TP(QT_INT8, 3, true); } else { if (one) FLM_LAUNCH_TP(QT_INT16, 1, true); else FLM_LAUNCH_TP(QT_INT16, 3, true); } }
    else       { if (i8) { if (one) FLM_LAUNCH_TP(QT_INT8, 1, false); else FLM_LAUNCH_TP(QT_INT8, 3, false); } else { if (one) FLM_LAUNCH_TP(QT_INT16, 1, false); else FLM_LAUNCH_TP(QT_INT16, 3, false); } }
#undef FLM_LAUNCH_TP
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

} // namespace fh
