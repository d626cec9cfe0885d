// flm_layers.hip -- host side of k_layers (flm_layer.h): ALL layers of a token in one launch.  A translation unit of its own because the kernel is compiled with
// FLM_OPAQUE_TID (flm_math.h: reads of threadIdx.x are opaque, so that nothing derived from it is hoisted out of the loop over layers and kept in registers).
#define FLM_OPAQUE_TID 1
#include "flm_host.h"

namespace fh {

template <int QT> int plan_layer(flm_ctx* c, int l, bool with_qkv, int G, LayerArgs& A, BackArgs& p, int& grid, int& r2, TailArgs* tail);     // flm_layerlaunch.hip
int launch_layers_tp(flm_ctx* c, hipStream_t st, const LayerArgs* LA, const BackArgs& p, int grid, int r2, int l0, int l1, int G);           // flm_layers_tp.hip

// the layers' argument blocks in device memory, per number of workgroups a head is spread over (G = 1 | hs / 32): built outside any stream capture, valid until an option changes
int layers_prepare(flm_ctx* c, int G) {
    const int key = G > 1 ? 1 : 0;
    if (!c->fuse_token || (c->world != 1 && !(c->p2p && c->grp_tpl))) return FLM_OK;            // (tensor parallel: the rank-spanning form, where the group agreed on it)
    if (c->la_valid[key]) return FLM_OK;
    const int L = c->d.n_layers, qt = c->d.quant_type;
    std::vector<LayerArgs> host((size_t)L);
    c->la_ok[key] = false; c->tail_ok[key] = false;
    const std::string err0 = c->err, gerr0 = g_last_error;
    TailArgs tail{};
    for (int l = 0; l < L; ++l) {
        BackArgs p; int grid = 0, r2 = 0;
        const int r = qt == FLM_QT_INT8 ? plan_layer<QT_INT8>(c, l, true, G, host[l], p, grid, r2, &tail) : plan_layer<QT_INT16>(c, l, true, G, host[l], p, grid, r2, &tail);
        if (r == FLM_ERR_UNSUPPORTED) { c->la_valid[key] = true; c->err = err0; g_last_error = gerr0; return FLM_OK; }      // (this shape runs one launch per layer, or per phase: a probe, not a failure -- flm_last_error keeps what it said)
        if (r) return r;
        c->la_p[key] = p; c->la_grid[key] = grid; c->la_r2[key] = r2;
    }
    HIPC(c, hipMemcpyAsync(c->la_dev[key], host.data(), sizeof(LayerArgs) * (size_t)L, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));                                           // (host goes out of scope)
    if (tail.gridc > 0) {   // the one-launch token's argument block (every layer planned with the classifier's layout below the stash: the same st_base throughout)
        HIPC(c, hipMemcpyAsync(c->tail_dev[key], &tail, sizeof(TailArgs), hipMemcpyHostToDevice, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        c->tail_ok[key] = true;
    }
    c->la_ok[key] = true; c->la_valid[key] = true;
    return FLM_OK;
}

// layers [l0, l1) of the token in one launch; FLM_ERR_UNSUPPORTED: not prepared / not possible for this shape
int launch_layers(flm_ctx* c, hipStream_t st, int l0, int l1, int G, bool tail) {
    const int key = G > 1 ? 1 : 0;
    if (!c->fuse_token || !c->la_valid[key] || !c->la_ok[key] || l1 <= l0) return FLM_ERR_UNSUPPORTED;
    if (tail && (!c->tail_ok[key] || l0 != 0 || l1 != c->d.n_layers)) return FLM_ERR_UNSUPPORTED;
    if (c->world > 1) {
        if (tail || !c->p2p || !c->grp_tpl || !c->la_p[key].tp.world) return FLM_ERR_UNSUPPORTED;
        return launch_layers_tp(c, st, (const LayerArgs*)c->la_dev[key], c->la_p[key], c->la_grid[key], c->la_r2[key], l0, l1, G);
    }
    {
        static std::mutex mu; static bool done[64] = {false};
        std::lock_guard<std::mutex> lk(mu);
        if (c->device >= 0 && c->device < 64 && !done[c->device]) {
            const void* fns[] = {(const void*)&k_layers<QT_INT8, 1, false>, (const void*)&k_layers<QT_INT8, 3, false>, (const void*)&k_layers<QT_INT16, 1, false>, (const void*)&k_layers<QT_INT16, 3, false>,
                                 (const void*)&k_layers<QT_INT8, 1, true>, (const void*)&k_layers<QT_INT8, 3, true>, (const void*)&k_layers<QT_INT16, 1, true>, (const void*)&k_layers<QT_INT16, 3, true>,
                                 (const void*)&k_layers<QT_INT8, 1, false, 3>, (const void*)&k_layers<QT_INT8, 3, false, 3>, (const void*)&k_layers<QT_INT16, 1, false, 3>, (const void*)&k_layers<QT_INT16, 3, false, 3>,
                                 (const void*)&k_layers<QT_INT8, 1, true, 3>, (const void*)&k_layers<QT_INT8, 3, true, 3>, (const void*)&k_layers<QT_INT16, 1, true, 3>, (const void*)&k_layers<QT_INT16, 3, true, 3>,
                                 (const void*)&k_layers<QT_INT8, 1, false, 3, true>, (const void*)&k_layers<QT_INT8, 3, false, 3, true>, (const void*)&k_layers<QT_INT16, 1, false, 3, true>, (const void*)&k_layers<QT_INT16, 3, false, 3, true>,
                                 (const void*)&k_layers<QT_INT8, 1, true, 3, true>, (const void*)&k_layers<QT_INT8, 3, true, 3, true>, (const void*)&k_layers<QT_INT16, 1, true, 3, true>, (const void*)&k_layers<QT_INT16, 3, true, 3, true>,
                                 (const void*)&k_layers<QT_INT8, 1, false, 0, true>, (const void*)&k_layers<QT_INT8, 3, false, 0, true>, (const void*)&k_layers<QT_INT16, 1, false, 0, true>, (const void*)&k_layers<QT_INT16, 3, false, 0, true>,
                                 (const void*)&k_layers<QT_INT8, 1, true, 0, true>, (const void*)&k_layers<QT_INT8, 3, true, 0, true>, (const void*)&k_layers<QT_INT16, 1, true, 0, true>, (const void*)&k_layers<QT_INT16, 3, true, 0, true>};
            for (const void* f : fns) HIPC(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
            done[c->device] = true;
        }
    }
    const dim3 g3(c->la_grid[key]), b3(kGemvBlock);
    const LayerArgs* LA = (const LayerArgs*)c->la_dev[key];
    BackArgs p = c->la_p[key];
    if (!tail) { p.gr = 0; if (c->back_pre13 == 99) p.pre13 = 16; if (c->tok_preq == 99) p.preq = 16; }                                                                // (granule tags count from the tail launch's epoch: flm_layer.h BackArgs::gr)
    const bool i8 = c->d.quant_type == FLM_QT_INT8, one = c->la_r2[key] <= 1;
    const TailArgs* TA = (const TailArgs*)c->tail_dev[key];
#define FLM_LAUNCH_LAYERS(QT, XR2, SP) do { if (tail && p.r5) hipLaunchKernelGGL((k_layers<QT, XR2, SP, 3, true>), g3, b3, kLdsMax, st, LA, p, l0, l1, TA); \
                                            else if (tail) hipLaunchKernelGGL((k_layers<QT, XR2, SP, 0, true>), g3, b3, kLdsMax, st, LA, p, l0, l1, TA); \
                                            else if (p.r5) hipLaunchKernelGGL((k_layers<QT, XR2, SP, 3>), g3, b3, kLdsMax, st, LA, p, l0, l1, (const TailArgs*)nullptr); \
                                            else hipLaunchKernelGGL((k_layers<QT, XR2, SP, 0>), g3, b3, kLdsMax, st, LA, p, l0, l1, (const TailArgs*)nullptr); } while (0)
    if (G > 1) { if (i8) { if (one) FLM_LAUNCH_LAYERS(QT_INT8, 1, true); else FLM_LAUNCH_LAYERS(QT_INT8, 3, true); } else { if (one) FLM_LAUNCH_LAYERS(QT_INT16, 1, true); else FLM_LAUNCH_LAYERS(QT_INT16, 3, true); } }
    else       { if (i8) { if (one) FLM_LAUNCH_LAYERS(QT_INT8, 1, false); else FLM_LAUNCH_LAYERS(QT_INT8, 3, false); } else { if (one) FLM_LAUNCH_LAYERS(QT_INT16, 1, false); else FLM_LAUNCH_LAYERS(QT_INT16, 3, false); } }
#undef FLM_LAUNCH_LAYERS
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

} // namespace fh
