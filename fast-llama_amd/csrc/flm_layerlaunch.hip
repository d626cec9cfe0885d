// flm_layerlaunch.hip -- host side of k_attn_ffn (flm_layer.h): the whole decoder layer, or attention .. FFN2, as one launch.
#include "flm_host.h"

namespace fh {

// attention + Wo + FFN13 + FFN2 of layer l in one launch (k_attn_ffn, flm_layer.h); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
// the argument blocks, the launch geometry and the stash sizes of layer l (the same for every layer but for the pointers): shared by k_attn_ffn's launch and k_layers' (flm_layers.hip)
template <int QT>
int plan_layer(flm_ctx* c, int l, bool with_qkv, int G, LayerArgs& A, BackArgs& p, int& grid, int& r2, TailArgs* tail) {     // tail (k_layers' one-launch token): in: non-null = wanted; out: tail->gridc > 0 = possible, filled
    const auto& d = c->d;
    constexpr int esz = QTraits<QT>::kEsz;
    const int all = c->cu_count < 256 ? c->cu_count : 256, parts = c->heads_local * G, wgs_o = all - parts;
    // tensor parallel (round 6): the rank-spanning form (k_layers<.., TP>) where the group agreed on it (flm_p2p_import: grp_tpl); the ranks' geometry is identical
    const bool tpl = c->world > 1 && c->p2p && c->grp_tpl && with_qkv && c->x_tlines_off != 0;
    if ((c->world != 1 && !tpl) || (G == 1 && !tpl && c->hs % kGroup != 0) || wgs_o < 8 || parts > 256) return FLM_ERR_UNSUPPORTED;
    if (tpl && (d.n_heads * G > kTpLinesPerRank || all > kTpLinesPerRank)) return FLM_ERR_UNSUPPORTED;
    if (G > 1 && !with_qkv) return FLM_ERR_UNSUPPORTED;                     // (split heads: only the whole-layer form is instantiated)
    GemvArgs aq = args_qkv(c, l), ao = args_o(c, l), a13 = args_ffn13(c, l), a2 = args_ffn2(c, l);
    GemvPlan Pq{}, Po, P13, P2;
    int r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, ao, wgs_o, Po); if (r) return r;
    if (with_qkv) {
        r = plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_ROPE_KV>(c, aq, all, Pq); if (r) return r;
        if ((aq.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4) > 1) return FLM_ERR_UNSUPPORTED;
    }
    r = plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_SWIGLU>(c, a13, all, P13); if (r) return r;
    r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, a2, all, P2); if (r) return r;
    const int r13 = (a13.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4); r2 = (a2.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (r13 > 1 || r2 > 3) return FLM_ERR_UNSUPPORTED;
    if ((G > 1 || tpl) && (ao.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4) > 1) return FLM_ERR_UNSUPPORTED;     // (the Wo workgroups quantize the heads' fp32 output themselves: one round)
    // a workgroup with a single pass needs one strip buffer: the LDS above the phases' own layouts is the stash
    auto one_pass = [&](GemvArgs& a, GemvPlan& P, bool two, bool norm, int rows_per_item = 1) {
        const int rows = a.items * rows_per_item, npass = (rows + P.Rm - 1) / P.Rm;
        if (npass <= P.grid) { a.nbuf = 1; P.nbuf = 1; P.lds = (size_t)gemv_lds_layout(a.n, esz, norm, P.Rm, 64 >> P.cb_shift, two, 1).total; }
    };
    one_pass(ao, Po, false, false); one_pass(a13, P13, true, true); one_pass(a2, P2, false, false);
    size_t own = Po.lds; if (P13.lds > own) own = P13.lds; if (P2.lds > own) own = P2.lds;
    if (with_qkv) { one_pass(aq, Pq, false, true, 2); if (Pq.lds > kLdsMax) return FLM_ERR_UNSUPPORTED; if (c->fuse_token && Pq.lds > own) own = Pq.lds; }   // (k_layers stashes [Wq; Wk; Wv] too: above its layout as well)
    // the one-launch token runs the classifier as a phase of the same launch: its layout must fit below the stash as well
    GemvArgs acls{}; GemvPlan Pc{}; bool tail_fits = false;
    if (tail) {
        tail->gridc = 0;
        // (round 6: the one-launch token IS the granule form of the hand-offs -- flm_layer.h GRM --: one sweep round per thread for dim, and its one-workgroup-per-head instantiation
        //  takes q and the new K / V row as granules with the head's <= 2 tiles requested at once: long contexts must be able to split.  "gr_edges" 0: the four-launch token on flag rounds)
        const bool gr_tail = c->gr_edges && c->xg && d.dim <= 4 * kGemvBlock && d.dim % 4 == 0 && d.hidden_dim % 4 == 0 && (d.max_seq_len <= kSplitFrom || attn_parts(c, d.max_seq_len) > 1);
        if (gr_tail && c->fuse_tail && with_qkv && c->fuse_token && c->world == 1 && !tpl && c->got_emb && c->emb_qt == 0 && c->got_cls) {
            acls = args_cls(c);
            if (plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_STORE>(c, acls, all, Pc) == FLM_OK && (acls.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4) <= 1 && Pc.grid <= 256 && Pc.lds + 8 * (kStepBlk * 1024 + 256) <= kLdsMax) {
                tail_fits = true; if (Pc.lds > own) own = Pc.lds;
            }
        }
    }
    own = (own + 255) & ~(size_t)255;
    const size_t lds_attn = attn_lds_bytes(d.max_seq_len, c->hs, G > 1);
    if (own > kLdsMax || lds_attn > kLdsMax) return FLM_ERR_UNSUPPORTED;
    const int slot = kStepBlk * 1024 + 256, fit = (int)((kLdsMax - own) / slot);
    auto slots = [&](int want) { int n = want < 0 ? fit : want; if (n > fit) n = fit; if (n > 32) n = 32; return n < 0 ? 0 : n; };
    AttnArgs aa = args_attn(c, l, G);
    if (G == 1 && !tpl) { aa.oq = c->att_q; aa.os = c->att_qs; aa.oqt = QT; ao.xq = c->att_q; ao.xs = c->att_qs; }   // the heads hand their output over quantized
    p = BackArgs{};
    p.n_heads = parts; p.grido = Po.grid; p.grid13 = P13.grid; p.grid2 = P2.grid;
    p.flag_h = c->flag_lines; p.flag_hd = c->flag_lines + 512 * 16; p.flag_x = c->flag_lines + 1024 * 16;
    p.gridq = with_qkv ? Pq.grid : 0; p.flag_q = c->flag_lines + 768 * 16;
    p.target = (unsigned)(l + 1); p.err = c->xwg_err;
    p.st_base = (unsigned)own; p.nst13 = slots(c->back_nst13); p.nst13_head = slots(c->back_nst13_head); p.nst2 = slots(c->back_nst2); p.pre13 = c->back_pre13 == 99 ? 16 : c->back_pre13 < 0 ? 0 : c->back_pre13 > 16 ? 16 : c->back_pre13; p.pre2 = c->back_pre2 < 0 ? 0 : c->back_pre2 > 16 ? 16 : c->back_pre2;
    {   // arrival-order hand-offs (GemvCtx::run_ao): one pass per workgroup, every step resident, a column block with <= 16 producers (PRO_QUANT: <= 256 elements)
        auto steps = [&](const GemvArgs& a, const GemvPlan& P, int rows) { const int RB = 64 >> P.cb_shift, RBP = P.Rm / RB, nbc = (a.n * esz / 16) >> P.cb_shift; return (rows + P.Rm - 1) / P.Rm <= P.grid ? ((RBP + kStepBlk - 1) / kStepBlk) * nbc : 1 << 30; };
        const int epb_o = (16 << Po.cb_shift) / esz, epb_2 = (16 << P2.cb_shift) / esz;
        const int ns_o = steps(ao, Po, ao.items), ns_2 = steps(a2, P2, a2.items);
        p.ao_o = (c->back_ao & 1) && G == 1 && ns_o <= 32 && ao.n < 65536 && epb_o / c->hs + 2 <= 16 ? 1 : 0;
        const int want2 = ns_2 > 32 ? ns_2 - 32 : 0;
        const bool ok2 = (c->back_ao & 2) && want2 <= fit && want2 <= 32 && a2.n < 65536 && epb_2 <= 256 && epb_2 / P13.Rm + 2 <= 16;
        p.ao_2 = ok2 ? (c->back_ao2 >= 1 && c->back_ao2 <= 3 ? c->back_ao2 : 2) : 0;
        p.nst2_ao = ok2 ? want2 : 0;
        // one instantiation carries both arrival-order forms (k_layers<.., R5 = 3>) or none: Wo where a head is one workgroup (with split heads Wo stays as it was), FFN2
        p.r5 = ((G > 1 || p.ao_o) && p.ao_2 && with_qkv && !tpl) ? 3 : 0;
        if (!p.r5) { p.ao_o = 0; p.ao_2 = 0; p.nst2_ao = 0; }
        // arrival-order Wo: its ns_o steps on the first ceil(ns_o / 2) waves (two register sets each), the [W1; W3] stash issued by the others (flm_layer.h BackArgs::nw_o); "back_nwo" 16: as before
        p.nw_o = kWavesPerBlock;
        if (p.ao_o && G == 1) { const int need = (ns_o + 1) / 2, want = c->back_nwo > 0 ? c->back_nwo : (need < 4 ? 4 : need); p.nw_o = (want >= need && want <= kWavesPerBlock - 2) ? want : kWavesPerBlock; }
    }
    if (kAblate && c->trace_class == 102 && l == 0) { p.trace = c->trace; a13.trace = c->trace + 256 * 16; a2.trace = c->trace + 2 * 256 * 16; }   // tools/trace_back.py
    if (kAblate && c->trace_class == 103) { p.trace = c->trace; if (l == (c->d.n_layers > 1 ? 1 : 0)) { a13.trace = c->trace + 256 * 16; a2.trace = c->trace + 2 * 256 * 16; aa.trace = c->trace + 5 * 4096; } }   // (k_layers: its second layer)
    grid = parts + Po.grid; if (P13.grid > grid) grid = P13.grid; if (P2.grid > grid) grid = P2.grid; if (with_qkv && Pq.grid > grid) grid = Pq.grid;
    if (grid > all) return FLM_ERR_UNSUPPORTED;
    // the granule hand-offs (flm_layer.h BackArgs::gr): one GPU -- the one-launch token (launch_layers clears gr for a launch without tail: its flag values repeat from token to token);
    // tensor parallel -- the rank-spanning launch, where every rank asked for them (its values count from the token's epoch base)
    p.xg_a = c->xg; p.xg_b = c->xg + d.dim; p.xg_att = c->xg + 2 * (size_t)d.dim; p.xg_hd = c->xg + 3 * (size_t)d.dim; p.gres_off = c->drow_begin;
    p.gr = (c->gr_edges && c->xg && with_qkv && d.dim <= 4 * kGemvBlock && d.dim % 4 == 0 && d.hidden_dim % 4 == 0 && (tpl ? c->grp_gr : (tail != nullptr && tail_fits))) ? 1 : 0;   // (one sweep round per thread for dim; r2 <= 3 rounds for hidden: checked above)
    // with the edges on granules the sweep IS the phase's activation fetch, and it returns behind whatever its own wave requested in front of it: fewer early register sets
    // at short contexts (tools/back_bench.py, 32 layers: pre13 16 / 12 / 8 / 4: 1571 / 1562 / 1561 / 1600 us per token; preq 16 / 12 / 8 with pre13 8: 1561 / 1545 / 1555); with
    // split heads (long contexts) 16 stays (pre13 8: +0.9 %)
    if (p.gr && tpl && c->back_pre13 == 99) p.pre13 = 12;                        // (rank-spanning launch, one-GPU rehearsal, 4 layers: world 2 / 4 251-253 / 261-262 -> 249 / 256-257 us per token with pre13 12, preq 12)
    if (p.gr && !tpl && G == 1 && c->back_pre13 == 99) p.pre13 = 10;            // (preq: below, where it is set; with preq 12: pre13 8 / 10 / 12 = 1542 / 1537 / 1539 us per token at position 14, the same order at 40 and 100)
    {
        auto gpeers = [&](unsigned long long* (&peer)[7], size_t off) { int k = 0; for (int r = 0; r < c->world; ++r) if (r != c->rank) peer[k++] = (unsigned long long*)(c->peer[r] + c->x_gran_off) + off; };
        ao.gout = p.xg_b + c->drow_begin; a2.gout = p.xg_a + c->drow_begin; a13.gout = p.xg_hd + c->plan.hidden_begin;
        aa.gout = p.xg_att + (size_t)c->plan.head_begin * c->hs;                   // (used where the launch says so: layer_body's gr_att)
        {   // q and this token's K / V row from the QKV phase to the heads (layer_body's gr_q): [q: dim][k: kv_dim][v: kv_dim] behind the four vectors
            granule_t* gq = c->xg + 3 * (size_t)d.dim + d.hidden_dim;
            aq.gout = gq; aq.gk = gq + d.dim; aq.gv = gq + d.dim + c->dim_local;
            aa.qg = gq; aa.kg = aq.gk; aa.vg = aq.gv;
        }
        if (tpl && p.gr) {
            gpeers(ao.gout_peer, (size_t)d.dim + c->drow_begin); gpeers(a2.gout_peer, (size_t)c->drow_begin); gpeers(a13.gout_peer, 3 * (size_t)d.dim + c->plan.hidden_begin);
            gpeers(aa.gout_peer, 2 * (size_t)d.dim + (size_t)c->plan.head_begin * c->hs);
        }
    }
    p.flag_x2 = c->flag_lines + 1280 * 16; p.nstq = slots(c->tok_nstq);
    {   // split heads: two K tiles per part by LDS-DMA under the QKV phase, above the phase's stash and above everything the attention itself keeps in LDS before its scores are done
        const size_t tile2 = (size_t)2 * 64 * attn_row_stride(c->hs) * 4, off = ((size_t)own + (size_t)p.nstq * slot + 255) & ~(size_t)255;
        p.kpre_off = (G > 1 && with_qkv && c->attn_kpre && off >= attn_lds_bytes(d.max_seq_len, c->hs, false) && off + tile2 <= kLdsMax) ? (unsigned)off : 0u;
    }
    // (12 early waves of [Wq; Wk; Wv] in front of the x sweep wherever the launch runs on granules, one workgroup per head or split heads: positions 516 / 900 1900 / 2106 -> 1891 / 2098 us per token)
    p.preq = c->tok_preq == 99 ? (p.gr ? 12 : 16) : c->tok_preq < 0 ? 0 : c->tok_preq > 16 ? 16 : c->tok_preq;
    if (tpl) {   // the cross-rank lines: regions of the ranks' exchange buffers (never cleared: epoch values); flag_q and the split heads' score lines stay local (k_embed clears them)
        BackArgs::Tp& t = p.tp;
        t.world = c->world; t.rank = c->rank;
        for (int r = 0; r < c->world; ++r) t.peer[r] = (unsigned*)(c->peer[r] + c->x_tlines_off);
        const unsigned per = (unsigned)c->world * kTpLinesPerRank * 16;
        t.off_h = 0; t.off_x = kTpLinesPerRank * 16; t.off_hd = t.off_x + per; t.off_x2 = t.off_hd + per; t.off_cls = t.off_x2 + per;
        t.head_line0 = c->plan.head_begin * G; t.n_heads_all = d.n_heads * G;
        t.base = c->eng_base; t.fence = c->tp_fence >= 0 ? c->tp_fence : (c->ranks_on_device == c->world ? 0 : 3);
        t.abort_off = (int)(((long long)c->x_flags_off + (long long)kXchgAbortLine * 64 - (long long)c->x_tlines_off) / 4);      // (the group's one abort line: in the exchange flags' region, in front of this one)
        unsigned* mine = t.peer[c->rank];
        p.flag_h = mine + t.off_h; p.flag_x = mine + t.off_x; p.flag_hd = mine + t.off_hd; p.flag_x2 = mine + t.off_x2;
    }
    A.aq = aq; A.ao = ao; A.a13 = a13; A.a2 = a2; A.aa = aa;
    if (tail && tail_fits && Pc.grid <= grid) {
        TailArgs& T = *tail;
        T.acls = acls; T.emb = (const float*)c->emb; T.tok_ptr = &c->state->tok; T.dim = d.dim; T.vocab = c->cls.rows;
        T.epoch = c->tail_mem; T.flag_cls = c->tail_mem + 16; T.slots = (float*)(c->tail_mem + 16 + 256 * 16); T.gridc = Pc.grid;
        T.st = c->state; T.out_tokens = c->out_tokens_dev; T.out_cap = c->out_cap;
    }
    return FLM_OK;
}
template int plan_layer<QT_INT8>(flm_ctx*, int, bool, int, LayerArgs&, BackArgs&, int&, int&, TailArgs*);
template int plan_layer<QT_INT16>(flm_ctx*, int, bool, int, LayerArgs&, BackArgs&, int&, int&, TailArgs*);

// attention + Wo + FFN13 + FFN2 of layer l (with_qkv: the whole layer) in one launch (k_attn_ffn, flm_layer.h); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
template <int QT>
int launch_attn_ffn(flm_ctx* c, hipStream_t st, int l, bool with_qkv, int G) {
    LayerArgs A; BackArgs p; int grid = 0, r2 = 0;
    int r = plan_layer<QT>(c, l, with_qkv, G, A, p, grid, r2, nullptr); if (r) return r;
    const GemvArgs &aq = A.aq, &ao = A.ao, &a13 = A.a13, &a2 = A.a2; const AttnArgs& aa = A.aa;
    {   // the stash takes the rest of the CU's 160 KiB: raise the kernels' dynamic-LDS limit, once per device
        static std::mutex mu; static bool done[64] = {false};
        std::lock_guard<std::mutex> lk(mu);
        if (c->device >= 0 && c->device < 64 && !done[c->device]) {
            const void* fns[] = {(const void*)&k_attn_ffn<QT_INT8, 1, false>, (const void*)&k_attn_ffn<QT_INT8, 3, false>, (const void*)&k_attn_ffn<QT_INT16, 1, false>, (const void*)&k_attn_ffn<QT_INT16, 3, false>,
                                 (const void*)&k_attn_ffn<QT_INT8, 1, true>, (const void*)&k_attn_ffn<QT_INT8, 3, true>, (const void*)&k_attn_ffn<QT_INT16, 1, true>, (const void*)&k_attn_ffn<QT_INT16, 3, true>,
                                 (const void*)&k_attn_ffn<QT_INT8, 1, true, true>, (const void*)&k_attn_ffn<QT_INT8, 3, true, true>, (const void*)&k_attn_ffn<QT_INT16, 1, true, true>, (const void*)&k_attn_ffn<QT_INT16, 3, true, true>};
            for (const void* f : fns) HIPC(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
            done[c->device] = true;
        }
    }
    const dim3 g3(grid), b3(kGemvBlock);
    if (G > 1) {
        if (r2 <= 1) hipLaunchKernelGGL((k_attn_ffn<QT, 1, true, true>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
        else         hipLaunchKernelGGL((k_attn_ffn<QT, 3, true, true>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
    }
    else if (with_qkv) {
        if (r2 <= 1) hipLaunchKernelGGL((k_attn_ffn<QT, 1, true>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
        else         hipLaunchKernelGGL((k_attn_ffn<QT, 3, true>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
    }
    else if (r2 <= 1) hipLaunchKernelGGL((k_attn_ffn<QT, 1, false>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
    else              hipLaunchKernelGGL((k_attn_ffn<QT, 3, false>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

int launch_layer(flm_ctx* c, hipStream_t st, int qt, int l, bool with_qkv, int G) {
    return qt == FLM_QT_INT8 ? launch_attn_ffn<QT_INT8>(c, st, l, with_qkv, G) : launch_attn_ffn<QT_INT16>(c, st, l, with_qkv, G);
}

} // namespace fh
