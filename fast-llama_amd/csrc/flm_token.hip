// flm_token.hip -- the per-token launch sequence: ParallelTransformer::forward at bs == 1 (transformer.cpp:105-161) as kernel launches whose position / token operands
// live in device memory (one hipGraph per token), the per-phase and fused decode launches, the tensor-parallel exchanges, and the per-kernel timing of bench.py.
#include "flm_host.h"

namespace fh {

// ---------------------------------------------------------------------------------------------
// GEMV dispatch
// ---------------------------------------------------------------------------------------------
// Pass geometry of k_gemv for one launch (see the kernel's header comment).
//   cb_shift : CB = largest power of two <= 64 dividing K/16, so every 1 KiB wave load is full
//   Rm       : rows (per matrix) per workgroup pass; bounded by LDS (two strip buffers) and by 64 chain
//              lanes; chosen so that the passes divide evenly over `wgs` workgroups (CU-level balance is
//              what matters for an HBM-bound kernel; inside a workgroup the waves draw steps from a counter)
GemvPlan gemv_plan(int n, int esz, int rows, bool two, bool pairs, bool norm, int wgs) {
    GemvPlan P{};
    const int nchunks = n * esz / 16;
    int cbs = 0; while (cbs < 6 && (nchunks % (2 << cbs)) == 0) ++cbs;       // nchunks % 4 == 0 always
    const int RB = 64 >> cbs;
    const int mult = (pairs && RB < 2) ? 2 : RB;                             // ROPE_KV: row pairs stay in one pass
    const int lds_budget = 150 * 1024;                                       // one 1024-thread workgroup per CU out of 160 KiB
    int rmax = 64;                                                           // one chain lane per row
    while (rmax > mult && gemv_lds_layout(n, esz, norm, rmax, RB, two).total > lds_budget) rmax -= mult;
    rmax = rmax / mult * mult; if (rmax < mult) rmax = mult;
    if (rows < 1) rows = 1;
    int ppw = (rows + wgs * rmax - 1) / (wgs * rmax);                        // passes per workgroup
    if (ppw < 1) ppw = 1;
    int Rm = (rows + wgs * ppw - 1) / (wgs * ppw);
    Rm = (Rm + mult - 1) / mult * mult; if (Rm > rmax) Rm = rmax; if (Rm < mult) Rm = mult;
    const int npass = (rows + Rm - 1) / Rm;
    P.Rm = Rm; P.cb_shift = cbs; P.grid = npass < wgs ? npass : wgs; if (P.grid < 1) P.grid = 1;
    P.nbuf = 2;
    P.lds = (size_t)gemv_lds_layout(n, esz, norm, Rm, RB, two, P.nbuf).total;
    return P;
}

// ---------------------------------------------------------------------------------------------
// One token: ParallelTransformer::forward at bs == 1 (transformer.cpp:105-161).
// Position and token are read from c->state on the device.
//   with_cls  : run the final norm + classifier (+ argmax)
//   advance   : 1 = greedy (tok <- argmax, pos++), 0 = leave state (caller copies logits), 2 = prompt feed
// ---------------------------------------------------------------------------------------------
// argument blocks of the five GEMVs and the attention of layer l (shared by the per-phase launches and k_token)
GemvArgs args_qkv(flm_ctx* c, int l) {
    const auto& d = c->d; LayerW& w = c->layers[l];
    const size_t kv_layer = (size_t)c->heads_local * c->kv_rows * c->hs;
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = w.qkv.q; a.sW = w.qkv.s; a.n = d.dim; a.items = w.qkv.rows / 2;
    a.x = c->x1; a.norm_w = w.att_norm;
    a.out = c->qbuf; a.kcache = c->kcache + (size_t)l * kv_layer; a.vcache = c->vcache + (size_t)l * kv_layer;
    a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.pos_ptr = &c->state->pos;
    a.dim = c->dim_local; a.kv_dim = c->dim_local; a.max_seq = c->kv_rows /* (the epilogue's only use: the stride between two heads' cache rows) */; a.hs = c->hs;
    return a;
}
// Parts per head for a token whose context is T positions: long contexts spread a head's K/V stream over 4 CUs (attn_head, G > 1).
// Below kSplitFrom the exchange of scores between the parts (one more cross-workgroup hand-off) costs more than it saves.

int attn_parts(const flm_ctx* c, int T) {
    // every part owns kSplitDims = 32 output dimensions (its whole V slice then fits the registers / LDS of one workgroup)
    const int Gfull = c->hs / kSplitDims;
    if (c->world > 1 && c->p2p) {   // a tensor-parallel group: what ALL ranks agreed on (flm_p2p_import): the head lines across ranks assume one G
        if (!c->grp_can_split || c->grp_split == 0) return 1;
        return (c->grp_split >= 2 || T >= kSplitFrom) ? Gfull : 1;
    }
    const bool can = c->hs % kSplitDims == 0 && Gfull >= 2 && c->hs <= 128 && c->d.max_seq_len <= kSplitMaxSeq && c->heads_local * Gfull + 8 <= c->cu_count && c->heads_local * Gfull <= 256;
    if (!can || c->attn_split == 0) return 1;
    return (c->attn_split >= 2 || T >= kSplitFrom) ? Gfull : 1;
}
AttnArgs args_attn(flm_ctx* c, int l, int G) {
    const auto& d = c->d;
    const size_t kv_layer = (size_t)c->heads_local * c->kv_rows * c->hs;
    AttnArgs a{};
    a.q = c->qbuf; a.kcache = c->kcache + (size_t)l * kv_layer; a.vcache = c->vcache + (size_t)l * kv_layer;
    a.out = c->att_out + (size_t)c->plan.head_begin * c->hs; a.pos_ptr = &c->state->pos; a.hs = c->hs; a.max_seq = d.max_seq_len; a.kv_rows = c->kv_rows;
    a.G = G; a.sc_global = c->att_sc; a.flag_sc = c->flag_lines + 256 * 16; a.epoch = (unsigned)(l + 1); a.err = c->xwg_err;
    set_peers(c, a, a.out);
    return a;
}
GemvArgs args_o(flm_ctx* c, int l) {
    LayerW& w = c->layers[l];
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = w.o.q; a.sW = w.o.s; a.n = c->d.dim; a.items = c->drow_count;
    a.x = c->att_out; a.out = c->x1 + c->drow_begin;
    set_peers(c, a, a.out);
    return a;
}
GemvArgs args_ffn13(flm_ctx* c, int l) {
    LayerW& w = c->layers[l];
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = w.w13.q; a.sW = w.w13.s; a.n = c->d.dim; a.items = c->hidden_local;
    a.x = c->x1; a.norm_w = w.ffn_norm; a.out = c->hd + c->plan.hidden_begin;
    set_peers(c, a, a.out);
    return a;
}
GemvArgs args_ffn2(flm_ctx* c, int l) {
    LayerW& w = c->layers[l];
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = w.w2.q; a.sW = w.w2.s; a.n = c->d.hidden_dim; a.items = c->drow_count;
    a.x = c->hd; a.out = c->x1 + c->drow_begin;
    set_peers(c, a, a.out);
    return a;
}
GemvArgs args_cls(flm_ctx* c) {
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = c->cls.q; a.sW = c->cls.s; a.n = c->d.dim; a.items = c->cls.rows;
    a.x = c->x1; a.norm_w = c->out_norm; a.out = c->logits + (c->world > 1 ? (size_t)c->rank * c->vocab_slot : 0);
    set_peers(c, a, a.out);
    return a;
}


// tensor parallel, peer to peer: the consuming GEMV of exchange (layer l, kind) does the flag round itself (xchg_fold)
void set_fold(flm_ctx* c, GemvArgs& a, int l, int kind) {
    a.xf.world = 0;
    if (!(c->world > 1 && c->p2p && c->fold_xchg)) return;
    a.xf.local_flags = (unsigned*)(c->xbuf + c->x_flags_off);
    for (int r = 0; r < c->world; ++r) a.xf.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_flags_off);
    a.xf.base = c->eng_base; a.xf.add = (unsigned)(4 * l + kind + 1);
    a.xf.rank = c->rank; a.xf.world = c->world; a.xf.slot = 4 + (kind == 3 ? 1 : kind); a.xf.err = c->xwg_err;      // kinds: 0 att, 1 x1 behind Wo, 2 hd, 3 x1 behind FFN2 (the x1 slot again)
}

// attention + Wo GEMV of layer l in one launch (k_attn_o); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
template <int QT>
int launch_attn_o(flm_ctx* c, hipStream_t st, int l, int G) {
    const auto& d = c->d;
    const int parts = c->heads_local * G, wgs = c->cu_count - parts;
    if (wgs < 1 || parts > 256) return FLM_ERR_UNSUPPORTED;
    GemvArgs a = args_o(c, l);
    if (kAblate && c->trace_class == 101 && l == 0) a.trace = c->trace;     // tools/trace_ao.py
    GemvPlan P;
    int r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, a, wgs, P); if (r) return r;
    const int rounds = (a.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (rounds > 3) return FLM_ERR_UNSUPPORTED;
    size_t lds = attn_lds_bytes(d.max_seq_len, c->hs, G > 1); if (P.lds > lds) lds = P.lds;
    AttnArgs aa = args_attn(c, l, G);
    unsigned* flag = c->flag_lines;                              // one 64-byte line per head part, value = layer + 1; k_embed clears them at the start of the token
    const dim3 grid(parts + P.grid), block(kGemvBlock);
    AoTp tp{};
    if (c->world > 1) {
        // across ranks: every rank's head parts raise their lines in every rank's array (in the exchange buffer); the Wo workgroups read the full att vector
        // from this rank's exchange region (the heads' stores went to every rank) with coherent loads and quantize it themselves
        if (d.n_heads * G > 256) return FLM_ERR_UNSUPPORTED;
        tp.world = c->world; tp.line0 = c->plan.head_begin * G; tp.n_lines = d.n_heads * G; tp.base = c->eng_base; tp.add = (unsigned)(4 * l + 1);
        for (int r = 0; r < c->world; ++r) tp.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_hflags_off);
        flag = (unsigned*)(c->xbuf + c->x_hflags_off);
        if (G > 1) {
            if (rounds <= 1) hipLaunchKernelGGL((k_attn_o<QT, 1, false, true>), grid, block, lds, st, aa, a, parts, flag, 0u, c->xwg_err, tp);
            else             hipLaunchKernelGGL((k_attn_o<QT, 3, false, true>), grid, block, lds, st, aa, a, parts, flag, 0u, c->xwg_err, tp);
        }
        else if (rounds <= 1) hipLaunchKernelGGL((k_attn_o<QT, 1, false>), grid, block, lds, st, aa, a, parts, flag, 0u, c->xwg_err, tp);
        else                  hipLaunchKernelGGL((k_attn_o<QT, 3, false>), grid, block, lds, st, aa, a, parts, flag, 0u, c->xwg_err, tp);
        HIPC(c, hipGetLastError());
        return FLM_OK;
    }
    if (c->hs % kGroup == 0 && G == 1) {
        // a head's output is whole quant groups: the head workgroups quantize it themselves (A3 on the 64 values a wave
        // holds), the GEMV workgroups fetch 1 (2) bytes per element and skip the quantize prologue
        aa.oq = c->att_q; aa.os = c->att_qs; aa.oqt = QT;
        a.xq = c->att_q; a.xs = c->att_qs;
        hipLaunchKernelGGL((k_attn_o<QT, 0, true>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
    }
    else if (G > 1) {
        if (rounds <= 1) hipLaunchKernelGGL((k_attn_o<QT, 1, false, true>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
        else             hipLaunchKernelGGL((k_attn_o<QT, 3, false, true>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
    }
    else if (rounds <= 1) hipLaunchKernelGGL((k_attn_o<QT, 1, false>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
    else                  hipLaunchKernelGGL((k_attn_o<QT, 3, false>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

// QKV + attention + Wo GEMV of layer l in one launch (k_qkv_attn_o); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
template <int QT>
int launch_qkv_attn_o(flm_ctx* c, hipStream_t st, int l, int G) {
    const auto& d = c->d;
    const int parts = c->heads_local * G, all = c->cu_count < 256 ? c->cu_count : 256, wgs = all - parts;
    if (wgs < 1 || parts > 256) return FLM_ERR_UNSUPPORTED;
    GemvArgs aq = args_qkv(c, l), a = args_o(c, l);
    GemvPlan Pq, P;
    int r = plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_ROPE_KV>(c, aq, all, Pq); if (r) return r;
    r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, a, wgs, P); if (r) return r;
    const int rq = (aq.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4), rounds = (a.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (rq > 1 || rounds > 3) return FLM_ERR_UNSUPPORTED;
    size_t lds = attn_lds_bytes(d.max_seq_len, c->hs, G > 1); if (P.lds > lds) lds = P.lds; if (Pq.lds > lds) lds = Pq.lds;
    AttnArgs aa = args_attn(c, l, G);
    unsigned* flag = c->flag_lines;                              // heads' lines (as k_attn_o)
    unsigned* flagq = c->flag_lines + 768 * 16;                  // the QKV workgroups' lines; value = layer + 1, cleared by k_embed
    const int gridx = parts + P.grid > Pq.grid ? parts + P.grid : Pq.grid;
    const dim3 grid(gridx), block(kGemvBlock);
    const unsigned tgt = (unsigned)(l + 1);
    AoTp tp{};
    if (c->world > 1) {
        // across ranks (see launch_attn_o); the QKV phase consumes the x1 exchange behind the previous layer's FFN2 (kind 3 of layer l - 1; layer 0 reads the embedding)
        if (d.n_heads * G > 256) return FLM_ERR_UNSUPPORTED;
        if (l > 0) set_fold(c, aq, l - 1, 3);
        tp.world = c->world; tp.line0 = c->plan.head_begin * G; tp.n_lines = d.n_heads * G; tp.base = c->eng_base; tp.add = (unsigned)(4 * l + 1);
        for (int r = 0; r < c->world; ++r) tp.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_hflags_off);
        flag = (unsigned*)(c->xbuf + c->x_hflags_off);
        if (G > 1) {
            if (rounds <= 1) hipLaunchKernelGGL((k_qkv_attn_o<QT, 1, false, true, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
            else             hipLaunchKernelGGL((k_qkv_attn_o<QT, 3, false, true, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
        }
        else if (rounds <= 1) hipLaunchKernelGGL((k_qkv_attn_o<QT, 1, false, false, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
        else                  hipLaunchKernelGGL((k_qkv_attn_o<QT, 3, false, false, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
        HIPC(c, hipGetLastError());
        return FLM_OK;
    }
    if (c->hs % kGroup == 0 && G == 1) {
        aa.oq = c->att_q; aa.os = c->att_qs; aa.oqt = QT;
        a.xq = c->att_q; a.xs = c->att_qs;
        hipLaunchKernelGGL((k_qkv_attn_o<QT, 0, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
    }
    else if (G > 1) {
        if (rounds <= 1) hipLaunchKernelGGL((k_qkv_attn_o<QT, 1, false, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
        else             hipLaunchKernelGGL((k_qkv_attn_o<QT, 3, false, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
    }
    else if (rounds <= 1) hipLaunchKernelGGL((k_qkv_attn_o<QT, 1, false>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
    else                  hipLaunchKernelGGL((k_qkv_attn_o<QT, 3, false>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

// FFN13 + FFN2 of layer l in one launch (k_ffn); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
template <int QT>
int launch_ffn(flm_ctx* c, hipStream_t st, int l) {
    GemvArgs a13 = args_ffn13(c, l), a2 = args_ffn2(c, l);
    const int wgs = c->cu_count < 256 ? c->cu_count : 256;       // every workgroup resident, one flag line each
    GemvPlan P13, P2;
    int r = plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_SWIGLU>(c, a13, wgs, P13); if (r) return r;
    r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, a2, wgs, P2); if (r) return r;
    const int r13 = (a13.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4), r2 = (a2.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (r13 > 1 || r2 > 3) return FLM_ERR_UNSUPPORTED;
    const size_t lds = P13.lds > P2.lds ? P13.lds : P2.lds;
    const int grid = P13.grid > P2.grid ? P13.grid : P2.grid;
    unsigned* flag = c->flag_lines + 512 * 16;                    // value = layer + 1; k_embed clears the lines at the start of the token
    FfnTp tp{};
    if (c->world > 1) {
        // across ranks: FFN13 consumes the x1 exchange behind the Wo launch (folded flag round, kind 1); one line per RANK for hd (in the exchange buffer, behind the head lines)
        set_fold(c, a13, l, 1);
        tp.world = c->world; tp.rank = c->rank; tp.base = c->eng_base; tp.add = (unsigned)(4 * l + 3); tp.counter = c->ffn_counter;
        for (int r = 0; r < c->world; ++r) tp.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_hflags_off) + 256 * 16;
        flag = (unsigned*)(c->xbuf + c->x_hflags_off) + 256 * 16;
        if (r2 <= 1) hipLaunchKernelGGL((k_ffn<QT, 1, true>), dim3(grid), dim3(kGemvBlock), lds, st, a13, a2, P13.grid, P2.grid, flag, 0u, c->xwg_err, tp);
        else         hipLaunchKernelGGL((k_ffn<QT, 3, true>), dim3(grid), dim3(kGemvBlock), lds, st, a13, a2, P13.grid, P2.grid, flag, 0u, c->xwg_err, tp);
        HIPC(c, hipGetLastError());
        return FLM_OK;
    }
    if (r2 <= 1) hipLaunchKernelGGL((k_ffn<QT, 1>), dim3(grid), dim3(kGemvBlock), lds, st, a13, a2, P13.grid, P2.grid, flag, (unsigned)(l + 1), c->xwg_err, tp);
    else         hipLaunchKernelGGL((k_ffn<QT, 3>), dim3(grid), dim3(kGemvBlock), lds, st, a13, a2, P13.grid, P2.grid, flag, (unsigned)(l + 1), c->xwg_err, tp);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}


// one activation exchange between the tensor-parallel ranks (the reference's threads share the vector in memory instead):
// peer-to-peer (the producer already stored its slice everywhere: flag round only) or an RCCL all-gather
int exchange(flm_ctx* c, hipStream_t st, int kind, float* full, float* mine, int count) {
    Tick t(c, st, KC_ALLREDUCE);
    if (c->p2p) {
        XchgArgs x{};
        x.local_flags = (unsigned*)(c->xbuf + c->x_flags_off);
        for (int r = 0; r < c->world; ++r) x.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_flags_off);
        x.epoch = c->xepoch + kind; x.err = c->xwg_err; x.rank = c->rank; x.world = c->world; x.kind = kind;
        hipLaunchKernelGGL(k_xchg, dim3(1), dim3(64), 0, st, x);
        HIPC(c, hipGetLastError());
        return FLM_OK;
    }
    if (!c->comm) return fail(c, FLM_ERR_STATE, "tensor parallel: neither flm_p2p_import was called nor an RCCL id was given");
    NCCLC(c, ncclAllGather(mine, full, count, ncclFloat, c->comm, st));
    return FLM_OK;
}

int enqueue_token(flm_ctx* c, hipStream_t st, bool with_cls, int advance, int G) {
    const auto& d = c->d;
    const int qt = d.quant_type, hs = c->hs, L = d.n_layers;
    const bool tp = c->world > 1 || (c->comm != nullptr && c->force_tp), coh = tp && c->p2p;   // (a 1-rank communicator takes the sharded path only on request: "force_tp")
    if (!tp && with_cls && advance == 1 && c->fuse_tail && c->fuse_token && c->fuse_layer && c->fuse_back && c->fuse_attn_o && c->fuse_ffn && !c->timing && (c->trace_class < 0 || c->trace_class == 103)) {   // (trace builds, class 103: the stamps of the one-launch token's second layer)
        // a greedy decode token as ONE launch: embedding row, all layers, classifier, argmax + state advance (k_layers<.., TAIL>)
        const int r = launch_layers(c, st, 0, L, G, true);
        if (r != FLM_ERR_UNSUPPORTED) return r;
    }
    {
        Tick t(c, st, KC_EMBED);
        hipLaunchKernelGGL(k_embed, dim3((d.dim + 255) / 256), dim3(256), 0, st, c->x1, (const void*)c->emb, (const float*)c->emb_s, c->emb_qt, d.dim, (const int*)&c->state->tok, c->flag_lines, c->eng_base, c->ffn_counter);
        HIPC(c, hipGetLastError());
    }
    const int wgs = gemv_grid(c->cu_count, c->wg_per_cu, 0, 0);
    auto traced = [&](GemvArgs a, int kc, int l) { if (kAblate && c->trace_class == kc && l == 0) a.trace = c->trace; return a; };
    int r;
    // tensor parallel, peer to peer: the flag rounds of the att / x1 / hd exchanges happen inside the launches that consume them (xchg_fold), not in
    // launches of their own; what stays a k_xchg is the logits' exchange and, for a token without classifier, the last x1 exchange (the next token's
    // k_embed rewrites x1: every peer's stores into it must have landed first)
    // (ranks sharing a device need a CU partition each -- "cu_parts" -- or a consumer that fills the device while it polls keeps its peers' producers out)
    const bool fold = tp && c->p2p && c->world > 1 && c->grp_fold;          // (agreed by the group at flm_p2p_import)
    auto folded = [&](GemvArgs a, int l, int kind) { if (fold && l >= 0) set_fold(c, a, l, kind); return a; };
    // launches that span the ranks wait across workgroups of one launch too: only where the census found one workgroup per CU resident (a CU partition
    // made for the tests is sized for it: launches are cut to the partition)
    const bool span = fold && c->grp_span;
    const int tpfa = span ? c->grp_tpfa : 0, tpff = span ? c->grp_tpff : 0;
    if (!tp && c->fuse_token && c->fuse_layer && c->fuse_back && c->fuse_attn_o && c->fuse_ffn && !c->timing && (c->trace_class < 0 || c->trace_class == 103)) {   // all layers in one launch
        r = launch_layers(c, st, 0, L, G);
        if (r == FLM_OK) goto layers_done; else if (r != FLM_ERR_UNSUPPORTED) return r;
    }
    if (span && c->grp_tpl && c->fuse_token && !c->timing && c->trace_class < 0) {   // tensor parallel: all layers in one launch that spans the ranks (k_layers<.., TP>)
        r = launch_layers(c, st, 0, L, G);
        if (r == FLM_OK) {
            // (a token without classifier: the next token's k_embed rewrites x1 -- every peer's stores of the last layer's rows into it must have landed first; with one, the
            //  classifier's folded flag round below is that exchange)
            if (!with_cls) { r = exchange(c, st, XK_X1, c->x1, c->x1 + c->drow_begin, c->drow_count); if (r) return r; }
            goto layers_done;
        } else if (r != FLM_ERR_UNSUPPORTED) return r;
    }
    for (int l = 0; l < L; ++l) {
        bool fused = false;
        const bool back_ok = !tp && c->fuse_back && c->fuse_attn_o && c->fuse_ffn && !c->timing && (c->trace_class < 0 || c->trace_class == 102);
        if (back_ok && c->fuse_layer) {   // the whole layer in one launch
            r = launch_layer(c, st, qt, l, true, G);
            if (r == FLM_OK) continue; else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        if (((!tp && c->fuse_attn_o && (c->fuse_qkv >= 2 || (c->fuse_qkv && G > 1))) || (span && tpfa >= 2)) && !c->timing && c->trace_class < 0) {   // QKV + attention + ATTN_O in one launch (tensor parallel: "tp_fuse_attn" 2)
            r = qt == FLM_QT_INT8 ? launch_qkv_attn_o<QT_INT8>(c, st, l, G) : launch_qkv_attn_o<QT_INT16>(c, st, l, G);
            if (r == FLM_OK) fused = true; else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        if (!fused) {   // QKV task + RoPE + KV append (transformer.cpp:132-135, execute_qkv :386-395, execute_attn :431-439): this rank's heads
            Tick t(c, st, KC_QKV);
            r = launch_gemv<PRO_RMSNORM_QUANT, EPI_ROPE_KV>(c, st, qt, folded(traced(args_qkv(c, l), KC_QKV, l), l - 1, 3), wgs, coh); if (r) return r;
        }
        if (!fused && back_ok && G == 1) {   // attention + ATTN_O + FFN13 + FFN2 in one launch
            r = launch_layer(c, st, qt, l, false);
            if (r == FLM_OK) continue; else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        if (!fused && ((!tp && c->fuse_attn_o) || (span && tpfa)) && !c->timing && (c->trace_class < 0 || c->trace_class == 101)) {   // attention + ATTN_O in one launch (tensor parallel: across the ranks)
            r = qt == FLM_QT_INT8 ? launch_attn_o<QT_INT8>(c, st, l, G) : launch_attn_o<QT_INT16>(c, st, l, G);
            if (r == FLM_OK) fused = true; else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        if (!fused) {   // ATTN task (execute_attn :441-449): local heads write their slice of the full att_out vector (on every rank, peer to peer)
            Tick t(c, st, KC_ATTN);
            AttnArgs aa = args_attn(c, l, G); if (kAblate && c->trace_class == KC_ATTN && l == 0) aa.trace = c->trace;
            if (G > 1) hipLaunchKernelGGL(k_attn_decode<true>, dim3(c->heads_local * G), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, hs, true), st, aa);
            else       hipLaunchKernelGGL(k_attn_decode<false>, dim3(c->heads_local), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, hs, false), st, aa);
            HIPC(c, hipGetLastError());
        }
        // every rank needs all heads' outputs: the reference's threads share x2 in memory (transformer.cpp:451-454)
        if (tp && !fold) { r = exchange(c, st, XK_ATT, c->att_out, c->att_out + (size_t)c->plan.head_begin * hs, c->dim_local); if (r) return r; }
        if (!fused) {   // ATTN_O task + residual (transformer.cpp:138-139, execute_attn_o :457-466): this rank's rows of Wo
            Tick t(c, st, KC_ATTN_O);
            r = launch_gemv<PRO_QUANT, EPI_RESIDUAL>(c, st, qt, folded(traced(args_o(c, l), KC_ATTN_O, l), l, 0), wgs, coh); if (r) return r;
        }
        if (tp && !fold) { r = exchange(c, st, XK_X1, c->x1, c->x1 + c->drow_begin, c->drow_count); if (r) return r; }
        if (((!tp && c->fuse_ffn) || (span && tpff)) && !c->timing && c->trace_class < 0) {   // FFN13 + FFN2 in one launch (tensor parallel: across the ranks)
            r = qt == FLM_QT_INT8 ? launch_ffn<QT_INT8>(c, st, l) : launch_ffn<QT_INT16>(c, st, l);
            if (r == FLM_OK) {
                if (tp && l == L - 1 && !with_cls) { r = exchange(c, st, XK_X1, c->x1, c->x1 + c->drow_begin, c->drow_count); if (r) return r; }   // (see below)
                continue;
            } else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        {   // FFN13 task + SwiGLU (transformer.cpp:144-147, execute_ffn13 :468-483): this rank's rows of W1/W3
            Tick t(c, st, KC_FFN13);
            r = launch_gemv<PRO_RMSNORM_QUANT, EPI_SWIGLU>(c, st, qt, folded(traced(args_ffn13(c, l), KC_FFN13, l), l, 1), wgs, coh); if (r) return r;
        }
        if (tp && !fold) { r = exchange(c, st, XK_HD, c->hd, c->hd + c->plan.hidden_begin, c->hidden_local); if (r) return r; }
        {   // FFN2 task + residual (transformer.cpp:149-150, execute_ffn2 :485-494): this rank's rows of W2
            Tick t(c, st, KC_FFN2);
            r = launch_gemv<PRO_QUANT, EPI_RESIDUAL>(c, st, qt, folded(traced(args_ffn2(c, l), KC_FFN2, l), l, 2), wgs, coh); if (r) return r;
        }
        if (tp && (!fold || (l == L - 1 && !with_cls))) { r = exchange(c, st, XK_X1, c->x1, c->x1 + c->drow_begin, c->drow_count); if (r) return r; }
    }
layers_done:
    if (with_cls) {
        {   // final norm + CLS task (transformer.cpp:154-160, execute_cls :496-505): this rank's rows of the classifier
            Tick t(c, st, KC_CLS);
            r = launch_gemv<PRO_RMSNORM_QUANT, EPI_STORE>(c, st, qt, folded(traced(args_cls(c), KC_CLS, 0), L - 1, 3), wgs, coh); if (r) return r;
        }
        if (tp) { r = exchange(c, st, XK_LOGITS, c->logits, c->logits + (size_t)c->rank * c->vocab_slot, c->vocab_slot); if (r) return r; }
        if (advance != 0) {
            Tick t(c, st, KC_ARGMAX);
            hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(1024), 0, st, (const float*)c->logits, d.vocab_size, c->state, c->out_tokens_dev, 1, c->out_cap);
            HIPC(c, hipGetLastError());
        }
    } else if (advance == 2) {
        hipLaunchKernelGGL(k_advance_prompt, dim3(1), dim3(64), 0, st, c->state, (const int*)c->prompt_dev);
        HIPC(c, hipGetLastError());
    }
    return FLM_OK;
}

} // namespace fh

extern "C" {

int flm_kernel_times(flm_ctx* c, int pos, int iters, float* avg_us, int32_t* count) {
    if (!avg_us || !count || iters < 1) return FLM_ERR_INVALID;
    int r = check_ready(c, 1, pos); if (r) return r;
    double tot[FLM_KCLASSES] = {0}; long cnt[FLM_KCLASSES] = {0};
    if (c->world > 1) {
        for (int it = 0; it < iters + 1; ++it) {
            r = set_state(c, pos, 1 % c->d.vocab_size, 0); if (r) return r;
            std::vector<TimedLaunch> tl; c->timing = &tl;
            r = enqueue_token(c, c->stream, true, 1, attn_parts(c, pos + 1));
            c->timing = nullptr;
            hipStreamSynchronize(c->stream);
            for (auto& t : tl) {
                float ms = 0.f;
                if (!r && it > 0 && hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) { tot[t.kclass] += ms * 1000.0; cnt[t.kclass] += 1; }
                hipEventDestroy(t.e0); hipEventDestroy(t.e1);
            }
            if (r) return r;
        }
        for (int k = 0; k < FLM_KCLASSES; ++k) { avg_us[k] = cnt[k] ? (float)(tot[k] / cnt[k]) : 0.f; count[k] = (int32_t)(cnt[k] / iters); }
        return FLM_OK;
    }
    const auto& d = c->d;
    const int qt = d.quant_type, L = d.n_layers, wgs = gemv_grid(c->cu_count, c->wg_per_cu, 0, 0);
    hipStream_t st = c->stream;
    r = set_state(c, pos, 1 % d.vocab_size, 0); if (r) return r;
    EvPair ev; HIPC(c, hipEventCreate(&ev.e0)); HIPC(c, hipEventCreate(&ev.e1));
    const hipEvent_t e0 = ev.e0, e1 = ev.e1;
    auto launch = [&](int kc, int l) -> int {
        switch (kc) {
        case KC_EMBED:  hipLaunchKernelGGL(k_embed, dim3((d.dim + 255) / 256), dim3(256), 0, st, c->x1, (const void*)c->emb, (const float*)c->emb_s, c->emb_qt, d.dim, (const int*)&c->state->tok, c->flag_lines, c->eng_base, c->ffn_counter); return FLM_OK;
        case KC_QKV:    return launch_gemv<PRO_RMSNORM_QUANT, EPI_ROPE_KV>(c, st, qt, args_qkv(c, l), wgs);
        case KC_ATTN:   { const int G = attn_parts(c, pos + 1);
                          if (G > 1) hipLaunchKernelGGL(k_attn_decode<true>, dim3(c->heads_local * G), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, c->hs, true), st, args_attn(c, l, G));
                          else       hipLaunchKernelGGL(k_attn_decode<false>, dim3(c->heads_local), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, c->hs, false), st, args_attn(c, l, 1));
                          return FLM_OK; }
        case KC_ATTN_O: return launch_gemv<PRO_QUANT, EPI_RESIDUAL>(c, st, qt, args_o(c, l), wgs);
        case KC_FFN13:  return launch_gemv<PRO_RMSNORM_QUANT, EPI_SWIGLU>(c, st, qt, args_ffn13(c, l), wgs);
        case KC_FFN2:   return launch_gemv<PRO_QUANT, EPI_RESIDUAL>(c, st, qt, args_ffn2(c, l), wgs);
        case KC_CLS:    return launch_gemv<PRO_RMSNORM_QUANT, EPI_STORE>(c, st, qt, args_cls(c), wgs);
        case KC_ARGMAX: hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(1024), 0, st, (const float*)c->logits, d.vocab_size, c->state, (int*)nullptr, 0, 0); return FLM_OK;   // (no id is recorded: the step counter runs on)
        // the fused launches the token path uses on a single GPU (FLM_ERR_UNSUPPORTED: this shape / option setting runs the phases separately)
        case KC_ATTN_WO: if (!c->fuse_attn_o || c->world > 1) return FLM_ERR_UNSUPPORTED;      // (across ranks the launch waits for its peers' heads: not timed in isolation)
                         return qt == FLM_QT_INT8 ? launch_attn_o<QT_INT8>(c, st, l, attn_parts(c, pos + 1)) : launch_attn_o<QT_INT16>(c, st, l, attn_parts(c, pos + 1));
        case KC_FFN:     if (!c->fuse_ffn || c->world > 1) return FLM_ERR_UNSUPPORTED;
                         return qt == FLM_QT_INT8 ? launch_ffn<QT_INT8>(c, st, l) : launch_ffn<QT_INT16>(c, st, l);
        case KC_QKV_ATTN_WO: { const int G = attn_parts(c, pos + 1);
                         if (c->world > 1) return FLM_ERR_UNSUPPORTED;
                         if (!c->fuse_attn_o || !(c->fuse_qkv >= 2 || (c->fuse_qkv && G > 1))) return FLM_ERR_UNSUPPORTED;
                         return qt == FLM_QT_INT8 ? launch_qkv_attn_o<QT_INT8>(c, st, l, G) : launch_qkv_attn_o<QT_INT16>(c, st, l, G); }
        case KC_LAYER: case KC_BACK: {
                         if (c->world > 1 || !c->fuse_back || !c->fuse_attn_o || !c->fuse_ffn || (kc == KC_BACK && attn_parts(c, pos + 1) != 1) || (kc == KC_LAYER) != (c->fuse_layer != 0)) return FLM_ERR_UNSUPPORTED;
                         return launch_layer(c, st, qt, l, kc == KC_LAYER, attn_parts(c, pos + 1)); }
        case KC_LAYERS: {   // ONE launch for the L layers (enqueued for l == 0)
                         if (c->world > 1 || !c->fuse_token || !c->fuse_layer || !c->fuse_back || !c->fuse_attn_o || !c->fuse_ffn) return FLM_ERR_UNSUPPORTED;
                         if (l != 0) return FLM_OK;
                         const int G = attn_parts(c, pos + 1);
                         const int rr = layers_prepare(c, G); if (rr) return rr;
                         return launch_layers(c, st, 0, L, G); }
        case KC_TOKEN: {    // ONE launch for the whole greedy token: embedding row, the L layers, classifier, argmax + state advance (enqueued for l == 0; it moves the state on: put back first)
                         if (c->world > 1 || !c->fuse_tail || !c->fuse_token || !c->fuse_layer || !c->fuse_back || !c->fuse_attn_o || !c->fuse_ffn) return FLM_ERR_UNSUPPORTED;
                         if (l != 0) return FLM_OK;
                         const int G = attn_parts(c, pos + 1);
                         const int rr = layers_prepare(c, G); if (rr) return rr;
                         return launch_layers(c, st, 0, L, G, true); }
        default: return FLM_OK;
        }
    };
    const int classes[] = {KC_EMBED, KC_QKV, KC_ATTN, KC_ATTN_O, KC_FFN13, KC_FFN2, KC_CLS, KC_ARGMAX, KC_ATTN_WO, KC_FFN, KC_QKV_ATTN_WO, KC_LAYER, KC_BACK, KC_LAYERS, KC_TOKEN};
    for (int kc : classes) {
        const bool fused = kc == KC_ATTN_WO || kc == KC_FFN || kc == KC_QKV_ATTN_WO || kc == KC_LAYER || kc == KC_BACK || kc == KC_LAYERS || kc == KC_TOKEN;
        const bool per_layer = (kc >= KC_QKV && kc <= KC_FFN2) || fused;
        const int n = per_layer ? L : 8;
        for (int it = 0; it < iters + 1 && !r; ++it) {          // first round: warm-up
            if (kc == KC_TOKEN) r = set_state(c, pos, 1 % d.vocab_size, 0);          // (the launch moves the decode state on: put it back, outside the timed region)
            if (kc == KC_ATTN || (fused && kc != KC_TOKEN)) r = launch(KC_EMBED, 0);   // (clears the flag lines the workgroups of a fused launch / the parts of a split head wait on)
            HIPC(c, hipEventRecord(e0, st));
            for (int i = 0; i < n && !r; ++i) r = launch(kc, per_layer ? i : 0);
            if (fused && r == FLM_ERR_UNSUPPORTED) { r = FLM_OK; cnt[kc] = 0; break; }
            HIPC(c, hipEventRecord(e1, st));
            HIPC(c, hipEventSynchronize(e1));
            float ms = 0.f; HIPC(c, hipEventElapsedTime(&ms, e0, e1));
            if (it > 0) { tot[kc] += ms * 1000.0 / (kc == KC_LAYERS || kc == KC_TOKEN ? 1 : n); cnt[kc] += 1; }
        }
        avg_us[kc] = cnt[kc] ? (float)(tot[kc] / cnt[kc]) : 0.f;
        count[kc] = cnt[kc] ? (per_layer && kc != KC_LAYERS && kc != KC_TOKEN ? L : 1) : 0;
    }
    avg_us[KC_ALLREDUCE] = 0.f; count[KC_ALLREDUCE] = 0;
    if (r) return r;
    r = xwg_check(c); if (r == FLM_RETRY) return fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out while timing");
    if (r) return r;
    return flm_reset_kv(c);
}

int flm_kernel_bytes(flm_ctx* c, int kclass, int pos, double* bytes) {
    if (!c || !bytes) return FLM_ERR_INVALID;
    const auto& d = c->d; const double e = c->esz, sb = 4.0 / kGroup;
    auto mat = [&](double rows, double cols) { return rows * cols * (e + sb); };
    switch (kclass) {
    case KC_EMBED:  *bytes = d.dim * 4.0; break;
    case KC_QKV:    *bytes = mat(3.0 * c->dim_local, d.dim) + d.dim * 4.0; break;               // + rmsnorm weight
    case KC_ATTN:   *bytes = 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1); break;             // fp32 K and V rows
    case KC_ATTN_O: *bytes = mat(c->drow_count, d.dim); break;
    case KC_FFN13:  *bytes = 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0; break;
    case KC_FFN2:   *bytes = mat(c->drow_count, d.hidden_dim); break;
    case KC_ATTN_WO: *bytes = 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim); break;
    case KC_FFN:    *bytes = 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0 + mat(c->drow_count, d.hidden_dim); break;
    case KC_QKV_ATTN_WO: *bytes = mat(3.0 * c->dim_local, d.dim) + d.dim * 4.0 + 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim); break;
    case KC_CLS:    *bytes = mat(c->cls.rows, d.dim) + d.dim * 4.0; break;
    case KC_BACK:   *bytes = 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim) + 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0 + mat(c->drow_count, d.hidden_dim); break;
    case KC_LAYER:  *bytes = mat(3.0 * c->dim_local, d.dim) + d.dim * 4.0 + 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim) + 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0 + mat(c->drow_count, d.hidden_dim); break;
    case KC_LAYERS: *bytes = (double)d.n_layers * (mat(3.0 * c->dim_local, d.dim) + d.dim * 4.0 + 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim) + 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0 + mat(c->drow_count, d.hidden_dim)); break;
    case KC_TOKEN:  { double lb = 0, cb = 0; flm_kernel_bytes(c, KC_LAYERS, pos, &lb); flm_kernel_bytes(c, KC_CLS, pos, &cb); *bytes = lb + cb + d.dim * 4.0; break; }   // the layers + the classifier + the embedding row
    default:        *bytes = 0; break;
    }
    return FLM_OK;
}

} // extern "C"
