// flm_prompt.hip -- the batched prompt path: all tokens of a prompt but the last through GEMM tiles on the int8 matrix cores and prompt attention on the fp32 matrix
// cores; they leave the K/V rows the token-by-token path would (flm_prefill.h).
#include "flm_host.h"

namespace fh {

// ---------------------------------------------------------------------------------------------
// Batched prefill of B prompt tokens at positions pos .. pos+B-1 (single GPU): leaves their K/V rows in the cache, exactly
// the rows the token-by-token path would write (flm_kernels.h, "Batched prefill").  The prompt's LAST token is not part
// of the batch: it runs through the decode kernels and produces the logits.
// ---------------------------------------------------------------------------------------------
template <int QT, int PRO>
int launch_rows(flm_ctx* c, hipStream_t st, const RowsArgs& r, int B, bool coh = false) {
    const size_t lds = (size_t)gemv_lds_layout(r.n, QTraits<QT>::kEsz, true, 4, 4, false).total;
    const int rounds = (r.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (coh) {   // the rows lie in the exchange buffer and were partly written by peer GPUs
        if (rounds <= 1)      hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 1, true>), dim3(B), dim3(kGemvBlock), lds, st, r);
        else if (rounds <= 3) hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 3, true>), dim3(B), dim3(kGemvBlock), lds, st, r);
        else                  hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 0, true>), dim3(B), dim3(kGemvBlock), lds, st, r);
    }
    else if (rounds <= 1) hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 1>), dim3(B), dim3(kGemvBlock), lds, st, r);
    else if (rounds <= 3) hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 3>), dim3(B), dim3(kGemvBlock), lds, st, r);
    else                  hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 0>), dim3(B), dim3(kGemvBlock), lds, st, r);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}
// use_mfma (int8): 1 tile shape by size, 2 (and 0) always 64 x 64, 3 always 128 x 128 matrix-core tiles; int16: always the hi / lo byte planes on the int8 matrix cores
template <int QT, int EPI>
int launch_gemm(flm_ctx* c, hipStream_t st, const GemmArgs& g, int use_mfma) {
    const int tiles = ((g.rows + 63) / 64) * ((g.B + 63) / 64);
    if (use_mfma == 0) use_mfma = 2;
    if (QT == QT_INT8) {   // exact int32 group dots on v_mfma_i32_32x32x32_i8
        using Big = GemmTile<4, 2, 2>; using Small = GemmTile<2, 2, 1>;
        {   // the 128 x 128 tiles stage 76 KiB of LDS: raise the kernels' dynamic-LDS limit, once per device
            static std::mutex mu; static bool done[64] = {false};
            int dev = 0; HIPC(c, hipGetDevice(&dev));
            std::lock_guard<std::mutex> lk(mu);
            if (dev >= 0 && dev < 64 && !done[dev]) {
                HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI_STORE, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Big::kLds));
                HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI_RESIDUAL, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Big::kLds));
                HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI_SWIGLU, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Big::kLds));
                HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI_ROPE_KV, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Big::kLds));
                done[dev] = true;
            }
        }
        if constexpr (EPI == EPI_SWIGLU) {   // g.rows = hidden: a tile = 64 rows of W1 and of W3 (gemm_fuses_swiglu decides)
            const int tiles = ((g.rows + Big::TR / 2 - 1) / (Big::TR / 2)) * ((g.B + Big::TT - 1) / Big::TT);
            hipLaunchKernelGGL((k_gemm_q8_mfma<EPI_SWIGLU, 4, 2, 2>), dim3(tiles), dim3(Big::NT), Big::kLds, st, g);
            HIPC(c, hipGetLastError());
            return FLM_OK;
        } else {
        const int tiles128 = ((g.rows + Big::TR - 1) / Big::TR) * ((g.B + Big::TT - 1) / Big::TT);
        // 128 x 128 tiles move 0.6x the LDS cycles and half the bytes per product; they pay once every CU has one (measured, 7B width:
        // 512 tokens qkv / ffn13 100.8 vs 115.3 us, Wo / ffn2 (128 tiles) 95.4 vs 65.4; 1000 tokens 171 vs 221 and 110 vs 117)
        if (use_mfma == 3 || (use_mfma == 1 && tiles128 >= 256)) hipLaunchKernelGGL((k_gemm_q8_mfma<EPI, 4, 2, 2>), dim3(tiles128), dim3(Big::NT), Big::kLds, st, g);
        else hipLaunchKernelGGL((k_gemm_q8_mfma<EPI, 2, 2, 1>), dim3(tiles), dim3(Small::NT), Small::kLds, st, g);
        }
    }
    else if constexpr (EPI == EPI_SWIGLU) {   // g.rows = hidden: a tile = 32 rows of W1 and the same rows of W3
        const int tiles2 = ((g.rows + 31) / 32) * ((g.B + 63) / 64);
        hipLaunchKernelGGL((k_gemm_q16_mfma<EPI_SWIGLU>), dim3(tiles2), dim3(256), Gemm16Tile::kLds, st, g);
    }
    else hipLaunchKernelGGL((k_gemm_q16_mfma<EPI>), dim3(tiles), dim3(256), Gemm16Tile::kLds, st, g);   // hi / lo byte planes on the int8 matrix cores
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

template <int QT>
int prefill_batched(flm_ctx* c, int B, int pos) {
    const auto& d = c->d;
    const int L = d.n_layers, dim = d.dim, hid = d.hidden_dim, hs = c->hs;
    hipStream_t st = c->stream;
    int r = B <= c->pf_cap ? FLM_OK : fail(c, FLM_ERR_INVALID, "prefill: more tokens than max_seq_len"); if (r) return r;
    const size_t kv_layer = (size_t)c->heads_local * c->kv_rows * hs;
    if (!c->st_ready) {   // once per set of weights: the scales group-major (no allocation: the copies' memory came with the matrices)
        for (auto& w : c->layers)
            for (QMat* m : {&w.qkv, &w.o, &w.w13, &w.w2}) {
                const size_t n = (size_t)m->rows * (m->cols / kGroup);
                hipLaunchKernelGGL(k_transpose_scales, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)m->s, m->st, m->rows, m->cols / kGroup);
            }
        HIPC(c, hipGetLastError());
        c->st_ready = true;
    }
    // Tensor parallel (peer-to-peer; the matrix-core kernels): this rank's heads / rows / hidden slice of every step, as in the decode path
    // (split_rows, transformer.cpp:264-287); the attention output, the residual stream and hd are full-width in every rank's exchange
    // region: the kernel that produces a column slice stores it into all of them, a flag round (k_xchg) closes the step, the row
    // prologues read with coherent loads.  Single GPU: dimL == dim, slices = everything, no peers.
    const bool tp = c->world > 1;
    const int dimL = c->dim_local, hidL = c->hidden_local, col_o = c->drow_begin, rows_o = c->drow_count, col_h = c->plan.hidden_begin, col_a = c->plan.head_begin * hs;
    auto peers = [&](GemmArgs& g, float* p) { g.n_peer = 0; if (tp) { const size_t off = (char*)p - c->xbuf; for (int r2 = 0; r2 < c->world; ++r2) if (r2 != c->rank) g.out_peer[g.n_peer++] = (float*)(c->peer[r2] + off); } };
    hipLaunchKernelGGL(k_embed_rows, dim3(B), dim3(256), 0, st, c->pf_x, (const void*)c->emb, (const float*)c->emb_s, c->emb_qt, dim, (const int*)c->prompt_dev);
    HIPC(c, hipGetLastError());
    for (int l = 0; l < L; ++l) {
        LayerW& w = c->layers[l];
        // x2 = rmsnorm(x1); qx = quantize(x2); q,k,v = W x; RoPE; cache rows   (transformer.cpp:132-135, 386-395, 431-439)
        RowsArgs ra{c->pf_x, w.att_norm, c->pf_xq, c->pf_xs, dim, c->pf_xst};
        r = launch_rows<QT, PRO_RMSNORM_QUANT>(c, st, ra, B, tp); if (r) return r;
        GemmArgs g{w.qkv.q, w.qkv.s, c->pf_xq, c->pf_xs, c->pf_qkv, 3 * dimL, dim, 3 * dimL, B, c->pf_xst, w.qkv.st};
        if ((QT == QT_INT16 || c->use_mfma) && dimL % 32 == 0 && hs % 2 == 0) {      // (int16: always the matrix-core tiles -- round 5: with the same epilogues as the int8 tiles)
            // RoPE and the cache rows as the epilogue of the matrix-core tiles: no [tokens][3 dim] round trip, no k_rope_kv_rows
            g.qout = c->pf_q; g.kcache = c->kcache + (size_t)l * kv_layer; g.vcache = c->vcache + (size_t)l * kv_layer;
            g.rope_cos = c->rope_cos; g.rope_sin = c->rope_sin; g.dim = dimL; g.hs = hs; g.max_seq = c->kv_rows /* (the stride between two heads' cache rows) */; g.pos0 = pos;
            r = launch_gemm<QT, EPI_ROPE_KV>(c, st, g, c->use_mfma); if (r) return r;
        } else {
            r = launch_gemm<QT, EPI_STORE>(c, st, g, c->use_mfma); if (r) return r;
            hipLaunchKernelGGL(k_rope_kv_rows, dim3(B), dim3(256), 0, st, (const float*)c->pf_qkv, c->pf_q, c->kcache + (size_t)l * kv_layer, c->vcache + (size_t)l * kv_layer,
                               (const float*)c->rope_cos, (const float*)c->rope_sin, dimL, hs, c->kv_rows, pos);
            HIPC(c, hipGetLastError());
        }
        if (l == L - 1) break;                            // the batch only has to fill the cache: nothing downstream of the last layer's K/V is needed
        // attention of every query over the cache rows 0 .. its own position   (execute_attn :441-449): the local heads' columns of att
        AttnArgs aa{}; aa.q = c->pf_q; aa.kcache = c->kcache + (size_t)l * kv_layer; aa.vcache = c->vcache + (size_t)l * kv_layer;
        aa.out = c->pf_att + col_a; aa.pos_ptr = &c->state->pos; aa.hs = hs; aa.max_seq = d.max_seq_len; aa.kv_rows = c->kv_rows;
        if (tp) { const size_t off = (char*)aa.out - c->xbuf; for (int r2 = 0; r2 < c->world; ++r2) if (r2 != c->rank) aa.out_peer[aa.n_peer++] = (float*)(c->peer[r2] + off); }
        // which kernels: the exps of a tile of queries (weighted sum on the matrix cores) or the scores of 8 queries (VALU) must fit the LDS;
        // one query per workgroup needs 4 bytes per position and always fits (flm_ctx_create checked max_seq_len against it)
        const bool mq_fits = attn_mq_lds_bytes(d.max_seq_len, hs) <= kLdsMax;
        const bool pv_mfma = c->use_pv_mfma && (hs & 1) == 0;
        if (hs <= 128 && c->use_prefill_mq && c->use_qk_mfma && c->pf_scores && (pv_mfma || mq_fits)) {
            // scores on the matrix cores (fp32 MFMA = the reference's chains, bit for bit), then softmax + weighted sum per tile of queries
            aa.sc_global = c->pf_scores;
            const dim3 gq(c->heads_local, (B + kQkQ - 1) / kQkQ);
            switch (hs >> 5) {
            case 1: hipLaunchKernelGGL(k_qk_mfma<1>, gq, dim3(256), 0, st, aa, pos, dimL, B); break;
            case 2: hipLaunchKernelGGL(k_qk_mfma<2>, gq, dim3(256), 0, st, aa, pos, dimL, B); break;
            case 3: hipLaunchKernelGGL(k_qk_mfma<3>, gq, dim3(256), 0, st, aa, pos, dimL, B); break;
            default: hipLaunchKernelGGL(k_qk_mfma<4>, gq, dim3(256), 0, st, aa, pos, dimL, B); break;
            }
            HIPC(c, hipGetLastError());
            if (pv_mfma) {   // ... and the weighted sum too (an accumulator element = the reference's chain of one (query, dimension))
                const int qw = pv_mfma_queries(pos + B, kLdsMax);     // 16 queries per workgroup up to ~2500 positions, fewer beyond
                hipLaunchKernelGGL(k_attn_pv_mfma, dim3(c->heads_local, (B + qw - 1) / qw), dim3(256), pv_mfma_lds_bytes(pos + B, qw), st, aa, pos, dim, B, qw);
            } else
                hipLaunchKernelGGL(k_attn_prefill_mq<true>, dim3(c->heads_local, (B + kMqQueries - 1) / kMqQueries), dim3(kAttnBlock), attn_mq_lds_bytes(d.max_seq_len, hs), st, aa, pos, dim, B);
        }
        else if (hs <= 128 && c->use_prefill_mq && mq_fits)      // kMqQueries queries per workgroup share every K/V tile
            hipLaunchKernelGGL(k_attn_prefill_mq<false>, dim3(c->heads_local, (B + kMqQueries - 1) / kMqQueries), dim3(kAttnBlock), attn_mq_lds_bytes(d.max_seq_len, hs), st, aa, pos, dim, B);
        else
            hipLaunchKernelGGL(k_attn_prefill, dim3(c->heads_local, B), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, hs), st, aa, pos, dim);
        HIPC(c, hipGetLastError());
        if (tp) { r = exchange(c, st, XK_ATT, nullptr, nullptr, 0); if (r) return r; }
        // x1 += Wo quantize(att)   (transformer.cpp:138-139, 457-466): this rank's rows of Wo = its columns of x1
        RowsArgs rq{c->pf_att, nullptr, c->pf_xq, c->pf_xs, dim, c->pf_xst};
        r = launch_rows<QT, PRO_QUANT>(c, st, rq, B, tp); if (r) return r;
        GemmArgs go{w.o.q, w.o.s, c->pf_xq, c->pf_xs, c->pf_x + col_o, dim, dim, rows_o, B, c->pf_xst, w.o.st};
        peers(go, go.out);
        r = launch_gemm<QT, EPI_RESIDUAL>(c, st, go, c->use_mfma); if (r) return r;
        if (tp) { r = exchange(c, st, XK_X1, nullptr, nullptr, 0); if (r) return r; }
        // hd = swiglu(W1 qx, W3 qx) with qx = quantize(rmsnorm(x1))   (transformer.cpp:144-147, 468-483): this rank's slice of hd
        RowsArgs rf{c->pf_x, w.ffn_norm, c->pf_xq, c->pf_xs, dim, c->pf_xst};
        r = launch_rows<QT, PRO_RMSNORM_QUANT>(c, st, rf, B, tp); if (r) return r;
        if (QT == QT_INT16 || (c->use_mfma && (tp || c->use_mfma == 3 || (c->use_mfma == 1 && ((hidL + 63) / 64) * ((B + 127) / 128) >= 256)))) {
            // 128 x 128 tiles of 64 gate + 64 up rows: the GEMM's epilogue is the SwiGLU
            GemmArgs g13{w.w13.q, w.w13.s, c->pf_xq, c->pf_xs, c->pf_hd + col_h, hid, dim, hidL, B, c->pf_xst, w.w13.st};
            peers(g13, g13.out);
            r = launch_gemm<QT, EPI_SWIGLU>(c, st, g13, c->use_mfma); if (r) return r;
        } else {
            GemmArgs g13{w.w13.q, w.w13.s, c->pf_xq, c->pf_xs, c->pf_gu, 2 * hidL, dim, 2 * hidL, B, c->pf_xst, w.w13.st};
            r = launch_gemm<QT, EPI_STORE>(c, st, g13, c->use_mfma); if (r) return r;
            SwigluPeers sp{}; { GemmArgs t{}; peers(t, c->pf_hd + col_h); sp.n = t.n_peer; for (int i = 0; i < t.n_peer; ++i) sp.p[i] = t.out_peer[i]; }
            hipLaunchKernelGGL(k_swiglu_rows, dim3(B), dim3(256), 0, st, c->pf_hd + col_h, (const float*)c->pf_gu, hidL, hid, sp);
            HIPC(c, hipGetLastError());
        }
        if (tp) { r = exchange(c, st, XK_HD, nullptr, nullptr, 0); if (r) return r; }
        // x1 += W2 quantize(hd)   (transformer.cpp:149-150, 485-494)
        RowsArgs rh{c->pf_hd, nullptr, c->pf_xq, c->pf_xs, hid, c->pf_xst};
        r = launch_rows<QT, PRO_QUANT>(c, st, rh, B, tp); if (r) return r;
        GemmArgs g2{w.w2.q, w.w2.s, c->pf_xq, c->pf_xs, c->pf_x + col_o, dim, hid, rows_o, B, c->pf_xst, w.w2.st};
        peers(g2, g2.out);
        r = launch_gemm<QT, EPI_RESIDUAL>(c, st, g2, c->use_mfma); if (r) return r;
        if (tp) { r = exchange(c, st, XK_X1, nullptr, nullptr, 0); if (r) return r; }
    }
    return FLM_OK;
}

int launch_gemm_store(flm_ctx* c, hipStream_t st, int qt, const GemmArgs& g, int use_mfma) {
    return qt == FLM_QT_INT8 ? launch_gemm<QT_INT8, EPI_STORE>(c, st, g, use_mfma) : launch_gemm<QT_INT16, EPI_STORE>(c, st, g, use_mfma);
}
int prefill_batched_qt(flm_ctx* c, int B, int pos) {
    return c->d.quant_type == FLM_QT_INT8 ? prefill_batched<QT_INT8>(c, B, pos) : prefill_batched<QT_INT16>(c, B, pos);
}

} // namespace fh
