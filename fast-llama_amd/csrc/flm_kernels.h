// flm_kernels.h -- hand-written gfx950 (CDNA4) kernels for the fast-llama per-token hot path.
//
// Everything here is written for wave64 / MI355X only.  Reference citations are paths inside
// CoderLSF/fast-llama (the CPU engine whose arithmetic these kernels reproduce).
//
// DESIGN RULE: every fp32 value is produced with the SAME operations in the SAME order as the
// reference's x86 build (-O3 -mfma, FMA-contracted), so results are BIT-IDENTICAL to the CPU path.
// This is not pedantry: the reference quantizer q = trunc(x / (max|x|/127)) puts the largest
// element of every 64-group exactly on a truncation boundary (x_max/scale = 127 +- 1 ulp), so a
// 1-ulp difference anywhere upstream flips int8 values 126 <-> 127 chaotically and logits drift by
// 1e-2 -- far outside the 1e-3 parity bound.  Integer work (the int8/int16 dots, max, argmax) is
// order-free and fully parallel; every fp32 accumulation is a chain in reference order:
//   * GEMV      : exact int32 group dots in parallel, then acc = fma(sW*sX, float(dot_g), acc), g ascending
//   * rmsnorm   : sum of squares as the reference's 4 strided SSE lanes, each a sequential FMA chain
//   * attention : q.k as 8 strided lanes + sequential lane sum; glibc-exact expf; sequential softmax
//                 sum; weighted V sum sequential over positions
//
// Kernel inventory.  One GREEDY decode token on a single GPU = ONE launch, k_layers<QT, XR2, SPLIT, R5 = 3, TAIL> (fp32 embedding table, head size a multiple of 64, W2's share of a
// workgroup resident: int8 7B; from 128 positions on the SPLIT instantiation: a head spread over hs / 32 workgroups); a token whose logits go to the host, int16 7B and other shapes run
// k_embed + k_layers (ALL L decoder layers) + k_gemv(cls) + k_argmax_advance.  What else is here runs under options, tensor parallelism or tests:
//   k_layers<QT,XR2,SPLIT,R5,TAIL> THE DEFAULT DECODE LAUNCH (flm_layer.h, host side flm_layers.hip): layer_body -- the whole decoder layer: QKV GEMV, attention heads, Wo,
//                         FFN13 + SwiGLU, FFN2 -- in a loop over the layers' argument blocks in device memory; every hand-off, the one between two layers included, is a flag
//                         round; [W1; W3] is stashed in LDS by LDS-DMA under the attention, the next layer's [Wq; Wk; Wv] requested in front of the layer edge ("fuse_token" 0: off).
//                         R5 = 3 (round 5): Wo and FFN2 consume their activation in ARRIVAL ORDER (GemvCtx::run_ao: a wave polls the producers of its own steps' column blocks only).
//                         TAIL (round 5): the embedding row is the first layer's input, the classifier a phase behind the last layer, the argmax + state advance the launch's last act;
//                         flag values count from a per-token epoch base ("fuse_tail" 0: off).  Round 6: the TAIL launch hands x, x1 (hd, the split heads' output) and q + the new
//                         K / V row from workgroup to workgroup as data-tagged 8-byte granules {value, tag} -- no drained stores, no flag line, the consumers sweep their own
//                         elements (flm_gemv.h granule_t, flm_layer.h gemv_preload_granules, attn_head<.., GRIN>; "gr_edges" 0: the four-launch token on flag rounds)
//   k_layers<..,TP[,GRT]> (round 6; flm_layers_tp.hip) the same launch SPANNING the tensor-parallel ranks: every rank its rows (the reference's row split), the four hand-offs of a
//                         layer across the ranks -- as granules in every rank's exchange buffer (GRT: no flag, no fence) or as flag rounds ("tp_fuse_layers", "gr_edges")
//   k_attn_ffn<QT,XR2,QKV,SPLIT> the same layer_body as one launch per layer ("fuse_token" 0; "fuse_layer" 0: without the QKV GEMV; "fuse_back" 0: off)
//   k_gemv<QT,PRO,EPI,XR> group-quantized GEMV, HBM-bound; fused prologue (rmsnorm+quantize | quantize) and epilogue (store | residual add | SwiGLU | RoPE + KV-cache
//                         append): the classifier of every token; every phase of a tensor-parallel rank's token; the per-phase fallback behind a timed-out hand-off
//   k_attn_decode         fp32 single-query attention over the fp32 KV cache (heads split over workgroups at long contexts): the stand-alone launch
//                         of the tensor-parallel token path and of "fuse_attn_o" 0
//   k_attn_o<QT,XR,PREQ>  attention heads and the Wo GEMV in one launch ("fuse_back" 0; head sizes that are not multiples of 64; across tensor-parallel ranks)
//   k_qkv_attn_o<...>     the same with the QKV GEMV in front (long contexts: a head waits for the workgroups that reduced its rows only; across ranks)
//   k_ffn<QT,XR2>         FFN13 (+ SwiGLU) and FFN2 (+ residual) in one launch (beside k_attn_o / k_qkv_attn_o; across ranks as an option)
//   batched prompt processing: k_rows_prologue, k_gemm_q8_mfma<EPI,WT,WR,NB> / k_gemm_q16_mfma (int8 matrix cores; epilogues store |
//                         residual | SwiGLU | RoPE + KV rows), k_qk_mfma + k_attn_pv_mfma (fp32 matrix cores) /
//                         k_attn_prefill_mq (VALU), k_rope_kv_rows, k_swiglu_rows
//   k_embed, k_argmax_advance, k_xchg (tensor-parallel exchange)
// plus small op-level kernels that expose the same __device__ functions to the parity tests.
// (Round 3's weight-streaming engine -- loader / consumer waves around an LDS ring -- was measured slower than these launches and left the library in round 4:
//  tools/experiments/engine/, numbers in profiles/r03_engine_timelines.txt.)
//
// The code lives in: flm_math.h (exact scalar / wave building blocks), flm_gemv.h, flm_attn.h, flm_layer.h, flm_prefill.h, flm_misc.h.
#pragma once
#include "flm_math.h"
#include "flm_gemv.h"
#include "flm_attn.h"
#include "flm_layer.h"
#include "flm_prefill.h"
#include "flm_misc.h"
