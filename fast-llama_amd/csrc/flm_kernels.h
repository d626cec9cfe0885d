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
// Kernel inventory (one decode token on a single GPU = embed + L x {qkv, attention + attn_o, ffn13 + ffn2} + cls + argmax):
//   k_gemv<QT,PRO,EPI,XR> group-quantized GEMV, HBM-bound; fused prologue (rmsnorm+quantize | quantize)
//                         and epilogue (store | residual add | SwiGLU | RoPE + KV-cache append)
//   k_attn_decode         fp32 single-query attention over the fp32 KV cache (heads split over workgroups at long contexts): the stand-alone launch
//                         of the tensor-parallel token path, of "engine" 2 and of "fuse_attn_o" 0; the single-GPU default runs it inside k_attn_o
//   k_attn_o<QT,XR,PREQ>  attention heads and the Wo GEMV in one launch (single GPU)
//   k_qkv_attn_o<...>     the same with the QKV GEMV in front (long contexts: a head waits for the workgroups that reduced its rows only)
//   k_ffn<QT,XR2>         FFN13 (+ SwiGLU) and FFN2 (+ residual) in one launch (single GPU)
//   k_engine<QT>          the weight-streaming engine (flm_engine.h): several dependent GEMVs in one launch, loader waves feeding an LDS ring
//                         with LDS-DMA across the phase edges, consumer waves doing prologues / dots / chains / hand-offs (single GPU, int8)
//   batched prompt processing: k_rows_prologue, k_gemm_q8_mfma<EPI,WT,WR,NB> / k_gemm_q16_mfma (int8 matrix cores; epilogues store |
//                         residual | SwiGLU | RoPE + KV rows), k_qk_mfma + k_attn_pv_mfma (fp32 matrix cores) /
//                         k_attn_prefill_mq (VALU), k_rope_kv_rows, k_swiglu_rows
//   k_embed, k_argmax_advance, k_xchg (tensor-parallel exchange)
// plus small op-level kernels that expose the same __device__ functions to the parity tests.
//
// The code lives in: flm_math.h (exact scalar / wave building blocks), flm_gemv.h, flm_attn.h, flm_engine.h, flm_prefill.h, flm_misc.h.
#pragma once
#include "flm_math.h"
#include "flm_gemv.h"
#include "flm_attn.h"
#include "flm_layer.h"
#include "flm_engine.h"
#include "flm_prefill.h"
#include "flm_misc.h"
