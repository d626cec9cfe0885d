// flm_tuning.h -- the experiment dials of libflm_gpu.so.  NOT part of the drop-in boundary (include/flm_gpu.h): a deployment never touches them, their optimum was
// measured and is the default.  flm_set_option refuses these keys (FLM_ERR_INVALID) until the context has been put into tuning mode with flm_set_option(ctx, "tuning", 1)
// -- tools/back_bench.py, tools/stress.py and the parity tests do that (through fast-llama_amd/capi.py) to sweep them and to prove that none of them changes a result bit.
//
//   "wg_per_cu"        workgroups per CU for the stand-alone GEMV launches (1)
//   "use_mfma"         int8 prefill GEMM tile shape on the matrix cores: 1 by problem size, 2 (or 0) always 64 x 64, 3 always 128 x 128 tiles (1)
//   "tok_preq"         k_layers: how many of a workgroup's 16 waves request their first register set of the NEXT layer's [Wq; Wk; Wv] in front of the layer edge's hand-off (by launch: 12 in the one-launch token below 128 positions, else 16)
//   "tok_nstq"         ... and how many LDS stash slots (4.25 KiB each; -1: as many as the LDS holds) it fills with it there (4)
//   "back_nst13"       stash slots a Wo workgroup fills with [W1; W3] under the attention (-1)
//   "back_nst13_head"  ... a head workgroup fills behind its head (-1)
//   "back_nst2"        ... every workgroup fills with W2 behind its rows of hd (0; the arrival-order FFN2 holds W2's whole share anyway)
//   "back_pre13"       waves that request their first register set of [W1; W3] in front of the x1 hand-off (by launch: 8 in the one-launch token below 128 positions, else 16)
//   "back_pre2"        waves that request their first register set of W2 in front of the hd flag round (16; the hand-off in its round-4 form)
//   "back_ao2"         arrival-order FFN2: what of W2 is requested in front of a wave's first look: 1 everything, 2 the first register sets (default), 3 first sets + stash
//   "attn_kpre"        split heads (long contexts) inside the whole-layer launches: a part's first two K tiles are brought into LDS by LDS-DMA under the layer's QKV phase (1) or requested when the
//                      part's attention starts (0)
//   "back_nwo"         arrival-order Wo: how many of a workgroup's 16 waves hold Wo's steps and look for their heads; the others issue the [W1; W3] stash (0 = ceil(steps / 2): 10 at 7B; 16 = every
//                      wave does both, rounds 4-5)
//   "inject_wait_failure" 1 = raise the "a cross-workgroup wait gave up" flag NOW (one shot): the next call's fused launches run through without waiting, the call is re-run on
//                      one kernel per phase and the context stays there ("fallback" 1) -- the error path of a 20 ms time-out, exercised by tests/test_gpu_model.py without waiting for one
// Numbers behind the defaults: DESIGN.md section 7c / 7d, tools/back_bench.py.
#pragma once
namespace fh {
constexpr const char* kTuningKeys[] = {"wg_per_cu", "use_mfma", "tok_preq", "tok_nstq", "back_nst13", "back_nst13_head", "back_nst2", "back_pre13", "back_pre2", "back_ao2", "attn_kpre", "back_nwo", "inject_wait_failure"};
}
