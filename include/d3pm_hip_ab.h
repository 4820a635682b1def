/* d3pm_hip_ab.h -- additions of libd3pm_hip_ab.so, the A/B (experiment) build of the library.
 *
 * NOT part of the product.  libd3pm_hip_ab.so is the same sources compiled with -DD3PM_ABLATIONS
 * (python -c "import __graft_entry__ as g; g.build_ab()"); it exports everything include/d3pm_hip.h declares PLUS the
 * entry points below, and is loaded only by the interleaved A/B scripts tests/ab_*.py and tests/ab_bit_identity.py.  It holds
 * the schedules and fusions that were built, measured and NOT shipped (DESIGN.md section 3 records each measurement), and
 * timing-only ablation builds of the shipped kernels.  Its knobs are process-wide state on purpose (an experiment toggles
 * them between interleaved arms); nothing in libd3pm_hip.so can reach any of this.
 */
#ifndef D3PM_HIP_AB_H
#define D3PM_HIP_AB_H

#include "d3pm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what this header declares is exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* D3PM_AB_GEMM_BIG_MODE  schedule of the big-tile GEMM: 1 (shipped) / 0 = hand-placed / compiler-placed fragment reads, 9 = the
 *                        output stores of a tile issued inside the next tile's k-steps, 513 = non-temporal output stores,
 *                        2049 = every DMA piece of a k-step issued at its top (all: same results, none faster); 32769 = the
 *                        epilogue's operands (bias, residual rows, frame mask) requested at the top of the tile (same results).
 *                        TIMING-ONLY builds whose results are wrong by construction (parts of the kernel removed; bits:
 *                        16 no DMA, 32 no MFMA, 64 no barriers, 128 no LDS reads, 256 clock stamp for d3pm_debug_gemm_clock,
 *                        4096 with 32 = the operand stream through registers; 1025 = row panels without their LayerNorm
 *                        arithmetic): 17 32 33 81 145 209 257 465 1025 4129.
 * D3PM_AB_ATTN_ARM       0 = shipped; 3 = three 16-query groups per wave (bf16 self-attention; same results); 164 =
 *                        hand-placed fragment reads, 228 = K / V tiles by direct-to-LDS DMA (same results); 201 / 202 = the shipped
 *                        kernel held to two / one workgroup per CU; 100 + bits 1..32 = TIMING-ONLY builds with parts of the
 *                        kernel removed (results wrong by construction).
 * D3PM_AB_GEMM_RING      1 = the five-slab ring schedule (csrc/d3pm_mfma_gemm_ring.hip; same results, measured slower) where it applies.
 * D3PM_AB_GELU_TABLE     1 = the bf16 GELU epilogue of the 192 x 256 big-tile and the latency GEMM reads rn_bf16(gelu(v)) from an
 *                        8.5 KiB LDS table (same results; 86 vs 73 us on fc1).
 * D3PM_AB_LN_PROLOGUE    1 = wherever the latency GEMM runs a LayerNorm-fed projection of a d_model = 512 block, the LayerNorm
 *                        is that launch's prologue (same results; 78 vs 53 ms p50).
 * D3PM_AB_FUSED_FINAL_SAMPLE  1 = inside d3pm_sample_loop the final projection, the posterior and the draw are one kernel and the
 *                        logits never reach HBM (same ids; 253 us vs 30 + 87 us). */
enum { D3PM_AB_GEMM_BIG_MODE = 0, D3PM_AB_ATTN_ARM = 1, D3PM_AB_GEMM_RING = 2, D3PM_AB_GELU_TABLE = 3, D3PM_AB_LN_PROLOGUE = 4,
       D3PM_AB_FUSED_FINAL_SAMPLE = 5 };
int d3pm_ab_set(int knob, int value);

/* LayerNorm-prologue projection (d_model = 512, latency regime): Y[M][N] = act(LN(X) W^T + bias) with X [.][512] the
 * UN-normalised residual stream -- the 64-row operand panels of the latency GEMM are whole rows, so each workgroup normalises
 * them in LDS before its MFMAs (ar_discrete.py:131-132 norm1 -> self-attention in-projection, :145-159 norm3 + FiLM -> fc1).
 * With ln2_w / ln2_b: M = 2 m rows, output rows >= m are source rows 0 .. m-1 under the second LayerNorm (:136-142).
 * Bit-identical to d3pm_op_layernorm followed by d3pm_op_linear.  W [N][512], Y [M][N], act 0 none / 1 GELU. */
int d3pm_op_linear_lnpro(int dtype, const void *X, const void *W, const void *bias, void *Y, int M, int N, int act,
                         const void *ln_w, const void *ln_b, const void *ln2_w, const void *ln2_b, const void *film,
                         float eps, void *stream);

/* Single-op entry of the fused final projection + posterior + draw (replaces `final` at ar_discrete.py:776 followed by
 * p_sample :401-420): hidden [batch * canvas][d_model] (model dtype, already multiplied by the frame mask) -> x_next.
 * Same arguments as d3pm_posterior_sample otherwise.  D3PM_E_SHAPE when the fused kernel does not apply. */
int d3pm_op_final_sample(const d3pm_shape *shape, const d3pm_weights *w, int batch, const void *hidden, const int32_t *x_t,
                         int32_t *x_next, int t, const d3pm_schedule *sched, uint64_t seed, uint32_t utt0, uint32_t flags,
                         void *stream);

/* After a big-tile GEMM launched with D3PM_AB_GEMM_BIG_MODE bit 8 set (and a device synchronisation): {shader clocks, 100 MHz
 * reference ticks} that workgroup 0 spent in the kernel: clocks / ticks * 100 MHz = the clock the chip held under that load
 * (MI355X_MICROARCH.md "DVFS give-back" item 6).  No output of the kernel depends on it. */
int d3pm_debug_gemm_clock(unsigned long long *clocks_and_ticks);

/* Shader-clock stamps of the 32 x 32 x 16 self-attention kernel launched under D3PM_AB_ATTN_ARM = 320 with attn_query_groups = 32
 * (d3pm_mfma_attn32.hip): out[(slot * 12 + tile) * 8 + point], slot 0 = the first workgroup, 1 = one in the middle of the grid,
 * wave 0 of each; points: 0 top of the tile, 1 next tile's global loads issued, 2 scores available, 3 softmax done,
 * 4 P.V issued, 5 global loads returned, 6 LDS stores issued, 7 barrier passed. */
int d3pm_debug_attn32_stamps(unsigned long long *out, int n);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* D3PM_HIP_AB_H */
