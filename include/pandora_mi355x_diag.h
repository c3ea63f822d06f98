/* Diagnostics build of the MI355X C-ABI library (libpandora_mi355x_diag.so = the same sources with -DPM_DIAG,
 * open-pandora_amd/build.py --diag).  Everything include/pandora_mi355x.h declares is exported with the same meaning;
 * on top of it this build
 *   - reads the PANDORA_* tuning switches from the environment (once per process: INTEGRATION.md lists them) - the
 *     shipped library ignores them all;
 *   - instantiates the older attention kernel variants and the ceiling probes and lets a measurement run pick them.
 * Used by tools/ (A/B runs, counters), by bench.py's `roofline_attention.ceiling` leg and by the tests that exercise every
 * kernel variant; never by the product path (open-pandora_amd/ loads libpandora_mi355x.so unless PANDORA_DIAG_LIB=1).
 * No reference counterpart. */
#ifndef PANDORA_MI355X_DIAG_H
#define PANDORA_MI355X_DIAG_H
#ifdef __cplusplus
extern "C" {
#endif

/* Overrides which kernel variant pm_attention / pm_attention_fp8 launch for single-segment calls: 0 = the production
 * kernel (initial value, unless PANDORA_ATTN_VARIANT is set: read once, on first use), 3 / 5 / 9 = older variants kept
 * for A/B runs, 1 / 16 = force the 32x32x16 / the 16x16x32-MFMA production form at any length, 11 / 12 / 13 = ceiling probes of the production kernel whose OUTPUT IS NOT AN
 * ATTENTION RESULT (no global traffic / no softmax / no LDS reads: csrc/attn.hip).  Process-wide, not thread-safe:
 * measurement runs only. */
void pm_debug_attn_variant(int variant);

/* In-kernel clock stamps of the single-segment attention kernels (variants 0 and 16): while `buf` is non-NULL every
 * workgroup b of a pm_attention launch writes buf[2 b] = d(s_memtime) (shader cycles) and buf[2 b + 1] = d(s_memrealtime)
 * (100 MHz ticks) over its kernel body; clock = buf[2 b] / buf[2 b + 1] x 100 MHz (MI355X_MICROARCH.md "DVFS give-back" item 6).
 * `buf`: device memory of >= 16 bytes x grid size, read by nothing else.  NULL switches the stamps off. */
void pm_debug_attn_stamps(void* buf);

/* Kernel choice of pm_gemm for the 256x256 assembly-loop kernel (csrc/gemm_wide.hip): 0 = never, 1 = by its rule (initial value,
 * unless PANDORA_GEMM_WIDE is set), 2 = wherever it is legal.  Process-wide: measurement runs and the forced-kernel parity tests. */
void pm_debug_gemm_wide(int mode);  /* (+ 16 x loop variant of tools/gen_wide_loop.py VARIANTS: timing-only ablations, 7 = stamps) */
/* The same switch for gemm_wide_stream (csrc/gemm_wide_stream.hip: the tile of gemm_wide as one assembly statement per workgroup, K
 * stream continuous across tiles): 0 = never, 1 = by its rule (initial value unless PANDORA_GEMM_WSTREAM is set), 2 = wherever legal. */
void pm_debug_gemm_wstream(int mode);
/* In-kernel stamps of gemm_wide's loop variant 7: per workgroup b, buf[8 b + 0..6] = shader cycles in the tile prologues / K loops /
 * epilogues, -, K-steps walked, end-of-kernel s_memtime and s_memrealtime.  `buf`: device memory of 64 bytes x grid size; NULL = off. */
void pm_debug_wide_stamps(void* buf);

/* The epilogues' erf / GELU approximant (csrc/common.hpp: Q(t) = 2^P(t), r06) element by element over n f32 values:
 * mode 0 = erf(x), mode 1 = gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) (attention.py:415-430, F.gelu).  Returns a PM_* code. */
int pm_debug_erf(const float* x, float* y, int64_t n, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif
