/* gpv_probe.h -- measurement-only entry points (tools/probe/libgpvprobe.so). NOT part of the product boundary: include/gpv.h and
 * libgpv.so contain none of this. The library is self-contained: it instantiates what it needs from the
 * product's device headers and creates its own streams; it does not link libgpv.so. Return value: 0 or a negative hipError. */
#ifndef GPV_PROBE_H
#define GPV_PROBE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Instruction issue-rate microbenchmark on `device`. which: 0 = v_mad_u64_u32, 1 = v_mul_lo_u32, 2 = v_mul_hi_u32, 3 = v_fma_f64,
 * 4 = v_add_co_u32 chain, 5 = v_mad_u32_u24, 6 = v_add_u32, 7 = v_lshl_add_u64, 8 = the instruction mix of the product's
 * column-scanning Fr row as one serial chain per lane at 4 waves per SIMD (rate = multiply-adds only).
 * Returns lane-operations per second over the whole chip. */
int gpvp_microbench(int device, int which, double* lane_ops_per_sec);
/* The same with the shader clock (GHz) sampled during the timed launch: rate / (1024 SIMDs x clock x 64) = wave-instructions per SIMD cycle. */
int gpvp_microbench_clocked(int device, int which, double* lane_ops_per_sec, double* ghz);
/* One vs two interleaved Fr-row chains per lane (the instruction mix of which = 8) at `waves` in {1, 2, 3, 4} resident waves per SIMD:
 * multiply-adds per second over the whole chip. Answers whether chain-level parallelism inside a lane can replace wave-level
 * parallelism (two interleaved rows per lane cost ~22 more VGPRs in the real kernels, i.e. the fourth wave). */
int gpvp_row_mix_rate(int device, int chains, int waves, double* lane_mads_per_sec);
/* Shader clock under load: starts a one-wave sampler on its own high-priority stream that spins for `microseconds` and reports the
 * shader cycles that elapsed. Call gpvp_clock_sample_begin while the workload is being enqueued / running, _end after it: *ghz =
 * shader cycles / wall time of the sampling window. */
int gpvp_clock_sample_begin(int device, unsigned microseconds);
int gpvp_clock_sample_end(double* ghz);
/* MFMA feasibility probe (EXPERIMENTS.md section B, profiles/r02d_mfma_probe.txt): one mix row sum_j C_j * X_j computed
 * `iters` times per lane the product's way (which = 0: 4 x 81 v_mad_u64_u32) or as a byte-plane Toeplitz GEMM on
 * v_mfma_i32_32x32x32_i8 (which = 1); 2 = both kernels concurrently on two streams; 3 / 4 = the MFMA path's two halves alone;
 * 5 / 6 / 7 = 1 / 3 / 2 with the A operands read from precomputed Toeplitz register images. x [n][4][9] radix-2^29 limbs, c_limbs
 * [4][9], q = 384 + 8192 bytes (tools/mfma_probe.py), out [n][18] normalised limbs of the exact integer, *ms = timed launch(es). */
int gpvp_mfma_probe(int device, int which, const uint32_t* x, const uint32_t* c_limbs, const uint8_t* q, uint64_t* out, size_t n, int iters,
                    double* ms);
/* The whole Poseidon-BN254 permutation with the rows of its 56 partial rounds on the matrix pipe (which = 1..3) against the product's
 * operand-scanning kernel (which = 0). states / out [n][4][4] canonical; *ms = best of `reps` launches. */
int gpvp_mfma_probe_permute(int device, int which, const uint64_t* states, uint64_t* out, size_t n, const uint8_t* images,
                            size_t images_bytes, int reps, double* ms);
/* ms3[0] = every wave runs 4 x iters MFMAs, ms3[1] = every wave a VALU multiply-add chain of similar length, ms3[2] = per SIMD one
 * wave does the MFMAs and the other the VALU chain. */
int gpvp_mfma_probe_overlap(int device, int iters, double* ms3, uint32_t* hw_ids, size_t n_ids);
/* Output-stream write pattern (round 4): n_streams streams `stride_words` apart in a buffer of n_streams * stride_words words; every
 * stream is written by lanes_per_stream adjacent lanes which, per step, each run `spin` dependent multiply-adds and then store chunk_words
 * consecutive words (16-byte stores), together appending lanes_per_stream * chunk_words contiguous words; `steps` steps. store = 0: the
 * same loop without the stores. stride_words must be even (16-byte alignment) and >= steps * lanes_per_stream * chunk_words.
 * *ms = best of 3 launches. */
int gpvp_stream_write(int device, int store, size_t n_streams, size_t stride_words, unsigned steps, unsigned lanes_per_stream, unsigned chunk_words,
                      unsigned spin, double* ms);
const char* gpvp_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
