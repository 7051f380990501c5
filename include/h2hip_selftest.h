/*
 * h2hip_selftest.h -- host-side self-test hooks of libh2hip.so (not part of the drop-in ABI).
 *
 * The field / curve templates in halo2_prover_amd/csrc are __host__ __device__; these entry
 * points run the HOST instantiation of exactly that source so that `pytest -m "not gpu"` can
 * compare it with the CPU oracle on a machine without a GPU.  They are not a compute path:
 * one element per call, no batching, never used by the library itself.
 */
#ifndef H2HIP_SELFTEST_H
#define H2HIP_SELFTEST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* field: 0 bn254_fq, 1 bn254_fr, 2 pasta_fp, 3 pasta_fq; op: 0 add, 1 sub, 2 mul, 3 inv, 4 to_mont,
 * 5 from_mont, 6 neg.  Operands / result: 4 x u64 Montgomery limbs. */
int h2_selftest_field_op(int field, int op, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
/* curve ops on the host instantiation of the XYZZ formulas.  op: 0 = affine p + affine q,
 * 1 = 2 * affine p, 2 = (p + q) + q via xyzz_add of two accumulators, 3 = [k] p (k < 2^32, in q[0])
 * by double-and-add.  p, q: affine (8 limbs); out: affine (8 limbs), identity = zeros. */
int h2_selftest_curve_op(int curve, int op, const uint64_t p[8], const uint64_t q[8], uint64_t out[8]);
/* The same field ops through the DEVICE instantiation (gfx950 Comba multiplier): n element pairs, host
 * pointers, one kernel launch.  op 7 = the portable CIOS product compiled for the device (cross-check),
 * op 9 = the product through the MSM's 29-bit working form. */
int h2_selftest_field_op_device(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* the MSM's working-form group law (csrc/h2_curve29.hpp, h2_curve_quad.hpp) run by a device kernel on n pairs of
 * affine points (host pointers, 64 bytes each, API form; out: affine), four lanes per pair.  op 0 / 1: the
 * 4-lanes-per-point addition / doubling, 2 / 3: the one-lane forms, 4: double then add, 5: [k]p with k = the low
 * 32 bits of q.x by the weight kernel's double-and-add (a different k per quad: divergent control flow). */
int h2_selftest_curve_op_device(int curve, int op, const uint64_t* p, const uint64_t* q, uint64_t* out, size_t n);
/* host run of the signed-digit window decomposition used by the MSM digits kernel.
 * scalar: Montgomery limbs; geometry chosen as for `n_for_geometry` registered bases.
 * out[0..3] = widest window c, windows W, buckets B, scalar bits; out[4 + w] = 0 or |d| | sign << 31;
 * out[4 + W + w] = first bit of window w; out[4 + 2W + w] = width of window w (cap >= 4 + 3W).
 * Returns 0, or a negative value (cap too small / carry out of the top window). */
int h2_selftest_digits(int curve, const uint64_t scalar[4], size_t n_for_geometry, uint32_t* out, uint32_t cap);
/* test hook: cap the entries one sort launch may hold, so that the grouped-columns path of wide batches is reached
 * at small sizes; 0 restores the default (2^31 - 1). */
int h2_selftest_set_msm_max_entries(uint64_t limit);
/* MSM workspace checks.
 * h2_selftest_msm_check: host only, no GPU -- lays out the scratch arena of a launch of m columns of n scalars
 *   (col_stride elements apart) against n_bases registered bases and runs the bounds proof msm_device_run runs before
 *   every launch (each kernel's largest index against the region it indexes).  out[0..7] = window bits, windows,
 *   buckets, scalars per sort tile, staged scatter?, two-level sort?, entries per accumulate thread, regions.
 * h2_selftest_msm_tiles: host only -- the one-level sort's block -> (column, tile) mapping is a bijection onto the
 *   live pairs and every surplus block of the rounded-up grid is dead.
 * h2_selftest_msm_guard(1): from now on every MSM launch lays its arena out with a 256-byte red zone behind every
 *   region, fills the arena with a pattern first and counts the red-zone bytes that changed afterwards (synchronous;
 *   tests only; guard(2) also writes one byte behind the second region itself, to test the checker; guard(3) also makes
 *   the two-level sort carry the key's low bits in its side array, the layout of SRS sizes whose entries have no spare bits).
 *   h2_selftest_msm_guard_report: out[0] = launches checked, out[1] = regions overrun since guard(1);
 *   `first` = a description of the first one. */
int h2_selftest_msm_check(int curve, size_t n_bases, size_t n, size_t m, size_t col_stride, int guard, uint64_t out[8]);
int h2_selftest_msm_tiles(uint32_t tiles, uint32_t m);
int h2_selftest_msm_guard(int on);
int h2_selftest_msm_guard_report(uint64_t out[2], char* first, size_t cap);
/* the integer ceiling the MSM kernels are priced against: dependent products of the MSM's working field form
 * (9 x 29-bit limbs) over `curve`'s base field, every CU busy with `waves_per_simd` waves per SIMD; measured
 * chip-wide modmul/s (best of three launches).  bench.py reports it as `modmul_ceiling`. */
int h2_selftest_modmul_rate(int curve, int waves_per_simd, int iters, double* modmul_per_s);
/* host-only pieces of the product surface, for the CPU tests (no GPU, no h2_init needed):
 * what = 0: Blake2b-512 of `in` with the transcript's personalisation ("Halo2-Transcript") -> 64 bytes;
 * 1: the Poseidon constants over bn256::Fr (68 x 3 round constants, MDS, inverse MDS; 32-byte canonical LE each);
 * 2 / 3 / 4: circuit 0 / 1 / 2's verifying-key digest for k = in[0] and the commitments in[1..] (64 canonical bytes
 *    x || y per fixed column then per permutation column, zero = identity) -> 32-byte transcript_repr || the Debug string;
 * 5: pairing check e(P1, Q1) e(P2, Q2) == 1 on two pairs of 64 + 128 canonical bytes -> one byte;
 * 6: the quotient program of circuit in[0] as the prover compiles it -> six u32 (instructions, products, column reads,
 *    live-value slots, constants, inserted reductions) followed by the instructions (3 x u32 each: op_dst, a, b). */
/* commit phases of the C++ prover / keygen that were spread over more than one context (h2_init_devices) so far */
uint64_t h2_selftest_sharded_commits(void);
/* rows per context from which the C++ prover spreads a commit phase over the contexts (default 1024; 0 restores it) */
int h2_selftest_set_shard_min_rows(size_t rows);
int h2_selftest_host(int what, const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* out_len);
/* scratch arenas of the current context: out = {allocations, cross-stream hand-overs (event waits), MSM slots taken
 * over by a further stream, NTT slots taken over} since h2_init */
int h2_selftest_arena_stats(uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
