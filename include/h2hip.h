/*
 * h2hip.h -- C ABI of the MI355X (gfx950) MSM / NTT backend for halo2.
 *
 * Drop-in boundary for the hot path of 0xWOLAND/halo2-prover: the calls
 * halo2_proofs::plonk::create_proof / keygen_{vk,pk} make into
 *   halo2_proofs::arithmetic::best_multiexp   (via ParamsKZG::commit / commit_lagrange)
 *   halo2_proofs::arithmetic::best_fft        (via EvaluationDomain::{lagrange_to_coeff,
 *                                              coeff_to_extended, extended_to_coeff})
 * when driven by the reference's circuits/ crate:
 *   /root/reference/circuits/src/utils.rs:63-70   (keygen)        -> fixed/sigma commitments
 *   /root/reference/circuits/src/utils.rs:83-91   (SHPLONK prove) -> create_proof
 *   /root/reference/circuits/src/utils.rs:105-120 (GWC prove)     -> create_proof
 *   /root/reference/circuits/src/wasm.rs:57-65,77-122             -> same, behind wasm-bindgen
 * The functions themselves live in the un-vendored dependency halo2_proofs @6b43b6b
 * (circuits/Cargo.toml:16-17, Cargo.lock:836-838); their Rust signatures are restated in
 * SURVEY.md section 8(b).  INTEGRATION.md shows the Rust-side `extern "C"` binding.
 *
 * Conventions (identical to halo2curves' in-memory layout, so a Rust slice can be passed as is):
 *   field element  = 4 x u64 little-endian limbs, Montgomery form (R = 2^256)      32 bytes
 *   affine point   = x || y, identity = (0, 0)                                     64 bytes
 *   Jacobian point = x || y || z, identity has z = 0                               96 bytes
 * All functions return 0 on success or a negative h2_status_t; they never abort or throw
 * across the ABI (the reference panics on length mismatch; here that is H2_EINVAL).
 *
 * Threading and streams.  Every entry point takes one process-wide lock while it validates and ENQUEUES, so calls
 * from several host threads are safe.  Host-pointer entry points run on the library's own stream of each device and
 * return when the result is in host memory.  The *_device entry points enqueue on the caller's `stream` (a
 * hipStream_t; NULL = the library's stream, a blocking stream, i.e. ordered against the legacy null stream that
 * PyTorch uses by default) and return without synchronising.  Different calls may use different streams.  The MSM
 * workspace and the multi-pass NTT's second buffer exist once per stream for up to four streams per device: MSMs (or
 * NTTs) enqueued on different streams run side by side -- two proofs in flight on one GPU, DESIGN.md section 5 -- and a
 * fifth stream takes over the scratch that has been idle longest, behind an event wait.  The division / prefix-scan
 * scratch is one per device: it remembers its last user and makes a call on another stream wait for it.  Results are
 * correct whatever streams are mixed (tests/test_gpu_multi.py).  Scratch is sized by the largest call seen on its
 * stream and kept until h2_shutdown.
 * The caller orders its own producers / consumers of the device buffers it passes, as with any HIP library.
 *
 * Devices.  h2_init(device) binds the process to one GPU (one process per GPU under torch.distributed / RCCL is how
 * the Python host layer scales out).  h2_init_devices(n, ids) gives ONE process several GPUs
 * (the shape a Rust host linking this library wants): bases are replicated on every device at registration,
 * h2_msm_batch / h2_ntt_batch shard their columns column j -> device j mod n, h2_msm splits one long MSM by
 * contiguous point range and adds the n partial sums; results travel as 64/96-byte points through host memory, so
 * no device-to-device collective is needed inside one process.  *_device entry points act on the context of the
 * calling thread's current HIP device.  The product surface uses the contexts too: h2_generate_proof (keygen and
 * create_proof) spreads every commit phase over them by point range -- context g commits rows [n g / G, n (g+1) / G)
 * of every column of the phase against its replica of the table; the other contexts' shares of the columns and the
 * G x m partial sums travel device to device (peer copies) and the sums are added on the prover's device -- and
 * produces the same proof bytes as with one context (tests/test_gpu_multi.py).
 *
 * Trust.  h2_generate_proof / h2_verify_proof keep the SRS tables and keys of the last few params blobs and recognise a
 * blob by a fast fingerprint (eight multiply-xorshift lanes over every byte, finished through Blake2b with the header
 * and the G2 tail): it guards against accidents, NOT against a party who crafts a second blob to collide with a cached
 * one.  Params must come from a source the caller trusts (as the reference's own flow assumes: the UI generates them);
 * a verifier that takes params from an untrusted party calls h2_params_cache_clear() before h2_verify_proof.
 */
#ifndef H2HIP_H
#define H2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { H2_BN254 = 0, H2_PALLAS = 1, H2_VESTA = 2 } h2_curve_t;

typedef enum {
  H2_OK = 0,
  H2_EINVAL = -1,   /* bad argument: length / log_n mismatch, null pointer, unknown curve */
  H2_ENOMEM = -2,   /* host or device allocation failed */
  H2_EDEVICE = -3,  /* HIP runtime error (no device, launch failure, ...) */
  H2_EHANDLE = -4,  /* unknown or released bases handle */
  H2_ENOTINIT = -5, /* h2_init has not been called */
  H2_EPROOF = -6    /* the product surface (h2_generate_proof / h2_verify_proof ...): malformed input or proof */
} h2_status_t;

/* ---- lifecycle -------------------------------------------------------------------------- */
/* Bind this process to one GPU (HIP device ordinal).  Idempotent for the same device. */
int h2_init(int device);
/* One process, several GPUs (SURVEY.md section 8(b) `h2_init(n_devices, ids)`): one context, stream and set of
 * arenas per listed device.  Idempotent for the same list.  An id may be listed twice (two contexts on one GPU:
 * how the sharded paths are tested on a one-GPU box). */
int h2_init_devices(int n_devices, const int* device_ids);
/* number of contexts, or H2_ENOTINIT */
int h2_device_count(void);
int h2_shutdown(void);
const char* h2_strerror(int status);
/* Text of the last HIP error seen by this process ("" if none). */
const char* h2_last_device_error(void);
/* ABI version: major * 1000 + minor. */
int h2_version(void);

/* ---- bases (the SRS: ParamsKZG::g / g_lagrange) ------------------------------------------
 * Registers n affine points and keeps them -- plus the table of their 2^(c*w) multiples that
 * lets every Pippenger window share one bucket set -- resident in HBM.  Replaces the `bases`
 * slice argument of best_multiexp for all later calls (ParamsKZG::commit_lagrange passes
 * g_lagrange, ParamsKZG::commit passes g).  `affine` is a host pointer.  Every point must be the
 * identity (0, 0) or lie on the curve with canonical coordinates (checked on the device while the table is built, as
 * the reference's ParamsKZG::read checks with SerdeFormat::RawBytes): otherwise H2_EINVAL. */
int h2_bases_register(h2_curve_t curve, const uint64_t* affine /* n*8 */, size_t n, uint64_t* handle_out);
/* Same, from a device pointer (the points are copied; the caller keeps ownership). */
int h2_bases_register_device(h2_curve_t curve, const void* d_affine, size_t n, uint64_t* handle_out);
int h2_bases_release(uint64_t handle);
/* Number of points registered under the handle, or a negative status. */
int64_t h2_bases_len(uint64_t handle);

/* ---- MSM == best_multiexp(coeffs, bases) -> C::Curve ------------------------------------
 * out_jac receives sum_i scalars[i] * bases[i] as a Jacobian point (any representative of
 * the group element; compare after normalisation).  n may be smaller than the registered
 * length (a prefix of the bases is used), never larger. */
int h2_msm(h2_curve_t curve, uint64_t bases_handle, const uint64_t* scalars /* n*4 */, size_t n,
           uint64_t out_jac[12]);
/* m columns against the same bases (the per-column commitments of one proof phase).
 * out_affine receives m normalised affine points (identity = (0,0)), column order. */
int h2_msm_batch(h2_curve_t curve, uint64_t bases_handle, const uint64_t* const* scalars /* m ptrs */,
                 size_t n, size_t m, uint64_t* out_affine /* m*8 */);
/* Device-resident form: d_scalars holds m columns of n scalars, column stride n*32 bytes;
 * d_out_jac receives m Jacobian points (m*96 bytes, device memory).  Enqueued on `stream`
 * (a hipStream_t, NULL = the library's stream); returns without synchronising. */
int h2_msm_device(h2_curve_t curve, uint64_t bases_handle, const void* d_scalars, size_t n, size_t m,
                  void* d_out_jac, void* stream);
/* m columns, column j against the bases registered under handles[j] (all of the same length and curve; m <= 16):
 * ONE launch sequence for commitments that do not wait for each other although they use different SRS vectors --
 * in create_proof the permutation products (over g_lagrange) and the vanishing argument's random polynomial (over g,
 * drawn from the RNG, not from the transcript).  first_base / n / col_stride as in h2_msm_device_range.  Every MSM call pays ~0.3 ms of sort + small-grid tail whatever m is. */
int h2_msm_device_multi(h2_curve_t curve, const uint64_t* bases_handles /* m */, const void* d_scalars,
                        size_t first_base, size_t n, size_t col_stride, size_t m, void* d_out_jac, void* stream);
/* Make `stream` wait until the bucket-accumulate kernel of the most recently enqueued MSM (on any stream) has
 * finished.  What follows it in an MSM -- the per-bucket fix-up, the bucket weights, the tree sums -- are chains of
 * dependent point operations on about one wave per SIMD: work queued on another stream behind this wait (the
 * Lagrange -> coefficient -> extended transforms of columns whose commitment is being computed) runs beside them
 * instead of beside the chip-filling accumulate kernel.  The first call only switches the marking on (no MSM has
 * recorded its point yet, so nothing is waited for); bench.py's warm-up steps absorb that. */
int h2_stream_wait_msm_tail(void* stream);
/* The same over a contiguous RANGE of the registered bases: result_j = sum_{i<n} scalars_j[i] * bases[first_base + i],
 * columns col_stride elements apart (col_stride >= n).  This is one rank's share of an MSM split by point range over
 * several GPUs (BASELINE config 4): every rank passes its slice of every column, the
 * partial sums are all-gathered (96 B each) and added with h2_points_sum_device. */
int h2_msm_device_range(h2_curve_t curve, uint64_t bases_handle, const void* d_scalars, size_t first_base, size_t n,
                        size_t col_stride, size_t m, void* d_out_jac, void* stream);
/* d_out_jac[j] = sum_{g < groups} d_in_jac[g * count + j] for j < count (Jacobian points, device memory): adds the
 * all-gathered partial sums of a range-split MSM.  d_out_jac must not alias d_in_jac. */
int h2_points_sum_device(h2_curve_t curve, const void* d_in_jac, size_t groups, size_t count, void* d_out_jac,
                         void* stream);

/* ---- NTT == best_fft(a, omega, log_n) ---------------------------------------------------
 * In place, natural order in and out, A[i] = sum_j a[j] * omega^(i*j), unscaled, over the
 * SCALAR field of `curve` (bn256::Fr for H2_BN254).  a must hold exactly 1 << log_n elements. */
int h2_ntt(h2_curve_t curve, uint64_t* a /* n*4, in place */, const uint64_t omega[4], uint32_t log_n);
int h2_ntt_batch(h2_curve_t curve, uint64_t* const* cols /* m ptrs */, size_t m, const uint64_t omega[4],
                 uint32_t log_n);
/* Device-resident form: m columns, column stride (1 << log_n) * 32 bytes, in place. */
int h2_ntt_device(h2_curve_t curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n,
                  void* stream);

/* ---- EvaluationDomain pieces (halo2_proofs/src/poly/domain.rs) ----
 * Device-resident columns (m columns, stride n*32 bytes) over the SCALAR field of `curve`; asynchronous on
 * `stream` (NULL = the library's stream).  With these a column stays in HBM from its Lagrange form through
 * lagrange_to_coeff / coeff_to_extended / divide_by_vanishing_poly / extended_to_coeff to its commitment. */
/* best_fft followed by a[i] *= scale, fused into the last pass: EvaluationDomain::ifft(a, omega_inv, k, n^-1) */
int h2_ntt_scaled_device(h2_curve_t curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n,
                         const uint64_t scale[4], void* stream);
/* a[i] *= c */
int h2_poly_scale_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const uint64_t c[4], void* stream);
/* a[i] *= g^i : distribute_powers_zeta / the coset shift before an extended-domain NTT (and its inverse) */
int h2_poly_coset_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const uint64_t g[4], void* stream);
/* a[i] *= t[i mod period], period a power of two: divide_by_vanishing_poly with t = t_evaluations */
int h2_poly_mul_periodic_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const void* d_t, size_t period,
                                void* stream);
/* a[i] = 1 / a[i] over n elements (zero stays zero): the permutation argument's denominators */
int h2_poly_inverse_device(h2_curve_t curve, void* d_a, size_t n, void* stream);
/* a[i] = a[i] op b[i] over n elements; op 0 = add, 1 = sub, 2 = mul */
int h2_poly_pointwise_device(h2_curve_t curve, int op, void* d_a, const void* d_b, size_t n, void* stream);
/* q = (a - a(z)) / (X - z) over n coefficients, q[n-1] = 0: halo2_proofs src/arithmetic.rs `kate_division(a, z)`
 * (the witness polynomials of the GWC / SHPLONK openings).  d_q must not alias d_a. */
int h2_poly_divide_linear_device(h2_curve_t curve, const void* d_a, size_t n, const uint64_t z[4], void* d_q,
                                 void* stream);
/* out[i] = prod_{j < i} a[j], out[0] = 1 (n elements; d_out may alias d_a): the running product of the permutation
 * argument, z[i+1] = z[i] * ratio[i] (halo2_proofs src/plonk/permutation/prover.rs `Argument::commit`; SURVEY.md
 * App. A.4) -- the caller multiplies by last_z and writes the blinding rows. */
int h2_poly_prefix_product_device(h2_curve_t curve, const void* d_a, size_t n, void* d_out, void* stream);
/* out[i] = Scalar::random(rng) number first_block + i of rng = ChaCha20Rng::from_seed(seed), Montgomery limbs:
 * the coefficients of the vanishing argument's random_poly (halo2_proofs src/plonk/vanishing/prover.rs
 * `Argument::commit`).  Each draw consumes one 64-byte ChaCha20 block. */
int h2_chacha20_scalars_device(h2_curve_t curve, const uint8_t seed[32], uint64_t first_block, size_t n, void* d_out,
                               void* stream);

/* ---- introspection used by bench.py's roofline (no effect on results) --------------------
 * Names the kernels launched by the last h2_msm* / h2_ntt* call and their geometry. */
typedef struct {
  uint32_t window_bits;   /* c */
  uint32_t windows;       /* W */
  uint32_t buckets;       /* 2^(c-1) */
  uint64_t table_bytes;   /* resident precomputed table */
} h2_msm_plan_t;
int h2_msm_plan(uint64_t bases_handle, h2_msm_plan_t* out);


/* ---- best_fft over group elements ------------------------------------------------------------
 * Replaces halo2_proofs::arithmetic::best_fft<Scalar, G> for G = C::Curve (the FftGroup impl of the curve's projective
 * points): the reference's only use is ParamsKZG::new -> g_to_lagrange, reached from
 * /root/reference/circuits/src/utils.rs:59-61 (g_lagrange = n^-1 * best_fft(g, omega^-1, k)).  n = 2^log_n Jacobian
 * points (96 bytes, Montgomery limbs, identity z = 0) transformed IN PLACE, natural order in and out, unscaled, omega in
 * the scalar field's Montgomery form.  Results are group elements: compare after affine normalisation.
 * h2_fft_group: host pointer, synchronous; h2_fft_group_device: device pointer, asynchronous on `stream`. */
int h2_fft_group(h2_curve_t curve, uint64_t* points_jac, const uint64_t omega[4], uint32_t log_n);
int h2_fft_group_device(h2_curve_t curve, void* d_points_jac, const uint64_t omega[4], uint32_t log_n, void* stream);

/* ---- SRS generation == the g vector of ParamsKZG::new(k) -----------------------------------
 * d_out_affine[i] = [s^i] G for i < n (device memory, n*64 bytes), s in Montgomery form.
 * Counterpart of the setup loop reached from /root/reference/circuits/src/utils.rs:59-61 and
 * wasm.rs:49-55; also how bench.py makes valid synthetic bases without the CPU oracle. */
int h2_srs_generate(h2_curve_t curve, const uint64_t s[4], size_t n, void* d_out_affine, void* stream);

/* d_out_affine[i] = [k_i] G for n scalars in device memory (Montgomery form): with k_i = L_i(s), the Lagrange
 * basis at the toxic scalar, this is the g_lagrange vector of ParamsKZG::new(k). */
int h2_fixed_base_mul(h2_curve_t curve, const void* d_scalars, size_t n, void* d_out_affine, void* stream);

/* ---- kernel timing for bench.py's roofline ---------------------------------------------------
 * While enabled, every MSM launch records HIP events (on the launch stream) around its
 * bucket-accumulate kernel.  h2_profile_read waits for them, returns the sums and resets. */
typedef struct {
  uint64_t launches;         /* accumulate-kernel launches timed */
  double kernel_ms;          /* sum of their durations */
  double algorithmic_bytes;  /* sum over launches of m*n*(32+64) + m*96 */
} h2_profile_t;
int h2_profile_enable(int on);
int h2_profile_read(h2_profile_t* out);

/* ---- the product surface: setup / prove / verify / simulate / count -----------------------------------
 * C counterparts of the reference's five wasm-bindgen exports (/root/reference/circuits/src/wasm.rs:49 setup,
 * :68 wasm_simulate_circuit, :77 wasm_generate_proof, :125 wasm_verify_proof, :182 get_circuit_count), i.e. of
 * utils.rs:59-158 (generate_params, generate_keys, generate_proof[_with_instance], verify[_with_instance]) over
 * KZG / BN254 with a Blake2b transcript: circuit 0 = Collatz (SHPLONK, no instance), 1 = arithmetic (GWC, public
 * inputs [constant, z]), anything else = Poseidon (GWC; prove takes hex_to_fr(output), verify recomputes the hash
 * of x).  Params and proofs use the reference's wire formats, so artefacts interoperate.
 * Everything numeric runs on the GPU of the current context; the SRS tables of the last few distinct params blobs
 * stay registered between calls.
 *
 * Randomness comes from the caller, call by call as the reference's RngCore is used: Fr::random = eight calls of
 * 8 bytes, the blinding polynomial's ChaCha20 seed = one call of 32 bytes.  rng = NULL draws from the OS.  With
 * the same stream the proof bytes equal the reference's.
 *
 * Output buffers are caller-owned; *out_len receives the size (also when the call returns H2_EINVAL because
 * `cap` is too small).  Malformed params / JSON / proofs give H2_EPROOF (the reference panics); a well-formed
 * proof that does not verify is *ok = 0 with H2_OK (the reference traps for the GWC circuits, utils.rs:150-157). */
typedef void (*h2_rng_fill_t)(void* ctx, uint8_t* out, size_t n);
int h2_setup(uint32_t k, h2_rng_fill_t rng, void* rng_ctx, uint8_t* out, size_t cap, size_t* out_len);
int h2_generate_proof(const uint8_t* params, size_t params_len, const char* json, int circuit, h2_rng_fill_t rng,
                      void* rng_ctx, uint8_t* out, size_t cap, size_t* out_len);
int h2_verify_proof(const uint8_t* params, size_t params_len, const uint8_t* proof, size_t proof_len,
                    const char* json, int circuit, int* ok);
/* the NUL-terminated result string of wasm_simulate_circuit ("N/A" for Collatz) */
int h2_simulate(const char* json, int circuit, char* out, size_t cap, size_t* out_len);
int h2_circuit_count(void);
/* The SRS of the last few distinct params blobs stays registered (MSM tables resident in HBM) between calls; this
 * releases them: the next call parses and registers its params again, as the reference does on every call. */
int h2_params_cache_clear(void);
/* Proving keys (fixed + permutation columns in all their forms, their commitments, the vk digest, the compiled
 * quotient program) depend only on the params and the circuit index; the reference rebuilds them on every prove and
 * verify call (wasm.rs:86,95,114,132), this library keeps them by default.  h2_key_cache(0) restores the
 * reference's behaviour (bench.py times both); returns the previous setting. */
int h2_key_cache(int enable);

#ifdef __cplusplus
}
#endif
#endif /* H2HIP_H */
