/*
 * h2_oracle.c -- CPU restatement of the reference's MSM / NTT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the timed CPU baseline
 * (bench.py "cpu_baseline", kind "port").  The product library (halo2_prover_amd/csrc)
 * never links, loads or calls it.
 *
 * The reference's arithmetic is NOT in /root/reference: it lives in the un-vendored git
 * dependency halo2_proofs @6b43b6bad3521a011afd26cf38fe28e317f27396
 * (/root/reference/circuits/Cargo.toml:16-17, Cargo.lock:836-838), halo2curves 0.3.2
 * (Cargo.lock:854-856) and pasta_curves 0.5.1 (Cargo.lock:1126-1128).  This file restates
 * the published algorithms of halo2_proofs/src/arithmetic.rs as recorded in SURVEY.md
 * Appendix A.1 (best_multiexp / multiexp_serial) and A.2 (best_fft /
 * recursive_butterfly_arithmetic); the call sites it stands in for are
 * /root/reference/circuits/src/utils.rs:83-91,105-120 (create_proof) and :63-70 (keygen).
 *
 * Parity pins (tests/test_oracle_pins.py): the 44 Pasta Poseidon vectors held by the
 * reference's tests (field arithmetic), and the params-file sha256 values recorded from
 * the reference's own build in SURVEY.md App. B.2 (BN254 group law + MSM/NTT relation
 * g_lagrange = iFFT_G(g)).  Pallas/Vesta CURVE results have no reference vector:
 * "parity unpinned" for those, self-consistency only.
 *
 * Layout conventions: field element = 4 x u64 little-endian
 * limbs in Montgomery form (R = 2^256); affine point = (x, y), identity = (0, 0);
 * Jacobian point = (x, y, z), identity has z = 0.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "h2_constants.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { fe x, y; } aff;
typedef struct { fe x, y, z; } jac;
typedef const h2o_field_t *F;

/* ------------------------------------------------------------------ field ---------- */
static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int ge_p(const uint64_t t[4], F f) {
  for (int i = 3; i >= 0; i--) {
    if (t[i] > f->p[i]) return 1;
    if (t[i] < f->p[i]) return 0;
  }
  return 1;
}
static inline void sub_p(uint64_t t[4], F f) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)t[i] - f->p[i] - (uint64_t)br;
    t[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
static inline void fe_add(fe *r, const fe *a, const fe *b, F f) {
  u128 c = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; i++) {
    c += (u128)a->l[i] + b->l[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  /* p < 2^255 so no carry out of 256 bits */
  if (ge_p(t, f)) sub_p(t, f);
  memcpy(r->l, t, 32);
}
static inline void fe_sub(fe *r, const fe *a, const fe *b, F f) {
  uint64_t t[4];
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;
    t[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)t[i] + f->p[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  memcpy(r->l, t, 32);
}
static inline void fe_neg(fe *r, const fe *a, F f) {
  fe z = {{0, 0, 0, 0}};
  fe_sub(r, &z, a, f);
}
static inline void fe_dbl(fe *r, const fe *a, F f) { fe_add(r, a, a, f); }

/* CIOS Montgomery product, 4 x 64-bit limbs */
static void fe_mul(fe *r, const fe *a, const fe *b, F f) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * f->inv;
    c = (u128)m * f->p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * f->p[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || ge_p(t, f)) sub_p(t, f);
  memcpy(r->l, t, 32);
}
static inline void fe_sqr(fe *r, const fe *a, F f) { fe_mul(r, a, a, f); }
static inline void fe_one(fe *r, F f) { memcpy(r->l, f->r, 32); }
static inline void fe_from_mont(fe *r, const fe *a, F f) {
  fe one = {{1, 0, 0, 0}};
  fe_mul(r, a, &one, f);
}
static inline void fe_to_mont(fe *r, const fe *a, F f) {
  fe r2;
  memcpy(r2.l, f->r2, 32);
  fe_mul(r, a, &r2, f);
}
/* r = a^e, e given as 4 little-endian limbs (plain integer) */
static void fe_pow(fe *r, const fe *a, const uint64_t e[4], F f) {
  fe acc;
  fe_one(&acc, f);
  for (int i = 255; i >= 0; i--) {
    fe_sqr(&acc, &acc, f);
    if ((e[i / 64] >> (i % 64)) & 1) fe_mul(&acc, &acc, a, f);
  }
  *r = acc;
}
static void fe_inv(fe *r, const fe *a, F f) {
  uint64_t e[4];
  memcpy(e, f->p, 32);
  e[0] -= 2; /* p is odd and > 2: no borrow */
  fe_pow(r, a, e, f);
}

/* ------------------------------------------------------------------ curve ---------- */
typedef struct { F f; fe b; } C;
static C curve_of(int cid) {
  C c;
  c.f = &H2O_FIELDS[H2O_CURVES[cid].base];
  memcpy(c.b.l, H2O_CURVES[cid].b, 32);
  return c;
}
static inline int aff_is_id(const aff *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline int jac_is_id(const jac *p) { return fe_is_zero(&p->z); }
static inline void jac_set_id(jac *p) { memset(p, 0, sizeof(*p)); }
static inline void jac_from_aff(jac *r, const aff *p, F f) {
  if (aff_is_id(p)) { jac_set_id(r); return; }
  r->x = p->x; r->y = p->y; fe_one(&r->z, f);
}
/* dbl-2009-l, a = 0 */
static void jac_double(jac *r, const jac *p, F f) {
  if (jac_is_id(p)) { jac_set_id(r); return; }
  fe a, b, c, d, e, g, t;
  fe_sqr(&a, &p->x, f);
  fe_sqr(&b, &p->y, f);
  fe_sqr(&c, &b, f);
  fe_add(&t, &p->x, &b, f);
  fe_sqr(&t, &t, f);
  fe_sub(&t, &t, &a, f);
  fe_sub(&t, &t, &c, f);
  fe_dbl(&d, &t, f);
  fe_dbl(&e, &a, f);
  fe_add(&e, &e, &a, f);
  fe_sqr(&g, &e, f);
  fe z3;
  fe_mul(&z3, &p->y, &p->z, f);
  fe_dbl(&z3, &z3, f);
  fe x3;
  fe_sub(&x3, &g, &d, f);
  fe_sub(&x3, &x3, &d, f);
  fe_sub(&t, &d, &x3, f);
  fe_mul(&t, &e, &t, f);
  fe_dbl(&c, &c, f); fe_dbl(&c, &c, f); fe_dbl(&c, &c, f);
  fe_sub(&r->y, &t, &c, f);
  r->x = x3;
  r->z = z3;
}
/* add-2007-bl with the exceptional cases resolved */
static void jac_add(jac *r, const jac *p, const jac *q, F f) {
  if (jac_is_id(p)) { *r = *q; return; }
  if (jac_is_id(q)) { *r = *p; return; }
  fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
  fe_sqr(&z1z1, &p->z, f);
  fe_sqr(&z2z2, &q->z, f);
  fe_mul(&u1, &p->x, &z2z2, f);
  fe_mul(&u2, &q->x, &z1z1, f);
  fe_mul(&s1, &p->y, &q->z, f); fe_mul(&s1, &s1, &z2z2, f);
  fe_mul(&s2, &q->y, &p->z, f); fe_mul(&s2, &s2, &z1z1, f);
  if (fe_eq(&u1, &u2)) {
    if (fe_eq(&s1, &s2)) { jac_double(r, p, f); return; }
    jac_set_id(r); return;
  }
  fe_sub(&h, &u2, &u1, f);
  fe_dbl(&i, &h, f); fe_sqr(&i, &i, f);
  fe_mul(&j, &h, &i, f);
  fe_sub(&rr, &s2, &s1, f); fe_dbl(&rr, &rr, f);
  fe_mul(&v, &u1, &i, f);
  fe x3, y3, z3;
  fe_sqr(&x3, &rr, f); fe_sub(&x3, &x3, &j, f); fe_sub(&x3, &x3, &v, f); fe_sub(&x3, &x3, &v, f);
  fe_sub(&t, &v, &x3, f); fe_mul(&y3, &rr, &t, f);
  fe_mul(&t, &s1, &j, f); fe_dbl(&t, &t, f); fe_sub(&y3, &y3, &t, f);
  fe_add(&z3, &p->z, &q->z, f); fe_sqr(&z3, &z3, f); fe_sub(&z3, &z3, &z1z1, f); fe_sub(&z3, &z3, &z2z2, f);
  fe_mul(&z3, &z3, &h, f);
  r->x = x3; r->y = y3; r->z = z3;
}
/* madd-2007-bl */
static void jac_add_mixed(jac *r, const jac *p, const aff *q, F f) {
  if (aff_is_id(q)) { *r = *p; return; }
  if (jac_is_id(p)) { jac_from_aff(r, q, f); return; }
  fe z1z1, u2, s2, h, hh, i, j, rr, v, t;
  fe_sqr(&z1z1, &p->z, f);
  fe_mul(&u2, &q->x, &z1z1, f);
  fe_mul(&s2, &q->y, &p->z, f); fe_mul(&s2, &s2, &z1z1, f);
  if (fe_eq(&p->x, &u2)) {
    if (fe_eq(&p->y, &s2)) { jac_double(r, p, f); return; }
    jac_set_id(r); return;
  }
  fe_sub(&h, &u2, &p->x, f);
  fe_sqr(&hh, &h, f);
  fe_dbl(&i, &hh, f); fe_dbl(&i, &i, f);
  fe_mul(&j, &h, &i, f);
  fe_sub(&rr, &s2, &p->y, f); fe_dbl(&rr, &rr, f);
  fe_mul(&v, &p->x, &i, f);
  fe x3, y3, z3;
  fe_sqr(&x3, &rr, f); fe_sub(&x3, &x3, &j, f); fe_sub(&x3, &x3, &v, f); fe_sub(&x3, &x3, &v, f);
  fe_sub(&t, &v, &x3, f); fe_mul(&y3, &rr, &t, f);
  fe_mul(&t, &p->y, &j, f); fe_dbl(&t, &t, f); fe_sub(&y3, &y3, &t, f);
  fe_add(&z3, &p->z, &h, f); fe_sqr(&z3, &z3, f); fe_sub(&z3, &z3, &z1z1, f); fe_sub(&z3, &z3, &hh, f);
  r->x = x3; r->y = y3; r->z = z3;
}
static void jac_neg(jac *r, const jac *p, F f) { r->x = p->x; r->z = p->z; fe_neg(&r->y, &p->y, f); }
static void jac_to_aff(aff *r, const jac *p, F f) {
  if (jac_is_id(p)) { memset(r, 0, sizeof(*r)); return; }
  fe zi, zi2, zi3;
  fe_inv(&zi, &p->z, f);
  fe_sqr(&zi2, &zi, f);
  fe_mul(&zi3, &zi2, &zi, f);
  fe_mul(&r->x, &p->x, &zi2, f);
  fe_mul(&r->y, &p->y, &zi3, f);
}
/* scalar given as canonical (non-Montgomery) little-endian limbs */
static void jac_mul_repr(jac *r, const jac *p, const uint64_t k[4], F f) {
  jac acc;
  jac_set_id(&acc);
  for (int i = 255; i >= 0; i--) {
    jac_double(&acc, &acc, f);
    if ((k[i / 64] >> (i % 64)) & 1) jac_add(&acc, &acc, p, f);
  }
  *r = acc;
}

/* ----------------------------------------------------- best_multiexp (App. A.1) ---- */
enum { B_NONE = 0, B_AFFINE = 1, B_PROJ = 2 };
typedef struct { int tag; aff a; jac p; } bucket_t;

static inline uint64_t get_at(int seg, int c, const uint64_t repr[4]) {
  /* bits [seg*c, seg*c+c) of the canonical little-endian representation, 0 beyond 256 */
  int skip = seg * c;
  if (skip >= 256) return 0;
  int limb = skip / 64, off = skip % 64;
  uint64_t v = repr[limb] >> off;
  if (off + c > 64 && limb + 1 < 4) v |= repr[limb + 1] << (64 - off);
  return v & ((1ULL << c) - 1);
}

static void multiexp_serial(const fe *coeffs, const aff *bases, size_t n, jac *acc, F fb, F fs) {
  fe *reprs = (fe *)malloc(n * sizeof(fe));
  for (size_t i = 0; i < n; i++) fe_from_mont(&reprs[i], &coeffs[i], fs);
  int c;
  if (n < 4) c = 1;
  else if (n < 32) c = 3;
  else c = (int)ceil(log((double)n));
  int segments = 256 / c + 1;
  size_t nb = ((size_t)1 << c) - 1;
  bucket_t *buckets = (bucket_t *)malloc(nb * sizeof(bucket_t));
  for (int seg = segments - 1; seg >= 0; seg--) {
    for (int k = 0; k < c; k++) jac_double(acc, acc, fb);
    for (size_t b = 0; b < nb; b++) buckets[b].tag = B_NONE;
    for (size_t i = 0; i < n; i++) {
      uint64_t d = get_at(seg, c, reprs[i].l);
      if (d == 0) continue;
      bucket_t *bk = &buckets[d - 1];
      if (bk->tag == B_NONE) { bk->tag = B_AFFINE; bk->a = bases[i]; }
      else if (bk->tag == B_AFFINE) {
        jac t;
        jac_from_aff(&t, &bk->a, fb);
        jac_add_mixed(&bk->p, &t, &bases[i], fb);
        bk->tag = B_PROJ;
      } else {
        jac_add_mixed(&bk->p, &bk->p, &bases[i], fb);
      }
    }
    jac running;
    jac_set_id(&running);
    for (size_t b = nb; b-- > 0;) {
      bucket_t *bk = &buckets[b];
      if (bk->tag == B_AFFINE) jac_add_mixed(&running, &running, &bk->a, fb);
      else if (bk->tag == B_PROJ) jac_add(&running, &running, &bk->p, fb);
      jac_add(acc, acc, &running, fb);
    }
  }
  free(buckets);
  free(reprs);
}

typedef struct { const fe *coeffs; const aff *bases; size_t n; jac acc; F fb, fs; } msm_job_t;
static void *msm_worker(void *arg) {
  msm_job_t *j = (msm_job_t *)arg;
  jac_set_id(&j->acc);
  multiexp_serial(j->coeffs, j->bases, j->n, &j->acc, j->fb, j->fs);
  return NULL;
}

int h2o_best_multiexp(int cid, const uint64_t *coeffs, const uint64_t *bases, size_t n, int threads,
                      uint64_t *out_jac) {
  if (cid < 0 || cid > 2 || threads < 1) return -1;
  F fb = &H2O_FIELDS[H2O_CURVES[cid].base], fs = &H2O_FIELDS[H2O_CURVES[cid].scalar];
  jac total;
  jac_set_id(&total);
  if (n > (size_t)threads && threads > 1) {
    size_t chunk = n / threads;
    size_t njobs = (n + chunk - 1) / chunk;
    msm_job_t *jobs = (msm_job_t *)calloc(njobs, sizeof(msm_job_t));
    pthread_t *th = (pthread_t *)calloc(njobs, sizeof(pthread_t));
    for (size_t k = 0; k < njobs; k++) {
      size_t lo = k * chunk, hi = lo + chunk > n ? n : lo + chunk;
      jobs[k].coeffs = (const fe *)coeffs + lo;
      jobs[k].bases = (const aff *)bases + lo;
      jobs[k].n = hi - lo;
      jobs[k].fb = fb; jobs[k].fs = fs;
      pthread_create(&th[k], NULL, msm_worker, &jobs[k]);
    }
    for (size_t k = 0; k < njobs; k++) {
      pthread_join(th[k], NULL);
      jac_add(&total, &total, &jobs[k].acc, fb);
    }
    free(jobs); free(th);
  } else {
    multiexp_serial((const fe *)coeffs, (const aff *)bases, n, &total, fb, fs);
  }
  memcpy(out_jac, &total, sizeof(total));
  return 0;
}

/* ----------------------------------------------------- best_fft (App. A.2) --------- */
static inline uint32_t bitrev(uint32_t k, uint32_t log_n) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < log_n; i++) { r = (r << 1) | (k & 1); k >>= 1; }
  return r;
}

typedef struct { fe *a; size_t n; size_t stride; const fe *tw; F f; int par; } fft_job_t;
static void recursive_butterfly(fe *a, size_t n, size_t stride, const fe *tw, F f, int par);
static void *fft_worker(void *arg) {
  fft_job_t *j = (fft_job_t *)arg;
  recursive_butterfly(j->a, j->n, j->stride, j->tw, j->f, j->par);
  return NULL;
}
static void recursive_butterfly(fe *a, size_t n, size_t stride, const fe *tw, F f, int par) {
  if (n == 2) {
    fe t = a[1];
    fe_sub(&a[1], &a[0], &t, f);
    fe_add(&a[0], &a[0], &t, f);
    return;
  }
  fe *L = a, *Rr = a + n / 2;
  if (par > 0) {
    pthread_t th;
    fft_job_t j = {L, n / 2, stride * 2, tw, f, par - 1};
    pthread_create(&th, NULL, fft_worker, &j);
    recursive_butterfly(Rr, n / 2, stride * 2, tw, f, par - 1);
    pthread_join(th, NULL);
  } else {
    recursive_butterfly(L, n / 2, stride * 2, tw, f, 0);
    recursive_butterfly(Rr, n / 2, stride * 2, tw, f, 0);
  }
  /* i = 0: twiddle is one, no multiplication */
  fe t = Rr[0];
  fe_sub(&Rr[0], &L[0], &t, f);
  fe_add(&L[0], &L[0], &t, f);
  for (size_t i = 1; i < n / 2; i++) {
    fe_mul(&t, &Rr[i], &tw[i * stride], f);
    fe_sub(&Rr[i], &L[i], &t, f);
    fe_add(&L[i], &L[i], &t, f);
  }
}

int h2o_best_fft(int fid, uint64_t *a_, const uint64_t *omega_, uint32_t log_n, int threads) {
  if (fid < 0 || fid > 3 || log_n > 30 || threads < 1) return -1;
  F f = &H2O_FIELDS[fid];
  fe *a = (fe *)a_;
  size_t n = (size_t)1 << log_n;
  if (n == 1) return 0;
  for (size_t k = 0; k < n; k++) {
    size_t rk = bitrev((uint32_t)k, log_n);
    if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; }
  }
  fe omega;
  memcpy(&omega, omega_, 32);
  fe *tw = (fe *)malloc((n / 2 ? n / 2 : 1) * sizeof(fe));
  fe_one(&tw[0], f);
  for (size_t i = 1; i < n / 2; i++) fe_mul(&tw[i], &tw[i - 1], &omega, f);
  int log_t = 0;
  while ((1 << (log_t + 1)) <= threads) log_t++;
  if (log_n <= (uint32_t)log_t) {
    /* iterative DIT stages, chunk = 2, 4, ..., n; twiddle stride n/chunk */
    size_t chunk = 2, tws = n / 2;
    for (uint32_t s = 0; s < log_n; s++) {
      for (size_t base = 0; base < n; base += chunk) {
        fe *L = a + base, *Rr = a + base + chunk / 2;
        for (size_t i = 0; i < chunk / 2; i++) {
          fe t;
          if (i == 0) t = Rr[0]; else fe_mul(&t, &Rr[i], &tw[i * tws], f);
          fe_sub(&Rr[i], &L[i], &t, f);
          fe_add(&L[i], &L[i], &t, f);
        }
      }
      chunk *= 2; tws /= 2;
    }
  } else {
    recursive_butterfly(a, n, 1, tw, f, log_t);
  }
  free(tw);
  return 0;
}

/* best_fft over group elements (FftGroup for the curve; SURVEY.md row a5, g_to_lagrange) */
int h2o_group_fft(int cid, uint64_t *pts_, const uint64_t *omega_, uint32_t log_n) {
  if (cid < 0 || cid > 2 || log_n > 24) return -1;
  F fb = &H2O_FIELDS[H2O_CURVES[cid].base], fs = &H2O_FIELDS[H2O_CURVES[cid].scalar];
  jac *a = (jac *)pts_;
  size_t n = (size_t)1 << log_n;
  for (size_t k = 0; k < n; k++) {
    size_t rk = bitrev((uint32_t)k, log_n);
    if (k < rk) { jac t = a[k]; a[k] = a[rk]; a[rk] = t; }
  }
  fe omega;
  memcpy(&omega, omega_, 32);
  fe *tw = (fe *)malloc((n / 2 ? n / 2 : 1) * sizeof(fe));
  fe_one(&tw[0], fs);
  for (size_t i = 1; i < n / 2; i++) fe_mul(&tw[i], &tw[i - 1], &omega, fs);
  /* store canonical representations for the scalar multiplications */
  for (size_t i = 0; i < n / 2; i++) fe_from_mont(&tw[i], &tw[i], fs);
  size_t chunk = 2, tws = n / 2;
  for (uint32_t s = 0; s < log_n; s++) {
    for (size_t base = 0; base < n; base += chunk) {
      jac *L = a + base, *Rr = a + base + chunk / 2;
      for (size_t i = 0; i < chunk / 2; i++) {
        jac t, nt;
        if (i == 0) t = Rr[0]; else jac_mul_repr(&t, &Rr[i], tw[i * tws].l, fb);
        jac_neg(&nt, &t, fb);
        jac_add(&Rr[i], &L[i], &nt, fb);
        jac_add(&L[i], &L[i], &t, fb);
      }
    }
    chunk *= 2; tws /= 2;
  }
  free(tw);
  return 0;
}

/* ----------------------------------------------------- helpers for tests / bench --- */
int h2o_field_op(int fid, int op, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  if (fid < 0 || fid > 3) return -1;
  F f = &H2O_FIELDS[fid];
  fe r;
  switch (op) {
    case 0: fe_add(&r, (const fe *)a, (const fe *)b, f); break;
    case 1: fe_sub(&r, (const fe *)a, (const fe *)b, f); break;
    case 2: fe_mul(&r, (const fe *)a, (const fe *)b, f); break;
    case 3: fe_inv(&r, (const fe *)a, f); break;
    case 4: fe_to_mont(&r, (const fe *)a, f); break;
    case 5: fe_from_mont(&r, (const fe *)a, f); break;
    case 6: fe_neg(&r, (const fe *)a, f); break;
    default: return -1;
  }
  memcpy(out, &r, 32);
  return 0;
}

/* sum_i coeffs[i] x^i by Horner's rule from the top coefficient down: halo2_proofs @6b43b6b src/arithmetic.rs
 * `eval_polynomial` (its serial branch; the parallel branch evaluates chunks the same way and recombines them with
 * powers of x, the same field element).  All values in Montgomery form. */
int h2o_eval_polynomial(int fid, const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out) {
  if (fid < 0 || fid > 3) return -1;
  F f = &H2O_FIELDS[fid];
  fe acc, xx;
  memset(&acc, 0, sizeof acc);
  memcpy(&xx, x, 32);
  for (size_t i = n; i-- > 0;) {
    fe_mul(&acc, &acc, &xx, f);
    fe_add(&acc, &acc, (const fe *)coeffs + i, f);
  }
  memcpy(out, &acc, 32);
  return 0;
}

/* n field multiplications a[i]*b[i] (for bulk cross-checks) */
int h2o_field_mul_many(int fid, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
  if (fid < 0 || fid > 3) return -1;
  F f = &H2O_FIELDS[fid];
  for (size_t i = 0; i < n; i++) fe_mul((fe *)out + i, (const fe *)a + i, (const fe *)b + i, f);
  return 0;
}

int h2o_to_affine(int cid, const uint64_t *jac_in, size_t n, uint64_t *aff_out) {
  if (cid < 0 || cid > 2) return -1;
  F fb = &H2O_FIELDS[H2O_CURVES[cid].base];
  for (size_t i = 0; i < n; i++) jac_to_aff((aff *)aff_out + i, (const jac *)jac_in + i, fb);
  return 0;
}

int h2o_is_on_curve(int cid, const uint64_t *aff_in, size_t n) {
  if (cid < 0 || cid > 2) return -1;
  C c = curve_of(cid);
  for (size_t i = 0; i < n; i++) {
    const aff *p = (const aff *)aff_in + i;
    if (aff_is_id(p)) continue;
    fe l, r;
    fe_sqr(&l, &p->y, c.f);
    fe_sqr(&r, &p->x, c.f); fe_mul(&r, &r, &p->x, c.f); fe_add(&r, &r, &c.b, c.f);
    if (!fe_eq(&l, &r)) return 0;
  }
  return 1;
}

/* out = [k] P, k in Montgomery form, P affine */
int h2o_scalar_mul(int cid, const uint64_t *k_mont, const uint64_t *aff_in, uint64_t *out_jac) {
  if (cid < 0 || cid > 2) return -1;
  F fb = &H2O_FIELDS[H2O_CURVES[cid].base], fs = &H2O_FIELDS[H2O_CURVES[cid].scalar];
  fe k;
  fe_from_mont(&k, (const fe *)k_mont, fs);
  jac p, r;
  jac_from_aff(&p, (const aff *)aff_in, fb);
  jac_mul_repr(&r, &p, k.l, fb);
  memcpy(out_jac, &r, sizeof(r));
  return 0;
}

/* r = p + q on Jacobian inputs (test helper for partial-sum combination) */
int h2o_jac_add(int cid, const uint64_t *p, const uint64_t *q, uint64_t *out) {
  if (cid < 0 || cid > 2) return -1;
  F fb = &H2O_FIELDS[H2O_CURVES[cid].base];
  jac r;
  jac_add(&r, (const jac *)p, (const jac *)q, fb);
  memcpy(out, &r, sizeof(r));
  return 0;
}

/* g[i] = [s^i] G for i < n, Jacobian out (ParamsKZG::new) */
typedef struct { int cid; fe s; size_t lo, hi; jac *out; } pow_job_t;
static void *pow_worker(void *arg) {
  pow_job_t *j = (pow_job_t *)arg;
  F fb = &H2O_FIELDS[H2O_CURVES[j->cid].base], fs = &H2O_FIELDS[H2O_CURVES[j->cid].scalar];
  aff g;
  memcpy(g.x.l, H2O_CURVES[j->cid].gx, 32);
  memcpy(g.y.l, H2O_CURVES[j->cid].gy, 32);
  jac gj;
  jac_from_aff(&gj, &g, fb);
  /* cur = s^lo */
  fe cur;
  uint64_t e[4] = {j->lo, 0, 0, 0};
  fe_pow(&cur, &j->s, e, fs);
  for (size_t i = j->lo; i < j->hi; i++) {
    fe k;
    fe_from_mont(&k, &cur, fs);
    jac_mul_repr(&j->out[i], &gj, k.l, fb);
    fe_mul(&cur, &cur, &j->s, fs);
  }
  return NULL;
}
int h2o_powers_of_s(int cid, const uint64_t *s_mont, size_t n, int threads, uint64_t *out_jac) {
  if (cid < 0 || cid > 2 || threads < 1) return -1;
  if ((size_t)threads > n) threads = (int)n;
  pow_job_t *jobs = (pow_job_t *)calloc(threads, sizeof(pow_job_t));
  pthread_t *th = (pthread_t *)calloc(threads, sizeof(pthread_t));
  for (int t = 0; t < threads; t++) {
    jobs[t].cid = cid;
    memcpy(&jobs[t].s, s_mont, 32);
    jobs[t].lo = n * t / threads;
    jobs[t].hi = n * (t + 1) / threads;
    jobs[t].out = (jac *)out_jac;
    pthread_create(&th[t], NULL, pow_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(jobs); free(th);
  return 0;
}

/* scale n Jacobian points by one scalar (Montgomery form) -- the n^-1 of g_to_lagrange */
int h2o_scale_points(int cid, const uint64_t *k_mont, uint64_t *pts, size_t n) {
  if (cid < 0 || cid > 2) return -1;
  F fb = &H2O_FIELDS[H2O_CURVES[cid].base], fs = &H2O_FIELDS[H2O_CURVES[cid].scalar];
  fe k;
  fe_from_mont(&k, (const fe *)k_mont, fs);
  jac *p = (jac *)pts;
  for (size_t i = 0; i < n; i++) jac_mul_repr(&p[i], &p[i], k.l, fb);
  return 0;
}

/* ---- synthetic inputs: SplitMix64 scalars, try-and-increment bases */
static inline uint64_t splitmix(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void synth_value(uint64_t *state, fe *v, F f) {
  for (int i = 0; i < 4; i++) v->l[i] = splitmix(state);
  v->l[3] &= (1ULL << 62) - 1;
  if (ge_p(v->l, f)) sub_p(v->l, f);
}
/* out[i] = Montgomery form of the i-th canonical synthetic value */
int h2o_synth_scalars(int fid, uint64_t seed, size_t n, uint64_t *out) {
  if (fid < 0 || fid > 3) return -1;
  F f = &H2O_FIELDS[fid];
  uint64_t st = seed;
  for (size_t i = 0; i < n; i++) {
    fe v;
    synth_value(&st, &v, f);
    fe_to_mont((fe *)out + i, &v, f);
  }
  return 0;
}
/* generic square root by Tonelli-Shanks; returns 0 when a is a non-residue */
static int fe_sqrt(fe *r, const fe *a, F f) {
  if (fe_is_zero(a)) { *r = *a; return 1; }
  uint32_t S = f->two_adicity;
  /* q = (p-1) >> S ; (q+1)/2 = (q >> 1) + 1 since q odd */
  uint64_t q[4], pm1[4];
  memcpy(pm1, f->p, 32);
  pm1[0] -= 1;
  for (int i = 0; i < 4; i++) {
    int sh = S % 64, w = S / 64;
    uint64_t lo = (i + w < 4) ? pm1[i + w] : 0, hi = (i + w + 1 < 4) ? pm1[i + w + 1] : 0;
    q[i] = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
  }
  uint64_t qh[4]; /* (q-1)/2 */
  for (int i = 0; i < 4; i++) qh[i] = (q[i] >> 1) | (i < 3 ? q[i + 1] << 63 : 0);
  fe w, x, b, z;
  fe_pow(&w, a, qh, f);            /* a^((q-1)/2) */
  fe_mul(&x, a, &w, f);            /* a^((q+1)/2) */
  fe_mul(&b, &x, &w, f);           /* a^q */
  memcpy(&z, f->root_of_unity, 32); /* generator^q: order 2^S */
  uint32_t v = S;
  fe one;
  fe_one(&one, f);
  while (!fe_eq(&b, &one)) {
    uint32_t k = 0;
    fe t = b;
    while (!fe_eq(&t, &one)) { fe_sqr(&t, &t, f); k++; if (k == v) return 0; }
    fe wv = z;
    for (uint32_t i = 0; i + k + 1 < v; i++) fe_sqr(&wv, &wv, f);
    fe_sqr(&z, &wv, f);
    fe_mul(&b, &b, &z, f);
    fe_mul(&x, &x, &wv, f);
    v = k;
  }
  *r = x;
  return 1;
}
typedef struct { int cid; uint64_t seed; size_t lo, hi; aff *out; } base_job_t;
static void *base_worker(void *arg) {
  base_job_t *j = (base_job_t *)arg;
  C c = curve_of(j->cid);
  fe one_c = {{1, 0, 0, 0}}, one_m;
  fe_to_mont(&one_m, &one_c, c.f);
  for (size_t i = j->lo; i < j->hi; i++) {
    /* independent stream per point so generation can be threaded: seed ^ golden*i */
    uint64_t st = j->seed + 0xD1B54A32D192ED03ULL * (uint64_t)(i + 1);
    fe xc, x, y, rhs;
    synth_value(&st, &xc, c.f);
    fe_to_mont(&x, &xc, c.f);
    for (;;) {
      fe_sqr(&rhs, &x, c.f); fe_mul(&rhs, &rhs, &x, c.f); fe_add(&rhs, &rhs, &c.b, c.f);
      if (fe_sqrt(&y, &rhs, c.f)) {
        fe chk;
        fe_sqr(&chk, &y, c.f);
        if (fe_eq(&chk, &rhs)) break;
      }
      fe_add(&x, &x, &one_m, c.f);
    }
    fe yc;
    fe_from_mont(&yc, &y, c.f);
    if (yc.l[0] & 1) fe_neg(&y, &y, c.f); /* even canonical parity */
    j->out[i].x = x;
    j->out[i].y = y;
  }
  return NULL;
}
int h2o_synth_bases(int cid, uint64_t seed, size_t n, int threads, uint64_t *out_aff) {
  if (cid < 0 || cid > 2 || threads < 1) return -1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  base_job_t *jobs = (base_job_t *)calloc(threads, sizeof(base_job_t));
  pthread_t *th = (pthread_t *)calloc(threads, sizeof(pthread_t));
  for (int t = 0; t < threads; t++) {
    jobs[t].cid = cid; jobs[t].seed = seed;
    jobs[t].lo = n * t / threads; jobs[t].hi = n * (t + 1) / threads;
    jobs[t].out = (aff *)out_aff;
    pthread_create(&th[t], NULL, base_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(jobs); free(th);
  return 0;
}
