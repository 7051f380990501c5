// h2_circuits.hpp -- the reference's three circuits as data (host): constraint systems, witness layouts, the
// verifying key's `{:?}` digest.
//
// Restates, for the C side of the product surface (h2_prover.hip), what halo2_prover_amd/prover.py holds in Python:
//   /root/reference/circuits/src/arithmetic_circuit.rs:187-267   (3 advice, 5 fixed, 1 instance; one gate)
//   /root/reference/circuits/src/collatz.rs:26-207               (3 advice, 2 selectors, 4 gates, 32 regions)
//   /root/reference/circuits/src/poseidon_circuit.rs:68-149 with halo2_gadgets' Pow5 chip (the vendored copy at
//       circuits/src/poseidon/pow5.rs:230-272,433-592) and the constant generation of
//       circuits/src/poseidon/primitives/grain.rs:52-137, mds.rs:5-102
// and, from the un-vendored halo2_proofs @6b43b6b: Assembly::copy of the permutation keygen and
// the Debug rendering of the pinned verifying key whose Blake2b digest opens every transcript.
#pragma once
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "h2_host.hpp"

namespace h2 {
namespace plonk {

// ---- halo2 Expression trees -----------------------------------------------------------------------------------
struct Expr;
using E = std::shared_ptr<const Expr>;
struct Expr {
  enum Kind { Const, Advice, Fixed, Instance, Neg, Sum, Prod, Scaled } kind;
  Fr c;                 // Const value / Scaled factor
  int qi = 0, col = 0, rot = 0;
  E a, b;
};
inline E e_const(const Fr& v) { auto e = std::make_shared<Expr>(); e->kind = Expr::Const; e->c = v; return e; }
inline E e_query(Expr::Kind k, int qi, int col, int rot) {
  auto e = std::make_shared<Expr>();
  e->kind = k; e->qi = qi; e->col = col; e->rot = rot;
  return e;
}
inline E e_adv(int qi, int col, int rot) { return e_query(Expr::Advice, qi, col, rot); }
inline E e_fix(int qi, int col, int rot) { return e_query(Expr::Fixed, qi, col, rot); }
inline E e_bin(Expr::Kind k, E a, E b) { auto e = std::make_shared<Expr>(); e->kind = k; e->a = std::move(a); e->b = std::move(b); return e; }
inline E e_neg(E a) { auto e = std::make_shared<Expr>(); e->kind = Expr::Neg; e->a = std::move(a); return e; }
inline E e_sum(E a, E b) { return e_bin(Expr::Sum, std::move(a), std::move(b)); }
inline E e_sub(E a, E b) { return e_sum(std::move(a), e_neg(std::move(b))); }     // Rust's a - b
inline E e_prod(E a, E b) { return e_bin(Expr::Prod, std::move(a), std::move(b)); }
inline E e_scaled(E a, const Fr& c) { auto e = std::make_shared<Expr>(); e->kind = Expr::Scaled; e->a = std::move(a); e->c = c; return e; }

inline void expr_debug(const E& e, std::string& out) {
  char buf[160];
  switch (e->kind) {
    case Expr::Const: out += "Constant(0x" + e->c.hex64() + ")"; break;
    case Expr::Advice: case Expr::Fixed: case Expr::Instance:
      snprintf(buf, sizeof buf, "%s { query_index: %d, column_index: %d, rotation: Rotation(%d) }",
               e->kind == Expr::Advice ? "Advice" : e->kind == Expr::Fixed ? "Fixed" : "Instance", e->qi, e->col, e->rot);
      out += buf;
      break;
    case Expr::Neg: out += "Negated("; expr_debug(e->a, out); out += ")"; break;
    case Expr::Sum: out += "Sum("; expr_debug(e->a, out); out += ", "; expr_debug(e->b, out); out += ")"; break;
    case Expr::Prod: out += "Product("; expr_debug(e->a, out); out += ", "; expr_debug(e->b, out); out += ")"; break;
    case Expr::Scaled: out += "Scaled("; expr_debug(e->a, out); out += ", 0x" + e->c.hex64() + ")"; break;
  }
}

inline Fr expr_eval(const E& e, const std::vector<Fr>& adv, const std::vector<Fr>& fix, const std::vector<Fr>& inst) {
  switch (e->kind) {
    case Expr::Const: return e->c;
    case Expr::Advice: return adv[e->qi];
    case Expr::Fixed: return fix[e->qi];
    case Expr::Instance: return inst[e->qi];
    case Expr::Neg: return -expr_eval(e->a, adv, fix, inst);
    case Expr::Sum: return expr_eval(e->a, adv, fix, inst) + expr_eval(e->b, adv, fix, inst);
    case Expr::Prod: return expr_eval(e->a, adv, fix, inst) * expr_eval(e->b, adv, fix, inst);
    case Expr::Scaled: return expr_eval(e->a, adv, fix, inst) * e->c;
  }
  return Fr::zero();
}

// ---- circuits ---------------------------------------------------------------------------------------------------
enum ColKind { ADVICE = 0, FIXED = 1, INSTANCE = 2 };
using ColRef = std::pair<ColKind, int>;
using Cell = std::pair<ColRef, uint32_t>;                 // (column, row)
using SparseCol = std::map<uint32_t, Fr>;                 // the rows a synthesis wrote; everything else is zero

struct Circuit {
  std::string name;
  int num_advice = 0, num_fixed = 0, num_instance = 0, num_selectors = 0, degree = 3;
  std::vector<ColRef> permutation_columns;
  std::vector<std::pair<int, int>> advice_queries, fixed_queries, instance_queries;   // (column, rotation)
  std::vector<int> constants;                             // fixed columns enabled for constants
  std::vector<E> gates;
  virtual ~Circuit() = default;
  virtual std::vector<SparseCol> synthesize_fixed() const = 0;
  virtual std::vector<SparseCol> synthesize_advice() const = 0;   // needs the witness
  virtual std::vector<std::pair<Cell, Cell>> copy_constraints() const = 0;
  int blinding_factors() const {
    std::map<int, int> per_col;
    int mx = 0;
    for (auto& q : advice_queries) mx = std::max(mx, ++per_col[q.first]);
    return std::max(3, mx) + 2;
  }
};

inline const char* kind_name(ColKind k) { return k == ADVICE ? "Advice" : k == FIXED ? "Fixed" : "Instance"; }

// -- arithmetic_circuit.rs: advice l, r, o; fixed sm, sl, sr, so, sc (creation order :196-200); instance PI
struct ArithmeticCircuit : Circuit {
  bool has_witness = false;
  Fr x, y, constant;
  enum { SM = 0, SL, SR, SO, SC };
  ArithmeticCircuit() {
    name = "arithmetic";
    num_advice = 3; num_fixed = 5; num_instance = 1; num_selectors = 0; degree = 3;
    permutation_columns = {{ADVICE, 0}, {ADVICE, 1}, {ADVICE, 2}, {INSTANCE, 0}};
    advice_queries = {{0, 0}, {1, 0}, {2, 0}};
    fixed_queries = {{1, 0}, {2, 0}, {3, 0}, {0, 0}, {4, 0}};
    instance_queries = {{0, 0}};
    E l = e_adv(0, 0, 0), r = e_adv(1, 1, 0), o = e_adv(2, 2, 0);
    E sl = e_fix(0, 1, 0), sr = e_fix(1, 2, 0), so = e_fix(2, 3, 0), sm = e_fix(3, 0, 0), sc = e_fix(4, 4, 0);
    // :216  l*sl + r*sr + l*r*sm + (o*so*(-1)) + sc
    gates = {e_sum(e_sum(e_sum(e_sum(e_prod(l, sl), e_prod(r, sr)), e_prod(e_prod(l, r), sm)),
                         e_scaled(e_prod(o, so), -Fr::one())), sc)};
  }
  std::vector<SparseCol> synthesize_fixed() const override {
    std::vector<SparseCol> f(5);
    for (uint32_t row = 0; row < 3; row++) f[SM][row] = f[SO][row] = Fr::one();
    f[SL][3] = f[SR][3] = f[SO][3] = Fr::one();
    return f;
  }
  std::vector<SparseCol> synthesize_advice() const override {
    const Fr xx = x * x, yy = y * y, prod = xx * yy;
    const Fr rows[4][3] = {{x, x, xx}, {y, y, yy}, {xx, yy, prod}, {prod, constant, prod + constant}};
    std::vector<SparseCol> a(3);
    for (uint32_t i = 0; i < 4; i++)
      for (int j = 0; j < 3; j++) a[j][i] = rows[i][j];
    return a;
  }
  std::vector<std::pair<Cell, Cell>> copy_constraints() const override {
    auto a = [](int col, uint32_t row) { return Cell{{ADVICE, col}, row}; };
    return {{a(0, 0), a(1, 0)}, {a(0, 1), a(1, 1)}, {a(2, 0), a(0, 2)}, {a(2, 1), a(1, 2)}, {a(2, 2), a(0, 3)},
            {a(1, 3), Cell{{INSTANCE, 0}, 0}}, {a(2, 3), Cell{{INSTANCE, 0}, 1}}};
  }
};

// -- Grain LFSR of the Poseidon reference parameter generation (poseidon/primitives/grain.rs:52-137)
class Grain {
 public:
  Grain(int num_bits, int t, int r_f, int r_p) : num_bits_(num_bits) {
    const int widths[6] = {2, 4, 12, 12, 10, 10}, values[6] = {1, 0, num_bits, t, r_f, r_p};
    for (int k = 0; k < 6; k++)
      for (int i = 0; i < widths[k]; i++) st_.push_back((values[k] >> (widths[k] - 1 - i)) & 1);
    for (int i = 0; i < 30; i++) st_.push_back(1);
    for (int i = 0; i < 160; i++) raw();
  }
  // num_bits bits, most significant first, as 32 little-endian bytes
  void take(uint8_t out[32]) {
    memset(out, 0, 32);
    for (int i = num_bits_ - 1; i >= 0; i--)
      if (bit()) out[i >> 3] |= (uint8_t)(1u << (i & 7));
  }

 private:
  int raw() {
    const int b = st_[pos_ + 62] ^ st_[pos_ + 51] ^ st_[pos_ + 38] ^ st_[pos_ + 23] ^ st_[pos_ + 13] ^ st_[pos_];
    st_.push_back((uint8_t)b);
    pos_++;
    return b;
  }
  int bit() {
    for (;;) {
      if (raw()) return raw();
      raw();
    }
  }
  std::vector<uint8_t> st_;
  size_t pos_ = 0;
  int num_bits_;
};

struct PoseidonConstants {
  std::vector<std::array<Fr, 3>> rcs;     // 68 rounds
  Fr mds[3][3], minv[3][3];
};
// round constants, Cauchy MDS and its inverse over bn256::Fr for t = 3, R_F = 8, R_P = 60 (primitives.rs:57-84,
// mds.rs:5-102; poseidon_circuit.rs:19-25,129-149)
inline const PoseidonConstants& poseidon_constants() {
  static const PoseidonConstants pc = [] {
    PoseidonConstants k;
    const int t = 3, r_f = 8, r_p = 60;
    Grain g(254, t, r_f, r_p);
    uint8_t b[32];
    for (int r = 0; r < r_f + r_p; r++) {
      std::array<Fr, 3> row;
      int have = 0;
      while (have < t) {
        g.take(b);
        Fr v;
        if (Fr::from_le_bytes_canonical(b, &v)) row[have++] = v;   // rejection sampling: values >= p are skipped
      }
      k.rcs.push_back(row);
    }
    Fr xs[3], ys[3];
    for (;;) {
      Fr vals[6];
      for (int i = 0; i < 6; i++) {
        g.take(b);
        vals[i] = Fr::from_le_bytes_reduce(b);                     // here the bits are reduced, not rejected
      }
      bool distinct = true;
      for (int i = 0; i < 6; i++)
        for (int j = i + 1; j < 6; j++)
          if (vals[i] == vals[j]) distinct = false;
      if (distinct) {
        for (int i = 0; i < 3; i++) { xs[i] = vals[i]; ys[i] = vals[3 + i]; }
        break;
      }
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) k.mds[i][j] = (xs[i] + ys[j]).inv();
    auto lag = [](const Fr* pts, int j, const Fr& x) {
      Fr acc = Fr::one();
      for (int m = 0; m < 3; m++)
        if (m != j) acc = acc * (x - pts[m]) * (pts[j] - pts[m]).inv();
      return acc;
    };
    Fr nys[3];
    for (int i = 0; i < 3; i++) nys[i] = -ys[i];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) k.minv[i][j] = (xs[j] - nys[i]) * lag(xs, j, nys[i]) * lag(nys, i, xs[j]);
    return k;
  }();
  return pc;
}

// -- poseidon_circuit.rs (:68-123) with the Pow5 chip: WIDTH 3, RATE 2, L 2.  advice state0..2 = 0..2,
// partial_sbox = 3; fixed rc_a = 0..2, rc_b = 3..5, selector columns 6..8
struct PoseidonCircuit : Circuit {
  bool has_witness = false;
  Fr message[2];
  PoseidonCircuit() {
    name = "poseidon";
    num_advice = 4; num_fixed = 9; num_instance = 1; num_selectors = 3; degree = 6;
    permutation_columns = {{INSTANCE, 0}, {FIXED, 3}, {ADVICE, 0}, {ADVICE, 1}, {ADVICE, 2}, {FIXED, 4}, {FIXED, 5}};
    advice_queries = {{0, 0}, {1, 0}, {2, 0}, {0, 1}, {1, 1}, {2, 1}, {3, 0}, {2, -1}, {0, -1}, {1, -1}};
    fixed_queries = {{3, 0}, {4, 0}, {5, 0}, {0, 0}, {1, 0}, {2, 0}, {6, 0}, {7, 0}, {8, 0}};
    instance_queries = {{0, 0}};
    constants = {3};
    const PoseidonConstants& pc = poseidon_constants();
    E s_cur[3], s_next[3], rc_a[3], rc_b[3];
    for (int i = 0; i < 3; i++) {
      s_cur[i] = e_adv(i, i, 0);
      s_next[i] = e_adv(3 + i, i, 1);
      rc_b[i] = e_fix(i, 3 + i, 0);
      rc_a[i] = e_fix(3 + i, i, 0);
    }
    E ps = e_adv(6, 3, 0);
    E s_prev[3] = {e_adv(8, 0, -1), e_adv(9, 1, -1), e_adv(7, 2, -1)};
    E s_full = e_fix(6, 6, 0), s_partial = e_fix(7, 7, 0), s_pad = e_fix(8, 8, 0);
    auto pow5 = [](const E& v) { E v2 = e_prod(v, v); return e_prod(e_prod(v2, v2), v); };
    for (int nx = 0; nx < 3; nx++) {
      E terms[3];
      for (int i = 0; i < 3; i++) terms[i] = e_scaled(pow5(e_sum(s_cur[i], rc_a[i])), pc.mds[nx][i]);
      gates.push_back(e_prod(s_full, e_sub(e_sum(e_sum(terms[0], terms[1]), terms[2]), s_next[nx])));
    }
    auto mid = [&](int i) {
      E acc = e_scaled(ps, pc.mds[i][0]);
      for (int c = 1; c < 3; c++) acc = e_sum(acc, e_scaled(e_sum(s_cur[c], rc_a[c]), pc.mds[i][c]));
      return acc;
    };
    auto nxt = [&](int i) {
      return e_sum(e_sum(e_scaled(s_next[0], pc.minv[i][0]), e_scaled(s_next[1], pc.minv[i][1])),
                   e_scaled(s_next[2], pc.minv[i][2]));
    };
    std::vector<E> partial = {e_sub(pow5(e_sum(s_cur[0], rc_a[0])), ps), e_sub(pow5(e_sum(mid(0), rc_b[0])), nxt(0))};
    for (int i = 1; i < 3; i++) partial.push_back(e_sub(e_sum(mid(i), rc_b[i]), nxt(i)));
    for (auto& g : partial) gates.push_back(e_prod(s_partial, g));
    std::vector<E> pad = {e_sub(e_sum(s_prev[0], s_cur[0]), s_next[0]), e_sub(e_sum(s_prev[1], s_cur[1]), s_next[1]),
                          e_sub(s_prev[2], s_next[2])};
    for (auto& g : pad) gates.push_back(e_prod(s_pad, g));
  }
  static Fr capacity() { return Fr::from_hex("0x20000000000000000"); }       // 2 << 64 (ConstantLength<2>)
  // the 1 + 4 + 30 + 4 state rows of the permutation and the partial rounds' S-box column
  void permutation_rows(std::vector<std::array<Fr, 3>>& rows, std::map<int, Fr>& sbox) const {
    const PoseidonConstants& pc = poseidon_constants();
    auto p5 = [](const Fr& v) { const Fr v2 = v * v; return v2 * v2 * v; };
    auto mix = [&](const std::array<Fr, 3>& v) {
      std::array<Fr, 3> o;
      for (int i = 0; i < 3; i++) o[i] = pc.mds[i][0] * v[0] + pc.mds[i][1] * v[1] + pc.mds[i][2] * v[2];
      return o;
    };
    std::array<Fr, 3> st = {message[0], message[1], capacity()};
    rows.push_back(st);
    for (int r = 0; r < 4; r++) {
      st = mix({p5(st[0] + pc.rcs[r][0]), p5(st[1] + pc.rcs[r][1]), p5(st[2] + pc.rcs[r][2])});
      rows.push_back(st);
    }
    for (int r = 0; r < 30; r++) {
      const int rnd = 4 + 2 * r;
      const std::array<Fr, 3> r0 = {p5(st[0] + pc.rcs[rnd][0]), st[1] + pc.rcs[rnd][1], st[2] + pc.rcs[rnd][2]};
      sbox[4 + r] = r0[0];
      const std::array<Fr, 3> m = mix(r0);
      st = mix({p5(m[0] + pc.rcs[rnd + 1][0]), m[1] + pc.rcs[rnd + 1][1], m[2] + pc.rcs[rnd + 1][2]});
      rows.push_back(st);
    }
    for (int r = 0; r < 4; r++) {
      st = mix({p5(st[0] + pc.rcs[64 + r][0]), p5(st[1] + pc.rcs[64 + r][1]), p5(st[2] + pc.rcs[64 + r][2])});
      rows.push_back(st);
    }
  }
  Fr output() const {   // poseidon::Hash::<_, S, ConstantLength<2>, 3, 2>::init().hash(message) (poseidon_circuit.rs:292-299)
    std::vector<std::array<Fr, 3>> rows;
    std::map<int, Fr> sbox;
    permutation_rows(rows, sbox);
    return rows.back()[0];
  }
  std::vector<SparseCol> synthesize_advice() const override {
    std::vector<SparseCol> adv(4);
    const Fr cap = capacity();
    adv[0][0] = message[0]; adv[1][0] = message[1];
    adv[2][1] = cap;
    adv[2][2] = cap;
    adv[0][3] = message[0]; adv[1][3] = message[1];
    adv[0][4] = message[0]; adv[1][4] = message[1]; adv[2][4] = cap;
    std::vector<std::array<Fr, 3>> rows;
    std::map<int, Fr> sbox;
    permutation_rows(rows, sbox);
    for (size_t off = 0; off < rows.size(); off++)
      for (int i = 0; i < 3; i++) adv[i][5 + (uint32_t)off] = rows[off][i];
    for (auto& kv : sbox) adv[3][5 + (uint32_t)kv.first] = kv.second;
    return adv;
  }
  std::vector<SparseCol> synthesize_fixed() const override {
    const PoseidonConstants& pc = poseidon_constants();
    std::vector<SparseCol> f(9);
    f[3][2] = capacity();
    f[8][3] = Fr::one();
    for (int r = 0; r < 4; r++) {
      for (int i = 0; i < 3; i++) f[i][5 + r] = pc.rcs[r][i];
      f[6][5 + r] = Fr::one();
    }
    for (int r = 0; r < 30; r++) {
      const int off = 4 + r, rnd = 4 + 2 * r;
      for (int i = 0; i < 3; i++) {
        f[i][5 + off] = pc.rcs[rnd][i];
        f[3 + i][5 + off] = pc.rcs[rnd + 1][i];
      }
      f[7][5 + off] = Fr::one();
    }
    for (int r = 0; r < 4; r++) {
      for (int i = 0; i < 3; i++) f[i][5 + 34 + r] = pc.rcs[64 + r][i];
      f[6][5 + 34 + r] = Fr::one();
    }
    return f;
  }
  std::vector<std::pair<Cell, Cell>> copy_constraints() const override {
    auto a = [](int col, uint32_t row) { return Cell{{ADVICE, col}, row}; };
    std::vector<std::pair<Cell, Cell>> out;
    for (int i = 0; i < 3; i++) out.push_back({Cell{{FIXED, 3}, (uint32_t)i}, a(i, 1)});
    for (int i = 0; i < 3; i++) out.push_back({a(i, 2), a(i, 1)});
    for (int i = 0; i < 2; i++) out.push_back({a(i, 3), a(i, 0)});
    for (int i = 0; i < 3; i++) out.push_back({a(i, 5), a(i, 4)});
    out.push_back({a(0, 43), Cell{{INSTANCE, 0}, 0}});
    return out;
  }
};

// -- collatz.rs: advice witness, is_odd, is_one; selectors final_entry (0) / selector (1) compressed into fixed
// columns 0 / 1; four gates, degree 4; region i of the SimpleFloorPlanner starts at row i(i+3)/2 and uses offsets
// i, i+1 (:119-134, :180-198); the last at 527
struct CollatzCircuit : Circuit {
  std::vector<Fr> x;                 // 32 values (padded with 1s, collatz.rs:256-261)
  std::vector<uint64_t> x_u64;       // the same as integers (parity / equality with one)
  CollatzCircuit() {
    name = "collatz";
    num_advice = 3; num_fixed = 2; num_instance = 0; num_selectors = 2; degree = 4;
    permutation_columns = {{ADVICE, 0}};
    advice_queries = {{0, 0}, {0, 1}, {1, 0}, {2, 0}};
    fixed_queries = {{0, 0}, {1, 0}};
    E xq = e_adv(0, 0, 0), y = e_adv(1, 0, 1), is_odd = e_adv(2, 1, 0), is_one = e_adv(3, 2, 0);
    E fin = e_fix(0, 0, 0), sel = e_fix(1, 1, 0), one = e_const(Fr::one());
    gates = {
        e_prod(sel, e_prod(e_sub(one, is_odd), e_sub(xq, e_prod(e_const(Fr::from_u64(2)), y)))),
        e_prod(e_prod(sel, e_sub(one, is_one)), e_prod(is_odd, e_sub(e_sum(e_prod(e_const(Fr::from_u64(3)), xq), one), y))),
        e_prod(e_prod(sel, is_one), e_sum(e_sub(xq, y), e_sub(xq, one))),
        e_prod(fin, e_sub(one, xq)),
    };
  }
  void set_sequence(const std::vector<uint64_t>& seq) {
    x_u64.assign(32, 1);
    for (size_t i = 0; i < seq.size() && i < 32; i++) x_u64[i] = seq[i];
    x.clear();
    for (uint64_t v : x_u64) x.push_back(Fr::from_u64(v));
  }
  std::vector<SparseCol> synthesize_advice() const override {
    std::vector<SparseCol> adv(3);
    for (uint32_t i = 0; i < 31; i++) {
      const uint32_t row = i * (i + 3) / 2 + i;
      adv[0][row] = x[i];
      adv[0][row + 1] = x[i + 1];
      adv[1][row] = Fr::from_u64(x_u64[i] & 1);
      adv[2][row] = Fr::from_u64(x_u64[i] == 1 ? 1 : 0);
    }
    adv[0][527 + 31] = x[31];
    return adv;
  }
  std::vector<SparseCol> synthesize_fixed() const override {
    std::vector<SparseCol> f(2);
    for (uint32_t i = 0; i < 31; i++) f[1][i * (i + 3) / 2 + i] = Fr::one();
    f[0][527 + 31] = Fr::one();
    return f;
  }
  std::vector<std::pair<Cell, Cell>> copy_constraints() const override { return {}; }
};

// ---- permutation keygen: Assembly::copy of halo2_proofs/src/plonk/permutation/keygen.rs ---------
// returns the cells that do not map to themselves: (column index in the permutation, row) -> (column index, row)
inline std::map<std::pair<int, uint32_t>, std::pair<int, uint32_t>> permutation_mapping(const Circuit& c) {
  using PC = std::pair<int, uint32_t>;
  std::map<ColRef, int> index;
  for (size_t i = 0; i < c.permutation_columns.size(); i++) index[c.permutation_columns[i]] = (int)i;
  std::map<PC, PC> mapping, aux;
  std::map<PC, uint32_t> sizes;
  for (auto& cc : c.copy_constraints()) {
    const PC left{index.at(cc.first.first), cc.first.second}, right{index.at(cc.second.first), cc.second.second};
    for (const PC& cell : {left, right})
      if (!mapping.count(cell)) {
        mapping[cell] = cell;
        aux[cell] = cell;
        sizes[cell] = 1;
      }
    if (aux[left] == aux[right]) continue;
    PC big = aux[left], small = aux[right];
    if (sizes[big] < sizes[small]) std::swap(big, small);
    sizes[big] += sizes[small];
    PC i = small;
    for (;;) {
      aux[i] = big;
      i = mapping[i];
      if (i == small) break;
    }
    std::swap(mapping[left], mapping[right]);
  }
  return mapping;
}

// ---- format!("{:?}", vk.pinned()) of halo2_proofs @6b43b6b ---------------------------------------
inline std::string point_debug(const G1& p) {
  if (p.inf) return "Infinity";
  return "(0x" + p.x.hex64() + ", 0x" + p.y.hex64() + ")";
}
inline std::string modulus_hex(bool base) {
  // "0x" + the modulus in lower-case hex without leading zeros (both start with the digit 3)
  uint8_t b[32];
  for (int i = 0; i < 8; i++) {
    const uint32_t w = base ? BN254_FQ::P(i) : BN254_FR::P(i);
    memcpy(b + 4 * i, &w, 4);
  }
  static const char* d = "0123456789abcdef";
  std::string s;
  for (int i = 31; i >= 0; i--) {
    s += d[b[i] >> 4];
    s += d[b[i] & 15];
  }
  size_t nz = s.find_first_not_of('0');
  return "0x" + s.substr(nz);
}
inline std::string vk_debug_string(const Circuit& c, uint32_t k, uint32_t extended_k, const Fr& omega,
                                   const std::vector<G1>& fixed_commitments, const std::vector<G1>& sigma_commitments) {
  char buf[256];
  auto col = [&](ColKind kind, int i) {
    snprintf(buf, sizeof buf, "Column { index: %d, column_type: %s }", i, kind_name(kind));
    return std::string(buf);
  };
  auto queries = [&](ColKind kind, const std::vector<std::pair<int, int>>& qs) {
    std::string s;
    for (size_t i = 0; i < qs.size(); i++) {
      if (i) s += ", ";
      s += "(" + col(kind, qs[i].first) + ", Rotation(" + std::to_string(qs[i].second) + "))";
    }
    return s;
  };
  std::string s = "PinnedVerificationKey { base_modulus: \"" + modulus_hex(true) + "\", scalar_modulus: \"" +
                  modulus_hex(false) + "\", domain: PinnedEvaluationDomain { k: " + std::to_string(k) +
                  ", extended_k: " + std::to_string(extended_k) + ", omega: 0x" + omega.hex64() + " }, ";
  s += "cs: PinnedConstraintSystem { num_fixed_columns: " + std::to_string(c.num_fixed) +
       ", num_advice_columns: " + std::to_string(c.num_advice) + ", num_instance_columns: " +
       std::to_string(c.num_instance) + ", num_selectors: " + std::to_string(c.num_selectors) + ", gates: [";
  for (size_t i = 0; i < c.gates.size(); i++) {
    if (i) s += ", ";
    expr_debug(c.gates[i], s);
  }
  s += "], advice_queries: [" + queries(ADVICE, c.advice_queries) + "], instance_queries: [" +
       queries(INSTANCE, c.instance_queries) + "], fixed_queries: [" + queries(FIXED, c.fixed_queries) +
       "], permutation: Argument { columns: [";
  for (size_t i = 0; i < c.permutation_columns.size(); i++) {
    if (i) s += ", ";
    s += col(c.permutation_columns[i].first, c.permutation_columns[i].second);
  }
  s += "] }, lookups: [], constants: [";
  for (size_t i = 0; i < c.constants.size(); i++) {
    if (i) s += ", ";
    s += col(FIXED, c.constants[i]);
  }
  s += "], minimum_degree: None }, fixed_commitments: [";
  for (size_t i = 0; i < fixed_commitments.size(); i++) {
    if (i) s += ", ";
    s += point_debug(fixed_commitments[i]);
  }
  s += "], permutation: VerifyingKey { commitments: [";
  for (size_t i = 0; i < sigma_commitments.size(); i++) {
    if (i) s += ", ";
    s += point_debug(sigma_commitments[i]);
  }
  s += "] } }";
  return s;
}
// vk.transcript_repr: Blake2b-512("Halo2-Verify-Key", len as u64 LE || the string) reduced mod r
inline Fr vk_transcript_repr(const std::string& s) {
  Blake2b h("Halo2-Verify-Key");
  const uint64_t len = s.size();
  h.update(&len, 8);
  h.update(s.data(), s.size());
  uint8_t d[64];
  h.digest(d);
  return Fr::from_le_bytes_wide(d);
}

}  // namespace plonk
}  // namespace h2
