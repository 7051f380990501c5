"""The verifier (halo2_prover_amd/verifier.py, mirror of utils.rs:125-158 and wasm.rs:125-179).

* CPU: the BN254 pairing's own properties (bilinearity, order, non-degeneracy).  The reference holds no vector for the
  pairing on its own -- "parity unpinned" for that piece; what pins the verifier as a whole are the proofs recorded
  from the reference's build, below.
* GPU: the proofs the reference's build produced (tests/golden/, sha256 in SURVEY.md App. B.2) are ACCEPTED -- all three
  circuits, GWC and SHPLONK; proofs at the recorded k = 11 and k = 16 hashes are accepted; proofs made
  with OsRng are accepted; the same proofs with one byte flipped, a wrong public input, a truncated or empty proof or
  another circuit's proof are REJECTED (the reference traps there, utils.rs:150-157; here `False`).
"""
import hashlib
import os

import pytest

import pyref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ARITH_INPUT = '{"x":6,"y":9,"constant":7,"z":2923}'
POSEIDON_HASH_1_2 = 0x152E960B5C9C8A624B2CDF4855250E8A54EE074254281310DC4A9704F78C1917
POSEIDON_INPUT = '{"x":[1,2],"output":"0x%064x"}' % POSEIDON_HASH_1_2
COLLATZ_SEQ = [9, 28, 14, 7, 22, 11, 34, 17, 52, 26, 13, 40, 20, 10, 5, 16, 8, 4, 2, 1]
COLLATZ_INPUT = '{"x":%s}' % str(COLLATZ_SEQ).replace(" ", "")
POSEIDON_K11_SHA256 = "8d2d9052b47d9c9b45f3e3c268cec30797f74990cb47367bdfa7fbe77832129c"
POSEIDON_K16_SHA256 = "4c4e7d9301b652969a92718b3183f0bda79be2aaab245b68033ca96bf27bdc3c"


class SurveyRng:
    def __init__(self, start=0):
        self.s = R.SurveyStream(start=start)

    def fill(self, n):
        return self.s.fill(n)

    def fr_random(self, _field=None):
        return self.s.fr_random(R.BN_FR)


def golden(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


# ---------------------------------------------------------------------------------------------- CPU: pairing ----
def _g1_add(p1, p2):
    q = R.BN_FQ.p
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
    x3 = (lam * lam - x1 - x2) % q
    return (x3, (lam * (x1 - x3) - y1) % q)


def _g1_mul(k, p):
    r = None
    while k:
        if k & 1:
            r = _g1_add(r, p)
        p = _g1_add(p, p)
        k >>= 1
    return r


def test_pairing_is_bilinear_and_non_degenerate():
    from halo2_prover_amd import pairing as PR
    from halo2_prover_amd.prover import _G2_GEN, _g2_scalar_mul
    g1 = (1, 2)
    assert PR.g2_is_on_curve(_G2_GEN)
    e = PR.pairing(g1, _G2_GEN)
    assert e != PR.F12_ONE
    assert PR.f12_pow(e, PR.R) == PR.F12_ONE                                   # lands in the order-r subgroup
    a, b = 0x1234567890ABCDEF1234, 0xFEDCBA9876543210FED
    assert PR.pairing(_g1_mul(a, g1), _g2_scalar_mul(b, _G2_GEN)) == PR.f12_pow(e, a * b % PR.R)
    assert PR.pairing(_g1_add(_g1_mul(a, g1), _g1_mul(b, g1)), _G2_GEN) == PR.f12_mul(PR.f12_pow(e, a), PR.f12_pow(e, b))
    assert PR.pairing_check([(_g1_mul(a, g1), _G2_GEN), (g1, PR.g2_neg(_g2_scalar_mul(a, _G2_GEN)))])
    assert not PR.pairing_check([(_g1_mul(a + 1, g1), _G2_GEN), (g1, PR.g2_neg(_g2_scalar_mul(a, _G2_GEN)))])
    assert PR.pairing(None, _G2_GEN) == PR.F12_ONE and PR.pairing(g1, None) == PR.F12_ONE


def test_simulate_and_circuit_count():
    from halo2_prover_amd import verifier as V
    assert V.get_circuit_count() == 3
    assert V.wasm_simulate_circuit(COLLATZ_INPUT, 0) == "N/A"
    assert V.wasm_simulate_circuit(ARITH_INPUT, 1) == str(6 * 6 * 9 * 9 + 7) == "2923"
    # recorded from the reference's build: Poseidon([1, 2]) over bn256::Fr
    assert V.wasm_simulate_circuit(POSEIDON_INPUT, 2) == "0x152e960b5c9c8a624b2cdf4855250e8a54ee074254281310dc4a9704f78c1917"


# ---------------------------------------------------------------------------------------------- GPU: proofs ----
def _flip_positions(n):
    return sorted(set(list(range(0, n, 97)) + [1, 31, 32, 63, n - 33, n - 32, n - 1]))


@pytest.mark.gpu
def test_accepts_the_recorded_arithmetic_proof_and_rejects_corruptions(h2):
    from halo2_prover_amd import verifier as V
    params, proof = golden("params_k4.bin"), golden("proof_arithmetic_k4.bin")
    assert V.wasm_verify_proof(params, proof, ARITH_INPUT, 1) is True
    assert V.wasm_verify_proof(params, proof, '{"x":6,"y":9,"constant":7,"z":2924}', 1) is False     # wrong public input
    assert V.wasm_verify_proof(params, proof, '{"x":6,"y":9,"constant":8,"z":2923}', 1) is False
    assert V.wasm_verify_proof(params, proof[:-32], ARITH_INPUT, 1) is False                           # truncated
    assert V.wasm_verify_proof(params, b"", ARITH_INPUT, 1) is False
    for pos in _flip_positions(len(proof)):
        bad = bytearray(proof)
        bad[pos] ^= 0x04
        assert V.wasm_verify_proof(params, bytes(bad), ARITH_INPUT, 1) is False, pos
    # a point at infinity where a commitment / an opening point belongs: the reference's read_point refuses it
    for at in (0, 64, len(proof) - 32):
        for enc in (bytes(31) + b"\x80", bytes(32)):
            assert V.wasm_verify_proof(params, proof[:at] + enc + proof[at + 32:], ARITH_INPUT, 1) is False, at


@pytest.mark.gpu
def test_accepts_the_recorded_poseidon_proof_and_rejects_corruptions(h2):
    from halo2_prover_amd import verifier as V
    params, proof = golden("params_k6.bin"), golden("proof_poseidon_k6.bin")
    assert V.wasm_verify_proof(params, proof, POSEIDON_INPUT, 2) is True
    # the public input is recomputed from x (wasm.rs:154-168): a wrong `output` field changes nothing, a wrong x does
    assert V.wasm_verify_proof(params, proof, '{"x":[1,2],"output":"0x01"}', 2) is True
    assert V.wasm_verify_proof(params, proof, '{"x":[1,3],"output":"0x01"}', 2) is False
    for pos in _flip_positions(len(proof)):
        bad = bytearray(proof)
        bad[pos] ^= 0x01
        assert V.wasm_verify_proof(params, bytes(bad), POSEIDON_INPUT, 2) is False, pos
    # another circuit's proof under this circuit's key
    assert V.wasm_verify_proof(params, golden("proof_arithmetic_k4.bin"), POSEIDON_INPUT, 2) is False


@pytest.mark.gpu
def test_accepts_the_recorded_collatz_shplonk_proof_and_rejects_corruptions(h2):
    from halo2_prover_amd import prover, verifier as V
    params = prover.generate_params(10, SurveyRng(0)).write()
    assert hashlib.sha256(params).hexdigest() == "24cef0fa77991930622fce4c51c7ddf40aaf3324779b1c6592c1c29e6043374b"
    proof = golden("proof_collatz_k10.bin")
    assert V.wasm_verify_proof(params, proof, COLLATZ_INPUT, 0) is True
    for pos in _flip_positions(len(proof)):
        bad = bytearray(proof)
        bad[pos] ^= 0x10
        assert V.wasm_verify_proof(params, bytes(bad), COLLATZ_INPUT, 0) is False, pos
    assert V.wasm_verify_proof(params, proof[:608], COLLATZ_INPUT, 0) is False


@pytest.mark.gpu
def test_accepts_fresh_proofs_made_with_os_randomness(h2):
    """off the recorded RNG stream nothing else can judge a proof: prove with OsRng, verify"""
    from halo2_prover_amd import prover, verifier as V
    p4, p6 = golden("params_k4.bin"), golden("params_k6.bin")
    js = '{"x":3,"y":5,"constant":11,"z":%d}' % (3 * 3 * 5 * 5 + 11)
    proof = prover.wasm_generate_proof(p4, js, 1)
    assert V.wasm_verify_proof(p4, proof, js, 1) is True
    assert V.wasm_verify_proof(p4, proof, ARITH_INPUT, 1) is False
    js = '{"x":[7,8],"output":"%s"}' % V.wasm_simulate_circuit('{"x":[7,8]}', 2)   # the prover takes the claimed output
    proof = prover.wasm_generate_proof(p6, js, 2)                                     # as its public input (wasm.rs:116)
    assert V.wasm_verify_proof(p6, proof, js, 2) is True
    lie = prover.wasm_generate_proof(p6, '{"x":[7,8],"output":"0x00"}', 2)             # a false claim: no valid proof
    assert V.wasm_verify_proof(p6, lie, js, 2) is False
    p10 = prover.generate_params(10).write()
    js = '{"x":[6,3,10,5,16,8,4,2,1]}'
    proof = prover.wasm_generate_proof(p10, js, 0)
    assert V.wasm_verify_proof(p10, proof, js, 0) is True
    other = prover.generate_params(10).write()                     # another SRS: the pairing must fail
    assert V.wasm_verify_proof(other, proof, js, 0) is False


@pytest.mark.gpu
@pytest.mark.parametrize("k", [11, 16])
def test_accepts_the_poseidon_proof_at_the_recorded_hash(h2, k):
    from halo2_prover_amd import prover, verifier as V
    rng = SurveyRng(0)
    params = prover.generate_params(k, rng)
    circuit = prover.PoseidonCircuit([1, 2])
    pk = prover.generate_keys(params, circuit)
    proof = prover.generate_proof_with_instance(params, pk, circuit, [circuit.output()], rng)
    assert hashlib.sha256(proof).hexdigest() == {11: POSEIDON_K11_SHA256, 16: POSEIDON_K16_SHA256}[k]
    assert V.verify_with_instance(params, pk, proof, [circuit.output()]) is True
    assert V.verify_with_instance(params, pk, proof, [circuit.output() + 1]) is False
    bad = bytearray(proof)
    bad[700] ^= 0x20
    assert V.verify_with_instance(params, pk, bytes(bad), [circuit.output()]) is False
