"""Whole-proof pin of the CPU oracle (no GPU): keygen + create_proof (GWC) of the reference's arithmetic circuit
at k = 4, restated in oracle/halo2_ref.py on top of the C oracle's best_multiexp / best_fft, must reproduce

* the six Fiat-Shamir challenge checkpoints recorded in SURVEY.md App. B.5, and
* the sha256 of the 1184-byte proof recorded in SURVEY.md App. B.2 from the reference's own build

for the input {"x":6,"y":9,"constant":7,"z":2923} (/root/reference/circuits/src/arithmetic_circuit.rs:39-45,
wasm.rs:90-97) on the pinned params file, under the deterministic RNG stream of App. B.2 (calls 8.. follow
setup's 0..7).  Every one of the 13 commitments in that proof is a best_multiexp output and every polynomial
behind them went through best_fft, so this pins the hot path's results on real prover data."""
import hashlib
import os

import halo2_ref as H
import pyref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHALLENGES_K4 = {
    "theta": 0x06C57C43FCF14EE6717DE3EB214D43B85EC12BEABE4A73C8182EF267520B9C46,
    "beta": 0x253DD018D7552790DEA33ACC2DB0552F638F7C56C88B64F115543457C35B706A,
    "gamma": 0x259882FFDCB2CB55430C87719C970FBF33D7FC570A333E3E3FE122CAEB872FD8,
    "y": 0x27BC4C5117E9A643409ED30353119317BE3365C1683698C9ED6B0714E27EF13D,
    "x": 0x0430D455419494B7C0B188FF7A8259816250E8940746F49D3ABD46AE4EC99D42,
    "v": 0x063068C66F8E811EF9D68C59D6B7F025AC38228DBBA77858E8BA6C89B059430A,
}
PROOF_SHA256_K4 = "31d427b9666777794f4a126fbde11584f28748005a32dcaf27e40974f3866f13"


def test_arithmetic_k4_proof_matches_reference_record():
    params = open(os.path.join(GOLDEN, "params_k4.bin"), "rb").read()
    be = H.OracleBackend(params)
    pk = H.ProvingKey(H.ArithmeticCircuit(6, 9, 7), be, H.TRANSCRIPT_REPR[("arithmetic", 4)])
    # keygen facts of SURVEY.md App. A.6: the all-zero fixed column sc commits to the identity
    assert pk.fixed_commitments[H.ArithmeticCircuit.SC] is None
    assert all(pt is not None and R.BN254.is_on_curve(pt) for pt in pk.sigma_commitments)
    rng = R.SurveyStream(start=8)
    trace = {}
    proof = H.create_proof(pk, be, [[7, 2923]], rng, trace)
    assert {k: trace[k] for k in CHALLENGES_K4} == CHALLENGES_K4
    assert len(proof) == 1184                       # 13 points + 24 scalars
    assert hashlib.sha256(proof).hexdigest() == PROOF_SHA256_K4
    assert rng.counter == 8 + 48 * 8 + 1            # 48 x Fr::random + one 32-byte ChaCha seed
    open(os.path.join(GOLDEN, "proof_arithmetic_k4.bin"), "wb").write(proof) if not os.path.exists(
        os.path.join(GOLDEN, "proof_arithmetic_k4.bin")) else None


def test_golden_proof_file_is_the_pinned_bytes():
    path = os.path.join(GOLDEN, "proof_arithmetic_k4.bin")
    if not os.path.exists(path):
        test_arithmetic_k4_proof_matches_reference_record()
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == PROOF_SHA256_K4


def test_a_different_witness_changes_the_proof_but_not_its_shape():
    params = open(os.path.join(GOLDEN, "params_k4.bin"), "rb").read()
    be = H.OracleBackend(params)
    pk = H.ProvingKey(H.ArithmeticCircuit(3, 5, 11), be, H.TRANSCRIPT_REPR[("arithmetic", 4)])
    proof = H.create_proof(pk, be, [[11, 3 * 3 * 5 * 5 + 11]], R.SurveyStream(start=8))
    assert len(proof) == 1184 and hashlib.sha256(proof).hexdigest() != PROOF_SHA256_K4
