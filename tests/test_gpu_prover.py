"""GPU: the reference's prove() surface on the HIP backend reproduces the reference's own proof, byte for byte.

halo2_prover_amd.prover.generate_proof_with_instance (every MSM, NTT and the quotient on the GPU through the C ABI)
on the pinned k = 4 params, input {"x":6,"y":9,"constant":7,"z":2923}, RNG stream of SURVEY.md App. B.2:
sha256(proof) must equal the value recorded from the reference's build, i.e. the golden
tests/golden/proof_arithmetic_k4.bin, and the six challenge checkpoints of App. B.5 must be hit on the way."""
import hashlib
import os

import pytest

import pyref as R

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROOF_SHA256_K4 = "31d427b9666777794f4a126fbde11584f28748005a32dcaf27e40974f3866f13"
INPUT = '{"x":6,"y":9,"constant":7,"z":2923}'


class SurveyRng:
    """the deterministic stream of SURVEY.md App. B.2, positioned after setup's 8 calls"""

    def __init__(self):
        self.s = R.SurveyStream(start=8)

    def fill(self, n):
        return self.s.fill(n)

    def fr_random(self, _field=None):
        return self.s.fr_random(R.BN_FR)


def test_gpu_proof_is_bit_identical_to_the_reference(h2):
    from halo2_prover_amd import prover
    params_bytes = open(os.path.join(GOLDEN, "params_k4.bin"), "rb").read()
    params = h2.ParamsKZG.read(params_bytes)
    circuit = prover.ArithmeticCircuit.from_json(INPUT)
    pk = prover.generate_keys(params, circuit)
    assert pk.fixed_commitments[prover.ArithmeticCircuit.SC] is None      # all-zero column -> identity
    trace = {}
    proof = prover.generate_proof_with_instance(params, pk, circuit, [7, 2923], SurveyRng(), trace)
    want = {"theta": 0x06C57C43FCF14EE6717DE3EB214D43B85EC12BEABE4A73C8182EF267520B9C46,
            "beta": 0x253DD018D7552790DEA33ACC2DB0552F638F7C56C88B64F115543457C35B706A,
            "gamma": 0x259882FFDCB2CB55430C87719C970FBF33D7FC570A333E3E3FE122CAEB872FD8,
            "y": 0x27BC4C5117E9A643409ED30353119317BE3365C1683698C9ED6B0714E27EF13D,
            "x": 0x0430D455419494B7C0B188FF7A8259816250E8940746F49D3ABD46AE4EC99D42,
            "v": 0x063068C66F8E811EF9D68C59D6B7F025AC38228DBBA77858E8BA6C89B059430A}
    for name in ("theta", "beta", "gamma", "y", "x", "v"):
        assert trace[name] == want[name], name
    assert len(proof) == 1184
    assert hashlib.sha256(proof).hexdigest() == PROOF_SHA256_K4
    assert proof == open(os.path.join(GOLDEN, "proof_arithmetic_k4.bin"), "rb").read()
    # the wasm-level entry point gives the same bytes
    assert prover.wasm_generate_proof(params_bytes, INPUT, 1, SurveyRng()) == proof


def test_gpu_prover_matches_the_oracle_prover_on_other_witnesses(h2):
    import halo2_ref as H
    from halo2_prover_amd import prover
    params_bytes = open(os.path.join(GOLDEN, "params_k4.bin"), "rb").read()
    params = h2.ParamsKZG.read(params_bytes)
    be = H.OracleBackend(params_bytes)
    for x, y, c in ((3, 5, 11), (0, 0, 0), (2**63, 2**64 - 1, 12345678901234567)):
        z = (x * x % H.P) * (y * y % H.P) % H.P + c
        circuit = prover.ArithmeticCircuit(x, y, c)
        pk = prover.generate_keys(params, circuit)
        got = prover.generate_proof_with_instance(params, pk, circuit, [c, z], SurveyRng())
        opk = H.ProvingKey(H.ArithmeticCircuit(x, y, c), be, H.TRANSCRIPT_REPR[("arithmetic", 4)])
        assert got == H.create_proof(opk, be, [[c, z]], R.SurveyStream(start=8)), (x, y, c)
