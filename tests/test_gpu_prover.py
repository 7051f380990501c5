"""GPU: the reference's prove() surface on the HIP backend reproduces the reference's own proofs, byte for byte.

halo2_prover_amd.prover (every MSM, NTT, the quotient and the SRS generation on the GPU through the C ABI), on the
deterministic RNG stream of SURVEY.md App. B.2, must reproduce the values recorded from the reference's own build
(SURVEY.md App. A.6, B.2, B.5):

  * params files (setup):            sha256 for k = 4, 6, 10, 11, 16
  * verifying-key digests:           transcript_repr of arithmetic k=4 and Poseidon k=6
  * proofs:                          sha256 of arithmetic k=4, Poseidon k=6, Poseidon k=11 and **Poseidon k=16**
                                     (BASELINE.json's headline configuration), plus the challenge checkpoints
"""
import hashlib
import os

import pytest

import pyref as R

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAMS_SHA256 = {
    4: "e410bf985e9327e7ea474d74907e4209b50678844fab11bc09bcd1e2a2ae1272",
    6: "3cd009bb91fe7f1d4c2cc54296263062e5a1d2359aacd68bfa1ce542b5a1169d",
    10: "24cef0fa77991930622fce4c51c7ddf40aaf3324779b1c6592c1c29e6043374b",
    11: "c071f033c580c8d827fb719c4d428a0d10673b4ab7da0c2ea62dec3ffc3fc6ca",
    16: "07d2055cadf19515cc5e2bdc14a46d54b8fccb37e5da5afa2a58cccb0012cee8",
}
PROOF_SHA256 = {
    ("arithmetic", 4): "31d427b9666777794f4a126fbde11584f28748005a32dcaf27e40974f3866f13",
    ("poseidon", 6): "6d235bf4637e1dce12559c44eaf77812bae2746d78331db3850e16b26234e63e",
    ("poseidon", 11): "8d2d9052b47d9c9b45f3e3c268cec30797f74990cb47367bdfa7fbe77832129c",
    ("poseidon", 16): "4c4e7d9301b652969a92718b3183f0bda79be2aaab245b68033ca96bf27bdc3c",
}
TRANSCRIPT_REPR = {
    ("arithmetic", 4): 0x29FDBC4FAA50E4E635114C86B4655A8CC4C5B56751D66E7F06C91C80076930F9,
    ("poseidon", 6): 0x0394952BB11B51B764C54781C76A552834CD31144A91BC35FDAC9CDD15070A39,
}
ARITH_INPUT = '{"x":6,"y":9,"constant":7,"z":2923}'
POSEIDON_HASH_1_2 = 0x152E960B5C9C8A624B2CDF4855250E8A54EE074254281310DC4A9704F78C1917
POSEIDON_INPUT = '{"x":[1,2],"output":"0x%064x"}' % POSEIDON_HASH_1_2


class SurveyRng:
    """the deterministic stream of SURVEY.md App. B.2: call i yields SHA256("seed0-" + str(i)) digests"""

    def __init__(self, start=0):
        self.s = R.SurveyStream(start=start)

    def fill(self, n):
        return self.s.fill(n)

    def fr_random(self, _field=None):
        return self.s.fr_random(R.BN_FR)


def test_gpu_arithmetic_proof_is_bit_identical_to_the_reference(h2):
    from halo2_prover_amd import prover
    params_bytes = open(os.path.join(GOLDEN, "params_k4.bin"), "rb").read()
    params = h2.ParamsKZG.read(params_bytes)
    circuit = prover.ArithmeticCircuit.from_json(ARITH_INPUT)
    pk = prover.generate_keys(params, circuit)
    assert pk.fixed_commitments[prover.ArithmeticCircuit.SC] is None      # all-zero column -> identity
    assert pk.transcript_repr == TRANSCRIPT_REPR[("arithmetic", 4)]
    trace = {}
    proof = prover.generate_proof_with_instance(params, pk, circuit, [7, 2923], SurveyRng(8), trace)
    want = {"theta": 0x06C57C43FCF14EE6717DE3EB214D43B85EC12BEABE4A73C8182EF267520B9C46,
            "beta": 0x253DD018D7552790DEA33ACC2DB0552F638F7C56C88B64F115543457C35B706A,
            "gamma": 0x259882FFDCB2CB55430C87719C970FBF33D7FC570A333E3E3FE122CAEB872FD8,
            "y": 0x27BC4C5117E9A643409ED30353119317BE3365C1683698C9ED6B0714E27EF13D,
            "x": 0x0430D455419494B7C0B188FF7A8259816250E8940746F49D3ABD46AE4EC99D42,
            "v": 0x063068C66F8E811EF9D68C59D6B7F025AC38228DBBA77858E8BA6C89B059430A}
    for name in ("theta", "beta", "gamma", "y", "x", "v"):
        assert trace[name] == want[name], name
    assert len(proof) == 1184
    assert hashlib.sha256(proof).hexdigest() == PROOF_SHA256[("arithmetic", 4)]
    assert proof == open(os.path.join(GOLDEN, "proof_arithmetic_k4.bin"), "rb").read()
    # the wasm-level entry point gives the same bytes
    assert prover.wasm_generate_proof(params_bytes, ARITH_INPUT, 1, SurveyRng(8)) == proof


def test_gpu_poseidon_k6_proof_is_bit_identical_to_the_reference(h2):
    from halo2_prover_amd import prover
    params_bytes = open(os.path.join(GOLDEN, "params_k6.bin"), "rb").read()
    params = h2.ParamsKZG.read(params_bytes)
    circuit = prover.PoseidonCircuit([1, 2])
    assert circuit.output() == POSEIDON_HASH_1_2                          # wasm_simulate_circuit's answer
    pk = prover.generate_keys(params, circuit)
    assert pk.transcript_repr == TRANSCRIPT_REPR[("poseidon", 6)]
    trace = {}
    proof = prover.generate_proof_with_instance(params, pk, circuit, [circuit.output()], SurveyRng(8), trace)
    assert trace["theta"] == 0x0B3C600455604EDA16B5DD3BDE867A7959D86F521C9BA096C0573C726193023D
    assert trace["y"] == 0x02829BCBB0AC70200F1DEC238DEE474499B2BE901A839A5042D66CA4D97B3DF3
    assert trace["x"] == 0x122555C65F6F0CB889DD51420A47AB236450C7EAF7F7A253ED11ECB1BF2E109D
    assert trace["v"] == 0x1DF4C5795BEAE4379F05C98DE19DE324BF5E5DEAB457E5A9D7AC646F788D9D75
    assert len(proof) == 1536
    assert hashlib.sha256(proof).hexdigest() == PROOF_SHA256[("poseidon", 6)]
    assert proof == open(os.path.join(GOLDEN, "proof_poseidon_k6.bin"), "rb").read()
    assert prover.wasm_generate_proof(params_bytes, POSEIDON_INPUT, 2, SurveyRng(8)) == proof


def test_gpu_collatz_shplonk_proof_is_bit_identical_to_the_reference(h2):
    """circuit 0: setup(10) then prove in one process, as the UI does (Circuits.tsx:90) and as recorded"""
    from halo2_prover_amd import prover
    rng = SurveyRng(0)
    params = prover.generate_params(10, rng)
    assert hashlib.sha256(params.write()).hexdigest() == PARAMS_SHA256[10]
    seq = [9, 28, 14, 7, 22, 11, 34, 17, 52, 26, 13, 40, 20, 10, 5, 16, 8, 4, 2, 1]
    circuit = prover.CollatzCircuit(seq)
    pk = prover.generate_keys(params, circuit)
    assert pk.transcript_repr == 0x174D961F4BE70218C76F49111B0E742F0EC7583D402762C15220F90E82809AB5
    proof = prover.generate_proof(params, pk, circuit, rng)
    assert len(proof) == 640
    assert hashlib.sha256(proof).hexdigest() == "8709c25ae65667b14921a4df48907cccc0d7d024ae2f56b2e9e25b6b4d679352"
    assert proof == open(os.path.join(GOLDEN, "proof_collatz_k10.bin"), "rb").read()
    js = '{"x":%s}' % str(seq).replace(" ", "")
    assert prover.wasm_generate_proof(params.write(), js, 0, SurveyRng(8)) == proof


@pytest.mark.parametrize("k", [4, 6, 10, 11])
def test_gpu_setup_reproduces_the_reference_params(h2, k):
    """generate_params(k) = ParamsKZG::new(k): g and g_lagrange by GPU fixed-base multiplications"""
    from halo2_prover_amd import prover
    params = prover.generate_params(k, SurveyRng(0))
    assert hashlib.sha256(params.write()).hexdigest() == PARAMS_SHA256[k]


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
        got = prover.generate_proof_with_instance(params, pk, circuit, [c, z], SurveyRng(8))
        opk = H.ProvingKey(H.ArithmeticCircuit(x, y, c), be)
        assert opk.transcript_repr == pk.transcript_repr
        assert got == H.create_proof(opk, be, [[c, z]], R.SurveyStream(start=8)), (x, y, c)
    # Poseidon with another message, k = 6
    params_bytes = open(os.path.join(GOLDEN, "params_k6.bin"), "rb").read()
    params = h2.ParamsKZG.read(params_bytes)
    be = H.OracleBackend(params_bytes)
    msg = [0xDEADBEEF, 2**64 - 59]
    circuit = prover.PoseidonCircuit(msg)
    pk = prover.generate_keys(params, circuit)
    got = prover.generate_proof_with_instance(params, pk, circuit, [circuit.output()], SurveyRng(8))
    ocirc = H.PoseidonCircuit(msg)
    opk = H.ProvingKey(ocirc, be)
    assert got == H.create_proof(opk, be, [[ocirc.output()]], R.SurveyStream(start=8))


@pytest.mark.parametrize("k", [11, 16])
def test_gpu_poseidon_proof_at_baseline_sizes_is_bit_identical_to_the_reference(h2, k):
    """BASELINE.json configs[1] (Poseidon k = 11) and the metric's configuration (Poseidon k = 16): setup then
    prove in one process on the App. B.2 stream, exactly as the reference's recorded runs; both the params and
    the proof must hash to the recorded values."""
    from halo2_prover_amd import prover
    rng = SurveyRng(0)
    params = prover.generate_params(k, rng)
    assert hashlib.sha256(params.write()).hexdigest() == PARAMS_SHA256[k]
    circuit = prover.PoseidonCircuit([1, 2])
    pk = prover.generate_keys(params, circuit)
    proof = prover.generate_proof_with_instance(params, pk, circuit, [circuit.output()], rng)
    assert len(proof) == 1536
    assert hashlib.sha256(proof).hexdigest() == PROOF_SHA256[("poseidon", k)]
