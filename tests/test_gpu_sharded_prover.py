"""GPU: the prover with its commitments sharded over ranks (one process per rank, world size 2 and 3) still produces
the reference's proofs byte for byte on EVERY rank.

The box has one GPU, so the ranks share cuda:0 and the collective runs over gloo (sharded.gather_columns goes
through host memory for any backend but RCCL); column sharding, padding of short shards, the gather order and the
transcript on each rank are exactly what runs under RCCL with one GPU per rank.
"""
import hashlib
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ARITH_INPUT = '{"x":6,"y":9,"constant":7,"z":2923}'
ARITH_SHA256 = "31d427b9666777794f4a126fbde11584f28748005a32dcaf27e40974f3866f13"
POSEIDON_K6_SHA256 = "6d235bf4637e1dce12559c44eaf77812bae2746d78331db3850e16b26234e63e"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, backend="gloo"):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyref as R
    import halo2_prover_amd as h2
    from halo2_prover_amd import prover, sharded
    if world == 1:
        sharded.FORCE_GATHER = True          # run the collective even alone: the RCCL code path

    class Rng:
        def __init__(self, start):
            self.s = R.SurveyStream(start=start)

        def fill(self, n):
            return self.s.fill(n)

        def fr_random(self, _field=None):
            return self.s.fr_random(R.BN_FR)

    h2.init(0)
    digests = []
    params = h2.ParamsKZG.read(open(os.path.join(GOLDEN, "params_k4.bin"), "rb").read())
    circuit = prover.ArithmeticCircuit.from_json(ARITH_INPUT)
    pk = prover.generate_keys(params, circuit)
    digests.append(hashlib.sha256(prover.generate_proof_with_instance(params, pk, circuit, [7, 2923], Rng(8))).hexdigest())
    params = h2.ParamsKZG.read(open(os.path.join(GOLDEN, "params_k6.bin"), "rb").read())
    circuit = prover.PoseidonCircuit([1, 2])
    pk = prover.generate_keys(params, circuit)
    digests.append(hashlib.sha256(prover.generate_proof_with_instance(params, pk, circuit, [circuit.output()], Rng(8))).hexdigest())
    open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write("\n".join(digests))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_prover_reproduces_the_recorded_proofs_on_every_rank(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split("\n")
        assert got == [ARITH_SHA256, POSEIDON_K6_SHA256], r


def test_sharded_prover_over_rccl_with_one_rank(tmp_path):
    """the same prover with the gather forced in a one-rank RCCL group: all_gather_into_tensor on device tensors"""
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), "nccl"), nprocs=1, join=True)
    got = open(os.path.join(str(tmp_path), "rank0.txt")).read().split("\n")
    assert got == [ARITH_SHA256, POSEIDON_K6_SHA256]
