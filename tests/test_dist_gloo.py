"""not-gpu: the N > 1 path of bench.py on CPU -- world_size 2 over gloo.  The ring is broadcast from rank 0, every
rank proves its own shard with rank-specific RNG seeds (no data-path collective), timings are max-reduced.  The
prover here is the oracle (CPU stand-in for the engine, test infrastructure only)."""
import hashlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    import coracle as CO
    import zkattest_ref as R
    S, nkeys, B, sec = 77, 4, 1, 20
    params = R.synth_params(S, sec)

    def xy(pt, w):
        x, y = pt.toAffine()
        return x.to_bytes(w, 'big') + y.to_bytes(w, 'big')
    ins = [R.synth_proof_input(S, b, nkeys) for b in range(B)]
    if rank == 0:
        ring = R.synth_ring_fast(S, nkeys)
        for m, s, p, w, d, seed in ins:
            ring[w] = R.keyToInt(p)
        ring_b = b''.join(v.to_bytes(32, 'big') for v in ring)
    else:
        ring_b = bytes(32 * nkeys)
    t = torch.frombuffer(bytearray(ring_b), dtype=torch.uint8)
    dist.broadcast(t, src=0)
    ring_b = bytes(t.numpy().tobytes())
    seeds = bench.rank_seeds(b''.join(i[5] for i in ins), rank)
    ctx = CO.OracleCtx(xy(params.NistGroup.h, 32), xy(params.ProofGroup.g, 36), xy(params.ProofGroup.h, 36), sec)
    ctx.set_ring(ring_b, nkeys)
    msg = b''.join(i[0] for i in ins)
    proofs, st = ctx.prove_batch(msg, b''.join(i[1] for i in ins), b''.join(i[2][1:] for i in ins), [i[3] for i in ins], seeds=seeds)
    ok, _ = ctx.verify_batch(msg, proofs)
    dt = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    q.put((rank, st, ok, hashlib.sha256(proofs[0]).hexdigest(), hashlib.sha256(ring_b).hexdigest(), float(dt.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_prove_over_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, st0, ok0, h0, ring0, t0), (r1, st1, ok1, h1, ring1, t1) = res
    assert st0 == st1 == [0] and ok0 == ok1 == [1]
    assert ring0 == ring1            # broadcast delivered the ring
    assert h0 != h1                  # rank-specific randomness -> distinct proofs of the same statement
    assert t0 == t1 == 1.5           # max over ranks
