"""Shared test inputs: the reference tests' deterministic generators."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
IQ_NAMES = ["iq2_xs", "iq2_s", "iq3_xxs", "iq3_s", "iq4_xs"]


def load_iq_fixture(name):
    f = json.loads((GOLDEN / f"{name}.json").read_text())
    raw = np.frombuffer(bytes.fromhex(f["raw_hex"]), dtype=np.uint8).copy()
    gold = np.asarray(f["golden_f32"], dtype=np.float32)
    return f, raw, gold


def tile_blocks(raw, type_size, n_fixture_blocks, n_blocks):
    """Repeat valid fixture blocks (rule of gguf/iq/cuda_tests.rs:132-136)."""
    blocks = raw.reshape(n_fixture_blocks, type_size)
    idx = np.arange(n_blocks) % n_fixture_blocks
    return np.ascontiguousarray(blocks[idx]).reshape(-1)


def ramp(m, n_in, mod, off, step):
    """x[i] = ((i % mod) - off) * step, the reference tests' activation ramps."""
    i = np.arange(m * n_in, dtype=np.int64)
    return (((i % mod).astype(np.float32) - np.float32(off)) * np.float32(step)).reshape(m, n_in)


def random_iq_blocks(rng, type_size, n_blocks):
    """Uniform random block bytes with the f16 super-scale forced finite & small
    (exponent field 0x2C..0x34 -> |d| in [2^-4, 2^-2))."""
    b = rng.integers(0, 256, size=(n_blocks, type_size), dtype=np.uint8)
    hi = rng.integers(0x2C, 0x34, size=n_blocks, dtype=np.uint8)
    sign = rng.integers(0, 2, size=n_blocks, dtype=np.uint8) << 7
    b[:, 1] = hi | sign
    return b.reshape(-1)


class Lcg:
    """rng = rng*1664525 + 1013904223 (mod 2^32); marlin_cuda.rs:2049-2053."""

    def __init__(self, seed=0x9E3779B9):
        self.s = seed & 0xFFFFFFFF

    def next(self, modulus):
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        return self.s % modulus


def awq_lcg_case(m, k, n, group, seed=0x9E3779B9):
    """qweight [K/8,N] sequential, qzeros [G,N/8], scales, inputs: fill order and
    value maps of marlin_cuda.rs:2057-2089."""
    r = Lcg(seed)
    pk, pn, g = k // 8, n // 8, k // group
    qw = np.empty(pk * n, np.uint32)
    for i in range(pk * n):
        w = 0
        for j in range(8):
            w |= (r.next(16) & 0xF) << (4 * j)
        qw[i] = w
    qz = np.empty(g * pn, np.uint32)
    for i in range(g * pn):
        w = 0
        for j in range(8):
            w |= (r.next(16) & 0xF) << (4 * j)
        qz[i] = w
    scales = np.array([(np.float32(r.next(2000)) - np.float32(1000.0)) * np.float32(1e-4) for _ in range(g * n)], np.float32)
    inputs = np.array([(np.float32(r.next(2000)) - np.float32(1000.0)) * np.float32(1e-3) for _ in range(m * k)], np.float32)
    return qw.reshape(pk, n), qz.reshape(g, pn), scales.reshape(g, n), inputs.reshape(m, k)


# ---------------------------------------------------------------- several ranks as processes on one GPU
def make_exchange(rank, world, conns):
    """exchange(mine) -> [every rank's object in rank order] over the pipes spawn_ranks made (conns[rank] = ends to the others)."""
    def exchange(mine):
        for c in conns[rank]:
            c.send((rank, mine))
        got = {rank: mine}
        for c in conns[rank]:
            r, b = c.recv()
            got[r] = b
        return [got[r] for r in range(world)]
    return exchange


def spawn_ranks(world, target, extra_args=(), timeout=240):
    """Run target(rank, world, conns, result_queue, *extra_args) in `world` spawned processes (they share the box's one GPU and
    join through peer-mapped memory); every process must put (rank, payload) on the queue.  Returns {rank: payload}."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    conns = {r: [] for r in range(world)}
    for i in range(world):
        for j in range(i + 1, world):
            a, b = ctx.Pipe()
            conns[i].append(a)
            conns[j].append(b)
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, conns, q) + tuple(extra_args)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            r, payload = q.get(timeout=timeout)
            results[r] = payload
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert set(results) == set(range(world)), f"ranks that reported: {sorted(results)}"
    return results
