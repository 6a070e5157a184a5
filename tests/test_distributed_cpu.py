"""CPU, world_size=2 gloo: the N>1 sharding + CLIP-vector all-gather logic used by bench.py / compress.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    import sgic_amd  # noqa
    from sgic_amd.dist import gather_vectors, shard_range
    lo, hi = shard_range(n_total, rank, world)
    full = torch.arange(n_total * 8, dtype=torch.float32).reshape(n_total, 8)
    got = gather_vectors(full[lo:hi].clone(), n_total, rank, world)
    q.put((rank, lo, hi, bool(torch.equal(got, full))))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    for n_total in (7, 8, 1):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
        [p.start() for p in ps]
        res = sorted(q.get(timeout=120) for _ in ps)
        [p.join(timeout=60) for p in ps]
        assert all(r[3] for r in res), res
        assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_total


def test_shard_range_partitions():
    import sgic_amd  # noqa
    from sgic_amd.dist import shard_range
    for n in (0, 1, 5, 10000):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_faiss_flat_ip_file_matches_reference_sample(golden_dir, tmp_path):
    import sgic_amd  # noqa
    from sgic_amd.faiss_io import FaissDB, read_index_flat_ip, write_index_flat_ip
    ref = open(os.path.join(golden_dir, "ref_index.faiss"), "rb").read()
    v = read_index_flat_ip(os.path.join(golden_dir, "ref_index.faiss"))
    assert v.shape == (1, 512)
    p = tmp_path / "i.faiss"
    write_index_flat_ip(str(p), v)
    assert p.read_bytes() == ref
    # the reference wrote FaissDB.add(np.load(apple.npy)): re-normalised copy of the CLIP vector
    db = FaissDB(str(tmp_path / "db"), 512)
    db.add(np.load(os.path.join(golden_dir, "ref_apple.npy")), "../IO/bitstreams/apple.c2df")
    db.persist()
    assert (tmp_path / "db" / "index.faiss").read_bytes() == ref
    assert (tmp_path / "db" / "ids.txt").read_text() == open(os.path.join(golden_dir, "ref_ids.txt")).read()
    s, i = db.search(v[0], 1)
    assert i[0, 0] == 0 and abs(s[0, 0] - 1.0) < 1e-5


def test_c2df_container_byte_identical_to_reference_sample(golden_dir):
    import sgic_amd  # noqa
    from sgic_amd import filemaker, zstd
    d = open(os.path.join(golden_dir, "ref_apple.c2df"), "rb").read()
    enc, hdr = filemaker.unpack_c2df(d)
    assert filemaker.pack_c2df(enc, hdr) == d
    assert list(enc.keys()) == ["z_bit_stream", "h_bit_stream", "img_shape", "feat_shape", "stack_shape", "token_length",
                                "z_indices_shape", "clip_stream", "clip_meta"]
    q = np.frombuffer(zstd.decompress(enc["clip_stream"]), dtype=np.uint8)
    v = np.load(os.path.join(golden_dir, "ref_apple.npy"))
    assert np.array_equal(q, np.clip(np.round((v * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8))
    assert zstd.decompress(zstd.Compressor(19).compress(q.tobytes())) == q.tobytes()
