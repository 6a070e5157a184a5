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


_STEMS = ["img_10", "img_2", "a", "a-1", "a.b", "B", "zeta", "a_2", "0007", "Alpha", "b"]


def _vec_of(stem, d=16):
    import zlib
    r = np.random.default_rng(zlib.crc32(stem.encode()))
    v = r.standard_normal(d).astype(np.float32)
    return v / np.linalg.norm(v)


def _index_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    import sgic_amd  # noqa
    from sgic_amd.compress import assemble_index, stem_of
    from sgic_amd.dist import gather_vectors, shard_range
    # what compress.py does per rank: sorted file list -> contiguous shard -> one vector per file -> gather -> rank 0 index
    files = sorted(os.path.join(root, "in", s + ext) for s, ext in zip(_STEMS, [".jpg", ".png"] * 6))
    lo, hi = shard_range(len(files), rank, world)
    local = torch.from_numpy(np.stack([_vec_of(stem_of(f)) for f in files[lo:hi]]))
    for f in files[lo:hi]:                                   # every rank wrote the .c2df files of its shard
        if stem_of(f) != "zeta":                             # ... except one that "failed": it must stay out of the index
            open(os.path.join(root, "bit", stem_of(f) + ".c2df"), "wb").close()
    dist.barrier()
    allv = gather_vectors(local, len(files), rank, world).numpy()
    ids = assemble_index(files, allv, os.path.join(root, "bit"), os.path.join(root, "idx"), 16) if rank == 0 else None
    q.put((rank, ids))
    dist.destroy_process_group()


def test_rank0_index_assembly_order_world2(tmp_path):
    """compress.py:295-306: ids.txt / index rows follow sorted(glob(clip_vecs/*.npy)) = stems ordered by "<stem>.npy",
    only stems whose .c2df exists, doc id = join(bit_dir, stem + ".c2df"); vectors come through the all-gather"""
    root = str(tmp_path)
    for d in ("in", "bit", "idx"):
        os.makedirs(os.path.join(root, d))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_index_worker, args=(r, 2, port, root, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    import sgic_amd  # noqa
    from sgic_amd.faiss_io import read_index_flat_ip
    want = [s for s in sorted(_STEMS, key=lambda s: s + ".npy") if s != "zeta"]
    # "<stem>.npy" order, not bare-stem order: "a.b.npy" < "a.npy" (b < n) although "a" < "a.b"
    assert want.index("a-1") < want.index("a.b") < want.index("a") < want.index("a_2") and sorted(want) != want
    ids = [ln for ln in open(os.path.join(root, "idx", "ids.txt")).read().splitlines()]
    assert ids == [os.path.join(root, "bit", s + ".c2df") for s in want] == res[0]
    v = read_index_flat_ip(os.path.join(root, "idx", "index.faiss"))
    assert v.shape == (len(want), 16)
    for row, s in zip(v, want):
        assert np.allclose(row, _vec_of(s), atol=1e-6), s


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


def test_index_assembly_duplicate_stems_and_empty(tmp_path):
    """two inputs that share a stem (a.jpg, a.png) collapse to ONE index entry, like their .npy files do in the reference
    (compress.py:286 overwrites, :296 globs the survivors); an empty corpus writes no index at all (compress.py:297)"""
    import sgic_amd  # noqa
    from sgic_amd.compress import assemble_index
    from sgic_amd.faiss_io import read_index_flat_ip
    bit, idx = tmp_path / "bit", tmp_path / "idx"
    bit.mkdir()
    files = sorted(str(tmp_path / "in" / n) for n in ("a.jpg", "a.png", "b.jpg"))
    for s in ("a", "b"):
        (bit / f"{s}.c2df").write_bytes(b"x")
    vecs = np.stack([_vec_of("a-jpg"), _vec_of("a-png"), _vec_of("b")])
    ids = assemble_index(files, vecs, str(bit), str(idx), 16)
    assert ids == [str(bit / "a.c2df"), str(bit / "b.c2df")]
    v = read_index_flat_ip(str(idx / "index.faiss"))
    assert v.shape == (2, 16) and np.allclose(v[0], _vec_of("a-png"), atol=1e-6)      # the later file in sorted order wins
    assert assemble_index([], np.zeros((0, 16), np.float32), str(bit), str(tmp_path / "none"), 16) == []
    assert not (tmp_path / "none").exists()
