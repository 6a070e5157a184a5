"""Host logic of the launch-mode cache (sgic_amd.ops): persistence, merge, nearest-M family lookup, and that the in-tree
cache file parses.  No GPU: nothing is launched."""
import json
import os

import pytest


@pytest.fixture()
def ops(tmp_path, monkeypatch):
    import sgic_amd  # noqa
    from sgic_amd import ops
    monkeypatch.setenv("SGIC_TILE_CACHE", str(tmp_path / "cache.json"))
    saved = (dict(ops._TILE), {k: dict(v) for k, v in ops._FAMILY.items()}, ops._DIRTY)
    ops._TILE.clear()
    ops._FAMILY.clear()
    yield ops
    ops._TILE.clear()
    ops._TILE.update(saved[0])
    ops._FAMILY.clear()
    ops._FAMILY.update(saved[1])
    ops._DIRTY = saved[2]


def test_in_tree_cache_is_well_formed():
    import sgic_amd  # noqa
    from sgic_amd import ops
    d = json.load(open(ops._INTREE_CACHE))
    assert d["device"] == "gfx950" and len(d["picks"]) >= 40
    for k, v in d["picks"].items():
        kind, *nums = k.split("|")
        assert kind in ("gemm", "gemm3", "conv3x3", "conv3", "attn", "attn3") and all(n.lstrip("-").isdigit() for n in nums)
        assert 0 <= int(v) <= {"attn": 7, "attn3": 7, "gemm3": 31, "conv3": 31}.get(kind, 15)
    # the dominant GEMM shapes of the benchmarked configuration are covered: no process races them again
    for key in ("gemm|9248|4096|1024|0|1", "gemm|9248|1024|4096|1|0", "gemm|9248|3072|1024|0|0", "attn|32|289|16|0",
                "gemm3|9248|4096|1024|0|1", "gemm3|9248|1024|4096|1|0", "gemm3|9248|3072|1024|0|0"):   # gemm3: the bf16x3 split GEMM
        assert key in d["picks"], key


def test_save_merge_and_reload(ops, tmp_path):
    ops._remember(("gemm", 9248, 1024, 4096, 1, 0), 12)
    ops._remember(("attn", 32, 289, 16, 0), 5)
    path = ops.save_tile_cache()
    assert path == str(tmp_path / "cache.json")
    d = json.load(open(path))
    assert d["picks"] == {"attn|32|289|16|0": 5, "gemm|9248|1024|4096|1|0": 12}
    # a second process adds its own picks without losing the first one's
    ops._TILE.clear()
    ops._FAMILY.clear()
    ops._remember(("gemm", 8192, 768, 3072, 1, 0), 13)
    ops.save_tile_cache()
    assert set(json.load(open(path))["picks"]) == {"attn|32|289|16|0", "gemm|9248|1024|4096|1|0", "gemm|8192|768|3072|1|0"}
    ops._TILE.clear()
    ops._FAMILY.clear()
    ops._load_tile_cache()
    assert ops._lookup(("gemm", 9248, 1024, 4096, 1, 0)) == 12 and ops._lookup(("attn", 32, 289, 16, 0)) == 5
    assert ops.save_tile_cache() is None or True          # nothing dirty: no rewrite needed


def test_nearest_m_family_lookup(ops):
    """a ragged last batch / another batch size borrows the pick of the nearest M of the same (N, K, epilogue) family
    (within 2x) instead of racing 12 modes; a different epilogue or a far M does not match"""
    ops._remember(("gemm", 9248, 4096, 1024, 0, 1), 12)
    ops._remember(("gemm", 2312, 4096, 1024, 0, 1), 13)
    assert ops._lookup(("gemm", 8959, 4096, 1024, 0, 1)) == 12            # batch 31 of 32
    assert ops._lookup(("gemm", 2601, 4096, 1024, 0, 1)) == 13            # nearest of the two
    assert ops._lookup(("gemm", 289, 4096, 1024, 0, 1)) is None           # 8x away: measure it
    assert ops._lookup(("gemm", 9248, 4096, 1024, 1, 1)) is None          # other epilogue
    assert ops._lookup(("gemm", 9248, 4096, 768, 0, 1)) is None           # other K
    # borrowed picks are not persisted as measurements
    ops.save_tile_cache()
    picks = json.load(open(os.environ["SGIC_TILE_CACHE"]))["picks"]
    assert "gemm|9248|4096|1024|0|1" in picks


def test_tile_of_reads_profile_keys(ops):
    ops._remember(("gemm", 9248, 4096, 1024, 0, 1), 12)
    assert ops.tile_of((9248, 4096, 1024, False, 1)) == 12
    assert ops.tile_of(("batched", 32, 256, 256, 64, False, 0)) == 0
    assert ops.tile_of((1, 2, 3, False, 0)) is None


def test_picks_of_another_build_are_dropped(ops, tmp_path):
    """a user cache holding a launch mode this library does not have (written by a newer build) must not reach the C ABI"""
    p = tmp_path / "cache.json"
    p.write_text(json.dumps({"device": "gfx950", "picks": {"gemm3|9248|4096|1024|0|1": 99, "gemm|64|64|512|0|0": 15, "bogus|1|2": 1,
                                                         "gemm3|x|1": 2}}))
    ops._TILE.clear()
    ops._FAMILY.clear()
    ops._load_tile_cache()
    assert ops._TILE.get(("gemm", 64, 64, 512, 0, 0)) == 15
    assert ops._TILE.get(("gemm3", 9248, 4096, 1024, 0, 1)) != 99
    assert not any(k[0] == "bogus" for k in ops._TILE)


def test_mode_ranges_agree_between_the_layers():
    """the launch-mode ranges are written down in three places: the kernels' dispatchers (csrc), the tuner's mode lists (ops)
    and the cache validator (ops._MAX_MODE): they must agree, or a tuned pick can be rejected by the C ABI"""
    import re
    import sgic_amd  # noqa
    from sgic_amd import ops
    csrc = os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "csrc")
    split_max = int(re.search(r"#define SGIC_SPLIT3_TILE_MODES (\d+)", open(os.path.join(csrc, "gemm_split.hip")).read()).group(1))
    f32_max = int(re.search(r"#define SGIC_TILE_MODES (\d+)", open(os.path.join(csrc, "gemm.hip")).read()).group(1))
    assert max(ops.SPLIT3_MODES) == split_max == ops._MAX_MODE["gemm3"] == ops._MAX_MODE["conv3"]
    assert max(ops.TUNE_MODES) == f32_max == ops._MAX_MODE["gemm"] == ops._MAX_MODE["conv3x3"]
    assert max(ops.ATTN_MODES) == ops._MAX_MODE["attn"] == ops._MAX_MODE["attn3"] == 7


def test_planes_rowmajor_inverts_the_slice_major_layout():
    """ops.Planes.rowmajor() (what the GPU tests compare producers' planes through) is the inverse of csrc/split3.h: s3_pack_off --
    element (row, col) of plane p sits at p * rows * cols + ((col / 32) * rows + row) * 32 + col % 32"""
    import torch
    import sgic_amd  # noqa
    from sgic_amd import ops
    rows, cols = 5, 96
    pl = ops.Planes(rows, cols, torch.device("cpu"))
    pl.t.copy_(torch.arange(3 * rows * cols, dtype=torch.int16))
    rm = pl.rowmajor()
    assert rm.shape == (3, rows, cols)
    for p in range(3):
        for r in range(rows):
            for c in (0, 1, 31, 32, 63, 64, 95):
                assert int(rm[p, r, c]) == p * rows * cols + ((c // 32) * rows + r) * 32 + c % 32
