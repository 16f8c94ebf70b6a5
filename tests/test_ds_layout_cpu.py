"""CPU: the index arithmetic of the dS hand-over (csrc/fa_bwd_dkdv_w4.hip DS variant -> csrc/fa_bwd_dq_ds.hip), restated in numpy.

The dK/dV kernel stores its packed dS registers as they are; the dQ kernel re-blocks the tile with the LDS-DMA's per-lane source
address, reads it transposed (ds_read_b64_tr_b16) and ends up with a permuted query on every lane.  This test walks one 32 x 32 tile
through those four maps with the formulas of the kernels' comments and checks that (1) the writer covers the 2-KiB tile exactly once,
1 KiB contiguous per store instruction, (2) the DMA re-blocking is a permutation of the tile's 16-byte chunks, (3) the transposed
reads hand every lane the 8 keys of its k-slots for ITS query — the query the epilogue then stores the row under —, (4) the two
16-lane groups of a half wave read disjoint LDS banks, and (5) the workspace sizes the C ABI reports are the tile grid's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-pytorch_amd"))


def _writer_tile(ds):
    """ds[query, key] (32 x 32 values) -> the 1024 16-bit elements of the stored tile, by the dK/dV kernel's register order:
    store instruction (half s), lane (r = key, h): 16 bytes at 1024 s + 32 r + 16 h, element j = query 16 s + 8 (j >> 2) + 4 h + (j & 3)"""
    tile = np.full(1024, np.nan)
    for s in range(2):
        chunks = set()
        for lane in range(64):
            r, h = lane & 31, lane >> 5
            byte = 1024 * s + 32 * r + 16 * h
            chunks.add(byte)
            for j in range(8):
                qrow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)
                assert np.isnan(tile[byte // 2 + j])
                tile[byte // 2 + j] = ds[qrow, r]
        assert chunks == set(range(1024 * s, 1024 * s + 1024, 16))   # one store instruction = one contiguous KiB
    assert not np.isnan(tile).any()
    return tile


def _dma_reblock(tile):
    """global tile -> LDS image: piece i, lane: 16 bytes from 1024 ((lane >> 3) & 1) + 32 (16 i + 4 (lane >> 4) + ((lane >> 1) & 3)) + 16 (lane & 1)
    land at LDS byte 1024 i + 16 lane"""
    lds = np.full(1024, np.nan)
    seen = set()
    for i in range(2):
        for lane in range(64):
            src = 1024 * ((lane >> 3) & 1) + 32 * (16 * i + 4 * (lane >> 4) + ((lane >> 1) & 3)) + 16 * (lane & 1)
            seen.add(src)
            lds[(1024 * i + 16 * lane) // 2:(1024 * i + 16 * lane) // 2 + 8] = tile[src // 2:src // 2 + 8]
    assert len(seen) == 128 and not np.isnan(lds).any()   # a permutation of the tile's 128 chunks
    return lds


def _tr_read(lds, base_of_lane):
    """ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns; lane 4 q + p supplies the address of row q, columns
    4 p .. 4 p + 3; lane i of the group receives column i of the 4 rows"""
    out = np.zeros((64, 4))
    for g in range(4):
        block = np.zeros((4, 16))
        for q in range(4):
            for p in range(4):
                a = base_of_lane(16 * g + 4 * q + p)
                assert a % 8 == 0
                block[q, 4 * p:4 * p + 4] = lds[a // 2:a // 2 + 4]
        for i in range(16):
            out[16 * g + i] = block[:, i]
    return out


def test_ds_tile_round_trip():
    rng = np.random.default_rng(0)
    ds = rng.standard_normal((32, 32))                   # ds[query, key]
    lds = _dma_reblock(_writer_tile(ds))
    for ks in range(2):                                  # the two 16-key MFMA steps of the stage
        def addr(lane, plus8=0):
            h, li, g16 = lane >> 5, lane & 15, (lane >> 4) & 1
            tq, tp = li >> 2, li & 3
            return 256 * h + 128 * g16 + 32 * tq + 8 * tp + 1024 * ks + 512 * plus8
        lo, hi = _tr_read(lds, lambda l_: addr(l_, 0)), _tr_read(lds, lambda l_: addr(l_, 1))
        for lane in range(64):
            c, h = lane & 31, lane >> 5
            query = 16 * (c >> 4) + 8 * ((c & 7) >> 2) + 4 * ((c >> 3) & 1) + (c & 3)   # the row the epilogue stores this lane's dQ under
            keys = [16 * ks + 4 * h + j for j in range(4)] + [16 * ks + 8 + 4 * h + j for j in range(4)]   # k-slots 8 h + 0 .. 7
            np.testing.assert_array_equal(np.concatenate([lo[lane], hi[lane]]), ds[query, keys])
        # banks (4-byte words modulo 64) of the two 16-lane groups of each half wave are disjoint: conflict free
        for h in range(2):
            banks = [set(), set()]
            for lane in range(32 * h, 32 * h + 32):
                a = addr(lane)
                banks[(lane >> 4) & 1].update({(a // 4) % 64, (a // 4 + 1) % 64})
            assert not (banks[0] & banks[1]) and len(banks[0]) == len(banks[1]) == 32
    # the lane -> query map is a permutation of the block's 32 rows
    assert sorted(16 * (c >> 4) + 8 * ((c & 7) >> 2) + 4 * ((c >> 3) & 1) + (c & 3) for c in range(32)) == list(range(32))


def test_workspace_sizes_are_the_tile_grid():
    import flashattention_lab_cuda as ext

    lib = ext._lib
    for bh, n in ((256, 4096), (80, 1000), (128, 8192), (300, 4100)):
        tiles = ((n + 31) // 32) * (8 * ((n + 255) // 256))          # 32-query blocks x 32-key blocks (keys padded to 256-key tiles)
        per_unit = tiles * 2048
        fit = (4 << 30) // per_unit
        parts = -(-bh // fit)
        units = bh if fit >= bh else -(-bh // parts)                 # equal chunks
        extra = lib.fa_backward_workspace_bytes_fast(bh, n, 128, 2, 0) - lib.fa_backward_workspace_bytes(bh, n, 128, 2)
        assert extra == units * per_unit, (bh, n)
