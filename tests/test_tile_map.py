"""The block -> tile map of the compositing kernels (render_common.h xcd_tile / xcd_grid / tile_chunk_code), restated in Python:
every tile is visited exactly once, blocks past the end map to nothing, an XCD (block % 8) gets chunks from all over the frame."""
import pytest


def chunk_code(ntiles, forced=-1):
    if forced == 0:
        return 0
    c = 64
    if forced > 0:
        c = forced
    else:
        while c > 1 and ntiles // c < 64:
            c >>= 1
    lg = 0
    while (1 << lg) < c:
        lg += 1
    return lg + 1


def grid(ntiles, code):
    if code == 0:
        return (ntiles + 7) // 8 * 8
    ch = 1 << (code - 1)
    nchunks = (ntiles + ch - 1) // ch
    return (nchunks + 7) // 8 * 8 * ch


def xcd_tile(block, ntiles, tiles_x, code, SH):
    if code == 0:
        per = (ntiles + 7) >> 3
        k = (block & 7) * per + (block >> 3)
    else:
        lg, j = code - 1, block >> 3
        k = ((((j >> lg) << 3) + (block & 7)) << lg) + (j & ((1 << lg) - 1))
    if k >= ntiles:
        return -1
    if SH == 0:
        return k
    tiles_y = ntiles // tiles_x
    strip = k // (SH * tiles_x)
    rem = k - strip * SH * tiles_x
    h = min(SH, tiles_y - SH * strip)
    bx = rem // h
    by = SH * strip + (rem - bx * h)
    return by * tiles_x + bx


@pytest.mark.parametrize("tiles_x,tiles_y", [(120, 68), (32, 13), (1, 1), (7, 3), (16, 16), (255, 255), (3, 100)])
@pytest.mark.parametrize("forced", [-1, 0, 1, 4, 64])
@pytest.mark.parametrize("SH", [0, 8])
def test_every_tile_once(tiles_x, tiles_y, forced, SH):
    ntiles = tiles_x * tiles_y
    code = chunk_code(ntiles, forced)
    seen = [xcd_tile(b, ntiles, tiles_x, code, SH) for b in range(grid(ntiles, code))]
    live = [t for t in seen if t >= 0]
    assert sorted(live) == list(range(ntiles))
    assert len(live) == ntiles and len(seen) - len(live) < 8 * max(1, 1 << max(code - 1, 0)) + 8   # idle blocks: less than one round of chunks


def test_an_xcd_sees_the_whole_frame():
    """1080p: with chunks every XCD's tiles span (nearly) all strips of the image; with contiguous eighths one band only."""
    tiles_x, tiles_y, SH = 120, 68, 8
    ntiles = tiles_x * tiles_y
    for forced, min_rows in ((-1, 60), (0, 0)):
        code = chunk_code(ntiles, forced)
        rows = {}
        for b in range(grid(ntiles, code)):
            t = xcd_tile(b, ntiles, tiles_x, code, SH)
            if t >= 0:
                rows.setdefault(b & 7, set()).add(t // tiles_x)
        spans = [len(r) for r in rows.values()]
        if forced == -1:
            assert min(spans) >= min_rows, spans
        else:
            assert max(spans) <= 24, spans   # a band of at most three strips each
    assert chunk_code(416) == 3      # the DAS3R shape (32 x 13 tiles): chunks of 4 -> 104 chunks, 13 per XCD
    assert chunk_code(8160) == 7     # 1080p: chunks of 64
