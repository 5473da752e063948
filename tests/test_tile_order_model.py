"""CPU model of the compositing kernels' block -> tile map (render_common.h xcd_tile): for every grid shape and strip height the
map must hit every tile exactly once (blocks past the tile count return -1), and inside an XCD's share consecutive positions must
be spatial neighbours (that is what the strip order is for)."""
import pytest


def xcd_tile(block, ntiles, tiles_x, SH):
    per = (ntiles + 7) >> 3
    k = (block & 7) * per + (block >> 3)
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


def xcd_grid(ntiles):
    return ((ntiles + 7) // 8) * 8


@pytest.mark.parametrize("tiles_x,tiles_y", [(120, 68), (32, 13), (16, 16), (1, 1), (7, 3), (3, 50), (240, 135), (1, 9), (9, 1)])
@pytest.mark.parametrize("SH", [0, 1, 2, 4, 8, 16, 64])
def test_every_tile_is_composited_exactly_once(tiles_x, tiles_y, SH):
    ntiles = tiles_x * tiles_y
    seen = [xcd_tile(b, ntiles, tiles_x, SH) for b in range(xcd_grid(ntiles))]
    tiles = [t for t in seen if t >= 0]
    assert sorted(tiles) == list(range(ntiles))
    assert len(seen) - len(tiles) == xcd_grid(ntiles) - ntiles


def test_strip_order_keeps_neighbours_close():
    tiles_x, tiles_y, SH = 120, 68, 8
    ntiles = tiles_x * tiles_y
    order = {}                                         # tile -> position in the locality order
    per = (ntiles + 7) >> 3
    for b in range(xcd_grid(ntiles)):
        t = xcd_tile(b, ntiles, tiles_x, SH)
        if t >= 0:
            order[t] = (b & 7) * per + (b >> 3)
    far_v = far_h = 0
    for by in range(tiles_y - 1):
        for bx in range(tiles_x - 1):
            t = by * tiles_x + bx
            far_v += abs(order[t + tiles_x] - order[t]) > 1
            far_h += abs(order[t + 1] - order[t]) > 8
    # vertical neighbours are adjacent except across strip boundaries (1 in 8 rows), horizontal ones 8 apart (5 in the last strip)
    assert far_v <= (tiles_y // SH) * tiles_x and far_h == 0
