"""CPU model of the index arithmetic of das3r_amd/csrc/render_bwd_rgn.hip (round 6: the backward compositing kernel with a DPP row per 2x2
pixel region).  The kernel deals a tile's 64 regions to 4 waves x 4 passes x 4 rows in three ways (template parameter GEO), gives every
lane one pixel of the tile to own (constants + replay state), and builds a wave's lists from one 16-bit reach mask per entry.  What must
hold whatever the geometry — and is asserted here on a restatement of the kernel's formulas — :
  * the (wave, pass, row) -> region map is a bijection onto the tile's 8 x 8 regions;
  * lane 16 row + s of a wave owns pixel U = s & 3 of region `row` of pass s >> 2: every pixel of the tile has exactly one owner;
  * bit 4 pass + row of the mask an entry gets from a wave is set iff the entry's cull box (|centre distance| <= half extent + 0.5 per
    axis: render_common.h region_mask, the forward's test) reaches that region — so no (pixel, entry) pair inside the box is dropped;
  * with GEO 2 (the default) horizontally, vertically and diagonally adjacent 4x4 groups always belong to different waves.
Replaces nothing upstream: upstream's renderCUDA (cuda_rasterizer/backward.cu) has no culling below the tile."""
import itertools

import numpy as np
import pytest


def strip_y8(geo, wave, B):
    return 2 * B + (wave & 1) if geo == 1 else 4 * (wave >> 1) + B


def strip_h(geo, wave, B):
    return (((wave - strip_y8(geo, wave, B)) & 3) >> 1) if geo == 1 else (wave & 1)


def region_rx8(geo, wave, B, r):
    return 2 * ((wave - 2 * B) & 3) + (r & 1) if geo == 2 else 4 * strip_h(geo, wave, B) + r


def region_ry8(geo, wave, B, r):
    return 2 * B + (r >> 1) if geo == 2 else strip_y8(geo, wave, B)


@pytest.mark.parametrize("geo", [0, 1, 2])
def test_regions_and_pixels_are_dealt_exactly_once(geo):
    regions, pixels = set(), set()
    for wave, B, r in itertools.product(range(4), range(4), range(4)):
        rx, ry = region_rx8(geo, wave, B, r), region_ry8(geo, wave, B, r)
        assert 0 <= rx < 8 and 0 <= ry < 8
        regions.add((rx, ry))
    assert len(regions) == 64
    for wave, lane in itertools.product(range(4), range(64)):
        row, s = lane >> 4, lane & 15
        px = 2 * region_rx8(geo, wave, s >> 2, row) + (s & 1)
        py = 2 * region_ry8(geo, wave, s >> 2, row) + ((s >> 1) & 1)
        pixels.add((px, py))
    assert pixels == set(itertools.product(range(16), range(16)))


def test_interleaved_groups_never_share_a_wave_with_a_neighbour():
    owner = {}
    for wave, B in itertools.product(range(4), range(4)):
        bx4, by4 = region_rx8(2, wave, B, 0) // 2, region_ry8(2, wave, B, 0) // 2
        owner[(bx4, by4)] = wave
        assert wave == (bx4 + 2 * by4) & 3
    assert len(owner) == 16
    for (x, y), w in owner.items():
        for dx, dy in ((1, 0), (0, 1), (1, 1), (1, -1)):
            if (x + dx, y + dy) in owner:
                assert owner[(x + dx, y + dy)] != w


def wave_mask(geo, wave, px, py, hx, hy):
    """The kernel's m16: xb / yb over the tile's eight region columns / rows, then the wave's sixteen bits (tile origin 0)."""
    xb = sum((1 << k) for k in range(8) if abs(px - (2 * k + 0.5)) <= hx + 0.5)
    yb = sum((1 << k) for k in range(8) if abs(py - (2 * k + 0.5)) <= hy + 0.5)
    m = 0
    for B in range(4):
        if geo == 2:
            xq, yq = (xb >> region_rx8(2, wave, B, 0)) & 3, (yb >> (2 * B)) & 3
            m |= ((xq if yq & 1 else 0) | ((xq << 2) if yq & 2 else 0)) << (4 * B)
        else:
            m |= (((xb >> (4 * strip_h(geo, wave, B))) & 15) << (4 * B)) if (yb >> strip_y8(geo, wave, B)) & 1 else 0
    return m


@pytest.mark.parametrize("geo", [0, 1, 2])
def test_reach_mask_lists_every_region_the_cull_box_touches(geo):
    rng = np.random.default_rng(5 + geo)
    for _ in range(400):
        px, py = rng.uniform(-6, 22, 2)
        hx, hy = rng.uniform(0.02, 9, 2) if rng.random() < 0.3 else rng.uniform(0.02, 2.5, 2)
        listed = set()
        for wave in range(4):
            m = wave_mask(geo, wave, px, py, hx, hy)
            for B, r in itertools.product(range(4), range(4)):
                if (m >> (4 * B + r)) & 1:
                    listed.add((region_rx8(geo, wave, B, r), region_ry8(geo, wave, B, r)))
        want = {(rx, ry) for rx, ry in itertools.product(range(8), range(8))
                if abs(px - (2 * rx + 0.5)) <= hx + 0.5 and abs(py - (2 * ry + 0.5)) <= hy + 0.5}
        assert listed == want
        # ... and the box test itself is conservative for the region's four pixel centres: a pixel inside the cull box lies in a listed region
        for x, y in itertools.product(range(16), range(16)):
            if abs(px - x) <= hx and abs(py - y) <= hy:
                assert (x // 2, y // 2) in listed
