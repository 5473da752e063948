"""CPU model of the depth buckets of the segmented binning path (das3r_amd/csrc/segkey.h, round 4): the numpy restatement of
depth_bin / depth_bucket, operation by operation (fp32 fma, fp32 multiply, truncation), and the two properties the path rests on:

  * MONOTONE: a smaller depth never lands in a later bucket — the one thing correctness needs (the segments are sorted exactly
    afterwards, segsort.hip; the GPU tests hold the resulting lists to the oracle bit for bit);
  * the sixteen bits BELOW the bucket (the fraction that rides in the key's low bits and orders a segment without a trip to the depth
    keys) are part of the same monotone value, and they separate the entries of a segment well: few ties;
  * BALANCED enough: whatever the depth distribution (uniform, one wall, two walls and a sky, five decades), no bucket holds much
    more than its share unless the depths themselves are (nearly) equal — what keeps the segments inside LDS.
"""
import numpy as np
import pytest

DBINS, DBIN_SHIFT, DBIN0, FRAC = 256, 19, (127 - 8) << 4, 16


def depth_bin(bits):
    raw = (bits >> DBIN_SHIFT).astype(np.int64) - DBIN0
    return np.clip(raw, 0, DBINS - 1)


def bucket_map(bits_all, weights, dbits):
    """-> function bits -> (bucket << 16 | fraction), built the way scan_emit_kernel builds it from the preprocess kernel's histogram."""
    hist = np.bincount(depth_bin(bits_all), weights=weights, minlength=DBINS).astype(np.uint64)
    tot = int(hist.sum())
    sh = 0
    while (tot >> sh) >= (1 << 24):
        sh += 1
    cs = (hist >> np.uint64(sh)).astype(np.uint64)
    ex = np.concatenate([[0], np.cumsum(cs)[:-1]]).astype(np.uint64)
    tot2 = int(cs.sum())
    cnt, cdf = cs.astype(np.float32), ex.astype(np.float32)
    scale = np.float32(np.float32(1 << (dbits + FRAC)) / np.float32(tot2)) if tot2 else np.float32(0)
    nb = 1 << (dbits + FRAC)   # bucket and sixteen bits of fraction below it (the key's low bits: segsort.hip)

    def f(bits):
        raw = (bits >> DBIN_SHIFT).astype(np.int64) - DBIN0
        b = np.clip(raw, 0, DBINS - 1)
        frac = ((bits & ((1 << DBIN_SHIFT) - 1)).astype(np.float32) * np.float32(1.0 / (1 << DBIN_SHIFT))).astype(np.float32)
        frac = np.where(raw < 0, np.float32(0), np.where(raw > DBINS - 1, np.float32(1), frac)).astype(np.float32)
        # fp32 fma: exact in float64 (24-bit x 24-bit product + a 24-bit addend fits 53 bits here), rounded once to fp32
        pos = (cnt[b].astype(np.float64) * frac.astype(np.float64) + cdf[b].astype(np.float64)).astype(np.float32)
        v = (pos * scale).astype(np.float32)
        return np.minimum(v.astype(np.uint32), nb - 1)
    return f


def _bits(z):
    return np.asarray(z, dtype=np.float32).view(np.uint32)


CASES = {
    "uniform_1_10": lambda g, n: g.uniform(1.0, 10.0, n),
    "one_wall": lambda g, n: np.full(n, 4.0),
    "thin_slab": lambda g, n: 4.0 + 1e-5 * g.random(n),
    "two_walls_and_sky": lambda g, n: np.concatenate([2.0 + 1e-3 * g.random(n // 3), 3.7 + 0.05 * g.random(n // 3), g.uniform(50, 90, n - 2 * (n // 3))]),
    "five_decades": lambda g, n: 10.0 ** g.uniform(-2.5, 2.5, n),
    "outside_the_bins": lambda g, n: np.concatenate([g.uniform(0.0011, 0.0035, n // 2), g.uniform(300, 5000, n - n // 2)]),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("dbits", [2, 7])
def test_buckets_are_monotone_in_the_depth_bits(name, dbits):
    g = np.random.default_rng(3)
    z = CASES[name](g, 200_000).astype(np.float32)
    w = g.integers(1, 5, z.shape[0])
    bits = _bits(z)
    f = bucket_map(bits, w, dbits)
    order = np.sort(bits)
    b = f(order)
    assert (np.diff(b.astype(np.int64)) >= 0).all(), "a smaller depth landed in a later bucket, or in an earlier fraction of its bucket"
    assert b.max() < (1 << (dbits + FRAC))
    # every representable depth between two neighbours of the set, too (the map is built from the histogram, not from the set)
    probe = np.sort(_bits(np.nextafter(z[:5000], np.float32(np.inf)))) if name != "one_wall" else order[:10]
    both = np.sort(np.concatenate([order[:5000], probe]))
    assert (np.diff(f(both).astype(np.int64)) >= 0).all()


CASES["two_rough_walls_and_sky"] = lambda g, n: np.concatenate([2.0 * (1 + 0.03 * g.random(n // 3)), 3.7 + 0.2 * g.random(n // 3), g.uniform(50, 90, n - 2 * (n // 3))])


CASES["slab_1_percent"] = lambda g, n: 4.0 * (1 + 0.01 * g.random(n))


@pytest.mark.parametrize("name", ["uniform_1_10", "two_rough_walls_and_sky", "five_decades", "slab_1_percent"])
def test_buckets_are_about_equally_full(name):
    """128 buckets: the fullest holds at most 8x its share when the depths are spread inside their histogram bins (a bin is 1/16
    octave = 4.4 % of the depth; inside it the map interpolates linearly in the mantissa bits, so content that fills only the first
    fifth of a bin — the edge of a wall, a slab 1 % thick — is five times as dense as the map assumes).  What the map cannot split is a
    spike much NARROWER than a bin next to other content in the same bin ("two_walls_and_sky": a wall 0.05 % thick, a third of the
    scene, ends in one bucket): its (tile, bucket) segments are then a third of a tile's list — still sorted exactly, by rank in LDS
    up to 3072 entries and by the global-memory network beyond, which also sends the shape back to the global sort (segsort.hip)."""
    g = np.random.default_rng(5)
    z = CASES[name](g, 400_000).astype(np.float32)
    bits = _bits(z)
    f = bucket_map(bits, np.ones_like(bits), 7)
    counts = np.bincount(f(bits) >> FRAC, minlength=128)
    assert counts.max() <= 8 * z.shape[0] / 128, (name, counts.max(), z.shape[0] / 128)


def test_counts_beyond_2_to_24_are_shifted_not_rounded():
    g = np.random.default_rng(7)
    z = g.uniform(1.0, 10.0, 100_000).astype(np.float32)
    bits = _bits(z)
    f = bucket_map(bits, np.full(bits.shape, 1 << 12, dtype=np.int64), 7)   # total 4.1e8 > 2^24
    b = f(np.sort(bits))
    assert (np.diff(b.astype(np.int64)) >= 0).all() and (b.max() >> FRAC) == 127 and (b.min() >> FRAC) == 0


def test_fractions_rarely_tie_inside_a_segment():
    """What the fraction bits buy: on a spread depth distribution two entries of the same (tile, bucket) segment share all sixteen
    fraction bits only rarely — segment_sort_kernel fetches exact depth bits for tie groups only."""
    g = np.random.default_rng(11)
    z = g.uniform(1.0, 10.0, 1_000_000).astype(np.float32)
    bits = _bits(z)
    v = bucket_map(bits, np.ones_like(bits), 7)(bits)
    tile = g.integers(0, 416, z.shape[0]).astype(np.uint64)
    seg = (tile << np.uint64(23)) | v.astype(np.uint64)          # (tile, bucket, fraction): equal values = a tie inside a segment
    _, counts = np.unique(seg, return_counts=True)
    tied = int(counts[counts > 1].sum())
    assert tied <= 0.02 * z.shape[0], tied


def test_depth_histogram_sample_covers_a_das3r_model():
    """preprocess.hip picks the workgroups that sample the segmented path's depth histogram with a full-avalanche hash of their index
    (round 5).  Round 4's rule — `(index & mask) == 0`, every 16th workgroup at the Sintel shape — is a biased sample of a DAS3R model:
    its Gaussians are the pixels of its frames in row-major order, a workgroup of 256 is half a 512-pixel row, and every 16th workgroup is
    the LEFT half of every 8th row: the histogram never saw the right half of the scene (docs/ledger.md (ba)).  numpy restatement of
    the hash and of launch_preprocess's mask: at the Sintel and DAVIS shapes every (tile row, image half) gets its share of the sample,
    and the old rule provably does not."""
    import numpy as np

    def mask_of(nblocks):
        m = 0
        while (nblocks >> bin(m).count("1")) > 512:
            m = (m << 1) | 1
        return m

    def hashed(b, m):
        h = (b * 0x9E3779B1) & 0xFFFFFFFF
        h ^= h >> 15
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        h ^= h >> 16
        return (h & m) == 0

    for frames, W, H in ((20, 512, 208), (45, 512, 288)):
        P = frames * W * H
        nblocks = (P + 255) // 256
        m = mask_of(nblocks)
        b = np.arange(nblocks, dtype=np.uint64)
        pix = (b * 256) % (W * H)                      # first pixel of the workgroup inside its frame
        group = (pix // W // 16) * 2 + (pix % W) // 256   # (tile row, left / right half of the image)
        ngroups = ((H + 15) // 16) * 2
        new = np.bincount(group[hashed(b, m)].astype(np.int64), minlength=ngroups)
        old = np.bincount(group[(b & np.uint64(m)) == 0].astype(np.int64), minlength=ngroups)
        expect = hashed(b, m).sum() / ngroups
        assert 256 <= hashed(b, m).sum() <= 700
        assert new.min() >= 0.3 * expect, (frames, W, H, new.tolist())      # every part of the image is in the sample
        assert (old == 0).sum() >= ngroups // 2, old.tolist()                # the old rule: half of the groups never sampled
