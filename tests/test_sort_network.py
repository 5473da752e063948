"""CPU models of the compositing kernels' LDS sorting networks (das3r_amd/csrc/render_common.h: local_order_tile).
513 .. 1024 entries: the always-ascending bitonic network, one thread per comparison, with the partner-out-of-range rule
instead of padding.  257 .. 512 entries (round 3): the same network run by ONE wave with eight words per lane in registers,
three levels per pass (wave_bitonic_sort<3>; the model below mirrors its index arithmetic loop for loop, also for the
sixteen-word variant), the words compared as doubles.  Checks (i) that it sorts every length, (ii) the claim the kernel's barrier placement rests on:
a step with partner distance < 128 (lj <= 6) only touches words of the 128-word chunk that the comparison's own wave owns."""
import numpy as np
import pytest

TILE_PIX = 256


def network_steps(n):
    """Yield (lk, lj, pairs) exactly as the kernel enumerates them: pairs = [(i, q)] with q < n."""
    N = 2 * TILE_PIX
    while N < n:
        N <<= 1
    lk = 1
    while (1 << (lk - 1)) < n:
        for lj in range(lk - 1, -1, -1):
            pairs = []
            for pr in range(N >> 1):
                i = ((pr >> lj) << (lj + 1)) | (pr & ((1 << lj) - 1))
                q = (i ^ ((2 << lj) - 1)) if lj == lk - 1 else (i | (1 << lj))
                if q < n:
                    pairs.append((pr, i, q))
            yield lk, lj, pairs
        lk += 1


@pytest.mark.parametrize("n", [2, 3, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 777, 1000, 1023, 1024])
def test_network_sorts_every_length(n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 20, size=n).astype(np.uint64)
    keys = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)   # (depth bits, position): unique, like the kernel's words
    a = keys.copy()
    for _, _, pairs in network_steps(n):
        for _, i, q in pairs:
            assert i < q < n
            if a[i] > a[q]:
                a[i], a[q] = a[q], a[i]
    assert np.array_equal(a, np.sort(keys))


@pytest.mark.parametrize("n", [257, 400, 512, 640, 1024])
def test_narrow_steps_stay_in_the_wave_chunk(n):
    """Comparison pr belongs to wave (pr % 256) // 64 and, for pr >= 256, to the same wave's second chunk: chunk = pr // 64.
    In a step with lj <= 6 both words of the comparison lie in words [128 * chunk, 128 * chunk + 128) — no other wave reads or
    writes them, so those steps need no workgroup barrier; steps with lj > 6 do cross chunks."""
    crossed_wide = False
    for lk, lj, pairs in network_steps(n):
        for pr, i, q in pairs:
            chunk = pr // 64
            inside = 128 * chunk <= i < 128 * chunk + 128 and 128 * chunk <= q < 128 * chunk + 128
            if lj <= 6:
                assert inside, (lk, lj, pr, i, q)
            else:
                crossed_wide |= not inside
    if n > 256:
        assert crossed_wide


def wave_sort_model(keys, G):
    """wave_bitonic_sort<G>: N = 64 << G words, 64 lanes, E = 1 << G words per lane and pass.  -> (sorted words, passes)"""
    E = 1 << G
    H, LOGN, N = E // 2, 6 + G, 64 * E
    key = keys.copy()
    assert len(key) == N

    def ce(lo, hi):
        assert lo < hi
        if key[lo] > key[hi]:
            key[lo], key[hi] = key[hi], key[lo]
    passes = 1
    for lane in range(64):   # pass 0: every lane sorts its E contiguous words (merges 1 .. G)
        base = lane * E
        for lk in range(1, G + 1):
            for i in range(E):
                if not i & (1 << (lk - 1)):
                    ce(base + i, base + (i ^ ((1 << lk) - 1)))
            for lj in range(lk - 2, -1, -1):
                for i in range(E):
                    if not (i >> lj) & 1:
                        ce(base + i, base + (i | (1 << lj)))
    for lk in range(G + 1, LOGN + 1):
        sh = lk - G
        touched = set()
        for lane in range(64):   # mirror level + butterflies on bits lk - 2 .. lk - G
            block, hbase = lane >> sh, lane & ((1 << sh) - 1)
            lo0, hi0 = (block << lk) + hbase, (block << lk) + ((1 << lk) - 1) - hbase
            x = [lo0 + (m << sh) for m in range(H)]
            y = [hi0 - (m << sh) for m in range(H)]
            touched.update(x + y)
            for m in range(H):
                ce(x[m], y[m])
            for t in range(G - 2, -1, -1):
                for m in range(H):
                    if not (m >> t) & 1:
                        ce(x[m], x[m | (1 << t)])
                        ce(y[m | (1 << t)], y[m])
        assert len(touched) == N, "every word belongs to exactly one lane in a pass"
        passes += 1
        R = lk - G
        while R > 0:
            g = min(G, R)
            touched = set()
            for lane in range(64):
                for c in range(E >> g):
                    gid = c * 64 + lane
                    base = ((gid >> (R - g)) << R) | (gid & ((1 << (R - g)) - 1))
                    e = [base + (m << (R - g)) for m in range(1 << g)]
                    assert not touched & set(e)
                    touched.update(e)
                    for t in range(g - 1, -1, -1):
                        for m in range(1 << g):
                            if not (m >> t) & 1:
                                ce(e[m], e[m | (1 << t)])
            assert len(touched) == N
            passes += 1
            R -= g
    return key, passes


@pytest.mark.parametrize("G, n", [(3, 257), (3, 300), (3, 320), (3, 400), (3, 511), (3, 512), (4, 513), (4, 777), (4, 1024)])
def test_wave_network_sorts_padded_lists(G, n):
    N = 64 << G
    rng = np.random.default_rng(100 * G + n)
    depth = rng.uniform(0.002, 90.0, size=n).astype(np.float32)
    depth[rng.integers(0, n, size=n // 8)] = depth[0]          # ties: the position decides
    words = (depth.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    pad = np.uint64(0x7FEFFFFFFFFFFFFF)
    full = np.concatenate([words, np.full(N - n, pad)])
    # the kernel compares the words as DOUBLES: positive finite floats' bits in the high half give positive finite doubles in the same order
    as_double = full.view(np.float64)
    assert np.all(np.isfinite(as_double)) and np.all(as_double > 0)
    assert np.array_equal(np.argsort(as_double[:n], kind="stable"), np.argsort(words, kind="stable")) and as_double[:n].max() < as_double[n:].min(initial=np.inf)
    out, passes = wave_sort_model(full, G)
    assert np.array_equal(out[:n], np.sort(words)) and np.all(out[n:] == pad)
    assert passes == (16 if G == 3 else 15)
