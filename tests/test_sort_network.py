"""CPU model of the compositing kernels' LDS sorting network (das3r_amd/csrc/render_common.h: local_order_tile, the
256 < n <= 1024 branch): the always-ascending bitonic network, one thread per comparison, with the partner-out-of-range
rule instead of padding.  Checks (i) that it sorts every length, (ii) the claim the kernel's barrier placement rests on:
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
