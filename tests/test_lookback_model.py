"""CPU model of the look-back index arithmetic of the radix passes (das3r_amd/csrc/sort_onesweep.hip, round 6): status rows PACKED four
digit counts per 8-byte word (group rows: two totals per word) and READ BY A WAVE — lane = row phase x word, row phases folded with xor
shuffles, lane l fetching field l % FPW of word l / FPW.  The model restates pack_row_word / pack_group_word / sum_published_wave lane
by lane in numpy and checks that lane l of wave w ends with the sum over the rows of digit 64 w + l's count — for every row count the
kernel meets (one window, several windows, a ragged last window), with flag bits that must not leak into the sums."""
import numpy as np
import pytest

RADIX = 256


def pack_rows(counts, bits):
    """counts [rows, 256] -> words [rows, 256 * bits / 64] (uint64), as the publishing threads assemble them with shuffles."""
    fpw = 64 // bits
    flag = 1 << (bits - 1)
    assert counts.max() < flag
    f = (counts.astype(np.uint64) | np.uint64(flag)).reshape(counts.shape[0], RADIX // fpw, fpw)
    words = np.zeros(f.shape[:2], dtype=np.uint64)
    for k in range(fpw):   # digit (fpw j + k) sits in bits [bits k, bits (k + 1)) of word j
        words |= f[:, :, k] << np.uint64(bits * k)
    return words


def wave_read(words, bits, count, lb=16):
    """sum_published_wave<bits>: -> [4 waves, 64 lanes] sums, emulated lane by lane."""
    fpw = 64 // bits
    wpw = 64 // fpw            # words of a row one wave owns
    ph_n = 64 // wpw           # rows one load instruction of the wave covers
    vmask = (1 << (bits - 1)) - 1
    out = np.zeros((4, 64), dtype=np.int64)
    for wave in range(4):
        acc = np.zeros((64, fpw), dtype=np.int64)
        for p in range(0, count, lb * ph_n):                      # windows
            for lane in range(64):
                wi, ph = lane & (wpw - 1), lane // wpw
                for j in range(lb):
                    row = p + j * ph_n + ph
                    if row < count:
                        x = int(words[row, wave * wpw + wi])
                        assert x & (1 << (bits - 1)), "an unpublished word would be polled again"
                        for f in range(fpw):
                            acc[lane, f] += (x >> (f * bits)) & vmask
        o = wpw
        while o < 64:                                            # xor shuffles fold the row phases
            acc = acc + acc[np.arange(64) ^ o]
            o <<= 1
        for lane in range(64):                                   # lane l: field l % fpw of word l / fpw (held by lane l / fpw)
            out[wave, lane] = acc[lane // fpw, lane % fpw]
    return out


@pytest.mark.parametrize("bits,limit", [(16, 4096), (32, 128 * 4096)])
@pytest.mark.parametrize("count", [0, 1, 3, 4, 31, 63, 64, 65, 127])
def test_wave_read_of_packed_rows_sums_every_digit(bits, limit, count):
    rng = np.random.default_rng(bits * 1000 + count)
    rows = max(count, 1)
    counts = rng.integers(0, limit + 1, size=(rows, RADIX))
    counts[rng.random(counts.shape) < 0.3] = 0
    if count:
        counts[0, 5] = limit            # the largest value a field ever holds: 256 IPL keys of one digit / a full group of them
    words = pack_rows(counts, bits)
    got = wave_read(words, bits, count)
    want = counts[:count].sum(0).reshape(4, 64)
    assert np.array_equal(got, want)


def test_a_packed_word_is_zero_until_it_is_published_and_never_after():
    """The poll's test: a word with every count zero still carries its flag bits (zeroed memory = nobody published)."""
    for bits in (16, 32):
        w = pack_rows(np.zeros((1, RADIX), dtype=np.int64), bits)
        assert (w != 0).all()
        fpw = 64 // bits
        for f in range(fpw):
            assert ((w >> np.uint64(bits * f + bits - 1)) & np.uint64(1)).all()
