"""CPU model of the arithmetic the four-lanes-per-pixel forward kernel (das3r_amd/csrc/render_lanes.hip) relies on, in numpy
float32 — every operation rounded as the GPU rounds it.

The reference's per-pixel loop (upstream forward.cu renderCUDA; oracle/raster_oracle.c):
    for every list entry with alpha >= 1/255 (others are skipped):  test_T = T (1 - alpha);  if test_T < 1e-4: stop
                                                                    C += c alpha T;  T = test_T;  last = position
The kernel takes FOUR consecutive entries per step, one per lane of a quad, alpha = 0 standing for a skipped entry:
(1) T in front of entry k = ((T f_0) f_1) f_2 with f_j = max(1 - alpha_j, m_jk), m_jk = 0 for j < k and 1 else: the factors of the
    lanes behind are exactly 1 (1 - alpha <= 1), so the product IS the sequential one, bit for bit;
(2) no stop inside the step <=> every lane's test_T = (T in front of it) (1 - alpha) is >= 1e-4; then T = lane 3's test_T;
(3) otherwise, with s_k = [test_T_k >= 1e-4]: the entries with s = 1 are exactly those in front of the first failure (test_T falls
    along the quad), they are blended as they are; T = min over the quad of (s_k ? test_T_k : T) is T behind the last entry taken;
    the pixel is live afterwards iff every s_k = 1;
(4) colour: each lane adds its own entries, the four partial sums are added at the end — the same terms in another order.
final T, the last contributor and the stop position must equal the sequential loop's exactly; the colour within 2e-6."""
import numpy as np

F = np.float32
THRESH = F(0.0001)


def _sequential(alpha, col):
    T, C, last, stopped_at = F(1.0), np.zeros(3, dtype=F), 0, None
    for i, a in enumerate(alpha):
        if a == 0:
            continue
        test_T = F(T * F(F(1.0) - a))
        if test_T < THRESH:
            stopped_at = i
            break
        w = F(a * T)
        for ch in range(3):
            C[ch] = F(np.float64(col[i, ch]) * np.float64(w) + np.float64(C[ch]))   # fma (the product of two floats is exact in double)
        T = test_T
        last = i + 1
    return T, C, last, stopped_at


def _quad(alpha, col):
    n = len(alpha)
    T, live = F(1.0), F(1.0)
    Cpart = np.zeros((4, 3), dtype=F)
    last = np.zeros(4, dtype=np.int64)
    stopped_at = None
    for t in range(0, n, 4):
        a = np.zeros(4, dtype=F)
        for k in range(4):
            if t + k < n:
                a[k] = F(alpha[t + k] * live)
        om = (F(1.0) - a).astype(F)
        x = np.zeros(4, dtype=F)
        for k in range(4):
            v = T
            for j in range(3):
                f = max(om[j], F(0.0) if k > j else F(1.0))
                v = F(v * f)
            x[k] = v
        tn = (x * om).astype(F)
        if not np.any(tn < THRESH):                      # (2)
            s = np.ones(4, dtype=F)
            T_next, live_next = tn[3], live
        else:                                            # (3)
            s = np.where(tn < THRESH, F(0.0), F(1.0)).astype(F)
            T_next = min(tn[k] if s[k] != 0 else T for k in range(4))
            live_next = F(live * s.min())
            if stopped_at is None and live_next == 0 and live != 0:
                stopped_at = t + int(np.argmin(s))
        for k in range(4):
            w = F(a[k] * s[k])
            wT = F(w * x[k])
            if t + k < n:
                for ch in range(3):
                    Cpart[k, ch] = F(np.float64(col[t + k, ch]) * np.float64(wT) + np.float64(Cpart[k, ch]))
                if w > 0:
                    last[k] = t + k + 1
        T, live = F(T_next), live_next
    C = (F(Cpart[0] + Cpart[1]) + F(Cpart[2] + Cpart[3])).astype(F)
    return T, C, int(last.max()), stopped_at


def _case(rng, n, strength, skip):
    alpha = np.minimum(F(0.99), (rng.random(n) ** 2 * strength).astype(F))
    alpha[alpha < F(1.0 / 255.0)] = 0                    # the kernel's alpha_if_visible: invisible entries carry alpha = 0
    alpha[rng.random(n) < skip] = 0
    col = rng.random((n, 3)).astype(F)
    return alpha, col


def test_quad_steps_reproduce_the_sequential_loop():
    rng = np.random.default_rng(5)
    stops = 0
    for trial in range(400):
        n = int(rng.integers(1, 90))
        alpha, col = _case(rng, n, strength=[0.05, 0.3, 0.9, 3.0][trial % 4], skip=[0.0, 0.3, 0.7][trial % 3])
        Ts, Cs, ls, ss = _sequential(alpha, col)
        Tq, Cq, lq, sq = _quad(alpha, col)
        assert Tq.tobytes() == Ts.tobytes(), (trial, Tq, Ts)          # bit for bit
        assert lq == ls and sq == ss, (trial, lq, ls, sq, ss)
        assert np.abs(Cq - Cs).max() <= 2e-6 * max(1.0, float(np.abs(Cs).max())), (trial, Cq, Cs)
        stops += ss is not None
    assert 50 < stops < 350                                           # both paths of a step were exercised


def test_a_stop_at_every_lane_of_a_quad_and_invisible_entries_behind_it():
    """The stopping entry at lane 0, 1, 2, 3 of its step, invisible entries (alpha = 0) in front of and behind it, a visible entry
    behind it in the same step (must not be blended), and a list that ends inside a step."""
    for lane in range(4):
        for tail in ([], [0.0], [0.5], [0.0, 0.7, 0.0]):
            for lead in (0, 1, 4, 6):
                front = [0.9] * 3 + [0.0] * lead             # T = 1e-3 after three entries, then `lead` invisible ones
                pad = (lane - len(front)) % 4
                alpha = np.array(front + [0.0] * pad + [0.95] + tail, dtype=F)   # 1e-3 * 0.05 < 1e-4: the 0.95 entry stops the pixel
                assert (len(front) + pad) % 4 == lane
                col = np.linspace(0.1, 0.9, 3 * len(alpha)).reshape(-1, 3).astype(F)
                Ts, Cs, ls, ss = _sequential(alpha, col)
                Tq, Cq, lq, sq = _quad(alpha, col)
                assert ss == len(front) + pad and sq == ss
                assert Tq.tobytes() == Ts.tobytes() and lq == ls == 3
                assert np.abs(Cq - Cs).max() <= 2e-6


def test_factors_of_the_lanes_behind_are_exactly_one():
    rng = np.random.default_rng(6)
    om = (F(1.0) - np.minimum(F(0.99), rng.random(10000).astype(F))).astype(F)
    assert np.all(np.maximum(om, F(1.0)) == F(1.0)) and np.all(np.maximum(om, F(0.0)) == om)
    T = rng.random(10000).astype(F)
    assert np.all((T * np.maximum(om, F(1.0))).astype(F) == T)       # x 1 is exact: the product of a lane is the sequential product
