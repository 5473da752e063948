// sort_onesweep.hip — single-pass-per-digit stable radix passes (two-level look-back over published counts) used by the
// rasterizer's binning: ONE kernel per 8-bit digit instead of histogram + row scan + scatter, and no second read of the
// keys.  At 100 k splats the binning was launch/latency bound (22 launches of ~4.5 us each); this path needs 11.
//
// Each 256-lane workgroup takes a ticket (arrival order => every predecessor is already running: no dependence on
// dispatch order), ranks its 256*IPL keys exactly like the classic scatter kernel (wave-private LDS counters, ballot
// match-any), publishes its per-digit counts and finds the number of equal-digit keys in all earlier workgroups from
// their published words (two levels: workgroups of its own group + totals of the earlier groups, see the kernel).
//
// Inter-workgroup hand-off follows the MI355X rule for XCD-private L2s (cdna_hip_programming.md §6 G16, recipe R2): the
// payload IS the flag — one naturally aligned 8-byte {tag, value} word per (workgroup, digit), written with ONE relaxed
// agent-scope atomic store (sc1, write-through) and polled with relaxed agent-scope atomic loads; no fences, no plain
// accesses to shared words.  Status words, tickets and global histograms are zeroed by an earlier kernel of the same
// forward (preprocess / a memset), never by a previous call.  Spins are bounded: on timeout an error word is raised and
// the forward reports DAS3R_ERR_HIP instead of hanging the GPU.
#include "granule.h"

namespace das3r {

// Round 6: PACKED status rows.  A workgroup's count of a digit is at most 4096: four digits share one 8-byte
// word (bit 15 of each 16-bit field = published, one store publishes all four), a row is 64 words = 512 bytes instead of 2 KB; a group's
// total (<= 2^19) takes 32 bits (bit 31 = published), two per word, 1 KB a row.  The four (two) threads whose digits share a word poll the
// same address: a wave's 64 loads are ONE 128-byte line instead of four.  What every workgroup reads in its look-back (~ 55 rows) was
// 70 MB per pass at 1 M splats, more than the keys; the rows are also zeroed by every forward.
constexpr int ROW_WORDS = RADIX_SIZE / 4;          // u64 words of a workgroup's status row
constexpr int GROUP_ROW_WORDS = RADIX_SIZE / 2;    // ... of a group's row

// Sum of digit `field`'s values in `count` published rows, `stride` words apart, starting at `col` (the word of the first row that
// holds the digit); a value is BITS wide, its top bit the flag.  Loads go out in windows of LB independent requests; a window with an
// unpublished word is re-polled as a whole.
// Slots past `count` are NOT loaded.  (Measured on MI355X: padding the window by re-reading the last word — up to LB
// back-to-back sc1 loads of one address per poll — made published words invisible to some pollers for seconds, i.e.
// look-back timeouts in ~30 % of 1M-splat forwards; sc1 loads are L2-served, MI355X_MICROARCH.md.)
template <int BITS, int LB = 16>
__device__ __forceinline__ uint32_t sum_published(const u64 *col, const int stride, const int field, const int count, uint32_t *err) {
    uint32_t sum = 0;
    unsigned spins = 0;
    const int sh = field * BITS;
    const u64 flag = 1ull << (sh + BITS - 1), vmask = (1ull << (BITS - 1)) - 1ull;
    for (int p = 0; p < count; p += LB) {
        u64 x[LB];
        while (true) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < LB; j++) x[j] = (p + j < count) ? granule_poll(col + (size_t)(p + j) * stride, spins) : ~0ull;
#pragma unroll
            for (int j = 0; j < LB; j++) ok &= (x[j] & flag) != 0;
            if (ok) break;
            if (spins == SOFT_SPINS) atomicOr(err, ERR_HARD_POLL);
            if (++spins > SPIN_LIMIT) {  // a predecessor never published: give up loudly instead of hanging
                atomicOr(err, ERR_TIMEOUT);
                return sum;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < LB; j++) sum += (p + j < count) ? (uint32_t)((x[j] >> sh) & vmask) : 0u;
    }
    return sum;
}

// thread d holds digit d's value (v < 2^15): the word of digits 4j .. 4j + 3 lands in lane 4j (other lanes: garbage)
__device__ __forceinline__ u64 pack_row_word(const uint32_t v) {
    const uint32_t f = v | 0x8000u;
    const uint32_t pair = f | ((uint32_t)__shfl_down((int)f, 1, 64) << 16);
    return (u64)pair | ((u64)(uint32_t)__shfl_down((int)pair, 2, 64) << 32);
}
// ... digit d's group total (v < 2^31): the word of digits 2j, 2j + 1 lands in lane 2j
__device__ __forceinline__ u64 pack_group_word(const uint32_t v) {
    const uint32_t f = v | 0x80000000u;
    return (u64)f | ((u64)(uint32_t)__shfl_down((int)f, 1, 64) << 32);
}

// Round 6: the look-back read by a WAVE, not by a thread.  Wave w owns digits 64 w .. 64 w + 63, i.e. WPW = 64 / FPW consecutive words of
// every row (FPW fields of BITS bits per word); a load instruction of the wave covers 64 / WPW ROWS at once (lane = row phase x word), so
// the 63 rows the last workgroup of a group of 64 sums are 16 loads per lane — ONE window, one trip to memory — where a thread reading its
// own digit's word of every row needed four windows of 16, back to back, and that workgroup's group total is what every later group
// waits for.  Row phases are folded with xor shuffles, then lane l fetches field l % FPW of word l / FPW: digit (64 w + l)'s sum in lane l,
// exactly what sum_published returns.  Same polling rules (no padding loads of an address already polled, soft / hard spin limits).
template <int BITS>
__device__ __forceinline__ uint32_t sum_published_wave(const u64 *rows, const int stride, const int count, uint32_t *err) {
    constexpr int FPW = 64 / BITS, WPW = 64 / FPW, PH = 64 / WPW, LB = 16;
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    const int wi = lane & (WPW - 1), ph = lane / WPW;
    const u64 *col = rows + wave * WPW + wi;
    constexpr u64 flag = 1ull << (BITS - 1), vmask = (1ull << (BITS - 1)) - 1ull;
    uint32_t acc[FPW];
#pragma unroll
    for (int f = 0; f < FPW; f++) acc[f] = 0u;
    unsigned spins = 0;
    for (int p = 0; p < count; p += LB * PH) {
        u64 x[LB];
        while (true) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < LB; j++) {
                const int row = p + j * PH + ph;
                x[j] = row < count ? granule_poll(col + (size_t)row * stride, spins) : ~0ull;
            }
#pragma unroll
            for (int j = 0; j < LB; j++) ok &= (x[j] & flag) != 0;
            if (__all(ok)) break;   // (wave-uniform: every lane's words are what the others' sums need)
            if (spins == SOFT_SPINS && lane == 0) atomicOr(err, ERR_HARD_POLL);
            if (++spins > SPIN_LIMIT) {  // a predecessor never published: give up loudly instead of hanging
                if (lane == 0) atomicOr(err, ERR_TIMEOUT);
                return 0u;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < LB; j++) {
            const int row = p + j * PH + ph;
            if (row < count) {
#pragma unroll
                for (int f = 0; f < FPW; f++) acc[f] += (uint32_t)((x[j] >> (f * BITS)) & vmask);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < FPW; f++) {
#pragma unroll
        for (int o = WPW; o < 64; o <<= 1) acc[f] += (uint32_t)__shfl_xor((int)acc[f], o, 64);
    }
    uint32_t out = 0u;
#pragma unroll
    for (int f = 0; f < FPW; f++) {
        const uint32_t v = (uint32_t)__shfl((int)acc[f], lane / FPW, 64);
        if (lane % FPW == f) out = v;
    }
    return out;
}

// global histograms of all four 8-bit digits of the depth keys (read once, LDS-privatised, few global atomics)
__global__ void __launch_bounds__(256) depth_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t h[4][RADIX_SIZE];
#pragma unroll
    for (int q = 0; q < 4; q++) h[q][threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t k = keys[i];
        atomicAdd(&h[0][k & 255u], 1u);
        atomicAdd(&h[1][(k >> 8) & 255u], 1u);
        atomicAdd(&h[2][(k >> 16) & 255u], 1u);
        atomicAdd(&h[3][k >> 24], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t c = h[q][threadIdx.x];
        if (c) atomicAdd(&ghist[q * RADIX_SIZE + threadIdx.x], c);
    }
}

// One digit of a stable LSD sort / partition.  keys_out may be null; vals_in null => payload = index.
// TWO (the tile partition): a second payload travels with every key — the emission slot e of the instance (vals2_in null: e = the
// input index, i.e. the first pass) beside the splat id; the last pass leaves the ids in point_list and the slots in slot_list
// (where the backward pass puts the instance's partial sums), both in list order, all loads and stores coalesced.  Round 2
// carried e alone and gathered gid_of[e] in the last pass: a 64-byte sector per 4-byte word, 167 of the pass's 240 MB at 1 M splats.
template <int IPL, bool TWO>
__global__ void __launch_bounds__(256) onesweep_pass_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                            uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                            uint32_t cap, const uint32_t *__restrict__ n_ptr, int shift, int bits,
                                                            const uint32_t *__restrict__ ghist /*[256] this digit*/,
                                                            u64 *__restrict__ status /*[nblocks][ROW_WORDS]*/,
                                                            u64 *__restrict__ group_status /*[ngroups][GROUP_ROW_WORDS]*/, int gs_log2,
                                                            uint32_t *__restrict__ ticket,
                                                            const uint32_t *__restrict__ vals2_in, uint32_t *__restrict__ vals2_out,
                                                            uint32_t *__restrict__ err, uint32_t *__restrict__ zero_ptr, uint32_t zero_words
#ifdef DAS3R_EXPERIMENTS
                                                            , unsigned long long *__restrict__ trace /*common.h WG_STAMP*/
#endif
                                                            ) {
    // side duty (last depth pass only): zero the control words of the binning buffer, which did not exist yet when the
    // preprocess kernel zeroed everything else; nothing in this kernel touches them
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < zero_words; i += gridDim.x * 256u) zero_ptr[i] = 0u;
    // (16-bit: a workgroup holds at most 4096 keys — with 32-bit counters the 16-key instantiation needs 54 308 B, a few hundred bytes
    //  past a third of the CU's 160 KB once the allocation is rounded up: two workgroups per CU instead of three)
    __shared__ uint16_t cnt[4][RADIX_SIZE];  // per-wave digit counts, then per-wave running offsets
    __shared__ uint32_t gdelta[RADIX_SIZE];
    __shared__ uint32_t sk[256 * IPL], sv[256 * IPL];  // staging: key, payload
    __shared__ uint32_t sv2[TWO ? 256 * IPL : 1];      //          second payload
    __shared__ uint32_t ws[8];
    __shared__ uint32_t s_block;
    const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6;
    const uint32_t n = n_ptr ? min(*n_ptr, cap) : cap;
    if (tid == 0) s_block = ticket ? atomicAdd(ticket, 1u) : blockIdx.x;   // null: the whole grid is resident (scan_emit.hip)
#pragma unroll
    for (int w = 0; w < 4; w++) cnt[w][tid] = 0;
    __syncthreads();
    const uint32_t b = s_block;
    // A workgroup past the end of the list has nothing to sort and nobody behind it that has: it leaves at once.  (The grid is
    // sized for the CAPACITY of the binning buffer — 25 % over the last count on the speculative path: at 1 M splats 799
    // workgroups for 639 with keys, and 768 fit the chip at a time; the idle ones used to run every phase, look-back included.)
    const uint32_t last_block = n > 0u ? (n - 1u) / (256u * (uint32_t)IPL) : 0u;
    if (b > last_block) return;
    WG_STAMP(0)   // ticket taken
    const uint32_t base = (b * 4u + (uint32_t)wave) * 64u * (uint32_t)IPL;
    const uint32_t mask = (1u << bits) - 1u;

    uint32_t k[IPL], v[IPL], v2[TWO ? IPL : 1];
#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = base + s * 64 + lane;
        k[s] = i < n ? keys_in[i] : 0u;
        v[s] = (vals_in && i < n) ? vals_in[i] : i;
        if (TWO) v2[s] = (vals2_in && i < n) ? vals2_in[i] : i;
    }
    // Count AND rank in one sweep, without LDS atomics (ds_add costs ~12 cycles per active lane on this chip: 16 of them per lane
    // were the longest phase of the pass).  Per row of 64 keys: match-any over the digit bits -> every key knows its peers; the
    // first peer reads the wave's running count of the digit and adds the group's size (plain read + write: one writer per digit
    // and row, rows in program order, LDS in order per wave).  local[s] = (same-digit keys of this wave before this one).
#ifdef DAS3R_EXPERIMENTS
    if (trace != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP(1)   // keys and payloads arrived
#endif
    uint32_t local[IPL];
    {
        volatile uint16_t *cw = cnt[wave];
#pragma unroll
        for (int s = 0; s < IPL; s++) {
            const uint32_t i = base + s * 64 + lane;
            const bool valid = i < n;
            const uint32_t digit = (k[s] >> shift) & mask;
            // match-any over the 8 digit bits (bits above `bits` are zero in every lane: harmless).  Kept in 32-bit halves so
            // that every step is v_xnor + v_and on VGPRs; rank = v_mbcnt of the peer mask
            const uint64_t vm = __ballot(valid);
            uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
#pragma unroll
            for (int bb = 0; bb < RADIX_BITS; bb++) {
                const uint32_t sel = 0u - ((digit >> bb) & 1u);          // all ones if my bit is set
                const uint64_t m = __ballot((digit >> bb) & 1u);
                plo &= ~((uint32_t)m ^ sel);
                phi &= ~((uint32_t)(m >> 32) ^ sel);
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t count = (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
            uint32_t before = 0;
            if (valid) before = cw[digit];
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) cw[digit] = (uint16_t)(before + count);
            __builtin_amdgcn_wave_barrier();
            local[s] = before + rank;
        }
    }
    __syncthreads();
    WG_STAMP(2)   // counted and ranked

    // thread d owns digit d.  Publish this workgroup's count first, then do everything that needs only LOCAL information
    // (ranking + staging, ~3 us); the look-back over the earlier workgroups' counts comes after it, when their words have
    // long been published, and only the final write-out needs its result.
    const uint32_t c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid], c3 = cnt[3][tid];
    const uint32_t total = c0 + c1 + c2 + c3;
    {
        const u64 w = pack_row_word(total);
        if ((tid & 3) == 0) granule_store(status + (size_t)b * ROW_WORDS + (tid >> 2), w);
    }
    const uint32_t g = ghist[tid];
    uint32_t digit_base, lstart;
    {
        // global: first output slot of digit d = (keys with smaller digits) + (same digit in earlier workgroups);
        // local: slot of this workgroup's first digit-d key inside its own LDS staging area (keys grouped by digit)
        const uint32_t incl = wave_incl_scan_u32(g), lincl = wave_incl_scan_u32(total);
        if (lane == 63) {
            ws[wave] = incl;
            ws[4 + wave] = lincl;
        }
        __syncthreads();
        uint32_t wbase = 0, lbase = 0;
#pragma unroll
        for (int w = 0; w < 4; w++)
            if (w < wave) {
                wbase += ws[w];
                lbase += ws[4 + w];
            }
        lstart = lbase + lincl - total;
        digit_base = wbase + incl - g;
        cnt[0][tid] = (uint16_t)lstart;
        cnt[1][tid] = (uint16_t)(lstart + c0);
        cnt[2][tid] = (uint16_t)(lstart + c0 + c1);
        cnt[3][tid] = (uint16_t)(lstart + c0 + c1 + c2);
    }
    __syncthreads();

    // stage every key at its local slot: the staging area ends up grouped by digit, in stable order
    //   slot = (digit's first slot in the workgroup + same-digit keys of earlier waves) [cnt[wave][digit] now] + local[s]
#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = base + s * 64 + lane;
        if (i < n) {
            const uint32_t slot = cnt[wave][(k[s] >> shift) & mask] + local[s];
            sk[slot] = k[s];
            sv[slot] = v[s];
            if (TWO) sv2[slot] = v2[s];
        }
    }
    WG_STAMP(3)   // published, staged
    {
        // Two-level look-back.  Workgroups are grouped GS = 2^gs_log2 at a time; the last workgroup of a group also
        // publishes the group's total.  Every workgroup then needs (its predecessors inside its group) + (the totals of
        // all earlier groups): <= GS - 1 + ngroups - 1 ~ 2 sqrt(nblocks) words, ALL INDEPENDENT loads (no chain of
        // prefixes to chase), i.e. two far-memory round trips per pass however many workgroups run at once.
        // (The one-level chained scan degenerates when every workgroup is resident from the start — the normal case for
        // 0.1-4 M keys on 256 CUs: nobody owns a prefix yet, so workgroup b walks over all b predecessors.)
        const uint32_t gs_mask = (1u << gs_log2) - 1u;
        const uint32_t grp = b >> gs_log2, r = b & gs_mask;
        // (Reading both columns in one sweep — the earlier groups' totals do not depend on this group's words — does not shorten
        //  the look-back: 8.8 vs 8.4 us median at 1 M splats, tools/wg_trace.py.  What a workgroup waits for is the total of the
        //  group just before its own, which that group's last workgroup publishes only after staging and summing its own group;
        //  and the 55 rows of 2 KB every workgroup reads are 70 MB per pass, more than the keys.)
        const uint32_t in_group = sum_published_wave<16>(status + (size_t)(b - r) * ROW_WORDS, ROW_WORDS, (int)r, err);
        if (r == gs_mask) {   // (uniform)
            const u64 w = pack_group_word(in_group + total);
            if ((tid & 1) == 0) granule_store(group_status + (size_t)grp * GROUP_ROW_WORDS + (tid >> 1), w);
        }
        const uint32_t before_group = sum_published_wave<32>(group_status, GROUP_ROW_WORDS, (int)grp, err);
        const uint32_t excl = in_group + before_group;
        if (b == last_block && excl + total != g) atomicOr(err, ERR_COUNTS);   // self-check: all counts add up to the histogram
        gdelta[tid] = digit_base + excl - lstart;   // destination of staged slot i holding digit d: i + gdelta[d]
    }
    __syncthreads();
    WG_STAMP(4)   // look-back done (the whole workgroup)

    // write out in staged order: neighbouring lanes hold neighbouring slots of the same digit => every digit's run of this
    // workgroup goes out as contiguous, coalesced stores (a direct scatter issues 64 separate 4-byte writes per
    // instruction, which the L2 cannot merge once 256 digits x hundreds of workgroups have open write heads)
    const uint32_t block_first = b * 256u * (uint32_t)IPL;
    const uint32_t nvalid = n > block_first ? min(n - block_first, 256u * (uint32_t)IPL) : 0u;
#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = s * 256 + tid;
        if (i < nvalid) {
            const uint32_t key = sk[i];
            const uint32_t dst = i + gdelta[(key >> shift) & mask];
            if (dst >= n) {  // never write out of bounds, whatever went wrong upstream
                atomicOr(err, ERR_RANGE);
                continue;
            }
            if (keys_out) keys_out[dst] = key;
            vals_out[dst] = sv[i];
            if (TWO) vals2_out[dst] = sv2[i];
        }
    }
    WG_STAMP(5)   // stores issued
#ifdef DAS3R_EXPERIMENTS
    if (trace != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WG_STAMP(6)   // stores acknowledged
        if (threadIdx.x == 0 && s_block < (uint32_t)TRACE_WGS) trace[((size_t)(shift >> 3) * TRACE_WGS + s_block) * TRACE_STAMPS + 7] = blockIdx.x;
    }
#endif
}

static int onesweep_group_log2(int nblocks);

#ifdef DAS3R_EXPERIMENTS
#define TRACE_ARG , wg_trace()
#else
#define TRACE_ARG
#endif
static int onesweep_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int64_t cap, const uint32_t *n_ptr,
                         int shift, int bits, const uint32_t *ghist, u64 *status, uint32_t *ticket, const uint32_t *v2in,
                         uint32_t *v2out, uint32_t *err, bool debug, hipStream_t s, uint32_t *zero_ptr = nullptr,
                         uint32_t zero_words = 0) {
    const int ipl = sort_items_per_lane(cap);
    const int nblocks = div_up(cap, (int64_t)256 * ipl);
    const int gs_log2 = onesweep_group_log2(nblocks);
    u64 *group_status = status + (size_t)nblocks * ROW_WORDS;
#define PASS(IPL, TWO)                                                                                                    \
    DAS3R_LAUNCH((onesweep_pass_kernel<IPL, TWO>), dim3(nblocks), dim3(256), 0, s, kin, vin, kout, vout, (uint32_t)cap, n_ptr,   \
                 shift, bits, ghist, status, group_status, gs_log2, grid_is_resident(nblocks) ? (uint32_t *)nullptr : ticket, v2in, v2out, \
                 err, zero_ptr, zero_words TRACE_ARG)
    if (v2out) {
        if (ipl == 4) PASS(4, true); else if (ipl == 8) PASS(8, true); else PASS(16, true);
    } else {
        if (ipl == 4) PASS(4, false); else if (ipl == 8) PASS(8, false); else PASS(16, false);
    }
#undef PASS
    KERNEL_CHECK(s, debug, "onesweep_pass");
    return DAS3R_OK;
}

// group size of the two-level look-back: the power of two nearest above sqrt(nblocks), within [4, 128]
static int onesweep_group_log2(int nblocks) {
    int l = 2;
    while (l < 7 && (1 << (2 * l)) < nblocks) l++;
    return l;
}

// per pass: one packed row per workgroup + one per group
size_t onesweep_status_bytes(int64_t n, int passes) {
    if (n <= 0) return 256;
    const int nblocks = div_up(n, (int64_t)256 * sort_items_per_lane(n));
    const int ngroups = div_up(nblocks, 1 << onesweep_group_log2(nblocks));
    return (size_t)passes * ((size_t)nblocks * ROW_WORDS + (size_t)ngroups * GROUP_ROW_WORDS) * sizeof(u64);
}

// Depth sort of the P splats: ctrl = [ghist 4x256 u32][tickets 4 u32 (+pad)][status 4 passes], zeroed by preprocess_kernel.
// part: 0 = histogram + passes 0..2 (enqueued before the host knows num_rendered), 1 = pass 3, which also zeroes
// zero_ptr[0..zero_words) — the binning buffer's control words
int launch_onesweep_depth_sort(int P, char *geom, const Layout &L, int part, uint32_t *zero_ptr, uint32_t zero_words, bool debug,
                               hipStream_t s) {
    uint32_t *keyA = (uint32_t *)(geom + L.g_keyA), *keyB = (uint32_t *)(geom + L.g_keyB);
    uint32_t *valA = (uint32_t *)(geom + L.g_valA), *valB = (uint32_t *)(geom + L.g_valB);
    uint32_t *ghist = (uint32_t *)(geom + L.g_ghist), *ticket = (uint32_t *)(geom + L.g_ticket);
    uint32_t *err = (uint32_t *)(geom + L.g_ticket) + 8;  // inside the zeroed control region
    u64 *status = (u64 *)(geom + L.g_status);
    const size_t per_pass = onesweep_status_bytes(P, 1) / sizeof(u64);
    int rc;
    if (part == 1)
        return onesweep_pass(keyB, valB, nullptr, valA, P, nullptr, 24, 8, ghist + 768, status + 3 * per_pass, ticket + 3, nullptr, nullptr, err,
                             debug, s, zero_ptr, zero_words);
    const int hist_blocks = div_up(P, 256) < 512 ? div_up(P, 256) : 512;
    DAS3R_LAUNCH(depth_hist_kernel, dim3(hist_blocks), dim3(256), 0, s, keyA, (uint32_t)P, ghist);
    KERNEL_CHECK(s, debug, "depth_hist");
    // A -> B -> A -> B -> A ; final ranks land in valA (== pub.sorted_idx)
    if ((rc = onesweep_pass(keyA, nullptr, keyB, valB, P, nullptr, 0, 8, ghist + 0, status + 0 * per_pass, ticket + 0, nullptr, nullptr, err, debug, s))) return rc;
    if ((rc = onesweep_pass(keyB, valB, keyA, valA, P, nullptr, 8, 8, ghist + 256, status + 1 * per_pass, ticket + 1, nullptr, nullptr, err, debug, s))) return rc;
    if ((rc = onesweep_pass(keyA, valA, keyB, valB, P, nullptr, 16, 8, ghist + 512, status + 2 * per_pass, ticket + 2, nullptr, nullptr, err, debug, s))) return rc;
    return DAS3R_OK;
}

// Stable partition of the emitted instances by tile id.  ctrl (binning buffer) = [ghist 2x256][tickets][status], zeroed by
// a memset before emit; emit_kernel accumulated the two digit histograms.
int launch_onesweep_partition(int64_t cap, char *geom, char *binning, const Layout &L, uint32_t **keys_final, bool debug, hipStream_t s,
                              uint32_t *ghist_override, uint32_t *err_override) {
    const uint32_t *n_ptr = (const uint32_t *)(geom + L.g_count);
    uint32_t *err = err_override ? err_override : (uint32_t *)(geom + L.g_ticket) + 8;  // inside the zeroed control region
    uint32_t *keyA = (uint32_t *)(binning + L.b_keyA), *keyB = (uint32_t *)(binning + L.b_keyB);
    uint32_t *valA = (uint32_t *)(binning + L.b_valA), *valB = (uint32_t *)(binning + L.b_valB);
    uint32_t *gid_of = (uint32_t *)(binning + L.b_gid_of), *slot_list = (uint32_t *)(binning + L.b_slot), *e_tmp = (uint32_t *)(binning + L.b_e2);
    uint32_t *ghist = ghist_override ? ghist_override : (uint32_t *)(binning + L.b_ghist), *ticket = (uint32_t *)(binning + L.b_ticket);
    u64 *status = (u64 *)(binning + L.b_status);
    const size_t per_pass = onesweep_status_bytes(cap, 1) / sizeof(u64);
    // payloads: the splat id (read from gid_of by the first pass, then ping-pong so that the LAST pass writes the buffer the layout calls
    // point_list, whatever the number of passes: the backward pass lays the buffers out from tile_passes alone) and the emission slot e
    // (the input index in the first pass, then ping-pong so that the last pass writes slot_list)
    uint32_t *const pl_final = (uint32_t *)(binning + L.pub.point_list), *const pl_other = pl_final == valA ? valB : valA;
    const int passes = L.part_passes;
    uint32_t *kin = keyA, *kout = keyB;
    const uint32_t *vin = gid_of, *v2in = nullptr;
    int shift = 0, rc;
    for (int p = 0; p < passes; p++) {
        const int dw = tile_digit_width(L.kbits), bits = (L.kbits - shift) < dw ? (L.kbits - shift) : dw;   // (kbits = tbits unless segmented: common.h)
        const int at = shift + L.kshift;   // (segmented: the digits sit above the fraction bits of the key)
        uint32_t *vout = ((passes - 1 - p) & 1) ? pl_other : pl_final;
        uint32_t *v2out = ((passes - 1 - p) & 1) ? e_tmp : slot_list;
        if ((rc = onesweep_pass(kin, vin, kout, vout, cap, n_ptr, at, bits, ghist + p * 256, status + p * per_pass, ticket + p,
                                v2in, v2out, err, debug, s)))
            return rc;
        shift += bits;
        uint32_t *t = kin; kin = kout; kout = t;
        vin = vout;
        v2in = v2out;
    }
    *keys_final = kin;
    return DAS3R_OK;
}

}  // namespace das3r
