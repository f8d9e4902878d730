// SURVEY §8(f) row N4, loader side: the data rows of an ESOM .lrn feature table, text -> integer keys.
//
// Reference behaviour restated here (reference = CK7/abawaca, src/ClusterData.cpp):
//   :36-111   four header lines, read with fgets into an 8 KiB buffer: "% <ndps>", "% <ncols>" (ncols counts the key
//             column), the weights line (only its field count is checked), "% Key<TAB>name<TAB>name..."
//   :117-152  every further line: chomp trailing white space; skip it when empty or starting with '%'; it must have
//             ncols tab-separated fields; field 0 is the datapoint name (atoi), the rest the values
//   :154-163  values = atof(field)
//   :127-130,164-167  the number of data lines must equal ndps
// abawaca-build prints every value as "%.3lf" of a 3-decimal truncation (src/abawaca-build.cpp:603), so a value is an
// integer number of thousandths; that integer is the key the separator works on (abw_sep_load).  This parser accepts
// exactly such fixed-point text -- [-]digits[.digits], more than 3 decimals only when the extra ones are zeros -- and
// reports anything else (exponents, inf/nan, blanks inside a field) as ABW_ERR_FORMAT with its line, so that a caller
// can fall back to the reference's loader for tables that did not come from abawaca-build.
//
// Kernels (byte work, HBM-bound: the text is read twice, 16 B per lane):
//   lrn_nlcount   newlines per 1 KiB block                          -> scan -> line numbers
//   lrn_lines     start offset of every line
//   lrn_classify  one thread per line: chomp, data line or not      -> scan -> row numbers
//   lrn_blocks    one wave per 1 KiB BLOCK of text, whatever lines it holds: a lane owns 16 bytes and knows the line and
//                 the field number of its first byte -- the line from the newline prefix sums, the field from a segmented
//                 wave scan of tab counts (restarting at every newline) plus the tabs between the start of the block's
//                 first line and the block (per-block tails, walked back to the last block with a newline) -- and
//                 converts every field that STARTS in its bytes (a field may run on into the next lanes' bytes; the LDS
//                 copy has a 32-byte halo).  Every lane has work whatever the line length is.  Handles the lines whose
//                 last field is followed directly by the newline (all of abawaca-build's output);
//   lrn_rows      one wave per data line, for the others (trailing white space, which the reference chomps)
#include "abw_internal.h"
#include <string.h>
#include <stdlib.h>
#include <ctype.h>

int abw_device_scan_u32(abw_ctx* ctx, const uint32_t* in, uint64_t n, uint64_t* out);   // abw_sig.hip

namespace {

constexpr int LBLK = 1008;             // bytes of text per block: 63 lanes x 16 (lane 63 of a wave holds the 16 bytes behind its block)
constexpr int RPIECE = 1024;           // lrn_rows: bytes of a line per pass
constexpr int MAX_FIELD = 24;          // longest field converted (a 3-decimal int32 key has at most 12 characters)
constexpr int REF_LINE = 8191;         // the reference reads lines in pieces of this many bytes (char line[8192])

__device__ __forceinline__ uint32_t eq_mask4(uint32_t x, uint32_t pat)     // bit i set iff byte i of x equals the pattern byte
{
    const uint32_t t = x ^ pat;
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
    return __builtin_amdgcn_udot4(z, 0x08040201u, 0u, false) >> 7;
}

__device__ __forceinline__ uint32_t eq_mask16(const uint8_t* __restrict__ p, uint64_t at, uint64_t len, uint32_t pat)
{
    if (at + 16 <= len) {
        uint4 v;
        memcpy(&v, p + at, 16);
        return eq_mask4(v.x, pat) | (eq_mask4(v.y, pat) << 4) | (eq_mask4(v.z, pat) << 8) | (eq_mask4(v.w, pat) << 12);
    }
    uint32_t m = 0;
    for (uint64_t i = at; i < len && i < at + 16; i++) m |= (uint32_t)(p[i] == (uint8_t)pat) << (i - at);
    return m;
}

// (wave_prefix_u32 / wave_last_u32 / lane_above / lane_below: abw_internal.h)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x) { return wave_prefix_u32(x); }
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t x) { return wave_last_u32(wave_prefix_u32(x)); }
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x, int) { return wave_prefix_u32(x); }

// per block: newlines, and the tabs after its last newline (all its tabs when it has none)
__global__ __launch_bounds__(256) void lrn_nlcount_kernel(const uint8_t* __restrict__ text, uint64_t len, uint32_t* __restrict__ blk_nl,
                                                         uint32_t* __restrict__ blk_tail, uint64_t nblk)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
    for (uint64_t b = wave; b < nblk; b += nwaves) {
        const uint64_t A = b * LBLK + (uint64_t)lane * 16;
        const uint64_t lim = (lane < 63) ? len : 0;                      // lane 63's bytes belong to the next block
        const uint32_t nl = eq_mask16(text, A, lim, 0x0A0A0A0Au), tb = eq_mask16(text, A, lim, 0x09090909u);
        const uint32_t c = __popc(nl);
        const uint32_t x = wave_incl_scan_u32(c, lane);
        const unsigned long long with_nl = __ballot(nl != 0);
        const int last = with_nl ? 63 - __builtin_clzll(with_nl) : -1;      // the last lane that holds a newline
        uint32_t t = 0;
        if (lane > last) t = __popc(tb);
        else if (lane == last) t = __popc(tb >> (32 - __builtin_clz(nl)));   // tabs above its highest newline bit
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 63) { blk_nl[b] = x; blk_tail[b] = t; }
    }
}

// line_start[k + 1] = offset after the k-th newline (0-based)
__global__ __launch_bounds__(256) void lrn_lines_kernel(const uint8_t* __restrict__ text, uint64_t len, const uint64_t* __restrict__ blk_prefix,
                                                       uint64_t nblk, uint64_t* __restrict__ line_start)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
    for (uint64_t b = wave; b < nblk; b += nwaves) {
        const uint64_t A = b * LBLK + (uint64_t)lane * 16;
        uint32_t m = eq_mask16(text, A, (lane < 63) ? len : 0, 0x0A0A0A0Au);
        const uint32_t c = __popc(m);
        uint64_t k = blk_prefix[b] + wave_incl_scan_u32(c, lane) - c;
        while (m) { const int i = __ffs(m) - 1; m &= m - 1; line_start[++k] = A + i + 1; }
    }
}

struct LineInfo { uint64_t begin, end; };     // [begin, end) after the chomp

__global__ __launch_bounds__(256) void lrn_classify_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_start,
                                                          uint64_t n_lines, LineInfo* __restrict__ info, uint32_t* __restrict__ is_row,
                                                          uint8_t* __restrict__ kind, unsigned long long* __restrict__ err)
{
    const uint64_t l = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= n_lines) return;
    const uint64_t b = line_start[l];
    uint64_t e = line_start[l + 1] - 1;                                  // the newline (or one past the end of the text)
    if (e - b > (uint64_t)REF_LINE - 1 && l >= 4) atomicMin(err, (unsigned long long)(l + 1) << 8 | 1);   // ref fgets would split it
    while (e > b) {                                                      // ref :119-120 chomp (isspace in the C locale)
        const uint8_t c = text[e - 1];
        if (c == ' ' || (c >= 9 && c <= 13)) e--; else break;
    }
    info[l].begin = b; info[l].end = e;
    const bool row = l >= 4 && e > b && text[b] != '%';                  // ref :123; the first four lines are the header
    is_row[l] = row ? 1u : 0u;
    kind[l] = !row ? 0 : (e == line_start[l + 1] - 1 ? 1 : 2);           // 1: nothing chomped (lrn_blocks), 2: lrn_rows
}

// error codes in the low byte of err (smallest line wins): 1 line too long, 2 field count, 3 number format, 4 key overflow
// One field, [-]digits[.digits] up to its terminator (a tab; `term2`: what else may end it), to the key table.  Returns the
// terminator found.
__device__ __forceinline__ uint8_t convert_field(const uint8_t* p, uint8_t term2, uint32_t f, uint64_t l, uint64_t row, int32_t ncols, int64_t stride,
                                                 int32_t* __restrict__ dp_names, int32_t* __restrict__ keys, unsigned long long* __restrict__ err)
{
    int j = 0;
    bool neg = false, bad = false;
    long long ip = 0;
    int frac = 0, nd = 0, nfr = 0;
    if (p[0] == '-') { neg = true; j = 1; }
    for (; j < MAX_FIELD && p[j] >= '0' && p[j] <= '9'; j++) { ip = ip * 10 + (p[j] - '0'); nd++; if (ip > 4000000000ll) bad = true; }
    if (f > 0 && p[j] == '.') {
        j++;
        for (; j < MAX_FIELD && p[j] >= '0' && p[j] <= '9'; j++) {
            if (nfr < 3) frac = frac * 10 + (p[j] - '0'); else if (p[j] != '0') bad = true;
            nfr++;
        }
    }
    if (p[j] != '\t' && p[j] != term2) bad = true;                      // anything else inside the field (or a field too long)
    if (nd + nfr == 0 && j > 0) bad = true;                              // "-" or "." alone; an empty field is atof("") = 0
    for (; nfr < 3; nfr++) frac *= 10;
    const long long key = (neg ? -1 : 1) * (ip * 1000 + frac);
    if (bad) atomicMin(err, (unsigned long long)(l + 1) << 8 | 3);
    else if (f == 0 ? (ip > 2147483647ll || neg) : (key > 2147483647ll || key < -2147483647ll)) atomicMin(err, (unsigned long long)(l + 1) << 8 | 4);
    else if (f == 0) dp_names[row] = (int32_t)ip;                        // ref :146 atoi
    else if (f < (uint32_t)ncols) keys[(int64_t)(f - 1) * stride + (int64_t)row] = (int32_t)key;
    return p[j];
}

// bit i set iff byte i of x is an ASCII digit
__device__ __forceinline__ uint32_t digit_mask4(uint32_t x)
{
    const uint32_t t = x ^ 0x30303030u;                                   // '0'..'9' -> 0..9
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x76767676u) | t) & 0x80808080u;
    return __builtin_amdgcn_udot4(z, 0x08040201u, 0u, false) >> 7;
}
__device__ __forceinline__ uint32_t digit_mask16(const uint4& v) { return digit_mask4(v.x) | (digit_mask4(v.y) << 4) | (digit_mask4(v.z) << 8) | (digit_mask4(v.w) << 12); }
__device__ __forceinline__ uint32_t eq_mask16r(const uint4& v, uint32_t pat) { return eq_mask4(v.x, pat) | (eq_mask4(v.y, pat) << 4) | (eq_mask4(v.z, pat) << 8) | (eq_mask4(v.w, pat) << 12); }
__device__ __forceinline__ uint32_t sel4(int q, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return q == 0 ? a : q == 1 ? b : q == 2 ? c : d; }
// four ASCII-masked digits (first = most significant in the lowest byte; bytes hold 0..9) -> their number
__device__ __forceinline__ uint32_t four_digits(uint32_t w)
{
    const uint32_t t = w * 10u + (w >> 8);                                // byte 0: d0 d1, byte 2: d2 d3 (no carries: 99 at most)
    return (t & 0xFFu) * 100u + ((t >> 16) & 0xFFu);
}

// One wave per block of LBLK = 1008 bytes of the text (see the head of the file): lanes 0..62 own 16 bytes each, lane 63
// holds the 16 bytes behind the block (the tail of a field that starts in lane 62).  kind[l] == 1: the lines this kernel
// converts.  A field of the usual shape -- [-] up to 8 digits [. up to 3 digits], at most 12 bytes, ended by a tab or a
// newline within the 32 bytes the lane sees -- is converted without a loop over its bytes: its parts are found from the
// lane's character-class masks, the digits taken from a byte-aligned window and multiplied out in parallel.  Everything
// else (and everything that is an error) goes through convert_field, the byte-by-byte form.
__global__ __launch_bounds__(256) void lrn_blocks_kernel(const uint8_t* __restrict__ text, uint64_t len, const uint64_t* __restrict__ blk_prefix,
                                                        const uint32_t* __restrict__ blk_nl, const uint32_t* __restrict__ blk_tail, uint64_t nblk,
                                                        const uint8_t* __restrict__ kind, const uint64_t* __restrict__ row_of, uint64_t n_lines,
                                                        int32_t ncols, int64_t stride, int32_t* __restrict__ dp_names, int32_t* __restrict__ keys,
                                                        unsigned long long* __restrict__ err)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[4][1024 + 48];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint8_t* lb = buf[wid];
    // Which wave takes which block decides how the key table is written: a column of the table holds consecutive rows, a
    // 128-byte line of it 32 rows = about 32 lines of text, and workgroup i runs on XCD i mod 8, each XCD with an L2 of its own.
    // Blocks dealt out one by one would have every line of the table written in pieces from all eight L2s (each piece a
    // partial write-back).  So the text goes to the XCDs in runs of 64 blocks (~56 rows): the k-th block of XCD x
    // is block ((k / 64) * 8 + x) * 64 + k % 64, and the waves of the XCD take k = slot, slot + waves, ...
    const uint64_t xcd = blockIdx.x & 7, slot = (uint64_t)(blockIdx.x >> 3) * 4 + wid, slots = (uint64_t)(gridDim.x >> 3) * 4;
    // 16 bytes of the text at `at`; past its end: newlines
    auto fetch = [&](uint64_t at) -> uint4 {
        uint4 v;
        if (at + 16 <= len) memcpy(&v, text + at, 16);
        else {
            uint8_t t[16];
            for (int i = 0; i < 16; i++) t[i] = (at + i < len) ? text[at + i] : (uint8_t)'\n';
            memcpy(&v, t, 16);
        }
        return v;
    };
    auto block_of = [&](uint64_t k) { return ((k >> 6) * 8 + xcd) * 64 + (k & 63); };
    // the text of the wave's next block is on its way while this one is converted
    uint4 nxt_mine = fetch(block_of(slot) * LBLK + (uint64_t)lane * 16), nxt_halo = make_uint4(0, 0, 0, 0);
    if (lane < 2) nxt_halo = fetch(block_of(slot) * LBLK + 1024 + (uint64_t)lane * 16);
    for (uint64_t k = slot;; k += slots) {
        const uint64_t b = block_of(k);
        if ((k >> 6) * 8 * 64 >= nblk) break;                            // past the last run of every XCD
        const uint64_t base = b * LBLK;
        const uint4 mine = nxt_mine, halo = nxt_halo;
        {
            const uint64_t nb = block_of(k + slots) * LBLK;
            nxt_mine = fetch(nb + (uint64_t)lane * 16);
            if (lane < 2) nxt_halo = fetch(nb + 1024 + (uint64_t)lane * 16);
        }
        if (b >= nblk || base >= len) continue;
        // the block, lane 63's 16 bytes and 32 bytes of halo into LDS (the byte-by-byte form reads there)
        *reinterpret_cast<uint4*>(lb + lane * 16) = mine;
        if (lane < 2) *reinterpret_cast<uint4*>(lb + 1024 + lane * 16) = halo;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t A = base + (uint64_t)lane * 16;
        // character classes of this lane's 16 bytes (bit i = byte i); the 16 bytes behind them are the next lane's
        const uint32_t tab16 = eq_mask16r(mine, 0x09090909u), nl16 = eq_mask16r(mine, 0x0A0A0A0Au);
        const uint32_t dig16 = digit_mask16(mine), dot16 = eq_mask16r(mine, 0x2E2E2E2Eu), min16 = eq_mask16r(mine, 0x2D2D2D2Du);
        const uint32_t sep32 = (tab16 | nl16) | (lane_above(tab16 | nl16) << 16);
        const uint32_t nl32 = nl16 | (lane_above(nl16) << 16);
        const uint32_t dig32 = dig16 | (lane_above(dig16) << 16);
        const uint32_t dot32 = dot16 | (lane_above(dot16) << 16);
        const uint32_t W0 = mine.x, W1 = mine.y, W2 = mine.z, W3 = mine.w;
        const uint32_t W4 = lane_above(mine.x), W5 = lane_above(mine.y), W6 = lane_above(mine.z);
        const int live = (lane == 63 || A >= len) ? 0 : (len - A >= 16 ? 16 : (int)(len - A));
        const uint32_t lm = (1u << live) - 1u;
        const uint32_t tabs = tab16 & lm, nls = nl16 & lm;
        // line of this lane's first byte
        const uint32_t cn = __popc(nls);
        const uint64_t line0 = blk_prefix[b] + wave_incl_scan_u32(cn, lane) - cn;
        // field of this lane's first byte: tabs since the last newline -- a segmented exclusive scan over the lanes ...
        uint32_t seg_t = nls ? __popc(tabs >> (32 - __builtin_clz(nls))) : __popc(tabs);     // this lane's tabs after its last newline
        uint32_t seg_f = nls ? 1u : 0u;                                                      // ... which restart the count
        // (the same six DPP steps as the prefix sum; a lane without a source receives t = 0, f = 0: the identity)
#define SEG_STEP(CTRL, ROWS, BOUND) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)seg_t, CTRL, ROWS, 0xF, BOUND), \
                                      f = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)seg_f, CTRL, ROWS, 0xF, BOUND); \
                                      seg_t += seg_f ? 0u : t; seg_f |= f; }
        SEG_STEP(0x111, 0xF, true) SEG_STEP(0x112, 0xF, true) SEG_STEP(0x114, 0xF, true) SEG_STEP(0x118, 0xF, true)
        SEG_STEP(0x142, 0xA, false) SEG_STEP(0x143, 0xC, false)
#undef SEG_STEP
        uint32_t in_t = lane_below(seg_t), in_f = lane_below(seg_f);
        if (lane == 0) { in_t = 0; in_f = 0; }
        // ... plus, while no newline of this block lies before the lane, the tabs between the start of the line and the block
        uint32_t carry = 0;
        for (uint64_t j = b; j-- > 0;) { carry += blk_tail[j]; if (blk_nl[j]) break; }      // wave-uniform; a line spans at most 9 blocks
        const uint32_t field0 = in_t + (in_f ? 0u : carry);
        // a field starts at the first byte of the text and after every tab or newline
        uint32_t prev_sep = (lane_below(tab16 | nl16) >> 15) & 1u;                     // (all lanes take part in the shuffle)
        if (lane == 0) prev_sep = (A == 0) ? 1u : (uint32_t)(text[A - 1] == '\t' || text[A - 1] == '\n');
        uint32_t starts = (((tabs | nls) << 1) | prev_sep) & lm;
        uint64_t cur_line = ~0ull, row = 0;
        uint8_t cur_kind = 0;
        while (starts) {
            const int i = __ffs(starts) - 1;
            starts &= starts - 1;
            const uint32_t below = (1u << i) - 1u;
            const uint32_t nl_below = nls & below;
            const uint64_t l = line0 + __popc(nl_below);
            const uint32_t f = nl_below ? (uint32_t)__popc((tabs & below) >> (32 - __builtin_clz(nl_below))) : field0 + __popc(tabs & below);
            if (l != cur_line) { cur_line = l; cur_kind = (l < n_lines) ? kind[l] : 0; row = (cur_kind == 1) ? row_of[l] : 0; }
            if (cur_kind != 1) continue;
            // the parts of the field from the masks: length L up to its separator, '-', the point, the digits
            const uint32_t sepw = sep32 >> i;
            const int L = sepw ? __ffs(sepw) - 1 : 32;
            const uint32_t Lm = (L >= 32) ? 0xFFFFFFFFu : (1u << L) - 1u;
            const uint32_t neg = (min16 >> i) & 1u;
            const uint32_t dotin = (dot32 >> i) & Lm;
            const int dp = dotin ? __ffs(dotin) - 1 : L;
            const int nint = dp - (int)neg, nfr = dotin ? L - dp - 1 : 0;
            const uint32_t need = Lm & ~neg & ~(dotin ? (1u << dp) : 0u);              // where digits must be
            const bool usual = L <= 12 && ((dig32 >> i) & need) == need && (dotin & (dotin - 1u)) == 0 && nint >= 0 && nint <= 8 && nfr <= 3 &&
                               (need != 0 || L == 0) && !(f == 0 && (dotin | neg));
            bool done = false;
            bool ends_line = false;
            if (usual) {
                // the 12 bytes from the field's first one: a0 a1 a2
                const int q = i >> 2, sh = i & 3;
                const uint32_t x0 = sel4(q, W0, W1, W2, W3), x1 = sel4(q, W1, W2, W3, W4), x2 = sel4(q, W2, W3, W4, W5), x3 = sel4(q, W3, W4, W5, W6);
                const uint32_t a0 = __builtin_amdgcn_alignbyte(x1, x0, sh), a1 = __builtin_amdgcn_alignbyte(x2, x1, sh), a2 = __builtin_amdgcn_alignbyte(x3, x2, sh);
                // the 8 bytes that end at the point (or at the end of the field): its integer digits, right-aligned; bytes before the
                // first digit read as 0.  Z = 8 zero bytes, then a0 a1 a2: bytes [dp, dp + 8) of Z
                const int qd = dp >> 2, shd = dp & 3;
                const uint32_t y0 = sel4(qd, 0u, 0u, a0, a1), y1 = sel4(qd, 0u, a0, a1, a2), y2 = sel4(qd, a0, a1, a2, 0u);
                uint32_t lo = __builtin_amdgcn_alignbyte(y1, y0, shd), hi = __builtin_amdgcn_alignbyte(y2, y1, shd);
                const int drop = 8 - nint;                                            // leading bytes that are not digits of this field
                const uint32_t mlo = drop >= 4 ? 0u : (0xFFFFFFFFu << (8 * drop)), mhi = drop >= 8 ? 0u : (drop > 4 ? (0xFFFFFFFFu << (8 * (drop - 4))) : 0xFFFFFFFFu);
                lo &= 0x0F0F0F0Fu & mlo; hi &= 0x0F0F0F0Fu & mhi;
                const uint32_t ip = four_digits(lo) * 10000u + four_digits(hi);
                // the 3 bytes behind the point: bytes [dp + 9, dp + 12) of Z (zero bytes behind a2)
                const int of = dp + 9, qf = (of >> 2) - 2, shf = of & 3;
                const uint32_t g0 = sel4(qf, a0, a1, a2, 0u), g1 = sel4(qf, a1, a2, 0u, 0u);
                const uint32_t fr = __builtin_amdgcn_alignbyte(g1, g0, shf) & 0x000F0F0Fu & ((1u << (8 * nfr)) - 1u);
                const uint32_t frac = (fr & 0xFFu) * 100u + ((fr >> 8) & 0xFFu) * 10u + ((fr >> 16) & 0xFFu);
                if (f == 0) { dp_names[row] = (int32_t)ip; done = true; }
                else if (ip < 2147483u) {
                    const int32_t key = (int32_t)(ip * 1000u + frac);
                    if (f < (uint32_t)ncols) keys[(int64_t)(f - 1) * stride + (int64_t)row] = neg ? -key : key;
                    done = true;
                }
                ends_line = (nl32 >> (i + L)) & 1u;
            }
            if (!done) ends_line = convert_field(lb + lane * 16 + i, '\n', f, l, row, ncols, stride, dp_names, keys, err) == '\n';
            if (ends_line && f + 1 != (uint32_t)ncols) atomicMin(err, (unsigned long long)(l + 1) << 8 | 2);   // ref :135-140
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void lrn_rows_kernel(const uint8_t* __restrict__ text, const LineInfo* __restrict__ info,
                                                      const uint8_t* __restrict__ kind, const uint64_t* __restrict__ row_of,
                                                      uint64_t n_lines, int32_t ncols, int64_t stride,
                                                      int32_t* __restrict__ dp_names, int32_t* __restrict__ keys,
                                                      unsigned long long* __restrict__ err)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[4][RPIECE + 48];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint8_t* lb = buf[wid];
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t l = (uint64_t)blockIdx.x * 4 + wid; l < n_lines; l += nwaves) {
        if (kind[l] != 2) continue;                                      // wave-uniform
        const uint64_t row = row_of[l];
        const uint64_t b = info[l].begin, e = info[l].end;
        uint32_t carry = 0;                                              // tabs of the line before this piece
        for (uint64_t base = b; base < e; base += RPIECE) {
            // piece [base, base + 1024) + 32 bytes of halo into LDS; bytes past the end of the line read as tab
            for (int o = lane * 16; o < RPIECE + 32; o += 1024) {
                const uint64_t at = base + o;
                uint4 v;
                if (at + 16 <= e) memcpy(&v, text + at, 16);
                else {
                    uint8_t t[16];
                    for (int i = 0; i < 16; i++) t[i] = (at + i < e) ? text[at + i] : (uint8_t)'\t';
                    memcpy(&v, t, 16);
                }
                *reinterpret_cast<uint4*>(lb + o) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 mine = *reinterpret_cast<const uint4*>(lb + lane * 16);
            uint32_t tabs = eq_mask4(mine.x, 0x09090909u) | (eq_mask4(mine.y, 0x09090909u) << 4) | (eq_mask4(mine.z, 0x09090909u) << 8) |
                            (eq_mask4(mine.w, 0x09090909u) << 12);
            const uint64_t A = base + (uint64_t)lane * 16;
            const int live = (A >= e) ? 0 : (e - A >= 16 ? 16 : (int)(e - A));     // bytes of the line among this lane's 16
            tabs &= (1u << live) - 1u;
            const uint32_t c = __popc(tabs);
            const uint32_t incl = wave_incl_scan_u32(c, lane);
            uint32_t field = carry + incl - c;                           // field number of this lane's first byte
            // a field starts at the first byte of the line and after every tab; the byte before this lane's first one
            // is the last byte of the previous lane (or of the previous piece)
            const uint32_t prev_tab = (A == b) ? 1u : (A > b && A <= e ? (uint32_t)(text[A - 1] == '\t') : 0u);
            uint32_t starts = ((tabs << 1) | prev_tab) & ((1u << live) - 1u);
            // a tab that is the very last byte of the line opens one more (empty) field at e: handled by the lane that owns e - 1
            while (starts) {
                const int i = __ffs(starts) - 1;
                starts &= starts - 1;
                const uint32_t f = field + __popc(tabs & ((1u << i) - 1u));
                convert_field(lb + lane * 16 + i, '\t', f, l, row, ncols, stride, dp_names, keys, err);
            }
            carry += __shfl(incl, 63, 64);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0 && carry + 1 != (uint32_t)ncols) atomicMin(err, (unsigned long long)(l + 1) << 8 | 2);   // ref :135-140
    }
}

// ---- writer side (ref src/abawaca-build.cpp:595-607): rows "<key>\t%.3lf\t%.3lf...\n" ------------------------------
// A value is printed as "%.3lf" of int(1000.0*x)/1000.0 (ref :603); with k = int(1000.0*x) that text is
// [-]<|k| / 1000>.<|k| % 1000, three digits> (k/1000.0 is the double nearest to the 3-decimal number, which "%.3lf"
// prints back).  One wave per row, a lane takes every 64th column: field lengths -> wave prefix sum -> characters
// into an LDS copy of the row -> coalesced bytes out.
__device__ __forceinline__ int dec_digits(uint32_t v)
{
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7 :
           v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
__device__ __forceinline__ int field_len(int32_t k)          // "\t" + [-] + integer digits + "." + 3 decimals
{
    const uint32_t a = k < 0 ? 0u - (uint32_t)k : (uint32_t)k;
    return 1 + (k < 0) + dec_digits(a / 1000u) + 4;
}
__device__ __forceinline__ int key_len(unsigned long long key)
{
    int n = 1;
    while (key >= 10ull) { key /= 10ull; n++; }
    return n;
}

constexpr int EMIT_COLS_PER_LANE = 16;                        // up to 1024 columns
constexpr int EMIT_ROW_MAX = 16384;                           // bytes of one row staged in LDS

// ---- cluster writer (ref src/ClusterWriter.cpp:92-98): rows "<dp>\t%lf\t%lf...\n" ---------------------------------
// "%lf" prints six decimals of the EXACT binary value, rounded to nearest, ties to even (the C library's conversion,
// which is what the reference's fprintf does).  |v| = m * 2^e exactly with m < 2^53, so v * 10^6 = (m * 10^6) * 2^e with
// m * 10^6 < 2^73: a 128-bit integer, shifted; the bits shifted out decide the rounding.  Magnitudes whose millionths pass
// 2^63 (|v| >= 9.2e12) and inf / nan are reported, not printed.
struct Fix6 { uint64_t q; uint32_t neg, bad; };
__device__ __forceinline__ Fix6 fix6_of(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    Fix6 r;
    r.q = 0; r.neg = (uint32_t)(b >> 63); r.bad = 0;
    const uint32_t ex = (uint32_t)(b >> 52) & 0x7FFu;
    uint64_t m = b & ((1ull << 52) - 1ull);
    if (ex == 0x7FFu) { r.bad = 1; return r; }
    int s = 1074;                                            // |v| = m * 2^-s
    if (ex != 0) { m |= 1ull << 52; s = 1075 - (int)ex; }
    if (m == 0) return r;
    const uint64_t lo = m * 1000000ull, hi = __umul64hi(m, 1000000ull);
    if (s <= 0) {                                            // an integer: no rounding
        const int k = -s;
        if (k >= 63 || hi != 0 || (lo >> (63 - k)) != 0) { r.bad = 1; return r; }
        r.q = lo << k;
        return r;
    }
    if (s >= 128) return r;                                  // below 2^73 / 2^128: rounds to zero
    uint64_t q, q_hi, rem_hi, rem_lo, half_hi, half_lo;
    if (s < 64) {
        q = (lo >> s) | (hi << (64 - s)); q_hi = hi >> s;
        rem_hi = 0; rem_lo = lo & ((1ull << s) - 1ull); half_hi = 0; half_lo = 1ull << (s - 1);
    } else if (s == 64) {
        q = hi; q_hi = 0; rem_hi = 0; rem_lo = lo; half_hi = 0; half_lo = 1ull << 63;
    } else {
        const int t = s - 64;
        q = hi >> t; q_hi = 0;
        rem_hi = hi & ((1ull << t) - 1ull); rem_lo = lo; half_hi = 1ull << (t - 1); half_lo = 0;
    }
    if (q_hi != 0 || (q >> 63) != 0) { r.bad = 1; return r; }
    const bool above = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
    const bool tie = rem_hi == half_hi && rem_lo == half_lo;
    if (above || (tie && (q & 1ull))) q++;
    r.q = q;
    return r;
}
__device__ __forceinline__ int dec_digits64(uint64_t v)
{
    int n = 1;
    while (v >= 10ull) { v /= 10ull; n++; }
    return n;
}
__device__ __forceinline__ uint32_t field6_len(const Fix6& f) { return 1u + f.neg + (uint32_t)dec_digits64(f.q / 1000000ull) + 7u; }

constexpr int EMIT_LDS_MAX = 80 * 1024;                        // a workgroup's rows in LDS: two workgroups per CU
constexpr int EMIT6_FIELD_MAX = 23;                           // "\t-" + 13 digits + "." + 6 decimals, rounded up

// ---- a field of the cluster writer without loops (|value| < 1e8: eight integer digits at most) ----------------------
__device__ __forceinline__ uint32_t digits4(uint32_t y)      // y < 10000 -> its four digits as bytes, byte 0 = the thousands
{
    const uint32_t a = (y * 5243u) >> 19, b = y - a * 100u;                 // y / 100 (exact below 43 699), y % 100
    const uint32_t t = a | (b << 16);
    const uint32_t tens = ((t * 103u) >> 10) & 0x000F000Fu;                 // both halves / 10 (exact below 179)
    return tens | ((t - tens * 10u) << 8);
}
struct Field6 { uint64_t head, tail; uint32_t head_hi, len, tail_at; };
// head (+ head_hi): "\t", "-" if negative, the integer digits -- 12 bytes of which the first tail_at count; tail: ".",
// six decimals and the "\t" that starts the next field (the row's last field: its place takes the "\n" afterwards).
__device__ __forceinline__ Field6 field6_of(uint32_t ip, uint32_t fr, uint32_t neg)
{
    const uint32_t hi4 = ip / 10000u, lo4 = ip - hi4 * 10000u;
    const uint64_t dig = (uint64_t)digits4(hi4) | (uint64_t)digits4(lo4) << 32;           // byte 0 = the highest of eight
    const uint32_t lead = (uint32_t)__builtin_ctzll(dig | 1ull << 56) >> 3;               // leading zeros, 7 at most
    const uint64_t body = (dig + 0x3030303030303030ull) >> (8u * lead);
    const uint32_t nd = 8u - lead, p = 1u + neg;
    const uint32_t fh = fr / 10000u, fl = fr - fh * 10000u, ft = (fh * 103u) >> 10;
    Field6 r;
    r.head = (uint64_t)(neg ? 0x2D09u : 0x09u) | body << (8u * p);
    r.head_hi = (uint32_t)(body >> (64u - 8u * p));
    r.tail = 0x2Eull | (uint64_t)(0x30u + ft) << 8 | (uint64_t)(0x30u + fh - ft * 10u) << 16 | (uint64_t)(digits4(fl) + 0x30303030u) << 24 | 0x09ull << 56;
    r.tail_at = p + nd;
    r.len = p + nd + 7u;
    return r;
}
// LDS takes a wave's stores in program order, so a byte keeps what the LAST instruction that touches it writes.  The head's
// 12 bytes hold rubbish behind the digits: in the lane's own field (the tail, stored later, covers it) or, for a
// 9-byte field, in the first three bytes of the next lane's field -- which that lane's `head` store, one instruction
// after every lane's `head_hi`, covers.  The tail holds no rubbish (its eighth byte is the next field's tab).
__device__ __forceinline__ void field6_store(uint8_t* p, const Field6& f)
{
    __builtin_memcpy(p + 8, &f.head_hi, 4);
    asm volatile("" ::: "memory");                           // three instructions, in this order
    __builtin_memcpy(p, &f.head, 8);
    asm volatile("" ::: "memory");
    __builtin_memcpy(p + f.tail_at, &f.tail, 8);
}

// One launch per call (round 3; before: a length pass, a three-launch scan and a print pass, the table gathered twice).
// A workgroup takes the next four rows by ticket, prints each into LDS -- which also gives its length --, publishes the
// four lengths' sum and finds the bytes in front of it by a decoupled look-back over the workgroups with lower tickets
// (state word: bit 62 = "my sum", bit 63 = "everything up to and including me", value below; a wave reads 64 predecessors
// at a time).  Every workgroup with a lower ticket has started and publishes its sum without waiting for anyone, so every
// wait ends.  The text goes out in aligned 16-byte stores (the row's place is known only after it is printed: the LDS
// words are shifted by the wave-uniform byte offset with v_alignbyte).  TEXT = false: lengths and offsets only.
constexpr uint64_t LB_SUM = 1ull << 62, LB_ALL = 1ull << 63, LB_VAL = (1ull << 62) - 1ull;
enum { CTL_TICKET = 0, CTL_ERR = 1, CTL_TOTAL = 2, CTL_WORDS = 4 };      // CTL_ERR: the maximum of ~(row + 1 << 8 | code), 0 = none

__device__ __forceinline__ void emit6_report(unsigned long long* ctl, int64_t row, unsigned code)
{
    atomicMax(ctl + CTL_ERR, ~((unsigned long long)(row + 1) << 8 | code));
}

// What follows a row's printing, for both writers: the workgroup's rows find their place in the text (look-back over the
// workgroups with lower tickets), the last ticket stores the total, the rows leave in aligned 16-byte stores.
template <int WPB, bool TEXT>
__device__ __forceinline__ void emit_place_and_store(uint64_t bid, int64_t row, int64_t n_rows, uint32_t total, const uint8_t* lb,
                                                     unsigned long long* __restrict__ ctl, unsigned long long* __restrict__ state,
                                                     uint8_t* __restrict__ text, uint64_t text_cap, uint64_t* __restrict__ row_off)
{
    __shared__ unsigned long long s_excl;
    __shared__ uint32_t s_len[WPB];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) s_len[wid] = total;
    __syncthreads();
    if (wid == 0) {
        uint64_t sum = lane < WPB ? (uint64_t)s_len[lane] : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        uint64_t excl = 0;
        if (bid > 0) {
            if (lane == 0) __hip_atomic_store(state + bid, LB_SUM | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t top = (int64_t)bid - 1;                  // the nearest workgroup not yet added
            for (;;) {
                const int64_t idx = top - lane;
                const uint64_t v = idx >= 0 ? __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_ALL;
                const uint64_t have_all = __ballot((v & LB_ALL) != 0), nothing = __ballot((v >> 62) == 0);
                const int stop = have_all ? __builtin_ctzll(have_all) : 64;        // the nearest that knows everything below it
                const uint64_t below = stop >= 64 ? ~0ull : (2ull << stop) - 1ull;
                if (nothing & below) { __builtin_amdgcn_s_sleep(2); continue; }    // one of them has not published yet
                uint64_t t = lane <= stop ? (v & LB_VAL) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                excl += t;
                if (stop < 64) break;
                top -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(state + bid, LB_ALL | (excl + sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
            if ((int64_t)bid == (n_rows + WPB - 1) / WPB - 1) { ctl[CTL_TOTAL] = excl + sum; if (row_off) row_off[n_rows] = excl + sum; }
        }
    }
    __syncthreads();
    if (row >= n_rows) return;
    uint64_t o = s_excl;
    for (int w = 0; w < wid; w++) o += s_len[w];
    if (row_off && lane == 0) row_off[row] = o;
    if (!TEXT || o + total > text_cap) return;
    // head up to the first 16-byte boundary of the text, aligned 16-byte pieces, tail
    const uint32_t h = min(total, (uint32_t)__builtin_amdgcn_readfirstlane((int)((16u - (uint32_t)(o & 15ull)) & 15u)));
    const uint32_t nvec = (total - h) >> 4, r = h & 3u;
    if ((uint32_t)lane < h) text[o + lane] = lb[lane];
    const uint32_t* lw = reinterpret_cast<const uint32_t*>(lb) + (h >> 2);
    uint4* tv = reinterpret_cast<uint4*>(text + o + h);
    for (uint32_t k = lane; k < nvec; k += 64) {
        const uint32_t* q = lw + 4 * k;
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
        uint4 out;
        out.x = __builtin_amdgcn_alignbyte(w1, w0, r); out.y = __builtin_amdgcn_alignbyte(w2, w1, r);
        out.z = __builtin_amdgcn_alignbyte(w3, w2, r); out.w = __builtin_amdgcn_alignbyte(w4, w3, r);
        tv[k] = out;
    }
    const uint32_t done = h + (nvec << 4);
    if (done + (uint32_t)lane < total) text[o + done + lane] = lb[done + lane];
}

template <int WPB>
__device__ __forceinline__ uint64_t emit_ticket(unsigned long long* __restrict__ ctl)
{
    __shared__ unsigned long long s_bid;
    if (threadIdx.x == 0) s_bid = atomicAdd(ctl + CTL_TICKET, 1ull);
    __syncthreads();
    return s_bid;
}

// The build writer's rows in the same one launch (round 3; before: lengths, a three-launch scan, the print).  A field is
// "\t[-]<|k| / 1000>" -- 12 bytes of which the first 2 .. 9 count -- and ".ddd"; the head leaves in three 4-byte LDS stores
// from the highest down, the tail after every lane's head: a field has at least six bytes, so the rubbish of a head lies
// in the lane's own tail or in the NEXT lane's field, in a word that lane stores in a later instruction.
template <bool TEXT, int WPB>
__global__ __launch_bounds__(64 * WPB) void lrn_emit3_kernel(const int32_t* __restrict__ milli, int64_t n_rows, int32_t n_cols, uint64_t first_key,
                                                            uint32_t row_stride, unsigned long long* __restrict__ ctl, unsigned long long* __restrict__ state,
                                                            uint8_t* __restrict__ text, uint64_t text_cap)
{
    extern __shared__ __align__(16) uint8_t emit3_lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // persistent workgroups (as many as the chip holds at once) take tile after tile by ticket: 25 000 workgroups of eight
    // rows each were started slower than they finish -- on average fewer than six waves per CU were in flight (PMC)
    const uint64_t nblk = ((uint64_t)n_rows + WPB - 1) / WPB;
    for (;;) {
        const uint64_t bid = emit_ticket<WPB>(ctl);
        if (bid >= nblk) break;                               // uniform: every thread of the workgroup read the same ticket
        const int64_t row = (int64_t)bid * WPB + wid;
        uint8_t* lb = emit3_lds + (size_t)wid * row_stride;
        uint32_t total = 0;
        if (row < n_rows) {
            const int32_t* r = milli + row * n_cols;
            const uint64_t key = first_key + (uint64_t)row;
            const int klen = key_len(key);
            if (TEXT && lane == 0) { unsigned long long k = key; for (int i = klen - 1; i >= 0; i--) { lb[i] = (uint8_t)('0' + k % 10ull); k /= 10ull; } }
            uint32_t carry = (uint32_t)klen;
            for (int c0 = 0; c0 < n_cols; c0 += 64) {            // columns c0 .. c0+63, one per lane, in order
                const int c = c0 + lane;
                const int32_t k = (c < n_cols) ? r[c] : 0;
                const uint32_t neg = k < 0, a = neg ? 0u - (uint32_t)k : (uint32_t)k;
                const uint32_t q = a / 1000u, f = a - q * 1000u;                         // q < 2 147 484: seven digits
                const uint32_t hi4 = q / 10000u, lo4 = q - hi4 * 10000u;
                const uint64_t dig = (uint64_t)digits4(hi4) | (uint64_t)digits4(lo4) << 32;
                const uint32_t lead = (uint32_t)__builtin_ctzll(dig | 1ull << 56) >> 3;
                const uint64_t body = (dig + 0x3030303030303030ull) >> (8u * lead);
                const uint32_t p = 1u + neg, at = p + 8u - lead;
                const uint32_t len = (c < n_cols) ? at + 4u : 0u;
                if (TEXT) {
                    const uint32_t incl = wave_incl_scan_dpp(len);
                    if (c < n_cols) {
                        uint8_t* dst = lb + carry + incl - len;
                        const uint64_t head = (uint64_t)(neg ? 0x2D09u : 0x09u) | body << (8u * p);
                        const uint32_t h2 = (uint32_t)(body >> (64u - 8u * p)), h1 = (uint32_t)(head >> 32), h0 = (uint32_t)head;
                        const uint32_t tail = (digits4(f) + 0x30303000u) | 0x2Eu;
                        __builtin_memcpy(dst + 8, &h2, 4);
                        asm volatile("" ::: "memory");           // four instructions, in this order: merged into wider stores, two
                        __builtin_memcpy(dst + 4, &h1, 4);       // lanes would write the same byte in one instruction
                        asm volatile("" ::: "memory");
                        __builtin_memcpy(dst, &h0, 4);
                        asm volatile("" ::: "memory");
                        __builtin_memcpy(dst + at, &tail, 4);
                    }
                    carry += __shfl(incl, 63, 64);
                } else {
                    carry += wave_sum_dpp(len);
                }
            }
            if (TEXT && lane == 0) lb[carry] = '\n';
            total = carry + 1u;
        }
        emit_place_and_store<WPB, TEXT>(bid, row, n_rows, total, lb, ctl, state, text, text_cap, nullptr);
        if ((uint64_t)gridDim.x >= nblk) break;               // a workgroup per tile: no second ticket to learn that
    }

}

// ROWMAJ: the table is the resident copy, [ld][n_cols] (a datapoint's values side by side: a row is one coalesced read
// instead of n_cols lines of 128 bytes for 8 bytes each -- what bounds the other form).
template <bool TEXT, int WPB, bool ROWMAJ>                    // WPB rows (waves) per workgroup
__global__ __launch_bounds__(64 * WPB) void lrn_emit6_kernel(const double* __restrict__ values, int64_t ld, int32_t n_cols,
                                                       const int64_t* __restrict__ dps, int64_t n_rows, uint32_t row_stride,
                                                       unsigned long long* __restrict__ ctl, unsigned long long* __restrict__ state,
                                                       uint8_t* __restrict__ text, uint64_t text_cap, uint64_t* __restrict__ row_off)
{
    extern __shared__ __align__(16) uint8_t emit6_lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // persistent workgroups (as many as the chip holds at once) take tile after tile by ticket: 25 000 workgroups of eight
    // rows each were started slower than they finish -- on average fewer than six waves per CU were in flight (PMC)
    const uint64_t nblk = ((uint64_t)n_rows + WPB - 1) / WPB;
    for (;;) {
        const uint64_t bid = emit_ticket<WPB>(ctl);
        if (bid >= nblk) break;                               // uniform: every thread of the workgroup read the same ticket
        const int64_t row = (int64_t)bid * WPB + wid;
        uint8_t* lb = emit6_lds + (size_t)wid * row_stride;
        uint32_t total = 0;
        if (row < n_rows) {
            int64_t dp = dps[row];
            if (dp < 1 || dp > ld) { if (lane == 0) emit6_report(ctl, row, 1); dp = 1; }
            const int klen = key_len((unsigned long long)dp);
            if (TEXT && lane == 0) { unsigned long long k = (unsigned long long)dp; for (int i = klen - 1; i >= 0; i--) { lb[i] = (uint8_t)('0' + k % 10ull); k /= 10ull; } }
            uint32_t carry = (uint32_t)klen;
            for (int cb = 0; cb < n_cols; cb += 256) {           // four gathers in flight per lane
                double vv[4];
    #pragma unroll
                for (int u = 0; u < 4; u++) { const int c = cb + 64 * u + lane; vv[u] = c < n_cols ? (ROWMAJ ? values[(dp - 1) * (int64_t)n_cols + c] : values[(int64_t)c * ld + dp - 1]) : 0.0; }
    #pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int c0 = cb + 64 * u, c = c0 + lane;
                    if (c0 >= n_cols) break;
                    const Fix6 f = fix6_of(vv[u]);
                    if (f.bad) emit6_report(ctl, row, 2);
                    uint64_t ip = f.q / 1000000ull;
                    const uint32_t fr = (uint32_t)(f.q - ip * 1000000ull);
                    if (__ballot(ip >= 100000000ull) == 0) {
                        // the usual wave: every |value| below 1e8.  No loop over the digits and no byte stores: the field
                        // is put together in registers and leaves in four stores -- see field6_store for their order
                        const Field6 fd = field6_of((uint32_t)ip, fr, f.neg);
                        const uint32_t len = (c < n_cols) ? fd.len : 0u;
                        if (TEXT) {
                            const uint32_t incl = wave_incl_scan_dpp(len);
                            if (c < n_cols) field6_store(lb + carry + incl - len, fd);
                            carry += __shfl(incl, 63, 64);
                        } else {
                            carry += wave_sum_dpp(len);
                        }
                        continue;
                    }
                    const int nd = ip < 0x100000000ull ? dec_digits((uint32_t)ip) : dec_digits64(ip);
                    const uint32_t len = (c < n_cols) ? 1u + f.neg + (uint32_t)nd + 7u : 0u;
                    if (TEXT) {
                        const uint32_t incl = wave_incl_scan_dpp(len);
                        if (c < n_cols) {
                            uint8_t* p = lb + carry + incl - len;
                            uint32_t r6 = fr;
                            *p++ = '\t';
                            if (f.neg) *p++ = '-';
                            for (int i = nd - 1; i >= 0; i--) { p[i] = (uint8_t)('0' + ip % 10ull); ip /= 10ull; }
                            p += nd;
                            *p++ = '.';
                            for (int i = 5; i >= 0; i--) { p[i] = (uint8_t)('0' + r6 % 10u); r6 /= 10u; }
                        }
                        carry += __shfl(incl, 63, 64);
                    } else {
                        carry += wave_sum_dpp(len);
                    }
                }
            }
            if (TEXT && lane == 0) lb[carry] = '\n';
            total = carry + 1u;
        }
        emit_place_and_store<WPB, TEXT>(bid, row, n_rows, total, lb, ctl, state, text, text_cap, row_off);
        if ((uint64_t)gridDim.x >= nblk) break;               // a workgroup per tile: no second ticket to learn that
    }

}

// [n_cols][ld] -> [ld][n_cols], 32 x 32 tiles through LDS (the resident copy of the cluster writer's table)
__global__ __launch_bounds__(256) void lrn_transpose_kernel(const double* __restrict__ in, int64_t ld, int32_t n_cols, double* __restrict__ out)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t dp0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < n_cols && dp0 + tx < ld) tile[j][tx] = in[(int64_t)(c0 + j) * ld + dp0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (dp0 + j < ld && c0 + tx < n_cols) out[(dp0 + j) * (int64_t)n_cols + c0 + tx] = tile[tx][j];
}

// header: the first four lines, on the host (ref :36-111)
struct LrnHeader { int64_t ndps; int32_t ncols; uint64_t names_begin, names_end; };

int parse_header(const uint8_t* text, uint64_t len, LrnHeader& h, std::string& why)
{
    uint64_t ls[5], n = 0, p = 0;
    ls[0] = 0;
    while (n < 4 && p < len) { if (text[p] == '\n') ls[++n] = p + 1; p++; }
    if (n < 4) { why = "fewer than four header lines"; return ABW_ERR_FORMAT; }
    auto num_after_percent = [&](uint64_t b, uint64_t e, long long& v) {
        if (b >= e || text[b] != '%') return false;                       // ref :48 "%c %u", c == '%'
        uint64_t q = b + 1;
        while (q < e && isspace(text[q])) q++;
        if (q >= e || !isdigit(text[q])) return false;
        v = 0;
        while (q < e && isdigit(text[q])) { v = v * 10 + (text[q] - '0'); if (v > (1ll << 40)) return false; q++; }
        return true;
    };
    long long a = 0, c = 0;
    if (!num_after_percent(ls[0], ls[1], a)) { why = "unexpected line 1"; return ABW_ERR_FORMAT; }
    if (!num_after_percent(ls[1], ls[2], c) || c < 2 || c > 100000) { why = "unexpected line 2"; return ABW_ERR_FORMAT; }
    uint32_t t = 1;
    for (uint64_t q = ls[2]; q < ls[3]; q++) t += text[q] == '\t';       // ref :74-82 the weights line has ncols fields
    if (t != (uint32_t)c) { why = "line 3 does not have as many fields as line 2 says"; return ABW_ERR_FORMAT; }
    uint64_t e = ls[4] - 1;
    while (e > ls[3] && isspace(text[e - 1])) e--;                       // ref :92-93
    t = 1;
    for (uint64_t q = ls[3]; q < e; q++) t += text[q] == '\t';
    if (t != (uint32_t)c) { why = "line 4 does not name as many dimensions as line 2 says"; return ABW_ERR_FORMAT; }   // ref :108-111
    h.ndps = a; h.ncols = (int32_t)c;
    uint64_t q = ls[3];
    while (q < e && text[q] != '\t') q++;
    h.names_begin = q + 1; h.names_end = e;                              // ref :97 the first column is the key's title
    return ABW_OK;
}

}  // namespace

extern "C" {

int abw_lrn_header(const uint8_t* text, uint64_t len, int64_t* ndps, int32_t* ndims, char* names_buf, uint64_t names_len)
{
    if (!text || !ndps || !ndims) return ABW_ERR_ARG;
    LrnHeader h;
    std::string why;
    const int rc = parse_header(text, len, h, why);
    if (rc) return rc;
    *ndps = h.ndps; *ndims = h.ncols - 1;
    if (names_buf) {                                                     // dimension names, NUL-separated
        const uint64_t need = h.names_end - h.names_begin + 1;
        if (names_len < need) return ABW_ERR_CAPACITY;
        for (uint64_t q = h.names_begin; q < h.names_end; q++) names_buf[q - h.names_begin] = text[q] == '\t' ? '\0' : (char)text[q];
        names_buf[need - 1] = '\0';
    }
    return ABW_OK;
}

int abw_lrn_parse(abw_ctx* ctx, const uint8_t* text, uint64_t len, int64_t ndps, int32_t ndims, int32_t* dp_names, int32_t* keys)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!text || !dp_names || !keys || ndps < 0 || ndims < 1) return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_parse: bad argument");
    LrnHeader h;
    std::string why;
    int rc = parse_header(text, len, h, why);
    if (rc) return abw_fail(ctx, rc, "abw_lrn_parse: %s", why.c_str());
    if (h.ndps != ndps || h.ncols - 1 != ndims)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_parse: the file declares %lld datapoints x %d dimensions, the caller expects %lld x %d",
                        (long long)h.ndps, h.ncols - 1, (long long)ndps, ndims);
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    const uint64_t nblk = len / LBLK + 1;
    // scratch[0]: text | scratch[1]: blk_nl u32 + blk_prefix u64 | scratch[2]: line_start u64 | scratch[3]: LineInfo
    // scratch[4]: is_row u32 + row_of u64 | scratch[6]: keys + dp_names + err
    if ((rc = abw_reserve(ctx, ctx->scratch[0], len + 64))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[1], 2 * (nblk * 4 + 8) + (nblk + 1) * 8))) return rc;
    uint8_t* d_text = (uint8_t*)ctx->scratch[0].p;
    uint32_t* blk_nl = (uint32_t*)ctx->scratch[1].p;
    uint32_t* blk_tail = (uint32_t*)((uint8_t*)blk_nl + ((nblk * 4 + 7) & ~(uint64_t)7));
    uint64_t* blk_prefix = (uint64_t*)((uint8_t*)blk_tail + ((nblk * 4 + 7) & ~(uint64_t)7));
    ABW_HIP(ctx, hipMemcpyAsync(d_text, text, len, hipMemcpyHostToDevice, st));
    const int grid = ctx->num_cu * 8;
    {
        ABW_PROF(ctx, "lrn_parse");                                      // kernels only; the copies are the caller's PCIe cost
        hipLaunchKernelGGL(lrn_nlcount_kernel, dim3(grid), dim3(256), 0, st, d_text, len, blk_nl, blk_tail, nblk);
        ABW_LAUNCH_CHECK(ctx);
        if ((rc = abw_device_scan_u32(ctx, blk_nl, nblk, blk_prefix))) return rc;
    }
    uint64_t n_nl = 0;
    ABW_HIP(ctx, hipMemcpyAsync(&n_nl, blk_prefix + nblk, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t n_lines = n_nl + ((len > 0 && text[len - 1] != '\n') ? 1 : 0);
    if ((rc = abw_reserve(ctx, ctx->scratch[2], (n_lines + 2) * 8))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[3], (n_lines + 1) * sizeof(LineInfo)))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[4], ((n_lines * 4 + 15) & ~(uint64_t)7) + (n_lines + 2) * 8 + n_lines + 16))) return rc;
    const size_t kb = (size_t)ndims * (size_t)std::max<int64_t>(ndps, 1) * 4, nb = (size_t)std::max<int64_t>(ndps, 1) * 4;
    if ((rc = abw_reserve(ctx, ctx->scratch[6], kb + nb + 64))) return rc;
    uint64_t* line_start = (uint64_t*)ctx->scratch[2].p;
    LineInfo* info = (LineInfo*)ctx->scratch[3].p;
    uint32_t* is_row = (uint32_t*)ctx->scratch[4].p;
    uint64_t* row_of = (uint64_t*)((uint8_t*)is_row + ((n_lines * 4 + 15) & ~(uint64_t)7));
    uint8_t* kind = (uint8_t*)(row_of + n_lines + 2);
    int32_t* d_keys = (int32_t*)ctx->scratch[6].p;
    int32_t* d_names = (int32_t*)((uint8_t*)d_keys + kb);
    unsigned long long* d_err = (unsigned long long*)((uint8_t*)d_names + ((nb + 7) & ~(size_t)7));
    const uint64_t zero = 0, endmark = len + 1;
    ABW_HIP(ctx, hipMemsetAsync(d_err, 0xFF, 8, st));
    ABW_HIP(ctx, hipMemcpyAsync(line_start, &zero, 8, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(line_start + n_lines, &endmark, 8, hipMemcpyHostToDevice, st));   // overwritten by the last newline's entry when the text ends with one
    {
        ABW_PROF(ctx, "lrn_parse");
        hipLaunchKernelGGL(lrn_lines_kernel, dim3(grid), dim3(256), 0, st, d_text, len, blk_prefix, nblk, line_start);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(lrn_classify_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, st, d_text, line_start, n_lines, info, is_row, kind, d_err);
        ABW_LAUNCH_CHECK(ctx);
        if ((rc = abw_device_scan_u32(ctx, is_row, n_lines, row_of))) return rc;
    }
    uint64_t n_rows = 0;
    ABW_HIP(ctx, hipMemcpyAsync(&n_rows, row_of + n_lines, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    if ((int64_t)n_rows != ndps)                                         // ref :127-130, :164-167
        return abw_fail(ctx, ABW_ERR_FORMAT, "abw_lrn_parse: %llu datapoint lines, the header says %lld", (unsigned long long)n_rows, (long long)ndps);
    if (n_rows > 0) {
        ABW_PROF(ctx, "lrn_parse");
        hipLaunchKernelGGL(lrn_blocks_kernel, dim3(grid), dim3(256), 0, st, d_text, len, blk_prefix, blk_nl, blk_tail, nblk, kind, row_of, n_lines,
                           h.ncols, (int64_t)ndps, d_names, d_keys, d_err);
        ABW_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(lrn_rows_kernel, dim3(grid), dim3(256), 0, st, d_text, info, kind, row_of, n_lines, h.ncols, (int64_t)ndps, d_names, d_keys, d_err);
        ABW_LAUNCH_CHECK(ctx);
    }
    unsigned long long err = 0;
    ABW_HIP(ctx, hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    if (err != ~0ull) {
        static const char* what[] = {"", "longer than the 8191 bytes the reference reads at once", "wrong number of fields",
                                     "a field is not [-]digits[.digits] with at most 3 decimals", "a key does not fit 32 bits"};
        const unsigned code = (unsigned)(err & 0xFF);
        return abw_fail(ctx, ABW_ERR_FORMAT, "abw_lrn_parse: line %llu: %s", (unsigned long long)(err >> 8), what[code < 5 ? code : 0]);
    }
    if (n_rows > 0) {
        ABW_HIP(ctx, hipMemcpyAsync(keys, d_keys, kb, hipMemcpyDeviceToHost, st));
        ABW_HIP(ctx, hipMemcpyAsync(dp_names, d_names, nb, hipMemcpyDeviceToHost, st));
        ABW_HIP(ctx, hipStreamSynchronize(st));
    }
    return ABW_OK;
}

// persistent workgroups of the writers: as many per CU as its 32 wave slots and its LDS hold
static unsigned emit_wgs_per_cu(int wpb, uint32_t row_stride)
{
    const unsigned by_waves = 32u / (unsigned)wpb, by_lds = (unsigned)((160u * 1024u) / ((size_t)wpb * row_stride + 256));
    return std::max(1u, std::min(by_waves, by_lds));
}

int abw_lrn_emit(abw_ctx* ctx, const int32_t* milli, int64_t n_rows, int32_t n_cols, uint64_t first_key,
                 uint8_t* text, uint64_t text_cap, uint64_t* text_len)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!milli || !text_len || n_rows < 0 || n_cols < 1 || n_cols > 64 * EMIT_COLS_PER_LANE)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_emit: bad argument (1 <= n_cols <= %d)", 64 * EMIT_COLS_PER_LANE);
    *text_len = 0;
    if (n_rows == 0) return ABW_OK;
    if ((uint64_t)n_cols * 17 + 24 > (uint64_t)EMIT_ROW_MAX) return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_emit: rows of %d columns do not fit the staging buffer", n_cols);
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    int rc;
    // scratch[6]: milli | scratch[4]: ticket, total + one look-back word per workgroup | scratch[0]: text, of the caller's
    // capacity (at most what the rows can take): one launch prints and sizes
    const size_t mb = (size_t)n_rows * n_cols * 4;
    const uint32_t row_stride = (uint32_t)(((uint64_t)n_cols * 13 + 24 + 16 + 15) & ~15ull);
    const int wpb = 8 * row_stride <= EMIT_LDS_MAX ? 8 : 4;
    const uint64_t nblk = ((uint64_t)n_rows + wpb - 1) / wpb;
    const size_t sb = (CTL_WORDS + (size_t)nblk) * 8;
    const unsigned grid = (unsigned)std::min<uint64_t>(nblk, (uint64_t)ctx->num_cu * emit_wgs_per_cu(wpb, row_stride));
    const bool with_text = text && text_cap > 0;
    const uint64_t cap = with_text ? std::min<uint64_t>(text_cap, (uint64_t)n_rows * ((uint64_t)n_cols * 13 + 21)) : 0;
    if ((rc = abw_reserve(ctx, ctx->scratch[6], mb))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[4], sb))) return rc;
    if (with_text && (rc = abw_reserve(ctx, ctx->scratch[0], cap + 64))) return rc;
    int32_t* d_milli = (int32_t*)ctx->scratch[6].p;
    unsigned long long* ctl = (unsigned long long*)ctx->scratch[4].p;
    unsigned long long* state = ctl + CTL_WORDS;
    uint8_t* d_text = with_text ? (uint8_t*)ctx->scratch[0].p : nullptr;
    ABW_HIP(ctx, hipMemcpyAsync(d_milli, milli, mb, hipMemcpyHostToDevice, st));
    auto launch = [&](auto kern, size_t lds) -> int {
        if (lds > 65536) ABW_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, EMIT_LDS_MAX));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), lds, st, (const int32_t*)d_milli, n_rows, n_cols, first_key, row_stride, ctl, state, d_text, cap);
        return ABW_OK;
    };
    {
        ABW_PROF(ctx, "lrn_emit");
        ABW_HIP(ctx, hipMemsetAsync(ctl, 0, sb, st));
        const size_t lds = (size_t)wpb * row_stride;
        if (with_text) rc = wpb == 8 ? launch(lrn_emit3_kernel<true, 8>, lds) : launch(lrn_emit3_kernel<true, 4>, lds);
        else rc = wpb == 8 ? launch(lrn_emit3_kernel<false, 8>, 0) : launch(lrn_emit3_kernel<false, 4>, 0);
        if (rc) return rc;
        ABW_LAUNCH_CHECK(ctx);
    }
    unsigned long long h_ctl[CTL_WORDS] = {0, 0, 0, 0};
    ABW_HIP(ctx, hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t total = h_ctl[CTL_TOTAL];
    *text_len = total;
    if (!text || total > text_cap) return abw_fail(ctx, ABW_ERR_CAPACITY, "abw_lrn_emit: the rows take %llu bytes, capacity %llu", (unsigned long long)total, (unsigned long long)text_cap);
    ABW_HIP(ctx, hipMemcpyAsync(text, d_text, total, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    return ABW_OK;
}

// d_values: [n_cols][ld], or with row_major the resident copy [ld][n_cols]
static int emit_rows_core(abw_ctx* ctx, const double* d_values, bool row_major, int64_t ld, int32_t n_cols, const int64_t* d_dps, int64_t n_rows,
                          uint8_t* d_text, uint64_t text_cap, uint64_t* text_len, uint64_t* d_row_off)
{
    if (!d_values || !text_len || n_rows < 0 || ld < 1 || n_cols < 1 || (n_rows > 0 && !d_dps))
        return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_emit_rows: bad argument");
    if ((uint64_t)n_cols * EMIT6_FIELD_MAX + 24 + 16 + 15 > (uint64_t)EMIT_ROW_MAX)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_emit_rows: rows of %d columns do not fit the staging buffer (at most %d)", n_cols, (EMIT_ROW_MAX - 24 - 31) / EMIT6_FIELD_MAX);
    *text_len = 0;
    if (n_rows == 0) return ABW_OK;
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    int rc;
    // scratch[4]: ticket, error, total | one look-back word per workgroup.  Eight rows (waves) per workgroup when two
    // workgroups' rows fit a CU's LDS, else four.  Measured at 189 columns, 4 / 8 / 16 rows per workgroup: 24.9 / 20.4 / 23.5 us
    // per 2 000-row cluster and 0.66 / 0.57 / 0.58 ms per 200 000 rows -- a workgroup's ticket and look-back are what costs
    const uint32_t row_stride = (uint32_t)(((uint64_t)n_cols * EMIT6_FIELD_MAX + 24 + 16 + 15) & ~15ull);      // 16: what a store may write behind the row's end
    const int wpb = 8 * row_stride <= EMIT_LDS_MAX ? 8 : 4;
    const uint64_t nblk = ((uint64_t)n_rows + wpb - 1) / wpb;
    const size_t sb = (CTL_WORDS + (size_t)nblk) * 8;
    const unsigned grid = (unsigned)std::min<uint64_t>(nblk, (uint64_t)ctx->num_cu * emit_wgs_per_cu(wpb, row_stride));
    if ((rc = abw_reserve(ctx, ctx->scratch[4], sb))) return rc;
    unsigned long long* ctl = (unsigned long long*)ctx->scratch[4].p;
    unsigned long long* state = ctl + CTL_WORDS;
    const bool with_text = d_text && text_cap > 0;
    auto launch = [&](auto kern, size_t lds) -> int {
        if (lds > 65536) ABW_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, EMIT_LDS_MAX));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), lds, st, d_values, ld, n_cols, d_dps, n_rows,
                           row_stride, ctl, state, with_text ? d_text : (uint8_t*)nullptr, with_text ? text_cap : (uint64_t)0, d_row_off);
        return ABW_OK;
    };
    {
        ABW_PROF(ctx, "lrn_emit_rows");
        ABW_HIP(ctx, hipMemsetAsync(ctl, 0, sb, st));
        const size_t lds = (size_t)wpb * row_stride;
#define EMIT6_PICK(T, R, L) (wpb == 8 ? launch(lrn_emit6_kernel<T, 8, R>, L) : launch(lrn_emit6_kernel<T, 4, R>, L))
        if (with_text) rc = row_major ? EMIT6_PICK(true, true, lds) : EMIT6_PICK(true, false, lds);
        else rc = row_major ? EMIT6_PICK(false, true, 0) : EMIT6_PICK(false, false, 0);
#undef EMIT6_PICK
        if (rc) return rc;
        ABW_LAUNCH_CHECK(ctx);
    }
    unsigned long long h_ctl[CTL_WORDS] = {0, 0, 0, 0};
    ABW_HIP(ctx, hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    if (h_ctl[CTL_ERR] != 0) {
        const unsigned long long err = ~h_ctl[CTL_ERR];
        if ((err & 0xFF) == 1) return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_emit_rows: row %llu names a datapoint outside [1, %lld]", (unsigned long long)(err >> 8), (long long)ld);
        return abw_fail(ctx, ABW_ERR_FORMAT, "abw_lrn_emit_rows: row %llu holds a value that is not finite or not below 9.2e12", (unsigned long long)(err >> 8));
    }
    const uint64_t total = h_ctl[CTL_TOTAL];
    *text_len = total;
    if (!d_text || total > text_cap)
        return abw_fail(ctx, ABW_ERR_CAPACITY, "abw_lrn_emit_rows: the rows take %llu bytes, capacity %llu", (unsigned long long)total, (unsigned long long)text_cap);
    return ABW_OK;
}

static int resident_table(abw_ctx* ctx, int64_t ld, int32_t n_cols)
{
    if (ctx->lrn_ld != ld || ctx->lrn_cols != n_cols)
        return abw_fail(ctx, ABW_ERR_STATE, "abw_lrn_emit_rows: no values given and no resident table of %lld x %d (abw_lrn_table_load)", (long long)ld, n_cols);
    return ABW_OK;
}

int abw_lrn_emit_rows_dev(abw_ctx* ctx, const double* d_values, int64_t ld, int32_t n_cols, const int64_t* d_dps, int64_t n_rows,
                          uint8_t* d_text, uint64_t text_cap, uint64_t* text_len, uint64_t* d_row_off)
{
    if (!ctx) return ABW_ERR_ARG;
    if (d_values) return emit_rows_core(ctx, d_values, false, ld, n_cols, d_dps, n_rows, d_text, text_cap, text_len, d_row_off);
    if (!text_len || ld < 1 || n_cols < 1) return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_emit_rows: bad argument");
    int rc;
    if ((rc = resident_table(ctx, ld, n_cols))) return rc;
    return emit_rows_core(ctx, (const double*)ctx->lrn_table.p, true, ld, n_cols, d_dps, n_rows, d_text, text_cap, text_len, d_row_off);
}

// the resident copy: [ld][n_cols], transposed on the device from the caller's [n_cols][ld]
static int table_keep(abw_ctx* ctx, const double* d_src, int64_t ld, int32_t n_cols)
{
    int rc;
    if ((rc = abw_reserve(ctx, ctx->lrn_table, (size_t)ld * (size_t)n_cols * 8))) return rc;
    hipLaunchKernelGGL(lrn_transpose_kernel, dim3((unsigned)((ld + 31) / 32), (unsigned)((n_cols + 31) / 32)), dim3(256), 0, ctx->stream,
                       d_src, ld, n_cols, (double*)ctx->lrn_table.p);
    ABW_LAUNCH_CHECK(ctx);
    ABW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->lrn_ld = ld; ctx->lrn_cols = n_cols;
    return ABW_OK;
}

int abw_lrn_table_load_dev(abw_ctx* ctx, const double* d_values, int64_t ld, int32_t n_cols)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_device_guard g(ctx->device);
    ctx->lrn_ld = 0; ctx->lrn_cols = 0;
    if (!d_values) { abw_release(ctx->lrn_table); return ABW_OK; }        // unload
    if (ld < 1 || n_cols < 1 || n_cols > 65535 * 32) return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_table_load: bad argument");
    return table_keep(ctx, d_values, ld, n_cols);
}

int abw_lrn_table_load(abw_ctx* ctx, const double* values, int64_t ld, int32_t n_cols)
{
    if (!ctx) return ABW_ERR_ARG;
    abw_device_guard g(ctx->device);
    ctx->lrn_ld = 0; ctx->lrn_cols = 0;
    if (!values) { abw_release(ctx->lrn_table); return ABW_OK; }          // unload
    if (ld < 1 || n_cols < 1 || n_cols > 65535 * 32) return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_table_load: bad argument");
    const size_t vb = (size_t)ld * (size_t)n_cols * 8;
    int rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[6], vb))) return rc;
    ABW_HIP(ctx, hipMemcpyAsync(ctx->scratch[6].p, values, vb, hipMemcpyHostToDevice, ctx->stream));
    return table_keep(ctx, (const double*)ctx->scratch[6].p, ld, n_cols);
}

int abw_lrn_emit_rows(abw_ctx* ctx, const double* values, int64_t ld, int32_t n_cols, const int64_t* dps, int64_t n_rows,
                      uint8_t* text, uint64_t text_cap, uint64_t* text_len)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!text_len || n_rows < 0 || ld < 1 || n_cols < 1 || (n_rows > 0 && !dps))
        return abw_fail(ctx, ABW_ERR_ARG, "abw_lrn_emit_rows: bad argument");
    int rc;
    if (!values && (rc = resident_table(ctx, ld, n_cols))) return rc;
    *text_len = 0;
    if (n_rows == 0) return ABW_OK;
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    // scratch[6]: the table (unless resident) | scratch[5]: the datapoint list | scratch[0]: text
    const size_t vb = (size_t)ld * (size_t)n_cols * 8, db = (size_t)n_rows * 8;
    const double* d_values = (const double*)ctx->lrn_table.p;
    if (values) {
        if ((rc = abw_reserve(ctx, ctx->scratch[6], vb))) return rc;
        ABW_HIP(ctx, hipMemcpyAsync(ctx->scratch[6].p, values, vb, hipMemcpyHostToDevice, st));
        d_values = (const double*)ctx->scratch[6].p;
    }
    if ((rc = abw_reserve(ctx, ctx->scratch[5], db))) return rc;
    ABW_HIP(ctx, hipMemcpyAsync(ctx->scratch[5].p, dps, db, hipMemcpyHostToDevice, st));
    // printed straight away into a device buffer of the caller's capacity (at most what the rows can take); without a
    // buffer, or when it is too small, the call only sizes the text
    uint64_t need = 0;
    uint8_t* d_text = NULL;
    uint64_t cap = 0;
    if (text && text_cap > 0) {
        cap = std::min<uint64_t>(text_cap, (uint64_t)n_rows * ((uint64_t)n_cols * EMIT6_FIELD_MAX + 24));
        if ((rc = abw_reserve(ctx, ctx->scratch[0], cap + 64))) return rc;
        d_text = (uint8_t*)ctx->scratch[0].p;
    }
    rc = emit_rows_core(ctx, d_values, !values, ld, n_cols, (const int64_t*)ctx->scratch[5].p, n_rows, d_text, cap, &need, NULL);
    *text_len = need;
    if (rc) return rc;
    ABW_HIP(ctx, hipMemcpyAsync(text, ctx->scratch[0].p, need, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    return ABW_OK;
}

}  // extern "C"
