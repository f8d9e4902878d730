// Row N1, the reader side — SAM text to the arrays abw_cov_accumulate takes.
//
// Reference semantics (restated in oracle/abw_oracle.c abw_or_sam_line, pinned to the SAM files of a reference run):
//   lines     src/ReadMappingReader.cpp:80-116  lines that start with '@' are skipped, every other line is a record;
//                                               "\n" and "\r" at its end are chomped (src/common.cpp:38-49)
//   fields    src/ReadMapping.cpp:23-72         split at tabs (src/common.cpp:95-108); FLAG = atoi(fs[1]), RNAME = fs[2],
//                                               POS = atoi(fs[3]) - 1 as size_t (POS 0 wraps), read length = SEQ.size(),
//                                               a lower-case n in SEQ throws (src/String.cpp:45-49)
//   SNPs      src/ReadMapping.cpp:78-185        the first field behind the 11th that holds "MD:Z:"; one SNP per mismatched
//                                               and per deleted base of it, one per I operation of the CIGAR string; every
//                                               malformation throws, as does a length that does not add up (:182-183)
//   filter    src/abawaca-build.cpp:546         unmapped (FLAG & 4), more than max_snps SNPs, secondary (FLAG & 0x100)
//   lookup    src/abawaca-build.cpp:549-551     RNAME among the scaffold names; unknown names are skipped
// A line the reference would not survive (it throws, or indexes a field that is not there), and the few things this parser
// is stricter about (FLAG / POS must be 1..9 digits; "MD:Z:" inside a field), make the call fail with ABW_ERR_FORMAT
// (ABW_ERR_LOWER_N for the n) and name the line: the caller falls back to the reference's reader, which then does what it
// does.  The `-c` sample's read-pair links (src/abawaca-build.cpp:531-545) stay with the caller.
//
// Kernels: newlines per 4 KiB block -> scan -> start of every line; one thread per line walks it once (fields counted at
// the tabs, FLAG / POS converted and RNAME hashed on the way, SEQ checked for n), then the MD:Z and CIGAR strings; the
// scaffold comes from a hash table of the names (FNV-1a, open addressing, built by the host per call; a hit is confirmed
// by comparing the bytes); two scans compact the records and the records that pass the filter, file order kept.
#include "abw_internal.h"
#include <string.h>
#include <string>
#include <vector>

namespace {

constexpr int SAM_BLK = 4096;                       // bytes of text per workgroup of the line kernels (16 per thread)
constexpr uint64_t FNV_OFFSET = 0xcbf29ce484222325ull, FNV_PRIME = 0x100000001b3ull;
constexpr uint32_t SLOT_EMPTY = 0xFFFFFFFFu;

struct SamNames {
    const uint64_t* slot_hash; const uint32_t* slot_id; uint64_t mask;
    const uint8_t* blob; const uint64_t* off; uint32_t n_scaf;
};

__device__ __forceinline__ uint32_t eq_mask4(uint32_t x, uint32_t pat)     // bit i set iff byte i of x equals the pattern byte
{
    const uint32_t t = x ^ pat;
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
    return __builtin_amdgcn_udot4(z, 0x08040201u, 0u, false) >> 7;
}
__device__ __forceinline__ uint32_t eq_mask16(const uint4& v, uint32_t pat)
{
    return eq_mask4(v.x, pat) | (eq_mask4(v.y, pat) << 4) | (eq_mask4(v.z, pat) << 8) | (eq_mask4(v.w, pat) << 12);
}

// 0x80 in every byte of x that equals the pattern byte, 0 elsewhere (exact: no borrow between the bytes)
__device__ __forceinline__ uint32_t eq_bytes4(uint32_t x, uint32_t pat)
{
    const uint32_t t = x ^ pat;
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}

// the 16 bytes of the text at `at` (a multiple of 16; past the end of the text: zeros)
__device__ __forceinline__ uint4 text16(const uint8_t* __restrict__ text, uint64_t at, uint64_t len)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (at + 16 <= len) memcpy(&v, text + at, 16);
    else if (at < len) {
        uint8_t b[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 16 && at + i < len; i++) b[i] = text[at + i];
        memcpy(&v, b, 16);
    }
    return v;
}

__global__ __launch_bounds__(256) void sam_nlcount_kernel(const uint8_t* __restrict__ text, uint64_t len, uint32_t* __restrict__ blk_nl)
{
    __shared__ uint32_t s_c;
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const uint4 v = text16(text, (uint64_t)blockIdx.x * SAM_BLK + (uint64_t)threadIdx.x * 16, len);
    uint32_t c = __popc(eq_bytes4(v.x, 0x0A0A0A0Au)) + __popc(eq_bytes4(v.y, 0x0A0A0A0Au)) + __popc(eq_bytes4(v.z, 0x0A0A0A0Au)) +
                 __popc(eq_bytes4(v.w, 0x0A0A0A0Au));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor((int)c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0) blk_nl[blockIdx.x] = s_c;
}

// line_start[0] = 0 (written by the host); the line behind the k-th newline of the text starts at its position + 1
__global__ __launch_bounds__(256) void sam_lines_kernel(const uint8_t* __restrict__ text, uint64_t len, const uint64_t* __restrict__ blk_prefix,
                                                       uint64_t* __restrict__ line_start)
{
    __shared__ uint32_t s_w[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t at = (uint64_t)blockIdx.x * SAM_BLK + (uint64_t)threadIdx.x * 16;
    const uint4 v = text16(text, at, len);
    uint32_t z[4] = {eq_bytes4(v.x, 0x0A0A0A0Au), eq_bytes4(v.y, 0x0A0A0A0Au), eq_bytes4(v.z, 0x0A0A0A0Au), eq_bytes4(v.w, 0x0A0A0A0Au)};
    const uint32_t c = __popc(z[0]) + __popc(z[1]) + __popc(z[2]) + __popc(z[3]);
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += y; }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    if (c == 0) return;
    uint64_t k = blk_prefix[blockIdx.x] + incl - c;
    for (int w = 0; w < wid; w++) k += s_w[w];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t m = z[q];
        while (m) {
            const int i = (__ffs(m) - 1) >> 3;
            m &= m - 1;
            line_start[++k] = at + (uint64_t)(4 * q + i) + 1;
        }
    }
}

// Returns 0 or the refusal code of oracle/abw_oracle.h (abw_or_sam_line); t[a, e): the line without its end.
// A thread per line, and the lanes of a wave must do the same KIND of work at the same time or each waits for the others'
// kind in turn: so the line is taken in phases -- (A) the ends of its first eleven fields, from tab to tab through aligned
// 16-byte loads and byte-equality masks (nine tenths of a record are QNAME, SEQ and QUAL, of which only the ends matter,
// and the n of SEQ falls out of a second mask); (B) the three short fields FLAG, RNAME, POS byte by byte; (C) the tags
// for "MD:Z:"; (D) the MD:Z and CIGAR strings.
// 64 bytes of a line in LDS, one window per thread: dword d of thread x at w[d * 256] (w already points at the thread's
// column, so the lanes of a wave that read the same dword of their windows hit 64 different banks).  The short fields are
// read byte by byte; from global memory every byte would be a round trip to the L2 with nothing else in flight.
struct Window {
    uint32_t* w;
    uint64_t base;                                                        // text index of the window's byte 0; ~0: nothing loaded
    __device__ __forceinline__ void load(const uint8_t* __restrict__ t, uint64_t from, uint64_t e)
    {
        const uintptr_t addr = (uintptr_t)(t + from);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(addr & ~(uintptr_t)15);
        base = from - (uint64_t)(addr & 15);
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = (base + 16 * (uint64_t)j < e) ? src[j] : make_uint4(0, 0, 0, 0);     // chunks that hold a byte of the line
#pragma unroll
        for (int j = 0; j < 4; j++) { w[(4 * j) * 256] = v[j].x; w[(4 * j + 1) * 256] = v[j].y; w[(4 * j + 2) * 256] = v[j].z; w[(4 * j + 3) * 256] = v[j].w; }
    }
    __device__ __forceinline__ bool holds(uint64_t i) const { return i - base < 64; }
    __device__ __forceinline__ uint8_t get(uint64_t i) const { const uint32_t o = (uint32_t)(i - base); return (uint8_t)(w[(o >> 2) * 256] >> (8 * (o & 3))); }
};

__device__ int sam_line(const uint8_t* __restrict__ t, uint64_t a, uint64_t e, const SamNames& nm, uint32_t* lds, uint32_t& flag, uint64_t& pos0,
                        uint32_t& seq_len, uint32_t& nsnps, uint32_t& scaf)
{
    if (e == a) return 1;                                                // split() of an empty line: no field at all
    // (A) tp[f]: where field f ends (a tab, or the end of the line), relative to a; ref src/ReadMapping.cpp:36-54.
    // Four chunks are loaded at a time: a thread has nothing else in flight while it waits for one.
    uint32_t tp[11];
    int nf = 0;
    bool lower_n = false;
    for (uint64_t i = a; i < e && nf < 11;) {
        const uintptr_t addr = (uintptr_t)(t + i);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(addr & ~(uintptr_t)15);   // the aligned 16 bytes that hold byte i: inside its page
        const int k0 = (int)(addr & 15);
        const uint64_t base0 = i - (uint64_t)k0;                         // text index of the first chunk's byte 0 (may lie before the line)
        uint4 vv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) vv[j] = (base0 + 16 * (uint64_t)j < e) ? src[j] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t base = base0 + 16 * (uint64_t)j;
            if (base < e && nf < 11) {
                const uint4 v = vv[j];
                int k = j == 0 ? k0 : 0;
                const int lim = (e - base < 16) ? (int)(e - base) : 16;  // the line's bytes of the chunk: [k, lim)
                const uint32_t valid = ((1u << lim) - 1u) & ~((1u << k) - 1u);
                uint32_t tabs = 0;
                if (eq_bytes4(v.x, 0x09090909u) | eq_bytes4(v.y, 0x09090909u) | eq_bytes4(v.z, 0x09090909u) | eq_bytes4(v.w, 0x09090909u))
                    tabs = eq_mask16(v, 0x09090909u) & valid;            // most chunks of a record hold no tab
                const uint32_t ens = (nf == 9 || (nf < 9 && tabs)) ? (eq_mask16(v, 0x6E6E6E6Eu) & valid) : 0u;    // only where SEQ can be
                while (tabs && nf < 11) {
                    const int p = __ffs(tabs) - 1;
                    tabs &= tabs - 1;
                    if (nf == 9) lower_n |= ((ens >> k) & ((1u << (p - k)) - 1u)) != 0;      // SEQ: src/String.cpp:45-49
                    tp[nf++] = (uint32_t)(base + (uint64_t)p - a);
                    k = p + 1;
                }
                if (nf == 9 && k < lim) lower_n |= (ens >> k) != 0;
                i = base + (uint64_t)lim;
            }
        }
    }
    if (nf < 11) tp[nf++] = (uint32_t)(e - a);                           // the last field ends with the line
    if (nf < 11) return 1;
    // the short fields through two windows: FLAG .. CIGAR (usually all of them), and the tags
    Window w1 = {lds, 0}, w2 = {lds + 16 * 256, 0};
    w1.load(t, a + tp[0] + 1, e);
    if (tp[10] + 1 < e - a) w2.load(t, a + tp[10] + 1, e); else w2.base = ~0ull - 64;
    auto rd = [&](uint64_t i) -> uint8_t { return w1.holds(i) ? w1.get(i) : w2.holds(i) ? w2.get(i) : t[i]; };
    // (B) FLAG = atoi(fs[1]) ref :39, POS = atoi(fs[3]) - 1 ref :41 (1..9 digits, nothing else), RNAME hashed
    uint64_t vflag = 0, vpos = 0, h = FNV_OFFSET;
    {
        const uint32_t f0 = tp[0] + 1, f1 = tp[1];
        if (f1 - f0 < 1 || f1 - f0 > 9) return 2;
        for (uint32_t j = f0; j < f1; j++) { const uint8_t c = rd(a + j); if (c < '0' || c > '9') return 2; vflag = vflag * 10 + (uint64_t)(c - '0'); }
        const uint32_t p0 = tp[2] + 1, p1 = tp[3];
        if (p1 - p0 < 1 || p1 - p0 > 9) return 2;
        for (uint32_t j = p0; j < p1; j++) { const uint8_t c = rd(a + j); if (c < '0' || c > '9') return 2; vpos = vpos * 10 + (uint64_t)(c - '0'); }
    }
    if (lower_n) return 3;
    const uint64_t rn_off = a + tp[1] + 1, rn_len = tp[2] - tp[1] - 1;
    const uint64_t cg_off = a + tp[4] + 1, cg_len = tp[5] - tp[4] - 1;
    const uint64_t sq_len = tp[9] - tp[8] - 1, ql_len = tp[10] - tp[9] - 1;
    for (uint64_t j = 0; j < rn_len; j++) h = (h ^ rd(rn_off + j)) * FNV_PRIME;
    flag = (uint32_t)vflag;
    pos0 = vpos - 1;
    seq_len = (uint32_t)sq_len;
    nsnps = 0;
    // the scaffold of RNAME (src/abawaca-build.cpp:549)
    scaf = nm.n_scaf;
    for (uint64_t s = h & nm.mask;; s = (s + 1) & nm.mask) {
        const uint32_t id = nm.slot_id[s];
        if (id == SLOT_EMPTY) break;
        if (nm.slot_hash[s] == h && nm.off[id + 1] - nm.off[id] == rn_len) {
            const uint8_t* __restrict__ name = nm.blob + nm.off[id];
            bool same = true;
            for (uint64_t j = 0; j < rn_len; j++) if (name[j] != rd(rn_off + j)) { same = false; break; }
            if (same) { scaf = id; break; }
        }
    }
    // (C) the MD:Z field, ref :57-69: the first field behind the 11th that holds "MD:Z:" (a five-state matcher)
    uint64_t md = 0, md_end = 0;
    bool have_md = false;
    {
        uint64_t fs = a + tp[10] + 1;
        int m = 0;
        for (uint64_t j = fs; j < e; j++) {
            const uint8_t c = rd(j);
            if (c == '\t') { fs = j + 1; m = 0; continue; }
            const uint8_t want = m == 0 ? 'M' : m == 1 ? 'D' : m == 3 ? 'Z' : ':';
            m = (c == want) ? m + 1 : (c == 'M' ? 1 : 0);
            if (m == 5) {
                if (j - 4 != fs) return 4;                               // inside a field: refused here
                have_md = true; md = fs;
                break;
            }
        }
        if (have_md) { md_end = md + 5; while (md_end < e && rd(md_end) != '\t') md_end++; }
    }
    if (!have_md) return 0;                                              // ref :80-81
    for (uint64_t j = md; j < md_end; j++) if (rd(j) == ' ') return 4;    // ref :60-66 keeps the blank, :102-103 throws on it
    // determine_snps, ref :85-128
    uint64_t q = md + 5, read_pos = 0;
    int nd = 0;
    for (uint8_t c; q < md_end && (c = rd(q)) >= '0' && c <= '9'; q++) { read_pos = read_pos * 10 + (uint64_t)(c - '0'); if (++nd > 9) return 4; }
    read_pos--;                                                          // ref :98
    uint32_t snps = 0;
    while (q < md_end) {
        const uint8_t c = rd(q);
        if (c != '^' && (c < 'A' || c > 'Z')) return 4;                  // ref :102-103
        if (c == '^') {
            q++;
            for (uint8_t d; q < md_end && ((d = rd(q)) < '0' || d > '9'); q++) snps++;          // ref :107-111
        } else {
            read_pos++;
            if (read_pos >= sq_len || read_pos >= ql_len) return 4;      // ref :115: get() / at() throw
            snps++;
            q++;
        }
        uint64_t o = 0;
        nd = 0;
        for (uint8_t d; q < md_end && (d = rd(q)) >= '0' && d <= '9'; q++) { o = o * 10 + (uint64_t)(d - '0'); if (++nd > 9) return 4; }
        if (nd == 0) return 4;                                           // ref :120-121
        read_pos += o;
    }
    // the CIGAR string, ref :132-179
    if (cg_len == 0) return 5;
    const uint8_t last = rd(cg_off + cg_len - 1);
    if (last != 'I' && last != 'D' && last != 'M') return 5;             // ref :134-137
    uint64_t read_pos2 = 0;
    for (uint64_t j = cg_off; j < cg_off + cg_len;) {
        uint64_t o = 0;
        nd = 0;
        uint8_t op;
        for (; (op = rd(j)) >= '0' && op <= '9'; j++) { o = o * 10 + (uint64_t)(op - '0'); if (++nd > 9) return 5; }   // the last character is a letter
        if (o == 0) return 5;                                            // ref :147-149
        j++;
        if (op == 'M') read_pos2 += o;
        else if (op == 'D') { }
        else if (op == 'I') { read_pos += o; snps++; read_pos2 += o; }   // ref :161-173
        else return 5;                                                   // ref :176-177
    }
    if (read_pos + 1 != sq_len || read_pos2 != sq_len) return 6;         // ref :182-183
    nsnps = snps;
    return 0;
}

// one thread per line slot (n_slots = newlines + 1; the last one may be empty)
__global__ __launch_bounds__(256) void sam_parse_kernel(const uint8_t* __restrict__ text, uint64_t len, const uint64_t* __restrict__ line_start,
                                                       uint64_t n_slots, SamNames nm, uint32_t max_snps,
                                                       uint32_t* __restrict__ l_scaf, uint64_t* __restrict__ l_pos, uint32_t* __restrict__ l_len,
                                                       uint32_t* __restrict__ l_flag, uint32_t* __restrict__ l_nsnps,
                                                       uint32_t* __restrict__ is_rec, uint32_t* __restrict__ keep, unsigned long long* __restrict__ err)
{
    __shared__ uint32_t s_win[2 * 16 * 256];                             // two windows of 64 bytes per thread (struct Window)
    const uint64_t l = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= n_slots) return;
    const uint64_t a = line_start[l];
    uint64_t e = (l + 1 < n_slots) ? line_start[l + 1] - 1 : len;        // the newline, or the end of the text
    uint32_t rec = 0, kp = 0;
    if (a < len && text[a] != '@') {                                     // src/ReadMappingReader.cpp:105-107
        while (e > a && (text[e - 1] == '\r' || text[e - 1] == '\n')) e--;           // chomp
        uint32_t flag = 0, seq_len = 0, nsnps = 0, scaf = nm.n_scaf;
        uint64_t pos0 = 0;
        const int rc = sam_line(text, a, e, nm, s_win + threadIdx.x, flag, pos0, seq_len, nsnps, scaf);
        if (rc) atomicMin(err, (unsigned long long)(l + 1) << 8 | (unsigned)rc);     // the first refused line wins
        rec = 1;
        kp = (rc == 0 && !(flag & 0x4u) && nsnps <= max_snps && !(flag & 0x100u)) ? 1u : 0u;    // src/abawaca-build.cpp:546
        l_scaf[l] = scaf; l_pos[l] = pos0; l_len[l] = seq_len; l_flag[l] = flag; l_nsnps[l] = nsnps;
    }
    is_rec[l] = rec;
    keep[l] = kp;
}

__global__ __launch_bounds__(256) void sam_compact_kernel(uint64_t n_slots, const uint32_t* __restrict__ is_rec, const uint32_t* __restrict__ keep,
                                                         const uint64_t* __restrict__ rec_at, const uint64_t* __restrict__ keep_at, uint64_t cap,
                                                         const uint32_t* __restrict__ l_scaf, const uint64_t* __restrict__ l_pos,
                                                         const uint32_t* __restrict__ l_len, const uint32_t* __restrict__ l_flag,
                                                         const uint32_t* __restrict__ l_nsnps,
                                                         uint32_t* __restrict__ read_scaf, uint64_t* __restrict__ read_pos, uint32_t* __restrict__ read_len,
                                                         uint32_t* __restrict__ rec_scaf, uint64_t* __restrict__ rec_pos, uint32_t* __restrict__ rec_len,
                                                         uint32_t* __restrict__ rec_flag, uint32_t* __restrict__ rec_nsnps)
{
    const uint64_t l = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= n_slots || !is_rec[l]) return;
    const uint64_t r = rec_at[l];
    if (r < cap) {
        if (rec_scaf) rec_scaf[r] = l_scaf[l];
        if (rec_pos) rec_pos[r] = l_pos[l];
        if (rec_len) rec_len[r] = l_len[l];
        if (rec_flag) rec_flag[r] = l_flag[l];
        if (rec_nsnps) rec_nsnps[r] = l_nsnps[l];
    }
    if (keep[l]) {
        const uint64_t k = keep_at[l];
        if (k < cap) {
            if (read_scaf) read_scaf[k] = l_scaf[l];
            if (read_pos) read_pos[k] = l_pos[l];
            if (read_len) read_len[k] = l_len[l];
        }
    }
}

uint64_t fnv1a(const uint8_t* p, uint64_t n)
{
    uint64_t h = FNV_OFFSET;
    for (uint64_t i = 0; i < n; i++) h = (h ^ p[i]) * FNV_PRIME;
    return h;
}

}  // namespace

namespace {
size_t up16(size_t b) { return (b + 15) & ~(size_t)15; }

// The scaffold names as an open-addressing table (at most half full; equal names keep the first, as an unordered_map insert
// does), built on the host and uploaded into `buf`: slot_hash u64[n_slots] | name_off u64[n_scaf + 1] | slot_id u32[n_slots] | blob.
int build_names(abw_ctx* ctx, abw_devbuf& buf, const uint8_t* names, const uint64_t* name_off, uint64_t n_scaf, uint64_t* n_slots_out)
{
    hipStream_t st = ctx->stream;
    uint64_t n_slots_tab = 16;
    while (n_slots_tab < 2 * n_scaf + 2) n_slots_tab <<= 1;
    std::vector<uint64_t> slot_hash((size_t)n_slots_tab, 0);
    std::vector<uint32_t> slot_id((size_t)n_slots_tab, SLOT_EMPTY);
    for (uint64_t s = 0; s < n_scaf; s++) {
        if (name_off[s + 1] < name_off[s]) return abw_fail(ctx, ABW_ERR_ARG, "abw_sam_parse: name_off not monotone");
        const uint64_t n = name_off[s + 1] - name_off[s], h = fnv1a(names + name_off[s], n);
        uint64_t at = h & (n_slots_tab - 1);
        bool dup = false;
        while (slot_id[at] != SLOT_EMPTY) {
            const uint32_t o = slot_id[at];
            if (slot_hash[at] == h && name_off[o + 1] - name_off[o] == n && memcmp(names + name_off[o], names + name_off[s], n) == 0) { dup = true; break; }
            at = (at + 1) & (n_slots_tab - 1);
        }
        if (!dup) { slot_hash[at] = h; slot_id[at] = (uint32_t)s; }
    }
    const uint64_t blob_len = n_scaf ? name_off[n_scaf] : 0;
    const size_t b_sh = n_slots_tab * 8, b_no = (n_scaf + 1) * 8, b_si = up16(n_slots_tab * 4), b_bl = up16(blob_len + 1);
    int rc;
    if ((rc = abw_reserve(ctx, buf, b_sh + b_no + b_si + b_bl))) return rc;
    uint8_t* p0 = (uint8_t*)buf.p;
    const uint64_t zero_off = 0;
    // (pageable sources: each copy has left the host buffer when the call returns)
    ABW_HIP(ctx, hipMemcpyAsync(p0, slot_hash.data(), b_sh, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(p0 + b_sh, n_scaf ? name_off : &zero_off, b_no, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipMemcpyAsync(p0 + b_sh + b_no, slot_id.data(), n_slots_tab * 4, hipMemcpyHostToDevice, st));
    if (blob_len) ABW_HIP(ctx, hipMemcpyAsync(p0 + b_sh + b_no + b_si, names, blob_len, hipMemcpyHostToDevice, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    *n_slots_out = n_slots_tab;
    return ABW_OK;
}
}  // namespace

// The names of an assembly once, for every abw_sam_parse[_dev] call that passes names == NULL afterwards (a SAM file read in
// pieces, several SAM files): the table stays on the device until the next load or abw_ctx_destroy.  n_scaf = 0 drops it.
extern "C" int abw_sam_names_load(abw_ctx* ctx, const uint8_t* names, const uint64_t* name_off, uint64_t n_scaf)
{
    if (!ctx) return ABW_ERR_ARG;
    if ((n_scaf && (!names || !name_off)) || n_scaf >= 0xFFFFFFFEull) return abw_fail(ctx, ABW_ERR_ARG, "abw_sam_names_load: bad argument");
    abw_device_guard g(ctx->device);
    ctx->sam_slots = 0; ctx->sam_n_scaf = 0;
    if (n_scaf == 0) return ABW_OK;
    uint64_t n_slots = 0;
    if (int rc = build_names(ctx, ctx->sam_names, names, name_off, n_scaf, &n_slots)) return rc;
    ctx->sam_slots = n_slots; ctx->sam_n_scaf = n_scaf;
    return ABW_OK;
}

extern "C" int abw_sam_parse_dev(abw_ctx* ctx, const uint8_t* d_text, uint64_t len, const uint8_t* names, const uint64_t* name_off,
                                 uint64_t n_scaf, uint32_t max_snps, uint64_t cap, uint32_t* d_read_scaf, uint64_t* d_read_pos,
                                 uint32_t* d_read_len, uint32_t* d_rec_scaf, uint64_t* d_rec_pos, uint32_t* d_rec_len, uint32_t* d_rec_flag,
                                 uint32_t* d_rec_nsnps, abw_sam_counts* counts)
{
    if (!ctx) return ABW_ERR_ARG;
    const bool resident = !names && !name_off && n_scaf != 0;          // the table of abw_sam_names_load
    if (!counts || (len && !d_text) || (n_scaf && !resident && (!names || !name_off)) || n_scaf >= 0xFFFFFFFEull)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_sam_parse: bad argument");
    if (resident && (ctx->sam_slots == 0 || ctx->sam_n_scaf != n_scaf))
        return abw_fail(ctx, ABW_ERR_STATE, "abw_sam_parse: names == NULL needs the table of abw_sam_names_load for these %llu scaffolds", (unsigned long long)n_scaf);
    counts->lines = counts->records = counts->kept = 0;
    if (len == 0) return ABW_OK;
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    int rc;
    uint64_t n_slots_tab = ctx->sam_slots;
    if (!resident && (rc = build_names(ctx, ctx->scratch[0], names, name_off, n_scaf, &n_slots_tab))) return rc;
    const uint64_t nblk = (len + SAM_BLK - 1) / SAM_BLK;
    const size_t b_sh = n_slots_tab * 8, b_no = (n_scaf + 1) * 8, b_si = up16(n_slots_tab * 4);
    uint8_t* p0 = (uint8_t*)(resident ? ctx->sam_names.p : ctx->scratch[0].p);
    uint64_t* d_slot_hash = (uint64_t*)p0; uint64_t* d_name_off = (uint64_t*)(p0 + b_sh);
    uint32_t* d_slot_id = (uint32_t*)(p0 + b_sh + b_no); uint8_t* d_blob = p0 + b_sh + b_no + b_si;
    const size_t b_bp = (nblk + 2) * 8, b_bn = up16(nblk * 4);
    if ((rc = abw_reserve(ctx, ctx->scratch[1], b_bp + b_bn + 16))) return rc;
    uint64_t* blk_prefix = (uint64_t*)ctx->scratch[1].p;
    uint32_t* blk_nl = (uint32_t*)((uint8_t*)blk_prefix + b_bp);
    unsigned long long* d_err = (unsigned long long*)((uint8_t*)blk_nl + b_bn);
    ABW_PROF(ctx, "sam_parse");
    hipLaunchKernelGGL(sam_nlcount_kernel, dim3((unsigned)nblk), dim3(256), 0, st, d_text, len, blk_nl);
    ABW_LAUNCH_CHECK(ctx);
    if ((rc = abw_device_scan_u32(ctx, blk_nl, nblk, blk_prefix))) return rc;
    if ((rc = abw_reserve_pinned(ctx, 64))) return rc;
    unsigned long long* h_back = (unsigned long long*)ctx->pinned;       // [0] newlines, [1] last byte, [2] error, [3] records, [4] kept
    ABW_HIP(ctx, hipMemcpyAsync(h_back, blk_prefix + nblk, 8, hipMemcpyDeviceToHost, st));
    h_back[1] = 0;
    ABW_HIP(ctx, hipMemcpyAsync(h_back + 1, d_text + len - 1, 1, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t n_nl = h_back[0], n_slots = n_nl + 1;
    counts->lines = n_nl + (((uint8_t)h_back[1] != (uint8_t)'\n') ? 1 : 0);
    // scratch[2]: line_start u64[n_slots] | l_pos u64 | rec_at, keep_at u64[n_slots + 1] | l_scaf, l_len, l_flag, l_nsnps, is_rec, keep u32[n_slots]
    const size_t b_l8 = n_slots * 8, b_at = (n_slots + 2) * 8, b_l4 = up16(n_slots * 4);
    if ((rc = abw_reserve(ctx, ctx->scratch[2], 2 * b_l8 + 2 * b_at + 6 * b_l4))) return rc;
    uint8_t* p2 = (uint8_t*)ctx->scratch[2].p;
    uint64_t* line_start = (uint64_t*)p2;           p2 += b_l8;
    uint64_t* l_pos = (uint64_t*)p2;                p2 += b_l8;
    uint64_t* rec_at = (uint64_t*)p2;               p2 += b_at;
    uint64_t* keep_at = (uint64_t*)p2;              p2 += b_at;
    uint32_t* l_scaf = (uint32_t*)p2;               p2 += b_l4;
    uint32_t* l_len = (uint32_t*)p2;                p2 += b_l4;
    uint32_t* l_flag = (uint32_t*)p2;               p2 += b_l4;
    uint32_t* l_nsnps = (uint32_t*)p2;              p2 += b_l4;
    uint32_t* is_rec = (uint32_t*)p2;               p2 += b_l4;
    uint32_t* keep = (uint32_t*)p2;
    ABW_HIP(ctx, hipMemsetAsync(line_start, 0, 8, st));
    ABW_HIP(ctx, hipMemsetAsync(d_err, 0xFF, 8, st));
    hipLaunchKernelGGL(sam_lines_kernel, dim3((unsigned)nblk), dim3(256), 0, st, d_text, len, (const uint64_t*)blk_prefix, line_start);
    ABW_LAUNCH_CHECK(ctx);
    SamNames nm = {d_slot_hash, d_slot_id, n_slots_tab - 1, d_blob, d_name_off, (uint32_t)n_scaf};
    const unsigned gl = (unsigned)((n_slots + 255) / 256);
    hipLaunchKernelGGL(sam_parse_kernel, dim3(gl), dim3(256), 0, st, d_text, len, (const uint64_t*)line_start, n_slots, nm, max_snps,
                       l_scaf, l_pos, l_len, l_flag, l_nsnps, is_rec, keep, d_err);
    ABW_LAUNCH_CHECK(ctx);
    if ((rc = abw_device_scan_u32(ctx, is_rec, n_slots, rec_at))) return rc;
    if ((rc = abw_device_scan_u32(ctx, keep, n_slots, keep_at))) return rc;
    hipLaunchKernelGGL(sam_compact_kernel, dim3(gl), dim3(256), 0, st, n_slots, (const uint32_t*)is_rec, (const uint32_t*)keep,
                       (const uint64_t*)rec_at, (const uint64_t*)keep_at, cap, (const uint32_t*)l_scaf, (const uint64_t*)l_pos,
                       (const uint32_t*)l_len, (const uint32_t*)l_flag, (const uint32_t*)l_nsnps,
                       d_read_scaf, d_read_pos, d_read_len, d_rec_scaf, d_rec_pos, d_rec_len, d_rec_flag, d_rec_nsnps);
    ABW_LAUNCH_CHECK(ctx);
    ABW_HIP(ctx, hipMemcpyAsync(h_back + 2, d_err, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipMemcpyAsync(h_back + 3, rec_at + n_slots, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipMemcpyAsync(h_back + 4, keep_at + n_slots, 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    counts->records = h_back[3];
    counts->kept = h_back[4];
    if (h_back[2] != ~0ull) {
        const int code = (int)(h_back[2] & 0xFF);
        static const char* what[] = {"", "fewer than 11 fields", "FLAG or POS is not a plain number", "a lower-case n in SEQ", "the MD:Z string",
                                     "the CIGAR string", "SNP offsets and CIGAR lengths do not add up to the read length"};
        counts->kept = 0;
        return abw_fail(ctx, code == 3 ? ABW_ERR_LOWER_N : ABW_ERR_FORMAT, "abw_sam_parse: line %llu: %s (code %d): the reference's reader does not survive "
                        "this line, or this parser is stricter -- use the reference's reader", (unsigned long long)(h_back[2] >> 8),
                        code >= 1 && code <= 6 ? what[code] : "?", code);
    }
    if (counts->records > cap) return abw_fail(ctx, ABW_ERR_CAPACITY, "abw_sam_parse: %llu records, capacity %llu", (unsigned long long)counts->records,
                                               (unsigned long long)cap);
    return ABW_OK;
}

extern "C" int abw_sam_parse(abw_ctx* ctx, const uint8_t* text, uint64_t len, const uint8_t* names, const uint64_t* name_off, uint64_t n_scaf,
                             uint32_t max_snps, uint64_t cap, uint32_t* read_scaf, uint64_t* read_pos, uint32_t* read_len, uint32_t* rec_scaf,
                             uint64_t* rec_pos, uint32_t* rec_len, uint32_t* rec_flag, uint32_t* rec_nsnps, abw_sam_counts* counts)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!counts || (len && !text)) return abw_fail(ctx, ABW_ERR_ARG, "abw_sam_parse: bad argument");
    counts->lines = counts->records = counts->kept = 0;
    if (len == 0) return ABW_OK;
    abw_device_guard g(ctx->device);
    hipStream_t st = ctx->stream;
    int rc;
    auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    // scratch[3]: the text | the eight output arrays at capacity
    const size_t b_text = up16(len + 16), b8 = up16(cap * 8), b4 = up16(cap * 4);
    if ((rc = abw_reserve(ctx, ctx->scratch[3], b_text + 2 * b8 + 6 * b4))) return rc;
    uint8_t* p = (uint8_t*)ctx->scratch[3].p;
    uint8_t* d_text = p;                            p += b_text;
    uint64_t* d_rpos = (uint64_t*)p;                p += b8;
    uint64_t* d_cpos = (uint64_t*)p;                p += b8;
    uint32_t* d_rscaf = (uint32_t*)p;               p += b4;
    uint32_t* d_rlen = (uint32_t*)p;                p += b4;
    uint32_t* d_cscaf = (uint32_t*)p;               p += b4;
    uint32_t* d_clen = (uint32_t*)p;                p += b4;
    uint32_t* d_cflag = (uint32_t*)p;               p += b4;
    uint32_t* d_csnps = (uint32_t*)p;
    ABW_HIP(ctx, hipMemcpyAsync(d_text, text, len, hipMemcpyHostToDevice, st));
    rc = abw_sam_parse_dev(ctx, d_text, len, names, name_off, n_scaf, max_snps, cap, read_scaf ? d_rscaf : nullptr, read_pos ? d_rpos : nullptr,
                           read_len ? d_rlen : nullptr, rec_scaf ? d_cscaf : nullptr, rec_pos ? d_cpos : nullptr, rec_len ? d_clen : nullptr,
                           rec_flag ? d_cflag : nullptr, rec_nsnps ? d_csnps : nullptr, counts);
    if (rc) return rc;
    const uint64_t nk = counts->kept, nr = counts->records;
    if (read_scaf && nk) ABW_HIP(ctx, hipMemcpyAsync(read_scaf, d_rscaf, nk * 4, hipMemcpyDeviceToHost, st));
    if (read_pos && nk) ABW_HIP(ctx, hipMemcpyAsync(read_pos, d_rpos, nk * 8, hipMemcpyDeviceToHost, st));
    if (read_len && nk) ABW_HIP(ctx, hipMemcpyAsync(read_len, d_rlen, nk * 4, hipMemcpyDeviceToHost, st));
    if (rec_scaf && nr) ABW_HIP(ctx, hipMemcpyAsync(rec_scaf, d_cscaf, nr * 4, hipMemcpyDeviceToHost, st));
    if (rec_pos && nr) ABW_HIP(ctx, hipMemcpyAsync(rec_pos, d_cpos, nr * 8, hipMemcpyDeviceToHost, st));
    if (rec_len && nr) ABW_HIP(ctx, hipMemcpyAsync(rec_len, d_clen, nr * 4, hipMemcpyDeviceToHost, st));
    if (rec_flag && nr) ABW_HIP(ctx, hipMemcpyAsync(rec_flag, d_cflag, nr * 4, hipMemcpyDeviceToHost, st));
    if (rec_nsnps && nr) ABW_HIP(ctx, hipMemcpyAsync(rec_nsnps, d_csnps, nr * 4, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    return ABW_OK;
}
