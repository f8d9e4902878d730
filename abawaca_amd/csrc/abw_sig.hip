// Stage 1 — signature build: segmentation of scaffolds into ~window non-N bp segments and the
// canonical 1..4-mer signature of every segment.
//
// Reference behaviour restated here (reference = CK7/abawaca):
//   segmentation   src/abawaca-build.cpp:198-228  (Scaf::Scaf)
//   counting       src/abawaca-build.cpp:103-174  (Scaf_segment::Scaf_segment)
//   column order   src/abawaca-build.cpp:75-100   (prepare_mer_indexes)
//   upper-casing   src/String.cpp:37-51           (DNAString::validate_string)
//   truncation     src/abawaca-build.cpp:603      (int(1000.0*x))
//
// Kernel design (gfx950): one wave64 per segment.  Each lane takes aligned 16-byte pieces of the
// segment (one coalesced 1 KiB read per wave-instruction) plus a 3-base halo, packs them into a
// 2-bit code stream and a validity bit stream (valid = ACGT and inside the segment), and counts every
// valid base ONCE in a wave-private 340-counter LDS table, by the length of the valid run it ends:
//   run >= 4 -> 4-mer table[w]           run == 3 -> head 3-mer table[w & 63]
//   run == 2 -> head 2-mer table[w & 15] run == 1 -> head 1-mer table[w & 3]
// The 4-mer table takes one LDS atomic per base (3 vector instructions: window, address, 0/1 addend);
// heads exist only at the start of a run (segment start, after an N) and go through a rare extra pass.
// The k = 3, 2, 1 counts the reference obtains by three more passes then follow exactly from suffix
// marginals: cnt3[s] = head3[s] + sum_x cnt4[x s], and so on down (a k-mer window that is not the
// suffix of a valid (k+1)-mer window is precisely a "head" at the start of a valid run).
// Finalisation divides in fp64 per mer and adds the two strands' quotients, like the reference.
#include "abw_internal.h"
#include <string.h>
#include <algorithm>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>

namespace {

// "hw code" of a base: (toupper(c) >> 1) & 3  ->  A=0 C=1 T=2 G=3
__host__ __device__ inline int hw_of_acgt(int d) { return d == 2 ? 3 : d == 3 ? 2 : d; }   // A,C,G,T=0..3 -> hw

// Wave-private counter table.  A window whose valid run has length r (1..4, r = 4: a full 4-mer) and
// whose last r bases have code w & m (m = 4^r - 1) counts at word (w & m) + 2*(255 - m):
//   T4 [0,256)   T3 [384,448)   T2 [480,496)   T1 [504,508)   word 508: always zero
constexpr int LDS_T4 = 0, LDS_T3 = 384, LDS_T2 = 480, LDS_T1 = 504, LDS_ZERO = 508, LDS_WORDS = 512;

struct SigTables {
    uint16_t dim_a[ABW_SIG_NDIMS];      // LDS word index of the first member mer of each dimension
    uint16_t dim_b[ABW_SIG_NDIMS];      // second member (reverse complement) or 0xFFFF for palindromes
    uint16_t cnt_perm[ABW_SIG_NMERS];   // oracle/ABI mer index -> LDS word index
};

__constant__ SigTables c_tabs;

int lds_index(int k, int mer_acgt)
{
    int hw = 0;
    for (int i = k - 1; i >= 0; i--) hw = (hw << 2) | hw_of_acgt((mer_acgt >> (2 * i)) & 3);
    const int base[5] = {0, LDS_T1, LDS_T2, LDS_T3, LDS_T4};
    return base[k] + hw;
}

void build_tables(SigTables& t, std::vector<std::string>* names)
{
    // ref: prepare_mer_indexes_recursive, src/abawaca-build.cpp:75-94 — ids in enumeration order
    // (k = 1..4, A<C<G<T) to whichever of {mer, reverse complement} shows up first.
    int next_id = 0, off = 0;
    for (int i = 0; i < ABW_SIG_NDIMS; i++) { t.dim_a[i] = 0xFFFF; t.dim_b[i] = 0xFFFF; }
    for (int k = 1; k <= 4; k++) {
        int n = 1 << (2 * k);
        std::vector<int> id(n, -1);
        for (int m = 0; m < n; m++) {
            int rc = 0;
            for (int i = 0; i < k; i++) rc = (rc << 2) | (3 - ((m >> (2 * i)) & 3));
            if (id[rc] >= 0) {
                id[m] = id[rc];
                t.dim_b[id[m]] = (uint16_t)lds_index(k, m);
            } else {
                id[m] = next_id++;
                t.dim_a[id[m]] = (uint16_t)lds_index(k, m);
                if (names) {
                    std::string s;
                    for (int i = k - 1; i >= 0; i--) s.push_back("ACGT"[(m >> (2 * i)) & 3]);
                    names->push_back(s);
                }
            }
            t.cnt_perm[off + m] = (uint16_t)lds_index(k, m);
        }
        off += n;
    }
}

// Built once per process, whichever thread asks first (contexts are independent: two host threads may make their first
// call on two contexts at the same moment).
const SigTables& host_tables()
{
    static SigTables t;
    static std::once_flag once;
    std::call_once(once, [] { build_tables(t, nullptr); });
    return t;
}

// ---------------------------------------------------------------------------------------------
// counting kernel
// ---------------------------------------------------------------------------------------------
// Four bases at once (one dword of ASCII, lowest address = oldest base = byte 0).
//   codes: byte i = 2-bit code of base i (A0 C1 T2 G3 = (toupper(c) >> 1) & 3)
//   miss:  byte i = 0x80 iff toupper(c) is NOT one of ACGT (not a valid base), else 0
// Both are turned into bit streams by v_dot4_u32_u8 with power-of-two weights (a 4-way gather).
__device__ __forceinline__ void classify4(uint32_t q, uint32_t& codes, uint32_t& miss)
{
    const uint32_t u = q & 0xDFDFDFDFu;                      // toupper for letters; only 'a'..'z' can map onto 'A'..'Z'
    codes = (u >> 1) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, codes);   // byte i = "ACTG"[code i]
    const uint32_t diff = u ^ expect;
    miss = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;     // exact nonzero-byte detector
}
// oldest base (byte 0) -> highest position
constexpr uint32_t DOT_PACK = 0x01041040u;      // codes  -> c0<<6 | c1<<4 | c2<<2 | c3
constexpr uint32_t DOT_BITS_LO = 0x01020408u;   // match  -> 128 * (b0<<3 | b1<<2 | b2<<1 | b3)
constexpr uint32_t DOT_BITS_HI = 0x10204080u;   // the same, 4 bits higher

// sum over the 64 lanes without touching LDS: butterflies inside each row of 16 (DPP), rows by readlane
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);   // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// double(n) / double(t) for integers, bit-identical to the IEEE division the reference performs
// (ref :171): this is the compiler's own fp64 division sequence (reciprocal + two Newton steps,
// quotient, one correction) with the reciprocal part hoisted -- a lane's dimensions share few
// totals -- and without the range scaling that integer operands below 2^32 never trigger.
struct Recip { double d, r; };
__device__ __forceinline__ Recip recip_of(uint32_t t)
{
    Recip R;
    R.d = (double)(int)t;
    double r = __builtin_amdgcn_rcp(R.d);
    double e = __builtin_fma(-R.d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-R.d, r, 1.0);
    R.r = __builtin_fma(r, e, r);
    return R;
}
__device__ __forceinline__ double quot(uint32_t n, const Recip& R)
{
    const double nn = (double)n;
    const double q = nn * R.r;
    const double e = __builtin_fma(-R.d, q, nn);
    return __builtin_fma(e, R.r, q);
}

// Known-answer aid for recip_of / quot: thread i takes the divisor t = t_lo + i * t_step and every dividend n in [0, t]
// (t <= n_max: all of them; above: the n_max + 1 values floor(j * t / n_max), j = 0..n_max), compares quot(n, recip_of(t))
// with the compiler's own IEEE division bit for bit, and sums the bit patterns of the quotients (mod 2^64) so that the
// host can check them against its own division.
__global__ __launch_bounds__(256) void div_selftest_kernel(uint32_t t_lo, uint32_t t_step, uint32_t count, uint32_t n_max,
                                                          unsigned long long* __restrict__ mismatches, unsigned long long* __restrict__ checksum)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t t = t_lo + i * t_step;
    const Recip R = recip_of(t);
    const double td = (double)t;
    unsigned long long bad = 0, sum = 0;
    const uint32_t steps = t <= n_max ? t : n_max;
    for (uint32_t j = 0; j <= steps; j++) {
        const uint32_t n = t <= n_max ? j : (uint32_t)((unsigned long long)j * t / n_max);
        const double q = quot(n, R);
        volatile double nd = (double)n;                       // keeps the reference division a plain runtime fp64 division
        const double ref = nd / td;
        bad += __double_as_longlong(q) != __double_as_longlong(ref);
        sum += (unsigned long long)__double_as_longlong(q);
    }
    if (bad) atomicAdd(mismatches, bad);
    if (checksum) checksum[i] = sum;
}

__global__ __launch_bounds__(256) void sig_count_kernel(
    const uint8_t* __restrict__ ascii, uint64_t total_len, const uint64_t* __restrict__ scaf_off,
    const uint32_t* __restrict__ seg_scaf, const uint64_t* __restrict__ seg_start,
    const uint64_t* __restrict__ seg_end, uint64_t n_seg,
    double* __restrict__ dims, int32_t* __restrict__ milli, uint32_t* __restrict__ counts,
    const uint64_t* __restrict__ n_seg_dev, uint32_t scaf_base)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][LDS_WORDS];
    // n_seg_dev: the count is still on the device (pipelined build: no read-back between segmentation and counting);
    // n_seg is then the capacity of the tables.  scaf_base: seg_scaf holds scaffold numbers of the whole job, the blob
    // and scaf_off are one chunk of it that starts at scaffold scaf_base.
    if (n_seg_dev) n_seg = min(n_seg, *n_seg_dev);
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: segment bookkeeping stays scalar
    uint32_t* tab = lds[wid];
    const uint64_t waves_total = (uint64_t)gridDim.x * 4;
    // this lane's (up to) three dimensions: table words of the two strands and which total divides them
    uint32_t dim_ia[3], dim_ib[3], dim_k[3];
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {
        const int dim = min(lane + 64 * rep, ABW_SIG_NDIMS - 1);
        dim_ia[rep] = c_tabs.dim_a[dim];
        dim_ib[rep] = c_tabs.dim_b[dim] == 0xFFFFu ? (uint32_t)LDS_ZERO : (uint32_t)c_tabs.dim_b[dim];
        dim_k[rep] = dim_ia[rep] >= LDS_T1 ? 0 : dim_ia[rep] >= LDS_T2 ? 1 : dim_ia[rep] >= LDS_T3 ? 2 : 3;
    }

    for (uint64_t s = (uint64_t)blockIdx.x * 4 + wid; s < n_seg; s += waves_total) {
        *reinterpret_cast<uint4*>(tab + 4 * lane) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(tab + 256 + 4 * lane) = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const uint64_t sb = scaf_off[seg_scaf[s] - scaf_base];
        const uint64_t seg_b = sb + seg_start[s] - 1;           // absolute, inclusive
        const uint64_t seg_e = sb + seg_end[s];                 // absolute, exclusive
        const uint64_t a0 = seg_b & ~(uint64_t)15;

        // the four bytes in front of a lane's piece (its 3-base halo) are the last word of the lane below's piece: they come
        // over by a wave shift, and lane 0 gets the last word of lane 63's piece of the step before (one scalar load per
        // segment for the first step) -- not a second load and its 64-bit address per lane and step
        uint32_t prev_tail = (a0 >= 4) ? *reinterpret_cast<const uint32_t*>(ascii + a0 - 4) : 0u;
        for (uint64_t base = a0; base < seg_e; base += 1024) {
            const uint64_t A = base + (uint64_t)lane * 16;
            uint32_t q3 = 0;
            // segment bounds relative to window 0 of lane 0 (base - 3), clamped so that 32-bit lane math is exact
            const int rel_b = (int)max((int64_t)-(1 << 20), min((int64_t)(1 << 20), (int64_t)(seg_b - base) + 3));
            const int rel_e = (int)max((int64_t)-(1 << 20), min((int64_t)(1 << 20), (int64_t)(seg_e - base) + 3));
            if (A < seg_e) {
                uint32_t q[4];
                if (A + 16 <= total_len) {
                    const uint4 v = *reinterpret_cast<const uint4*>(ascii + A);
                    q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
                } else {
                    q[0] = q[1] = q[2] = q[3] = 0;
                    for (uint64_t i = 0; A + i < total_len; i++) q[i >> 2] |= (uint32_t)ascii[A + i] << (8 * (i & 3));
                }
                q3 = q[3];
                const uint32_t halo = (uint32_t)__builtin_amdgcn_update_dpp((int)prev_tail, (int)q3, 0x138, 0xF, 0xF, false);   // wave_shr:1
                // Window t = 0..18 is the base at A - 3 + t (3-base halo, then the 16 bases of this
                // piece).  Code stream S (2 bits per base) and validity stream V (1 bit per base),
                // both with the newest base in the lowest bits: base t sits at bit 18 - t of V.
                uint32_t ch, zh, c0, z0, c1, z1, c2, z2, c3, z3;
                classify4(halo, ch, zh);
                classify4(q[0], c0, z0);
                classify4(q[1], c1, z1);
                classify4(q[2], c2, z2);
                classify4(q[3], c3, z3);
                const uint32_t S_lo = (__builtin_amdgcn_udot4(c0, DOT_PACK, 0u, false) << 24) | (__builtin_amdgcn_udot4(c1, DOT_PACK, 0u, false) << 16) |
                                      (__builtin_amdgcn_udot4(c2, DOT_PACK, 0u, false) << 8) | __builtin_amdgcn_udot4(c3, DOT_PACK, 0u, false);
                const uint32_t S_hi = __builtin_amdgcn_udot4(ch, DOT_PACK & 0xFFFFFF00u, 0u, false);     // the halo's byte 0 is not a halo base
                const uint32_t v_lo = __builtin_amdgcn_udot4(z3, DOT_BITS_LO, __builtin_amdgcn_udot4(z2, DOT_BITS_HI, 0u, false), false);
                const uint32_t v_mi = __builtin_amdgcn_udot4(z1, DOT_BITS_LO, __builtin_amdgcn_udot4(z0, DOT_BITS_HI, 0u, false), false);
                const uint32_t v_hi = __builtin_amdgcn_udot4(zh, DOT_BITS_LO & 0xFFFFFF00u, 0u, false);
                const uint32_t Vmiss = ((v_hi << 16) + (v_mi << 8) + v_lo) >> 7;   // bit 18 - t: base t is not a valid base
                // only bases inside [seg_b, seg_e) count, as window members too (ref :105-173 works on
                // the segment's own substring): t in [tl, th)
                const int tl = min(max(rel_b - 16 * lane, 0), 19);
                const int th = min(max(rel_e - 16 * lane, 0), 19);
                const uint32_t V = ~Vmiss & ((1u << (19 - tl)) - 1u) & ~((1u << (19 - th)) - 1u);
                // full 4-mer windows: base (bit sh) and the three before it (bits sh+1..sh+3) all valid
                const uint32_t M1 = V & (V >> 1), M2 = M1 & (V >> 2), M3 = M2 & (V >> 3);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int sh = 15 - i;                                       // bit of base i in V
                    const uint32_t w = (sh <= 12) ? __builtin_amdgcn_ubfe(S_lo, 2 * sh, 8)
                                                  : (__builtin_amdgcn_alignbit(S_hi, S_lo, 2 * sh) & 0xFFu);
                    atomicAdd(tab + LDS_T4 + w, __builtin_amdgcn_ubfe(M3, sh, 1));   // adds 0 unless the window is a whole 4-mer
                }
                // heads: valid bases whose run is shorter than 4 (the first three of a run: after the segment's start, an N
                // or any other letter) -- rare.  H1/H2/H3: run of exactly 1/2/3; word (w & m) + 2*(255 - m), m = 4^r - 1.
                // A run start puts one bit into each of the three, so one pass takes one head of each kind; more passes
                // only for a lane with several run starts in its 16 bases.
                uint32_t H1 = V & ~M1 & 0xFFFFu, H2 = M1 & ~M2 & 0xFFFFu, H3 = M2 & ~M3 & 0xFFFFu;
                if (__builtin_amdgcn_ballot_w64((H1 | H2 | H3) != 0u) != 0ull) {
                    const uint64_t S64 = ((uint64_t)S_hi << 32) | S_lo;
                    do {
                        if (H1) { const int sh = __builtin_ctz(H1); H1 &= H1 - 1u; atomicAdd(tab + LDS_T1 + ((uint32_t)(S64 >> (2 * sh)) & 3u), 1u); }
                        if (H2) { const int sh = __builtin_ctz(H2); H2 &= H2 - 1u; atomicAdd(tab + LDS_T2 + ((uint32_t)(S64 >> (2 * sh)) & 15u), 1u); }
                        if (H3) { const int sh = __builtin_ctz(H3); H3 &= H3 - 1u; atomicAdd(tab + LDS_T3 + ((uint32_t)(S64 >> (2 * sh)) & 63u), 1u); }
                    } while (H1 | H2 | H3);
                }
            }
            prev_tail = (uint32_t)__builtin_amdgcn_readlane((int)q3, 63);   // (a next step exists only if lane 63 had a piece)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // suffix marginals, in place: T3 += sum_x T4[x|s], T2 += sum_x T3[x|s], T1 += sum_x T2[x|s].
        // Branch-free: every lane computes, lanes outside a table write the always-zero word back.
        const uint32_t h3 = tab[LDS_T3 + lane];
        const uint32_t t4 = tab[LDS_T4 + lane] + tab[LDS_T4 + 64 + lane] + tab[LDS_T4 + 128 + lane] + tab[LDS_T4 + 192 + lane];
        const uint32_t c3 = h3 + t4;
        tab[LDS_T3 + lane] = c3;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int l16 = lane & 15, l4 = lane & 3;
        const uint32_t h2 = (lane < 16) ? tab[LDS_T2 + l16] : 0u;
        const uint32_t c2 = (lane < 16) ? h2 + tab[LDS_T3 + l16] + tab[LDS_T3 + 16 + l16] + tab[LDS_T3 + 32 + l16] + tab[LDS_T3 + 48 + l16] : 0u;
        tab[(lane < 16) ? LDS_T2 + lane : LDS_ZERO] = c2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t h1 = (lane < 4) ? tab[LDS_T1 + l4] : 0u;
        const uint32_t c1 = (lane < 4) ? h1 + tab[LDS_T2 + l4] + tab[LDS_T2 + 4 + l4] + tab[LDS_T2 + 8 + l4] + tab[LDS_T2 + 12 + l4] : 0u;
        tab[(lane < 4) ? LDS_T1 + lane : LDS_ZERO] = c1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // per-k totals (number of valid windows; ref: `total`, src/abawaca-build.cpp:120,136,153):
        // tot4 = sum T4, tot3 = tot4 + (3-base heads), ...  Two packed reductions while every count
        // fits 16 bits, i.e. for any segment shorter than 65536 bases.
        uint32_t tot[4];
        if (seg_e - seg_b < 65536) {
            const uint32_t a = wave_sum(t4 | (h3 << 16)), b = wave_sum(h2 | (h1 << 16));
            tot[3] = a & 0xFFFFu;
            tot[2] = tot[3] + (a >> 16);
            tot[1] = tot[2] + (b & 0xFFFFu);
            tot[0] = tot[1] + (b >> 16);
        } else {
            tot[3] = wave_sum(t4);
            tot[2] = wave_sum(c3);
            tot[1] = wave_sum(c2);
            tot[0] = wave_sum(c1);
        }

        // ref :171 — dimensions[idx] += double(count)/total, once per strand.  A palindrome's second
        // strand reads the always-zero word; an empty total divides zeros by one: both give +0.0.
        const Recip R4 = recip_of(max(tot[3], 1u));
#pragma unroll
        for (int rep = 0; rep < 3; rep++) {
            const int dim = lane + 64 * rep;
            Recip R = R4;
            if (rep == 0) R = recip_of(max(dim_k[0] == 0 ? tot[0] : dim_k[0] == 1 ? tot[1] : dim_k[0] == 2 ? tot[2] : tot[3], 1u));
            const double x = quot(tab[dim_ia[rep]], R) + quot(tab[dim_ib[rep]], R);
            if (rep < 2 || dim < ABW_SIG_NDIMS) {
                if (dims) dims[s * ABW_SIG_NDIMS + dim] = x;
                if (milli && dim > 0) milli[s * ABW_SIG_NCOLS + dim - 1] = (int)(1000.0 * x);   // ref :603
            }
        }
        if (counts) {
            for (int i = lane; i < ABW_SIG_NMERS; i += 64) counts[s * ABW_SIG_NMERS + i] = tab[c_tabs.cnt_perm[i]];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// segmentation on the device
//   pass 1: non-'N' count of every 1 KiB block of the blob + lower-case 'n' detection
//   pass 2: exclusive scan of the block counts (single workgroup, decoupled chunks)
//   pass 3: per scaffold: nonN, nseg, nbps, number of segments produced
//   pass 4: exclusive scan over scaffolds -> segment offsets
//   pass 5: one thread per segment: locate its last non-N base by rank (block prefix sums)
// ---------------------------------------------------------------------------------------------
constexpr int SEGBLK = 1024;

// 0x80 in every byte of x that equals the byte replicated in pat (exact)
__device__ __forceinline__ uint32_t bytes_eq(uint32_t x, uint32_t pat)
{
    const uint32_t t = x ^ pat;
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}

#ifndef SEGU
#define SEGU 8
#endif
__global__ __launch_bounds__(256) void seg_blockcount_kernel(const uint8_t* __restrict__ ascii, uint64_t total_len,
                                                            uint32_t* __restrict__ blk_nonN, uint64_t nblk,
                                                            int* __restrict__ flag_lower_n)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t any_n = 0;
    // SEGU blocks per turn: that many 16-byte loads in flight per lane before the first reduction (one at a time: 4.6 TB/s).
    // The pass is close to vector-instruction-bound beside the reads: the 'N's are counted (one population count per
    // word), a lower-case 'n' is only looked for (its detector words are ORed, one test at the end), and the four blocks'
    // counts travel through two wave reductions, two 16-bit counts in a word.
    for (uint64_t b0 = SEGU * wave; b0 < nblk; b0 += SEGU * nwaves) {
        uint4 v[SEGU];
        bool whole[SEGU];
#pragma unroll
        for (int u = 0; u < SEGU; u++) {
            const uint64_t b = b0 + (uint64_t)u, A = b * SEGBLK + (uint64_t)lane * 16;
            whole[u] = b < nblk && A + 16 <= total_len;
            typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));
            if (whole[u]) {
                const u32x4n t = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(ascii + A));
                v[u] = make_uint4(t[0], t[1], t[2], t[3]);
            } else {
                v[u] = make_uint4(0, 0, 0, 0);
            }
        }
        uint32_t nonN[SEGU];
#pragma unroll
        for (int u = 0; u < SEGU; u++) {
            const uint64_t b = b0 + (uint64_t)u, A = b * SEGBLK + (uint64_t)lane * 16;
            uint32_t nN = 0, len = 0;
            if (whole[u]) {
                nN = __popc(bytes_eq(v[u].x, 0x4E4E4E4Eu)) + __popc(bytes_eq(v[u].y, 0x4E4E4E4Eu)) + __popc(bytes_eq(v[u].z, 0x4E4E4E4Eu)) +
                     __popc(bytes_eq(v[u].w, 0x4E4E4E4Eu));
                any_n |= bytes_eq(v[u].x, 0x6E6E6E6Eu) | bytes_eq(v[u].y, 0x6E6E6E6Eu) | bytes_eq(v[u].z, 0x6E6E6E6Eu) | bytes_eq(v[u].w, 0x6E6E6E6Eu);
                len = 16;
            } else if (b < nblk) {
                for (uint64_t i = A; i < total_len; i++) { nN += ascii[i] == 'N'; any_n |= ascii[i] == 'n'; len++; }
            }
            nonN[u] = len - nN;                                 // <= 16 per lane, <= 1024 per block: 16 bits hold a block's sum
        }
#pragma unroll
        for (int u = 0; u < SEGU; u += 2) {
            const uint32_t s2 = wave_sum(nonN[u] | (nonN[u + 1] << 16));
            if (lane == 0) {
                if (b0 + u < nblk) blk_nonN[b0 + u] = s2 & 0xFFFFu;
                if (b0 + u + 1 < nblk) blk_nonN[b0 + u + 1] = s2 >> 16;
            }
        }
    }
    if (any_n) atomicOr(flag_lower_n, 1);
}

// exclusive scan -> uint64: scan_lookback_kernel (below, with device_scan_tmp).  out has n + 1 entries.

// rank (number of non-N bytes) of blob prefix [0, pos): whole blocks from blk_prefix + a partial block
__device__ __forceinline__ uint64_t prefix_nonN(const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ blk_prefix,
                                                uint64_t pos, int lane)
{
    const uint64_t b = pos / SEGBLK;
    const uint64_t before = blk_prefix[b];
    if (pos == b * SEGBLK) return before;
    if (blk_prefix[b + 1] - before == SEGBLK) return before + (pos - b * SEGBLK);   // a block without 'N': nothing to read
    const uint64_t A = b * SEGBLK + (uint64_t)lane * 16;
    uint32_t c = 0;
    for (uint64_t i = A; i < A + 16 && i < pos; i++) c += ascii[i] != 'N';
    return before + wave_sum(c);
}

struct ScafSeg { uint64_t rank0; uint64_t nbps; uint64_t nseg; };

__global__ __launch_bounds__(256) void seg_scaf_kernel(const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ scaf_off,
                                                      uint64_t n_scaf, const uint64_t* __restrict__ blk_prefix, uint32_t window,
                                                      ScafSeg* __restrict__ info, uint64_t* __restrict__ nseg_out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t s = wave; s < n_scaf; s += nwaves) {
        const uint64_t b = scaf_off[s], e = scaf_off[s + 1];
        const uint64_t r0 = prefix_nonN(ascii, blk_prefix, b, lane);
        const uint64_t r1 = prefix_nonN(ascii, blk_prefix, e, lane);
        const uint64_t nonN = r1 - r0;
        uint64_t nsegments = nonN / window;                      // ref :202
        if (nsegments == 0) nsegments = 1;                       // ref :204-205
        const uint64_t nbps = nonN / nsegments;                  // ref :206
        // ref :209-225 — a cut every nbps non-N bases; nbps == 0 cuts after every base
        const uint64_t produced = (nbps == 0) ? (e - b) : nonN / nbps;
        if (lane == 0) { info[s].rank0 = r0; info[s].nbps = nbps; info[s].nseg = produced; nseg_out[s] = produced; }
    }
}

// inclusive prefix sum over the 64 lanes, DPP only (row shifts, then the two row broadcasts of gfx9)
__device__ __forceinline__ uint32_t wave_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);    // row_shr:1, zero fill
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

// bit i (i = 0..3) set iff byte i of q is not 'N'
__device__ __forceinline__ uint32_t nonN4(uint32_t q)
{
    const uint32_t t = q ^ 0x4E4E4E4Eu;
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);      // 0x80 where the byte is 'N'
    return (__builtin_amdgcn_udot4(z, 0x08040201u, 0u, false) >> 7) ^ 0xFu;
}

// index of the j-th (0-based) set bit of a 16-bit mask that has more than j bits set
__device__ __forceinline__ int nth_set_bit16(uint32_t m, uint32_t j)
{
    int pos = 0;
    uint32_t c = __popc(m & 0xFFu);
    if (j >= c) { j -= c; pos = 8; m >>= 8; }
    c = __popc(m & 0xFu);
    if (j >= c) { j -= c; pos += 4; m >>= 4; }
    c = __popc(m & 0x3u);
    if (j >= c) { j -= c; pos += 2; m >>= 2; }
    if (j >= (m & 1u)) pos += 1;
    return pos;
}

// One thread per segment.  Segment j (0-based) of scaffold s is closed by the base whose 1-based rank among
// the scaffold's non-N bases is (j+1)*nbps (ref :212-224), i.e. by the byte at which the count of non-N bytes
// of the whole blob reaches R = rank0 + (j+1)*nbps.  Its 1 KiB block b is the one with
// blk_prefix[b] < R <= blk_prefix[b+1]: the block the byte would sit in if the scaffold had no N so far is a
// lower bound and almost always the answer, otherwise a binary search over the scaffold's blocks.  Inside a
// block without N the position is arithmetic; a block with N is read once, by the whole wave, for each
// segment that ends in it (ballot loop).  Nothing else of the sequence is read again.
__global__ __launch_bounds__(256) void seg_emit_kernel(const uint8_t* __restrict__ ascii, uint64_t total_len,
                                                      const uint64_t* __restrict__ scaf_off, uint64_t n_scaf,
                                                      const uint64_t* __restrict__ blk_prefix,
                                                      const ScafSeg* __restrict__ info, const uint64_t* __restrict__ seg_off,
                                                      uint64_t n_seg, uint32_t* __restrict__ seg_scaf,
                                                      uint64_t* __restrict__ seg_start, uint64_t* __restrict__ seg_end,
                                                      uint32_t scaf_base)
{
    const int lane = threadIdx.x & 63;
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    // n_seg = the launch's capacity; the count itself is on the device (seg_off[n_scaf]) -- a caller that read it back
    // passes it as the capacity, one that did not (abw_sig_segment_static) launches over its whole capacity
    n_seg = min(n_seg, seg_off[n_scaf]);
    const bool active = g < n_seg;
    uint64_t s = 0, j = 0, sb = 0, nseg = 0, pos = 0;     // pos: blob offset of the closing byte
    bool need = false;
    uint64_t need_b = 0; uint32_t need_k = 0;
    if (active) {
        // scaffold of segment g: the s with seg_off[s] <= g < seg_off[s+1] (scaffolds without segments are skipped)
        uint64_t lo = 0, hi = n_scaf;
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (seg_off[mid] <= g) lo = mid; else hi = mid; }
        s = lo; j = g - seg_off[s];
        const ScafSeg si = info[s];
        sb = scaf_off[s]; nseg = si.nseg;
        if (si.nbps == 0) pos = sb + j;                    // all-N scaffold: every base is a segment (ref :206 nbps = 0)
        else {
            const uint64_t k = (j + 1) * si.nbps;          // rank of the closing base inside the scaffold
            const uint64_t R = si.rank0 + k;
            uint64_t b = (sb + k - 1) / SEGBLK;
            if (blk_prefix[b + 1] < R) {
                uint64_t l = b + 1, h = (scaf_off[s + 1] - 1) / SEGBLK;          // answer in [l, h]
                while (l < h) { const uint64_t mid = (l + h) >> 1; if (blk_prefix[mid + 1] >= R) h = mid; else l = mid + 1; }
                b = l;
            }
            const uint64_t before = blk_prefix[b];
            if (blk_prefix[b + 1] - before == SEGBLK) pos = b * SEGBLK + (R - before) - 1;
            else { need = true; need_b = b; need_k = (uint32_t)(R - before); }
        }
    }
    unsigned long long todo = __ballot(need);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint64_t bb = __shfl(need_b, src, 64);
        const uint32_t kk = __shfl(need_k, src, 64);       // 1-based rank inside block bb
        const uint64_t A = bb * SEGBLK + (uint64_t)lane * 16;
        uint32_t nmask = 0;
        if (A + 16 <= total_len) {
            const uint4 v = *reinterpret_cast<const uint4*>(ascii + A);
            nmask = nonN4(v.x) | (nonN4(v.y) << 4) | (nonN4(v.z) << 8) | (nonN4(v.w) << 12);
        } else {
            for (uint64_t i = A; i < total_len && i < A + 16; i++) nmask |= (uint32_t)(ascii[i] != 'N') << (i - A);
        }
        const uint32_t c = __popc(nmask);
        const uint32_t x = wave_scan(c);
        const bool mine = (x - c < kk) && (kk <= x);
        const uint64_t found = mine ? A + (uint64_t)nth_set_bit16(nmask, kk - (x - c) - 1) : 0;
        const unsigned long long who = __ballot(mine);
        const uint64_t p = __shfl(found, who ? __ffsll((long long)who) - 1 : 0, 64);
        if (lane == src) pos = p;
    }
    if (active) {
        const uint64_t rel = pos - sb;                     // 0-based inside the scaffold
        seg_scaf[g] = (uint32_t)s + scaf_base;
        seg_end[g] = rel + 1;
        if (j == 0) seg_start[g] = 1;                      // ref :209 sit = seq.begin()
        if (j + 1 < nseg && g + 1 < n_seg) seg_start[g + 1] = rel + 2;      // ref :221-222 sit = it + 1
    }
}

// Round 3: ONE pass (plus the memset of its look-back words) instead of three launches.  A workgroup takes the next 4096
// elements by ticket, scans them (a thread owns four consecutive ones), publishes their sum and finds what lies in front by a
// decoupled look-back over the workgroups with lower tickets (state word: bit 62 "my sum", bit 63 "everything up to and
// including me", the value below; a wave reads 64 predecessors per step).  A lower ticket has started and publishes its
// sum without waiting for anyone, so every wait ends.  Sums stay below 2^62 (they count bytes, segments, records).
constexpr uint64_t SCAN_SUM = 1ull << 62, SCAN_ALL = 1ull << 63, SCAN_VAL = (1ull << 62) - 1ull;
constexpr int SCAN_ITEMS = 4, SCAN_TILE = 1024 * SCAN_ITEMS;

template <typename TIN>
__global__ __launch_bounds__(1024) void scan_lookback_kernel(const TIN* __restrict__ in, uint64_t n, uint64_t* __restrict__ out,
                                                            unsigned long long* __restrict__ tmp)
{
    __shared__ unsigned long long s_bid, s_excl;
    __shared__ uint64_t wsum[16];
    unsigned long long* state = tmp + 2;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_bid = atomicAdd(tmp, 1ull);
    __syncthreads();
    const uint64_t bid = s_bid;
    const uint64_t i0 = bid * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    if (sizeof(TIN) == 4 && i0 + SCAN_ITEMS <= n && ((uintptr_t)in & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4*>(in + i0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) v[k] = (i0 + k < n) ? (uint64_t)in[i0 + k] : 0ull;
    }
    const uint64_t mine = v[0] + v[1] + v[2] + v[3];
    uint64_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    if (wid == 0) {
        uint64_t sum = lane < 16 ? wsum[lane] : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        uint64_t excl = 0;
        if (bid > 0) {
            if (lane == 0) __hip_atomic_store(state + bid, SCAN_SUM | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t top = (int64_t)bid - 1;                  // the nearest workgroup not yet added
            for (;;) {
                const int64_t idx = top - lane;
                const uint64_t w = idx >= 0 ? __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SCAN_ALL;
                const uint64_t have_all = __ballot((w & SCAN_ALL) != 0), nothing = __ballot((w >> 62) == 0);
                const int stop = have_all ? __builtin_ctzll(have_all) : 64;        // the nearest that knows everything below it
                const uint64_t below = stop >= 64 ? ~0ull : (2ull << stop) - 1ull;
                if (nothing & below) { __builtin_amdgcn_s_sleep(2); continue; }    // one of them has not published yet
                uint64_t t = lane <= stop ? (w & SCAN_VAL) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                excl += t;
                if (stop < 64) break;
                top -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(state + bid, SCAN_ALL | (excl + sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
            if (bid == (n + SCAN_TILE - 1) / SCAN_TILE - 1) out[n] = excl + sum;
        }
    }
    __syncthreads();
    uint64_t before = s_excl + x - mine;                     // everything in front of this thread's four
    for (int k = 0; k < wid; k++) before += wsum[k];
    if (i0 + SCAN_ITEMS <= n && ((uintptr_t)out & 15) == 0) {
        ulonglong2 a, b2;
        a.x = before; a.y = before + v[0]; b2.x = a.y + v[1]; b2.y = b2.x + v[2];
        reinterpret_cast<ulonglong2*>(out + i0)[0] = a;
        reinterpret_cast<ulonglong2*>(out + i0)[1] = b2;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { if (i0 + k < n) out[i0 + k] = before; before += v[k]; }
    }
}

// tmp: (2 * ceil(n / 1024) + 2) uint64 (the size the two-level scan of rounds 1-2 needed; this one uses ceil(n / 4096) + 2)
template <typename TIN>
int device_scan_tmp(abw_ctx* ctx, const TIN* in, uint64_t n, uint64_t* out, uint64_t* tmp)
{
    if (n == 0) { ABW_HIP(ctx, hipMemsetAsync(out, 0, 8, ctx->stream)); return ABW_OK; }
    const uint64_t nblk = (n + SCAN_TILE - 1) / SCAN_TILE;
    ABW_HIP(ctx, hipMemsetAsync(tmp, 0, (nblk + 2) * 8, ctx->stream));
    hipLaunchKernelGGL(scan_lookback_kernel<TIN>, dim3((unsigned)nblk), dim3(1024), 0, ctx->stream, in, n, out, (unsigned long long*)tmp);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

// scratch[8] is used for the temporary
template <typename TIN>
int device_scan(abw_ctx* ctx, const TIN* in, uint64_t n, uint64_t* out)
{
    const uint64_t ngroups = (n + 1023) / 1024;
    int rc = abw_reserve(ctx, ctx->scratch[8], (2 * ngroups + 2) * 8);
    if (rc) return rc;
    return device_scan_tmp<TIN>(ctx, in, n, out, (uint64_t*)ctx->scratch[8].p);
}

// The arrays the segmentation works in, wherever they live (grow-only buffers of the context, or the static workspace).
struct SegWork {
    uint32_t* blk_nonN;      // [nblk]
    uint64_t* blk_prefix;    // [nblk + 1]
    ScafSeg*  info;          // [n_scaf]
    uint64_t* nseg;          // [n_scaf]
    uint64_t* seg_off;       // [n_scaf + 1], then the lower-case-n flag (8 bytes)
    uint64_t* scan_tmp;      // [2 * ceil(max(nblk, n_scaf) / 1024) + 2]
};

// passes 1-4: block counts, their scan, per-scaffold segment counts, their scan.  Leaves seg_off[n_scaf] = number of
// segments and, behind it, the lower-case-n flag.  Asynchronous.
int segment_counts(abw_ctx* ctx, const SegWork& w, const uint8_t* d_ascii, uint64_t total_len, const uint64_t* d_scaf_off,
                   uint64_t n_scaf, uint32_t window)
{
    const uint64_t nblk = total_len / SEGBLK + 1;
    hipStream_t st = ctx->stream;
    int* flag = (int*)(w.seg_off + n_scaf + 1);
    ABW_HIP(ctx, hipMemsetAsync(flag, 0, 8, st));
    const int grid = ctx->num_cu * 8;
    int rc;
    hipLaunchKernelGGL(seg_blockcount_kernel, dim3(grid), dim3(256), 0, st, d_ascii, total_len, w.blk_nonN, nblk, flag);
    ABW_LAUNCH_CHECK(ctx);
    if ((rc = device_scan_tmp<uint32_t>(ctx, w.blk_nonN, nblk, w.blk_prefix, w.scan_tmp))) return rc;
    hipLaunchKernelGGL(seg_scaf_kernel, dim3(grid), dim3(256), 0, st, d_ascii, d_scaf_off, n_scaf, w.blk_prefix, window, w.info, w.nseg);
    ABW_LAUNCH_CHECK(ctx);
    if ((rc = device_scan_tmp<uint64_t>(ctx, w.nseg, n_scaf, w.seg_off, w.scan_tmp))) return rc;
    return ABW_OK;
}

// ---- abw_sig_segment_static: the same work without any allocation.  Its arrays live in `scratch`, a file-scope device
// array sized for ABW_SIG_STATIC_MAX_LEN bytes of sequence and ABW_SIG_STATIC_MAX_SCAF scaffolds per call (about 11 MB,
// held by every GPU the library runs on until the process ends).  Host code never uses the array's name as an address:
// abw_sig_static_init resolves this device's copy once (hipGetSymbolAddress) and the context keeps the pointer.
constexpr uint64_t align256(uint64_t x) { return (x + 255) & ~(uint64_t)255; }
constexpr uint64_t STATIC_NBLK = ABW_SIG_STATIC_MAX_LEN / SEGBLK + 1;
constexpr uint64_t STATIC_SCAN_N = (STATIC_NBLK > ABW_SIG_STATIC_MAX_SCAF ? STATIC_NBLK : ABW_SIG_STATIC_MAX_SCAF);
constexpr uint64_t STATIC_OFF_BLK = 0;
constexpr uint64_t STATIC_OFF_PREFIX = STATIC_OFF_BLK + align256(STATIC_NBLK * 4);
constexpr uint64_t STATIC_OFF_INFO = STATIC_OFF_PREFIX + align256((STATIC_NBLK + 1) * 8);
constexpr uint64_t STATIC_OFF_NSEG = STATIC_OFF_INFO + align256(ABW_SIG_STATIC_MAX_SCAF * sizeof(ScafSeg));
constexpr uint64_t STATIC_OFF_SEGOFF = STATIC_OFF_NSEG + align256(ABW_SIG_STATIC_MAX_SCAF * 8);
constexpr uint64_t STATIC_OFF_SCAN = STATIC_OFF_SEGOFF + align256((ABW_SIG_STATIC_MAX_SCAF + 2) * 8);
constexpr uint64_t STATIC_BYTES = STATIC_OFF_SCAN + align256((2 * ((STATIC_SCAN_N + 1023) / 1024) + 2) * 8);
static __device__ __align__(256) unsigned char scratch[STATIC_BYTES];

// One copy of the constant tables per device, uploaded under a lock by the first context that counts on that device.
bool g_tables_uploaded[64] = {};
std::mutex g_tables_mtx;

int upload_tables(abw_ctx* ctx)
{
    std::lock_guard<std::mutex> lock(g_tables_mtx);
    if (ctx->device < 64 && g_tables_uploaded[ctx->device]) return ABW_OK;
    const SigTables& t = host_tables();
    ABW_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_tabs), &t, sizeof(t)));
    if (ctx->device < 64) g_tables_uploaded[ctx->device] = true;
    return ABW_OK;
}

// ---------------------------------------------------------------------------------------------
// abw_sig_build: host buffers in, host buffers out, as a PIPELINE over chunks of whole scaffolds.
//   host worker threads copy chunk i+1 of the caller's (pageable) sequence into a pinned staging buffer
//   | copy stream: H2D of chunk i+1 | compute stream: segmentation + counting of chunk i (no read-back in between: the
//   segment count stays on the device, seg_emit and sig_count run over the chunk's capacity) | copy-back stream: D2H of
//   chunk i-1's tables into pinned memory | host worker threads copy them to the caller's arrays.
// PCIe runs both ways at once and the kernels (1.4 ms per Gbp) hide behind the copies; what bounds the rate is the
// host's copy into / out of pinned memory, hence several threads.  Chunks hold whole scaffolds, so the segmentation is
// exactly the one of the unchunked call.
// ---------------------------------------------------------------------------------------------
constexpr uint64_t PIPE_CHUNK = 48ull << 20;      // bytes of sequence per chunk (a longer scaffold makes a longer chunk)
constexpr int PIPE_SLOTS = 4;

// A small persistent pool of host threads for the staging copies (created on first use, lives until the process ends).
class CopyPool {
public:
    static CopyPool& get() { static CopyPool* p = new CopyPool(); return *p; }
    // memcpy(dst, src, n) cut into slices that the workers run; *pending counts the slices not yet done
    void submit(void* dst, const void* src, size_t n, std::atomic<int>* pending)
    {
        if (n == 0) return;
        const size_t slice = 2u << 20;
        std::vector<std::function<void()>> jobs;
        for (size_t o = 0; o < n; o += slice) {
            const size_t len = std::min(slice, n - o);
            pending->fetch_add(1);
            jobs.push_back([=] { memcpy((char*)dst + o, (const char*)src + o, len); pending->fetch_sub(1); });
        }
        { std::lock_guard<std::mutex> l(m_); for (auto& j : jobs) q_.push_back(std::move(j)); }
        cv_.notify_all();
    }
    // the caller helps with whatever is queued until its own slices are done
    void wait(std::atomic<int>* pending)
    {
        while (pending->load() != 0) {
            std::function<void()> job;
            { std::lock_guard<std::mutex> l(m_); if (!q_.empty()) { job = std::move(q_.front()); q_.pop_front(); } }
            if (job) job(); else std::this_thread::yield();
        }
    }
private:
    CopyPool()
    {
        const unsigned nt = std::max(2u, std::min(8u, std::thread::hardware_concurrency() / 2));
        for (unsigned t = 0; t + 1 < nt; t++) {
            std::thread([this] {
                for (;;) {
                    std::function<void()> job;
                    { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [this] { return !q_.empty(); }); job = std::move(q_.front()); q_.pop_front(); }
                    job();
                }
            }).detach();
        }
    }
    std::deque<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_;
};

}  // namespace

struct PipeSlot {
    uint8_t* pin_in = nullptr;  size_t pin_in_cap = 0;      // sequence of the chunk + its scaffold offsets
    uint8_t* pin_out = nullptr; size_t pin_out_cap = 0;     // count/flag, segment tables, rows
    abw_devbuf d_in, d_out, d_work;
    hipEvent_t e_h2d = nullptr, e_compute = nullptr, e_count = nullptr, e_d2h = nullptr;
    std::atomic<int> out_pending{0};                        // slices of this slot's copy-out still running
    // the chunk in flight
    uint64_t first_scaf = 0, n_scaf = 0, bytes = 0, cap = 0, count = 0, out_base = 0;
    bool used = false, wrote = false;
};

struct abw_sig_pipe {
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    PipeSlot slot[PIPE_SLOTS];
};

namespace {

int pin_reserve(abw_ctx* ctx, uint8_t*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap) return ABW_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    ABW_HIP(ctx, hipHostMalloc((void**)&p, want, hipHostMallocDefault));
    cap = want;
    return ABW_OK;
}

int sig_build_pipeline(abw_ctx* ctx, const uint8_t* ascii, const uint64_t* scaf_off, uint64_t n_scaf, uint32_t window,
                       uint64_t seg_cap, uint32_t* seg_scaf, uint64_t* seg_start, uint64_t* seg_end,
                       double* dims, int32_t* milli, uint64_t* n_seg)
{
    int rc;
    if ((rc = upload_tables(ctx))) return rc;
    if (!ctx->sig_pipe) {
        abw_sig_pipe* P = new abw_sig_pipe();
        ctx->sig_pipe = P;
        ABW_HIP(ctx, hipStreamCreateWithFlags(&P->s_h2d, hipStreamNonBlocking));
        ABW_HIP(ctx, hipStreamCreateWithFlags(&P->s_d2h, hipStreamNonBlocking));
        for (auto& sl : P->slot) {
            ABW_HIP(ctx, hipEventCreateWithFlags(&sl.e_h2d, hipEventDisableTiming));
            ABW_HIP(ctx, hipEventCreateWithFlags(&sl.e_compute, hipEventDisableTiming));
            ABW_HIP(ctx, hipEventCreateWithFlags(&sl.e_count, hipEventDisableTiming));
            ABW_HIP(ctx, hipEventCreateWithFlags(&sl.e_d2h, hipEventDisableTiming));
        }
    }
    abw_sig_pipe& P = *ctx->sig_pipe;
    CopyPool& pool = CopyPool::get();
    hipStream_t st = ctx->stream;
    const size_t row_bytes = (dims ? ABW_SIG_NDIMS * 8ull : 0) + (milli ? ABW_SIG_NCOLS * 4ull : 0);
    for (auto& sl : P.slot) { sl.used = false; sl.wrote = false; }

    // the chunks: whole scaffolds, about PIPE_CHUNK bytes each
    std::vector<uint64_t> cut{0};
    while (cut.back() < n_scaf) {
        const uint64_t s0 = cut.back();
        uint64_t s1 = s0 + 1;
        while (s1 < n_scaf && scaf_off[s1 + 1] - scaf_off[s0] <= PIPE_CHUNK && s1 - s0 < (1u << 22)) s1++;
        cut.push_back(s1);
    }
    const int64_t nchunks = (int64_t)cut.size() - 1;

    uint64_t total_seg = 0;          // segments of the chunks whose count has come back
    bool over = false;               // more segments than seg_cap: keep counting, stop writing
    int flag_n = 0;
    std::atomic<int> in_pending{0};

    // stage A: chunk i into its slot -- staging copy, H2D, kernels; nothing here waits for the GPU
    auto submit = [&](int64_t i) -> int {
        PipeSlot& sl = P.slot[i % PIPE_SLOTS];
        const uint64_t s0 = cut[i], s1 = cut[i + 1];
        const uint64_t bytes = scaf_off[s1] - scaf_off[s0], ns = s1 - s0;
        const uint64_t blob_bytes = ((bytes + 15) & ~15ull) + 16, off_bytes = (ns + 1) * 8;
        const uint64_t cap = bytes / window + ns + 64;
        const uint64_t nblk = bytes / SEGBLK + 1;
        int rc2;
        pool.wait(&sl.out_pending);                               // the slot's previous tables have left its pinned buffer
        if ((rc2 = pin_reserve(ctx, sl.pin_in, sl.pin_in_cap, blob_bytes + off_bytes))) return rc2;
        if ((rc2 = pin_reserve(ctx, sl.pin_out, sl.pin_out_cap, 64 + cap * (4 + 8 + 8 + row_bytes)))) return rc2;
        if ((rc2 = abw_reserve(ctx, sl.d_in, blob_bytes + off_bytes))) return rc2;
        if ((rc2 = abw_reserve(ctx, sl.d_out, 64 + cap * (4 + 8 + 8 + row_bytes)))) return rc2;
        const size_t w_blk = (nblk * 4 + 255) & ~(size_t)255, w_pre = ((nblk + 1) * 8 + 255) & ~(size_t)255, w_info = (ns * sizeof(ScafSeg) + 255) & ~(size_t)255,
                     w_nseg = (ns * 8 + 255) & ~(size_t)255, w_off = ((ns + 2) * 8 + 255) & ~(size_t)255,
                     w_scan = (2 * ((std::max(nblk, ns) + 1023) / 1024) + 2) * 8;
        if ((rc2 = abw_reserve(ctx, sl.d_work, w_blk + w_pre + w_info + w_nseg + w_off + w_scan))) return rc2;
        // sequence + rebased offsets into pinned memory (worker threads + this one), then one H2D
        pool.submit(sl.pin_in, ascii + scaf_off[s0], bytes, &in_pending);
        memset(sl.pin_in + bytes, 0, blob_bytes - bytes);
        uint64_t* h_off = (uint64_t*)(sl.pin_in + blob_bytes);
        for (uint64_t j = 0; j <= ns; j++) h_off[j] = scaf_off[s0 + j] - scaf_off[s0];
        pool.wait(&in_pending);
        uint8_t* d_ascii = (uint8_t*)sl.d_in.p;
        uint64_t* d_off = (uint64_t*)(d_ascii + blob_bytes);
        if (sl.used) ABW_HIP(ctx, hipStreamWaitEvent(P.s_h2d, sl.e_compute, 0));      // the slot's previous chunk has been counted
        ABW_HIP(ctx, hipMemcpyAsync(d_ascii, sl.pin_in, blob_bytes + off_bytes, hipMemcpyHostToDevice, P.s_h2d));
        ABW_HIP(ctx, hipEventRecord(sl.e_h2d, P.s_h2d));
        // kernels: segmentation (the count stays on the device), emit over the capacity, counting up to the device-side count
        ABW_HIP(ctx, hipStreamWaitEvent(st, sl.e_h2d, 0));
        uint8_t* wb = (uint8_t*)sl.d_work.p;
        SegWork w = {(uint32_t*)wb, (uint64_t*)(wb + w_blk), (ScafSeg*)(wb + w_blk + w_pre), (uint64_t*)(wb + w_blk + w_pre + w_info),
                     (uint64_t*)(wb + w_blk + w_pre + w_info + w_nseg), (uint64_t*)(wb + w_blk + w_pre + w_info + w_nseg + w_off)};
        {
            ABW_PROF(ctx, "sig_segment");
            if ((rc2 = segment_counts(ctx, w, d_ascii, bytes, d_off, ns, window))) return rc2;
        }
        uint8_t* d = (uint8_t*)sl.d_out.p;
        ABW_HIP(ctx, hipMemcpyAsync(d, w.seg_off + ns, 16, hipMemcpyDeviceToDevice, st));       // count, flag
        ABW_HIP(ctx, hipMemcpyAsync(sl.pin_out, d, 16, hipMemcpyDeviceToHost, st));
        ABW_HIP(ctx, hipEventRecord(sl.e_count, st));
        uint64_t* d_start = (uint64_t*)(d + 64); uint64_t* d_end = d_start + cap;
        uint8_t* rows = (uint8_t*)(d_end + cap);
        double* d_dims = dims ? (double*)rows : nullptr;
        int32_t* d_milli = milli ? (int32_t*)(rows + (dims ? cap * ABW_SIG_NDIMS * 8ull : 0)) : nullptr;
        uint32_t* d_scaf = (uint32_t*)(rows + cap * row_bytes);
        {
            ABW_PROF(ctx, "sig_segment");
            hipLaunchKernelGGL(seg_emit_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, d_ascii, bytes, d_off, ns,
                               w.blk_prefix, w.info, w.seg_off, cap, d_scaf, d_start, d_end, (uint32_t)s0);
            ABW_LAUNCH_CHECK(ctx);
        }
        if (row_bytes) {
            const uint64_t blocks = std::min<uint64_t>((cap + 3) / 4, (uint64_t)ctx->num_cu * 16);
            ABW_PROF(ctx, "sig_count");
            hipLaunchKernelGGL(sig_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_ascii, bytes, d_off, d_scaf, d_start, d_end, cap,
                               d_dims, d_milli, (uint32_t*)nullptr, (const uint64_t*)w.seg_off + ns, (uint32_t)s0);
            ABW_LAUNCH_CHECK(ctx);
        }
        ABW_HIP(ctx, hipEventRecord(sl.e_compute, st));
        sl.first_scaf = s0; sl.n_scaf = ns; sl.bytes = bytes; sl.cap = cap; sl.used = true; sl.wrote = false;
        return ABW_OK;
    };

    // stage B: chunk i's count is back -> its place in the caller's arrays is known -> exact-size D2H into pinned memory
    auto copy_back = [&](int64_t i) -> int {
        PipeSlot& sl = P.slot[i % PIPE_SLOTS];
        ABW_HIP(ctx, hipEventSynchronize(sl.e_count));
        const uint64_t* res = (const uint64_t*)sl.pin_out;
        const uint64_t cnt = res[0];
        if (res[1]) flag_n = 1;
        sl.count = cnt; sl.out_base = total_seg; sl.wrote = false;
        total_seg += cnt;
        if (flag_n || over) return ABW_OK;
        if (total_seg > seg_cap) { over = true; return ABW_OK; }
        if (cnt == 0) return ABW_OK;
        if (!seg_scaf || !seg_start || !seg_end) return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_build: null segment output");
        if (cnt > sl.cap) {
            // more segments than the chunk's tables hold (all-N scaffolds make one per base): this chunk alone again with
            // tables of the right size, synchronously -- rare
            ABW_HIP(ctx, hipStreamSynchronize(st));
            const uint64_t blob_bytes = ((sl.bytes + 15) & ~15ull) + 16;
            uint8_t* d_ascii = (uint8_t*)sl.d_in.p;
            uint64_t* d_off = (uint64_t*)(d_ascii + blob_bytes);
            int rc2;
            abw_devbuf big;
            if ((rc2 = abw_reserve(ctx, big, 64 + cnt * (4 + 8 + 8 + row_bytes)))) return rc2;
            uint64_t* d_start = (uint64_t*)((uint8_t*)big.p + 64); uint64_t* d_end = d_start + cnt;
            uint8_t* rows = (uint8_t*)(d_end + cnt);
            uint32_t* d_scaf = (uint32_t*)(rows + cnt * row_bytes);
            double* d_dims = dims ? (double*)rows : nullptr;
            int32_t* d_milli = milli ? (int32_t*)(rows + (dims ? cnt * ABW_SIG_NDIMS * 8ull : 0)) : nullptr;
            uint64_t n2 = 0;
            rc2 = abw_sig_segment_dev(ctx, d_ascii, sl.bytes, d_off, sl.n_scaf, window, cnt, d_scaf, d_start, d_end, &n2);
            if (!rc2 && row_bytes) rc2 = abw_sig_count_dev(ctx, d_ascii, sl.bytes, d_off, d_scaf, d_start, d_end, cnt, d_dims, d_milli, nullptr);
            if (!rc2) {
                std::vector<uint32_t> h_scaf(cnt);
                hipError_t e = hipMemcpyAsync(h_scaf.data(), d_scaf, cnt * 4, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipMemcpyAsync(seg_start + sl.out_base, d_start, cnt * 8, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipMemcpyAsync(seg_end + sl.out_base, d_end, cnt * 8, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess && dims) e = hipMemcpyAsync(dims + sl.out_base * ABW_SIG_NDIMS, d_dims, cnt * ABW_SIG_NDIMS * 8ull, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess && milli) e = hipMemcpyAsync(milli + sl.out_base * ABW_SIG_NCOLS, d_milli, cnt * ABW_SIG_NCOLS * 4ull, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e != hipSuccess) rc2 = abw_fail(ctx, ABW_ERR_HIP, "abw_sig_build: copy back failed: %s", hipGetErrorString(e));
                else for (uint64_t j = 0; j < cnt; j++) seg_scaf[sl.out_base + j] = h_scaf[j] + (uint32_t)sl.first_scaf;
            }
            abw_release(big);
            return rc2;
        }
        const uint8_t* d = (const uint8_t*)sl.d_out.p;
        uint8_t* h = sl.pin_out;
        const size_t o_start = 64, o_end = o_start + sl.cap * 8, o_rows = o_end + sl.cap * 8, o_scaf = o_rows + sl.cap * row_bytes;
        const size_t o_milli = o_rows + (dims ? sl.cap * ABW_SIG_NDIMS * 8ull : 0);
        ABW_HIP(ctx, hipStreamWaitEvent(P.s_d2h, sl.e_compute, 0));
        ABW_HIP(ctx, hipMemcpyAsync(h + o_start, d + o_start, cnt * 8, hipMemcpyDeviceToHost, P.s_d2h));
        ABW_HIP(ctx, hipMemcpyAsync(h + o_end, d + o_end, cnt * 8, hipMemcpyDeviceToHost, P.s_d2h));
        if (dims) ABW_HIP(ctx, hipMemcpyAsync(h + o_rows, d + o_rows, cnt * ABW_SIG_NDIMS * 8ull, hipMemcpyDeviceToHost, P.s_d2h));
        if (milli) ABW_HIP(ctx, hipMemcpyAsync(h + o_milli, d + o_milli, cnt * ABW_SIG_NCOLS * 4ull, hipMemcpyDeviceToHost, P.s_d2h));
        ABW_HIP(ctx, hipMemcpyAsync(h + o_scaf, d + o_scaf, cnt * 4, hipMemcpyDeviceToHost, P.s_d2h));
        ABW_HIP(ctx, hipEventRecord(sl.e_d2h, P.s_d2h));
        sl.wrote = true;
        return ABW_OK;
    };

    // stage C: chunk i's tables are in pinned memory -> worker threads move them into the caller's arrays
    auto hand_over = [&](int64_t i) -> int {
        PipeSlot& sl = P.slot[i % PIPE_SLOTS];
        if (!sl.wrote) return ABW_OK;
        ABW_HIP(ctx, hipEventSynchronize(sl.e_d2h));
        const uint64_t cnt = sl.count, at = sl.out_base;
        uint8_t* h = sl.pin_out;
        const size_t o_start = 64, o_end = o_start + sl.cap * 8, o_rows = o_end + sl.cap * 8, o_scaf = o_rows + sl.cap * row_bytes;
        const size_t o_milli = o_rows + (dims ? sl.cap * ABW_SIG_NDIMS * 8ull : 0);
        pool.submit(seg_start + at, h + o_start, cnt * 8, &sl.out_pending);
        pool.submit(seg_end + at, h + o_end, cnt * 8, &sl.out_pending);
        if (dims) pool.submit(dims + at * ABW_SIG_NDIMS, h + o_rows, cnt * ABW_SIG_NDIMS * 8ull, &sl.out_pending);
        if (milli) pool.submit(milli + at * ABW_SIG_NCOLS, h + o_milli, cnt * ABW_SIG_NCOLS * 4ull, &sl.out_pending);
        pool.submit(seg_scaf + at, h + o_scaf, cnt * 4, &sl.out_pending);
        sl.wrote = false;
        return ABW_OK;
    };

    rc = ABW_OK;
    for (int64_t i = 0; i < nchunks + 2 && !rc && !flag_n; i++) {
        if (i < nchunks) rc = submit(i);
        if (!rc && i >= 1 && i - 1 < nchunks) rc = copy_back(i - 1);
        if (!rc && i >= 2 && i - 2 < nchunks) rc = hand_over(i - 2);
    }
    // quiesce whatever the outcome: nothing of this call may still touch the caller's memory or the slots afterwards
    (void)hipStreamSynchronize(P.s_h2d);
    (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(P.s_d2h);
    for (auto& sl : P.slot) pool.wait(&sl.out_pending);
    if (rc) return rc;
    if (flag_n) return abw_fail(ctx, ABW_ERR_LOWER_N, "lower-case 'n' in sequence data (the reference throws Illegal_DNAString)");
    *n_seg = total_seg;
    if (over || total_seg > seg_cap)
        return abw_fail(ctx, ABW_ERR_CAPACITY, "segment capacity %llu < %llu", (unsigned long long)seg_cap, (unsigned long long)total_seg);
    return ABW_OK;
}

}  // namespace

void abw_sig_pipe_free(abw_ctx* ctx)
{
    abw_sig_pipe* P = ctx->sig_pipe;
    if (!P) return;
    for (auto& sl : P->slot) {
        if (sl.pin_in) (void)hipHostFree(sl.pin_in);
        if (sl.pin_out) (void)hipHostFree(sl.pin_out);
        abw_release(sl.d_in); abw_release(sl.d_out); abw_release(sl.d_work);
        for (hipEvent_t e : {sl.e_h2d, sl.e_compute, sl.e_count, sl.e_d2h}) if (e) (void)hipEventDestroy(e);
    }
    if (P->s_h2d) (void)hipStreamDestroy(P->s_h2d);
    if (P->s_d2h) (void)hipStreamDestroy(P->s_d2h);
    delete P;
    ctx->sig_pipe = nullptr;
}

namespace {

}  // namespace

// Called by abw_ctx_create with the context's device current: this device's address of the static workspace.
int abw_sig_static_init(abw_ctx* ctx)
{
    void* p = nullptr;
    ABW_HIP(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    ctx->sig_static = p;
    return ABW_OK;
}

// shared with abw_cov.hip
int abw_device_scan_u32(abw_ctx* ctx, const uint32_t* in, uint64_t n, uint64_t* out) { return device_scan<uint32_t>(ctx, in, n, out); }

extern "C" {

int abw_sig_dim_names(char* buf, uint64_t buf_len)
{
    if (!buf) return ABW_ERR_ARG;
    SigTables t;
    std::vector<std::string> names;
    build_tables(t, &names);
    uint64_t need = 0;
    for (auto& s : names) need += s.size() + 1;
    if (buf_len < need) return ABW_ERR_CAPACITY;
    char* p = buf;
    for (auto& s : names) { memcpy(p, s.c_str(), s.size() + 1); p += s.size() + 1; }
    return ABW_OK;
}

int abw_sig_selftest_division(abw_ctx* ctx, uint32_t t_lo, uint32_t t_step, uint32_t count, uint32_t n_max,
                              uint64_t* mismatches, uint64_t* checksums)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!mismatches || t_lo == 0 || count == 0 || n_max == 0 || t_step == 0 || (uint64_t)t_lo + (uint64_t)(count - 1) * t_step > 0x7FFFFFFFull)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_selftest_division: divisors must lie in [1, 2^31)");
    abw_device_guard g(ctx->device);
    int rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[0], 8 + (size_t)count * 8))) return rc;
    unsigned long long* d_bad = (unsigned long long*)ctx->scratch[0].p;
    unsigned long long* d_sum = d_bad + 1;
    hipStream_t st = ctx->stream;
    ABW_HIP(ctx, hipMemsetAsync(d_bad, 0, 8, st));
    hipLaunchKernelGGL(div_selftest_kernel, dim3((count + 255) / 256), dim3(256), 0, st, t_lo, t_step, count, n_max, d_bad, d_sum);
    ABW_LAUNCH_CHECK(ctx);
    ABW_HIP(ctx, hipMemcpyAsync(mismatches, d_bad, 8, hipMemcpyDeviceToHost, st));
    if (checksums) ABW_HIP(ctx, hipMemcpyAsync(checksums, d_sum, (size_t)count * 8, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    return ABW_OK;
}

int abw_sig_segment_host(const uint8_t* ascii, const uint64_t* scaf_off, uint64_t n_scaf, uint32_t window,
                         uint64_t seg_cap, uint32_t* seg_scaf, uint64_t* seg_start, uint64_t* seg_end,
                         uint64_t* n_seg)
{
    if ((!ascii && n_scaf && scaf_off && scaf_off[n_scaf]) || !scaf_off || !n_seg || window == 0) return ABW_ERR_ARG;
    // pass 1 (parallel): per scaffold non-N count -> number of segments; also the lower-case 'n' rule
    std::vector<uint64_t> nseg(n_scaf + 1, 0), nbps_v(n_scaf, 0);
    std::vector<int> bad(1, 0);
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto pass1 = [&](unsigned t) {
        for (uint64_t s = t; s < n_scaf; s += nt) {
            const uint8_t* p = ascii + scaf_off[s];
            const uint64_t len = scaf_off[s + 1] - scaf_off[s];
            uint64_t Ns = 0;
            for (uint64_t i = 0; i < len; i++) { Ns += p[i] == 'N'; if (p[i] == 'n') bad[0] = 1; }
            uint64_t nonN = len - Ns, k = nonN / window;
            if (k == 0) k = 1;
            const uint64_t nbps = nonN / k;
            nbps_v[s] = nbps;
            nseg[s] = nbps == 0 ? len : nonN / nbps;
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(pass1, t);
        for (auto& x : th) x.join();
    }
    if (bad[0]) return ABW_ERR_LOWER_N;
    uint64_t total = 0;
    for (uint64_t s = 0; s < n_scaf; s++) { uint64_t c = nseg[s]; nseg[s] = total; total += c; }
    nseg[n_scaf] = total;
    *n_seg = total;
    if (total > seg_cap) return ABW_ERR_CAPACITY;
    if (total && (!seg_scaf || !seg_start || !seg_end)) return ABW_ERR_ARG;
    auto pass2 = [&](unsigned t) {
        for (uint64_t s = t; s < n_scaf; s += nt) {
            const uint8_t* p = ascii + scaf_off[s];
            const uint64_t len = scaf_off[s + 1] - scaf_off[s], nbps = nbps_v[s];
            uint64_t o = nseg[s], acc = 0, sit = 0;
            for (uint64_t it = 0; it < len; it++) {                 // ref :209-225
                if (p[it] != 'N') acc++;
                if (acc == nbps) {
                    seg_scaf[o] = (uint32_t)s; seg_start[o] = sit + 1; seg_end[o] = it + 1;
                    o++; sit = it + 1; acc = 0;
                }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(pass2, t);
        for (auto& x : th) x.join();
    }
    return ABW_OK;
}

int abw_sig_segment_dev(abw_ctx* ctx, const uint8_t* d_ascii, uint64_t total_len, const uint64_t* d_scaf_off,
                        uint64_t n_scaf, uint32_t window, uint64_t seg_cap, uint32_t* d_seg_scaf,
                        uint64_t* d_seg_start, uint64_t* d_seg_end, uint64_t* n_seg)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!n_seg || window == 0 || (n_scaf && !d_scaf_off)) return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_segment_dev: bad argument");
    abw_device_guard g(ctx->device);
    *n_seg = 0;
    if (n_scaf == 0) return ABW_OK;
    const uint64_t nblk = total_len / SEGBLK + 1;                 // + partial / empty tail block
    // scratch[0]: blk_nonN u32[nblk] | scratch[1]: blk_prefix u64[nblk+1] | scratch[2]: ScafSeg[n_scaf]
    // scratch[3]: nseg u64[n_scaf] | scratch[4]: seg_off u64[n_scaf+1] + the lower-case-n flag (one copy brings both back)
    // scratch[8]: temporary of the scans
    int rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[0], nblk * 4))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[1], (nblk + 1) * 8))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[2], n_scaf * sizeof(ScafSeg)))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[3], n_scaf * 8))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[4], (n_scaf + 2) * 8))) return rc;
    if ((rc = abw_reserve(ctx, ctx->scratch[8], (2 * ((std::max(nblk, n_scaf) + 1023) / 1024) + 2) * 8))) return rc;
    if ((rc = abw_reserve_pinned(ctx, 16))) return rc;
    SegWork w = {(uint32_t*)ctx->scratch[0].p, (uint64_t*)ctx->scratch[1].p, (ScafSeg*)ctx->scratch[2].p,
                 (uint64_t*)ctx->scratch[3].p, (uint64_t*)ctx->scratch[4].p, (uint64_t*)ctx->scratch[8].p};
    hipStream_t st = ctx->stream;
    ABW_PROF(ctx, "sig_segment");
    if ((rc = segment_counts(ctx, w, d_ascii, total_len, d_scaf_off, n_scaf, window))) return rc;
    uint64_t* back = (uint64_t*)ctx->pinned;                      // [0] number of segments, [1] flag
    ABW_HIP(ctx, hipMemcpyAsync(back, w.seg_off + n_scaf, 16, hipMemcpyDeviceToHost, st));
    ABW_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t total = back[0];
    const int h_flag = (int)back[1];
    if (h_flag) return abw_fail(ctx, ABW_ERR_LOWER_N, "lower-case 'n' in sequence data (the reference throws Illegal_DNAString)");
    *n_seg = total;
    if (total > seg_cap) return abw_fail(ctx, ABW_ERR_CAPACITY, "segment capacity %llu < %llu", (unsigned long long)seg_cap, (unsigned long long)total);
    if (total == 0) return ABW_OK;
    if (!d_seg_scaf || !d_seg_start || !d_seg_end) return abw_fail(ctx, ABW_ERR_ARG, "null segment output");
    hipLaunchKernelGGL(seg_emit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_ascii, total_len, d_scaf_off, n_scaf,
                       w.blk_prefix, w.info, w.seg_off, total, d_seg_scaf, d_seg_start, d_seg_end, 0u);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

int abw_sig_segment_static(abw_ctx* ctx, const uint8_t* d_ascii, uint64_t total_len, const uint64_t* d_scaf_off,
                           uint64_t n_scaf, uint32_t window, uint64_t seg_cap, uint32_t* d_seg_scaf,
                           uint64_t* d_seg_start, uint64_t* d_seg_end, uint64_t* d_result)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!d_result || window == 0 || (n_scaf && (!d_ascii || !d_scaf_off)) || (seg_cap && (!d_seg_scaf || !d_seg_start || !d_seg_end)))
        return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_segment_static: bad argument");
    if (total_len > ABW_SIG_STATIC_MAX_LEN || n_scaf > ABW_SIG_STATIC_MAX_SCAF || seg_cap > 0xFFFFFF00ull * 256)
        return abw_fail(ctx, ABW_ERR_CAPACITY, "abw_sig_segment_static: %llu bytes / %llu scaffolds pass the static workspace "
                        "(at most %llu bytes and %llu scaffolds per call; use abw_sig_segment_dev)", (unsigned long long)total_len,
                        (unsigned long long)n_scaf, (unsigned long long)ABW_SIG_STATIC_MAX_LEN, (unsigned long long)ABW_SIG_STATIC_MAX_SCAF);
    if (!ctx->sig_static) return abw_fail(ctx, ABW_ERR_STATE, "abw_sig_segment_static: the static workspace of device %d was not resolved", ctx->device);
    abw_device_guard g(ctx->device);
    if (n_scaf == 0) { ABW_HIP(ctx, hipMemsetAsync(d_result, 0, 16, ctx->stream)); return ABW_OK; }
    // the same arrays as abw_sig_segment_dev, carved out of this device's copy of `scratch` (sized for the maxima)
    uint8_t* base = (uint8_t*)ctx->sig_static;
    SegWork w;
    w.blk_nonN = (uint32_t*)(base + STATIC_OFF_BLK);
    w.blk_prefix = (uint64_t*)(base + STATIC_OFF_PREFIX);
    w.info = (ScafSeg*)(base + STATIC_OFF_INFO);
    w.nseg = (uint64_t*)(base + STATIC_OFF_NSEG);
    w.seg_off = (uint64_t*)(base + STATIC_OFF_SEGOFF);
    w.scan_tmp = (uint64_t*)(base + STATIC_OFF_SCAN);
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = segment_counts(ctx, w, d_ascii, total_len, d_scaf_off, n_scaf, window))) return rc;
    // the count and the flag stay on the device (no read-back, no synchronisation): seg_off[n_scaf], flag -> d_result[0], [1]
    ABW_HIP(ctx, hipMemcpyAsync(d_result, w.seg_off + n_scaf, 16, hipMemcpyDeviceToDevice, st));
    if (seg_cap) {
        hipLaunchKernelGGL(seg_emit_kernel, dim3((unsigned)((seg_cap + 255) / 256)), dim3(256), 0, st, d_ascii, total_len, d_scaf_off, n_scaf,
                           w.blk_prefix, w.info, w.seg_off, seg_cap, d_seg_scaf, d_seg_start, d_seg_end, 0u);
        ABW_LAUNCH_CHECK(ctx);
    }
    return ABW_OK;
}

int abw_sig_count_dev(abw_ctx* ctx, const uint8_t* d_ascii, uint64_t total_len, const uint64_t* d_scaf_off,
                      const uint32_t* d_seg_scaf, const uint64_t* d_seg_start, const uint64_t* d_seg_end,
                      uint64_t n_seg, double* d_dims, int32_t* d_milli, uint32_t* d_counts)
{
    if (!ctx) return ABW_ERR_ARG;
    if (n_seg == 0) return ABW_OK;
    if (!d_ascii || !d_scaf_off || !d_seg_scaf || !d_seg_start || !d_seg_end)
        return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_count_dev: null input");
    if (((uintptr_t)d_ascii & 15) != 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_count_dev: d_ascii must be 16-byte aligned");
    abw_device_guard g(ctx->device);
    int rc = upload_tables(ctx);
    if (rc) return rc;
    uint64_t blocks = (n_seg + 3) / 4;
    const uint64_t maxb = (uint64_t)ctx->num_cu * 16;
    if (blocks > maxb) blocks = maxb;
    ABW_PROF(ctx, "sig_count");
    hipLaunchKernelGGL(sig_count_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_ascii, total_len, d_scaf_off,
                       d_seg_scaf, d_seg_start, d_seg_end, n_seg, d_dims, d_milli, d_counts, (const uint64_t*)nullptr, 0u);
    ABW_LAUNCH_CHECK(ctx);
    return ABW_OK;
}

int abw_sig_build(abw_ctx* ctx, const uint8_t* ascii, const uint64_t* scaf_off, uint64_t n_scaf, uint32_t window,
                  uint64_t seg_cap, uint32_t* seg_scaf, uint64_t* seg_start, uint64_t* seg_end,
                  double* dims, int32_t* milli, uint64_t* n_seg)
{
    if (!ctx) return ABW_ERR_ARG;
    if (!scaf_off || !n_seg || window == 0) return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_build: null argument");
    abw_device_guard g(ctx->device);
    *n_seg = 0;
    if (n_scaf == 0) return ABW_OK;
    if (scaf_off[n_scaf] && !ascii) return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_build: null sequence");
    for (uint64_t i = 0; i < n_scaf; i++)
        if (scaf_off[i + 1] < scaf_off[i]) return abw_fail(ctx, ABW_ERR_ARG, "abw_sig_build: scaf_off must ascend");
    return sig_build_pipeline(ctx, ascii, scaf_off, n_scaf, window, seg_cap, seg_scaf, seg_start, seg_end, dims, milli, n_seg);
}

}  // extern "C"
