// kmer_device.hpp -- device-side k-mer arithmetic for gfx950 (wave64).
//
// Re-derivations of the reference's scalar helpers in a form that suits the
// GPU: no rolling state, no tables in memory.  Each helper cites the reference
// code whose RESULT it must reproduce bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(SHK_FAKE_HASH) && !defined(SHK_TIMING_ONLY)
#error "-DSHK_FAKE_HASH builds a library whose results are WRONG (timing-only ablation): say so with -DSHK_TIMING_ONLY as well"
#endif

namespace shk {

// ---------------------------------------------------------------------------
// XXH64 of the 8 little-endian bytes of a k-mer, seed 0.
// Must equal _get_hash(kmer) (kmer_utils.hpp:81-83 -> xxhash.hpp:495-500; for
// len == 8 only the short-input branch :487-489, one 8-byte lane :427-433 and
// the avalanche :449-453 execute).  Primes: xxhash.hpp:349.
// ---------------------------------------------------------------------------
constexpr uint64_t XP1 = 0x9E3779B185EBCA87ull;
constexpr uint64_t XP2 = 0xC2B2AE3D27D4EB4Full;
constexpr uint64_t XP3 = 0x165667B19E3779F9ull;
constexpr uint64_t XP4 = 0x85EBCA77C2B2AE63ull;
constexpr uint64_t XP5 = 0x27D4EB2F165667C5ull;

// rotate left by a constant 0 < r < 32; on the device two v_alignbit_b32 (hipcc's own lowering uses a
// 64-bit shift plus two more instructions)
__host__ __device__ __forceinline__ uint64_t rotl64(uint64_t x, int r)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 32 - r) << 32) | __builtin_amdgcn_alignbit(lo, hi, 32 - r);
#else
  return (x << r) | (x >> (64 - r));
#endif
}

__host__ __device__ __forceinline__ uint64_t xxh64_u64(uint64_t v)
{
#if defined(SHK_FAKE_HASH) && defined(__HIP_DEVICE_COMPILE__)
  // (timing-only ablation, results are WRONG: what a classify kernel would cost if a k-mer's filter position came for one multiply --
  //  the upper bound of what any scheme that takes XXH64 out of a probe can save; profiles/README.md, round 6)
  return (v * XP2) ^ (v >> 29);
#endif
  uint64_t h = XP5 + 8ull;          // seed(0) + PRIME5 + len
  uint64_t k1 = v * XP2;            // round(0, v)
#if defined(__HIP_DEVICE_COMPILE__)
  // keep the product as a value: hipcc otherwise rewrites rotl(v*P2, 31) as two more multiplies by
  // shifted constants (17 multiply instructions per hash instead of 15)
  asm("" : "+v"(k1));
#endif
  k1 = rotl64(k1, 31);
  k1 *= XP1;
  h ^= k1;
  h = rotl64(h, 27) * XP1 + XP4;
  h ^= h >> 33;
  h *= XP2;
  h ^= h >> 29;
  h *= XP3;
  h ^= h >> 32;
  return h;
}

// Low word of a position-table slot: multi(1) | overflow(1) | gene-or-rank(30).  TAB_OVERFLOW is kept in slot 0 of a
// bucket and says that some key whose HOME this bucket is was placed further down its probe path: a search that finds
// nothing at home and no such mark is over, however full the bucket is (at load 0.3 a bucket is full for 12 % of the
// probes but overflowed for 2 %).
constexpr uint32_t TAB_OVERFLOW = 1u << 30, TAB_PAYLOAD = 0x3FFFFFFFu;

// ---------------------------------------------------------------------------
// The k-mer keyed table (DeviceIndex::ktab): where a canonical k-mer lives and the word its slot is compared with.
// fwd = the k-mer as kmer_utils.hpp:67-69 packs it (first base most significant), rc = revcompl(fwd, k) (kmer_utils.hpp:47-55);
// the key is their minimum (KmerBuilder.hpp:49).  The MINIMISER is the smallest hash among the k - w + 1 canonical w-mers of the
// k-mer (w-mer i from the left of fwd and w-mer i from the right of rc are each other's reverse complement, so the set of
// canonical w-mers -- and with it the minimiser -- is the same whichever strand a read shows); the hash is a multiplication by
// an odd constant modulo 2^32, injective on w-mers of up to 16 bases, so the smallest hash names the w-mer.  Consecutive k-mers
// of a sequence share k - 1 bases and, more often than not, their minimiser: they are looked up in the same 128-byte line.
// w <= 15, k - w <= 3.
// ---------------------------------------------------------------------------
constexpr uint32_t KTAB_C1 = 0x9E3779B1u, KTAB_C2 = 0x85EBCA6Bu;
__host__ __device__ __forceinline__ void ktab_home(const uint64_t fwd, const uint64_t rc, const uint32_t k, const uint32_t w, const uint32_t line_lg,
                                                   uint32_t &bucket, uint32_t &want)
{
  const uint32_t wmask = (1u << (2u * w)) - 1u;
  const uint32_t nw = k - w;          // w-mers per k-mer, less one
  uint32_t mh = 0xFFFFFFFFu;
#pragma unroll
  for (uint32_t i = 0; i < 4u; ++i) {
    const uint32_t sa = i <= nw ? 2u * (nw - i) : 0u;
    const uint32_t a = (uint32_t)(fwd >> sa) & wmask, b = (uint32_t)(rc >> (2u * i)) & wmask;
    const uint32_t h = (a < b ? a : b) * KTAB_C1;
    mh = (i <= nw && h < mh) ? h : mh;
  }
  // Which of a line's eight buckets: three bits that depend on the bases AROUND the minimiser.  The keys that share a minimiser
  // differ only in where it sits in them and in the k - w bases beside it -- bases at the two ENDS of the k-mer -- so the low bits of
  // the key alone would be the minimiser's own for every key that ends with it (one bucket of the line taking them all: paths of
  // 30 lines on the 60 000-gene index).  The key is therefore passed through a bijection first -- its top three bases xored into its
  // last three, the upper of those six bits into the lower --: equal images, equal keys, and the image's low three bits mix all six
  // bases.  The bucket names those three bits, the slot holds the other 31 (k <= 17) and "taken".
  const uint64_t key = fwd < rc ? fwd : rc;
  uint32_t lo = (uint32_t)key ^ ((uint32_t)(key >> (2u * k - 6u)) & 63u);
  lo ^= (lo >> 3) & 7u;
  const uint32_t line = (mh * KTAB_C2) >> (32u - line_lg);
  bucket = (line << 3) | (lo & 7u);
  want = ((((uint32_t)(key >> 32) << 29) | (lo >> 3)) << 1) | 1u;
}

// ---------------------------------------------------------------------------
// Base classification, 4 ASCII bytes at a time (SWAR).
// Result must agree with to_int (kmer_utils.hpp:29-41): A/a C/c G/g T/t are
// valid with codes 0..3 (= to_int-1, kmer_utils.hpp:68), every other byte --
// including bytes >= 128, which are undefined behaviour in the reference -- is
// invalid.
//   code4 : 2-bit code in the low bits of each byte
//   inv4  : bit 0 of each byte set when that byte is NOT one of ACGTacgt
// ---------------------------------------------------------------------------
__device__ __forceinline__ void classify4(uint32_t w, uint32_t &code4, uint32_t &inv4)
{
  const uint32_t t = w & 0xDFDFDFDFu;                       // fold lower case onto upper case
  code4 = ((t >> 1) ^ (t >> 2)) & 0x03030303u;              // A->0 C->1 G->2 T->3
  const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, code4);
  uint32_t x = expect ^ t;                                  // zero byte <=> valid base
  x |= x >> 4;
  x |= x >> 2;
  x |= x >> 1;
  inv4 = x & 0x01010101u;
}

// 4 x 2-bit codes (one per byte, first character in the LOW byte) -> 8 bits,
// first character most significant (kmer_utils.hpp:67-69 packs MSB first).
__device__ __forceinline__ uint32_t pack4(uint32_t code4) { return (code4 * 0x40100401u) >> 24; }

// bit 0 of each byte -> 4 contiguous bits, byte 0 -> bit 0
__device__ __forceinline__ uint32_t gather4(uint32_t flags4) { return ((flags4 * 0x00204081u) >> 21) & 0xFu; }

// quality mask (FastqSplitter.hpp:70,:104-109): a base is masked when its
// quality character, compared as a SIGNED char, is below mq = Q + 33.
// Returns bit 0 of each byte set where masked.
__device__ __forceinline__ uint32_t qmask4(uint32_t q4, int mq)
{
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = (int)(int8_t)(q4 >> (8 * i));
    m |= (q < mq ? 1u : 0u) << (8 * i);
  }
  return m;
}

// ---------------------------------------------------------------------------
// Reverse complement of a k-mer held LEFT-ALIGNED in 64 bits (first base in
// bits 63:62).  Returns it right-aligned in the low 2k bits; must equal
// revcompl(kmer, k) (kmer_utils.hpp:47-55).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t revcomp_left_aligned(uint64_t top, uint32_t k)
{
  uint64_t x = ~top;
  // reverse all 64 bits, then swap the two bits of every base back
  uint32_t lo = __builtin_bitreverse32((uint32_t)(x >> 32));
  uint32_t hi = __builtin_bitreverse32((uint32_t)x);
  lo = ((lo >> 1) & 0x55555555u) | ((lo & 0x55555555u) << 1);
  hi = ((hi >> 1) & 0x55555555u) | ((hi & 0x55555555u) << 1);
  const uint64_t r = ((uint64_t)hi << 32) | lo;
  return r & ((1ull << (2 * k)) - 1ull);
}

// canonical k-mer = min(kmer, revcompl) (KmerBuilder.hpp:49, ReadAnalyzer.hpp:55)
__device__ __forceinline__ uint64_t canonical_from_top(uint64_t top, uint32_t k)
{
  const uint64_t fw = top >> (64 - 2 * k);
  const uint64_t rc = revcomp_left_aligned(top, k);
  return fw < rc ? fw : rc;
}

// hash % bits for bits = m * 2^s with s >= 32 and m < 2^32 -- every `-b` size of the CLI is a multiple of
// 2^33 bits (argument_parser.hpp:150).  h % (m 2^s) = ((h >> s) % m) << s | (h mod 2^s), and the quotient
// q = h >> s fits 32 bits, so the inner remainder is Lemire's direct computation (Faster Remainder by
// Direct Computation, 2019): c = floor((2^64-1)/m) + 1, r = ((c*q mod 2^64) * m) >> 64 -- exact for all
// 32-bit q and m; four multiply instructions instead of a 64-bit division loop.
__device__ __forceinline__ uint64_t bf_pos_fastmod(uint64_t h, uint32_t s, uint32_t m, uint64_t c)
{
  const uint32_t q = (uint32_t)(h >> s);
  const uint64_t lowbits = c * (uint64_t)q;
  const uint64_t p0 = (uint64_t)(uint32_t)lowbits * m;
  const uint64_t p1 = (uint64_t)(uint32_t)(lowbits >> 32) * m + (p0 >> 32);
  const uint32_t r = (uint32_t)(p1 >> 32);
  return ((uint64_t)r << s) | (h & ((1ull << s) - 1ull));
}

// rank of a set bit = number of ones in [0,pos) (bloomfilter.h:70 _brank(bf_idx);
// :90 uses rank(pos+1), 1-based -- the same list).  One directory word per
// 64-bit filter word, so the word that was probed is all that is needed.
__device__ __forceinline__ uint32_t bf_rank(const uint32_t *__restrict__ rank_w, uint64_t word, uint64_t pos)
{
  const uint64_t below = word & ((1ull << (pos & 63u)) - 1ull);
  return rank_w[pos >> 6] + (uint32_t)__builtin_popcountll(below);
}

// wave64 reductions on the DPP network (no LDS traffic; all 64 lanes must be
// active).  quad_perm swaps, row mirrors, then row_bcast15 / row_bcast31 fold
// the four 16-lane rows; lane 63 ends up with the result.
#define SHK_DPP(old_, v_, ctrl_, rmask_) __builtin_amdgcn_update_dpp((int)(old_), (int)(v_), (ctrl_), (rmask_), 0xF, false)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
  constexpr uint32_t I = 0xFFFFFFFFu;
  uint32_t t;
  t = (uint32_t)SHK_DPP(I, v, 0xB1, 0xF); v = t < v ? t : v;    // quad_perm [1,0,3,2]
  t = (uint32_t)SHK_DPP(I, v, 0x4E, 0xF); v = t < v ? t : v;    // quad_perm [2,3,0,1]
  t = (uint32_t)SHK_DPP(I, v, 0x141, 0xF); v = t < v ? t : v;   // row_half_mirror
  t = (uint32_t)SHK_DPP(I, v, 0x140, 0xF); v = t < v ? t : v;   // row_mirror
  t = (uint32_t)SHK_DPP(I, v, 0x142, 0xA); v = t < v ? t : v;   // row_bcast15 -> rows 1,3
  t = (uint32_t)SHK_DPP(I, v, 0x143, 0xC); v = t < v ? t : v;   // row_bcast31 -> rows 2,3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
  v += (uint32_t)SHK_DPP(0, v, 0xB1, 0xF);
  v += (uint32_t)SHK_DPP(0, v, 0x4E, 0xF);
  v += (uint32_t)SHK_DPP(0, v, 0x141, 0xF);
  v += (uint32_t)SHK_DPP(0, v, 0x140, 0xF);
  v += (uint32_t)SHK_DPP(0, v, 0x142, 0xA);
  v += (uint32_t)SHK_DPP(0, v, 0x143, 0xC);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#undef SHK_DPP

}  // namespace shk
