// classify_common.hpp -- device helpers shared by the classify kernels' translation units (classify.hip: the fast / general
// kernels and the small kernels behind them; classify_uni_u*.hip: classify_uni_kernel, one file per unroll so that they
// compile in parallel).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "kmer_device.hpp"
#include "shark_internal.hpp"

// waves per SIMD the classify_uni_kernel instantiations are compiled for (LDS-summary modes / table modes); the launcher sizes its grid by them
#ifndef SHK_UNI_WAVES
#define SHK_UNI_WAVES 6   // 75 VGPRs, nothing spilled; at 8 waves per SIMD (64 VGPRs) the loop reloads spilled lane constants from scratch and measures 1-6 % slower
#endif
// (table modes at 6: 80 VGPRs -- the quality-mask instantiations do not spill -- measured against 8 waves / 64 VGPRs: configs[4] shape
//  11.2 -> 10.4 ms, configs[2] index 41.8 -> 40.6 ms, everything else within 0.5 %)
#ifndef SHK_TAB_WAVES
#define SHK_TAB_WAVES 6
#endif
// (the instantiations that probe the minimiser-bucketed table, up to U = 5)
#ifndef SHK_KT_WAVES
#define SHK_KT_WAVES 6
#endif

// (-DSHK_NO_ACCEPT=1: a build without the early decision, for A/B timing)
#ifndef SHK_NO_ACCEPT
#define SHK_NO_ACCEPT 0
#endif
// waves of the one workgroup per CU that holds the LDS-resident table (16 = the most a workgroup can have; 8 for the occupancy experiment)
#ifndef SHK_LX_WAVES
#define SHK_LX_WAVES 16
#endif

namespace shk {

constexpr int CF_WAVES = 4;              // wavefronts per workgroup
constexpr int CF_THREADS = CF_WAVES * 64;
constexpr uint32_t GENE_INF = 0xFFFFFFFFu;

// timing-only ablation switches (env SHK_ABLATE) exist only in builds made with -DSHK_ABLATION;
// the shipped kernels carry neither the branches nor the SGPRs
#ifdef SHK_ABLATION
#define SHK_ABL(P, bit) ((P).ablate & (bit))
#else
#define SHK_ABL(P, bit) false
#endif

// how a k-mer's filter position is looked up (chosen per index at finalize time)
enum ProbeMode {
  PM_BV_MOD = 0,   // filter word, position = hash % size (non power-of-two sizes)
  PM_BV = 1,       // filter word, position = hash & (size-1)
  PM_BV_SUM = 2,   // summary level, then filter word
  PM_TAB = 3,      // position table (exact sparse encoding of the set bits)
  PM_TAB_SUM = 4,  // summary level, then position table
  PM_LDS_TAB = 5,  // 2^18-bit summary held in LDS, then position table (small indices)
  PM_TAB_MOD = 6,      // PM_TAB for a filter size that is not a power of two (position = hash % size)
  PM_LDS_TAB_MOD = 7,  // PM_LDS_TAB, likewise
  PM_KTAB = 8          // the k-mer keyed, minimiser-bucketed table (classify_uni_kernel only; the other kernels take PM_TAB on such an index)
};
__host__ __device__ constexpr bool pm_pow2(int m) { return m != PM_BV_MOD && m != PM_TAB_MOD && m != PM_LDS_TAB_MOD; }
__host__ __device__ constexpr bool pm_lds(int m) { return m == PM_LDS_TAB || m == PM_LDS_TAB_MOD; }
__host__ __device__ constexpr bool pm_tab(int m) { return m == PM_TAB || m == PM_TAB_SUM || m == PM_TAB_MOD || m == PM_KTAB || pm_lds(m); }

// position of a hash in a filter whose size is not a power of two: hash % _size (bloomfilter.h:58,:66,:88)
__device__ __forceinline__ uint64_t bf_pos_np(uint64_t h, const ClassifyParams &P)
{
  return P.mod_fast ? bf_pos_fastmod(h, P.mod_shift, P.mod_m, P.mod_c) : h % P.bf_bits;
}

// the smallest integer t with (double)t >= c * (double)len: an integer coverage passes the reference's test (ReadAnalyzer.hpp:104)
// iff it is >= t.  0 when the product is not positive (or NaN): nothing can be ruled out
__device__ __forceinline__ uint32_t cov_threshold(const double c, const uint32_t len)
{
  const double x = c * (double)len;
  if (!(x > 0.0)) return 0u;
  if (x >= 4294967295.0) return 0xFFFFFFFFu;
  return (uint32_t)ceil(x);
}

// Per-wave staging area of a read (LDS in the fast kernel, a global scratch slice in the general
// kernel).  The read's bases live at PACKED positions: mate 1 at [0, L1), mate 2 at [P2, P2+L2)
// with P2 = L1 rounded up to 8.  A k-mer SLOT is a packed position pp: slot pp is the k-mer that
// starts at packed position pp; it exists when pp < nk1 or 0 <= pp-P2 < nk2.  Two streams of 2-bit
// codes are kept so that both orientations of a k-mer are plain right-shift extractions
// (v_alignbit_b32) -- the per-k-mer bit reversal of the first versions is gone:
//   fw : base at position p  -> bits [2(p&15), +2) of dword p>>4
//   rv : base at position p  -> same rule applied to the mirrored position rcap-1-p
// A window of k bases read from `rv` at rcap-k-pp is the k-mer MSB-first (kmer_utils.hpp:67-69);
// the window read from `fw` at pp holds the bases in reverse order, so its complement is the
// reverse complement (kmer_utils.hpp:47-55).
struct WaveStore {
  uint32_t *fw;         // forward code stream
  uint32_t *rv;         // mirrored code stream
  uint32_t rcap;        // mirror length in bases (= stage_cap_bases(S))
  uint64_t *vbits;      // validity, 64 packed positions per word, LSB first
  uint32_t *rec_start;  // per slot: cursor into csr_ids
  uint32_t *rec_end;    // per slot: end of its list
  uint32_t *cur;        // per slot: gene at the cursor, GENE_INF when exhausted / no hit
};

// 8 bytes starting at p, of which only `rem` (>= 1) belong to the read.  Reads are packed back
// to back, so p has no alignment.  Unaligned 8-byte loads are slow on this path and a byte loop for
// the tail of a mate serialises one memory latency per byte (measured: 4.5 of 15 ms on the all-miss
// workload), so the bytes are fetched as up to three ALIGNED dwords and realigned with
// v_alignbyte_b32.  An aligned dword that contains at least one byte of the read cannot leave the
// caller's allocation, so nothing outside the buffers is ever touched.
struct Raw8 {
  uint32_t d0, d1, d2;   // up to three aligned dwords
  uint32_t shn;          // byte shift (bits 1:0) | number of wanted bytes << 4 (0 = nothing fetched)
};

__device__ __forceinline__ Raw8 load8_issue(const uint8_t *p, uint32_t rem)
{
  const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(p) & 3u;
  const uint32_t *q = reinterpret_cast<const uint32_t *>(p - sh);
  const uint32_t nbytes = rem < 8u ? rem : 8u;
  const uint32_t last = sh + nbytes - 1u;            // index of the last wanted byte relative to q
  Raw8 r;
  r.d0 = q[0];
  r.d1 = last >= 4u ? q[1] : 0u;
  r.d2 = last >= 8u ? q[2] : 0u;
  r.shn = sh | (nbytes << 4);
  return r;
}

// realign (first use of the loaded dwords: this is where the wait lands)
__device__ __forceinline__ uint64_t load8_finish(const Raw8 r)
{
  const uint32_t sh = r.shn & 3u, nbytes = r.shn >> 4;
  const uint32_t lo = __builtin_amdgcn_alignbyte(r.d1, r.d0, sh);
  const uint32_t hi = __builtin_amdgcn_alignbyte(r.d2, r.d1, sh);
  uint64_t w = ((uint64_t)hi << 32) | lo;
  if (nbytes < 8u) w &= (1ull << (8u * nbytes)) - 1ull;
  return w;
}

// wait for outstanding vector loads and hand `r` back as plain register values
__device__ __forceinline__ void retire_loads(Raw8 &r)
{
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.d0), "+v"(r.d1), "+v"(r.d2));
}

// the result/queue pointers, re-read from device memory where they are needed (scalar loads from
// the constant cache); the empty asm stops LICM from hoisting them into loop-long SGPRs
__device__ __forceinline__ const ClassifyOut *out_ptrs(const ClassifyParams &P)
{
  const ClassifyOut *o = P.out;
  asm volatile("" : "+s"(o));
  return o;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// offsets / lengths of one read (pair)
struct ReadMeta {
  uint64_t o1, o2;
  uint32_t L1, L2;
};

// a wave-uniform 64-bit value, forced into SGPRs so that loads addressed by it are scalar loads
// (a vector load of the offsets would put an s_waitcnt vmcnt(0) right behind the prefetched bases)
__device__ __forceinline__ uint64_t uniform64(uint64_t x)
{
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ ReadMeta fetch_meta(const ClassifyParams &P, uint64_t read_in)
{
  const uint64_t read = uniform64(read_in);
  // (the loads may be vector loads; readfirstlane makes the VALUES scalar, so that the per-read
  // geometry derived from them is computed on the scalar unit in the callers' loops)
  ReadMeta m;
  m.o1 = uniform64(P.off1[read]);
  m.L1 = __builtin_amdgcn_readfirstlane((uint32_t)(P.off1[read + 1] - m.o1));
  m.o2 = 0;
  m.L2 = 0;
  if (P.seq2) {
    m.o2 = uniform64(P.off2[read]);
    m.L2 = __builtin_amdgcn_readfirstlane((uint32_t)(P.off2[read + 1] - m.o2));
  }
  return m;
}

// Offsets of a read fetched one iteration ahead.  The loads are ordinary (vector) loads of a
// wave-uniform address, so the compiler tracks them: any use, spill or move waits for them first.
// (An earlier version issued s_load_dwordx4 in inline asm to keep these loads off the vmcnt
// counter; the compiler cannot know that such a result is still in flight, and under SGPR pressure
// it spilled the destination registers right behind the load -- saving stale values.)  The caller
// retires them together with the prefetched bases at the end of its loop, where they have long
// landed, and turns them into scalars there.
struct ReadMetaRaw {
  uint64_t a0, a1, b0, b1;   // off1[read], off1[read+1] ; off2[read], off2[read+1]
};

__device__ __forceinline__ ReadMetaRaw fetch_meta_issue(const ClassifyParams &P, uint64_t read_in)
{
  const uint64_t read = uniform64(read_in);
  ReadMetaRaw r;
  r.a0 = P.off1[read];
  r.a1 = P.off1[read + 1];
  r.b0 = 0;
  r.b1 = 0;
  if (P.seq2) {
    r.b0 = P.off2[read];
    r.b1 = P.off2[read + 1];
  }
  return r;
}

// wait for the outstanding vector loads and hand the offsets back as plain register values
__device__ __forceinline__ void retire_meta(ReadMetaRaw &r)
{
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.b0), "+v"(r.b1));
}

__device__ __forceinline__ ReadMeta meta_finish(const ReadMetaRaw &r)
{
  ReadMeta m;
  m.o1 = uniform64(r.a0);
  m.L1 = __builtin_amdgcn_readfirstlane((uint32_t)(r.a1 - r.a0));
  m.o2 = uniform64(r.b0);
  m.L2 = __builtin_amdgcn_readfirstlane((uint32_t)(r.b1 - r.b0));
  return m;
}

// issue the loads of the 8 bases (and qualities) that lane `gi` stages for this read (group gi of
// the packed layout); nothing here consumes the loaded dwords, so no wait is placed here
template <bool HASQ>
__device__ __forceinline__ void fetch_group(const ClassifyParams &P, const ReadMeta &m, uint32_t gi, Raw8 &w, Raw8 &q)
{
  const uint32_t g2 = (m.L1 + 7) >> 3;
  const uint32_t n_groups = g2 + ((m.L2 + 7) >> 3);
  w = Raw8{0u, 0u, 0u, 0u};
  q = Raw8{0u, 0u, 0u, 0u};
  if (SHK_ABL(P, 4u)) { w.d0 = 0x43414754u; w.d1 = 0x43415447u; w.shn = 8u << 4; return; }   // ablation 4: no base loads
  if (gi < n_groups) {
    const bool m2 = gi >= g2;
    const uint32_t b = (m2 ? gi - g2 : gi) << 3;
    const uint32_t L = m2 ? m.L2 : m.L1;
    if (b < L) {
      w = load8_issue((m2 ? P.seq2 + m.o2 : P.seq1 + m.o1) + b, L - b);
      if (HASQ) q = load8_issue((m2 ? P.qual2 + m.o2 : P.qual1 + m.o1) + b, L - b);
    }
  }
}

// bucket `bi` of a table.  SMALL: the table is known to be below 4 GiB (every index with an LDS summary: < 2^22 buckets),
// so the byte offset fits 32 bits and the load takes the scalar base + one VGPR of offset instead of a 64-bit VGPR address.
template <bool SMALL>
__device__ __forceinline__ uint4 load_bucket(const uint4 *__restrict__ tab16, const uint32_t bi)
{
  if (SMALL) return *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(tab16) + (bi << 4));
  return tab16[bi];
}

// The probe path behind a full home bucket, one probe after the other (process_read; classify_uni_kernel walks all its probes
// per round).  A probe that finds its key gets it moved into (x, y) of bk[j] in home form.  POSKEY: the position table's
// slots carry their displacement in the compared word.
template <int U, bool POSKEY, typename WantOf, typename BucketOf>
__device__ __forceinline__ void walk_probe_paths(const uint4 *__restrict__ tab16, uint4 (&bk)[U], bool (&more)[U], bool &lane_any,
                                                 const WantOf want_of, const BucketOf bucket_of, const uint32_t jlo = 0u,
                                                 const uint32_t jhi = (uint32_t)U)
{
#pragma unroll
  for (int j = 0; j < U; ++j) {
    if ((uint32_t)j < jlo || (uint32_t)j >= jhi) continue;   // (wave-uniform: the probes of the rounds being worked on)
    uint32_t d = 0;
    while (more[j]) {
      ++d;
      const uint4 b2 = tab16[bucket_of(j, d)];
      const uint32_t w0 = want_of(j);
      const uint32_t want = POSKEY ? (w0 | d) : w0;
      const bool n0 = b2.y == want, n1 = b2.w == want;
      if (n0 | n1) {
        bk[j].x = n0 ? b2.x : b2.z;
        bk[j].y = w0;
        lane_any = true;
        more[j] = false;
      } else if ((b2.y == 0u) | (b2.w == 0u) | (d >= 63u)) {   // a free slot ends every search
        more[j] = false;
      }
    }
  }
}

}  // namespace shk
