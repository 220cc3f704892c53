// gzip_parallel.hpp -- parallel inflate of ORDINARY gzip files (any writer, any level, one member or many) for the shark CLI.
//
// The reference reads compressed samples through gzopen / gzread on the parsing thread (main.cpp:88,:129,:202, kseq.h:228): one
// core inflates, everything else waits.  A deflate stream cannot simply be cut into pieces: a block may start at any BIT, and its
// back-references reach up to 32 KiB into text that the piece's decoder has never seen.  The way round both is known (pugz,
// rapidgzip); this is an implementation of it written for this repository:
//
//   1. the compressed file is cut into chunks of CHUNK bytes.  The first chunk starts behind the gzip header.  Every other chunk
//      SEARCHES for a deflate block that starts at or behind its first byte: bit by bit, a candidate must carry a non-final
//      dynamic-Huffman header whose three codes are complete prefix codes, must decode to its end-of-block symbol into text bytes
//      only (tab, LF, CR, 0x20-0x7e: samples are FASTQ / FASTA), and a valid block header must follow it.
//   2. pass 1 (parallel): every chunk inflates from its block start to the next chunk's block start into 16-BIT SYMBOLS: a literal is
//      itself, a back-reference into the 32 KiB in front of the chunk -- which this decoder has not seen -- is the symbol
//      0x8000 | its position in that window (the output buffer is primed with those 32 768 markers, so a copy is a copy).
//   3. the windows are resolved in file order (32 KiB per chunk: the last symbols of a chunk, translated with the window before
//      it, are the window of the next one), and
//   4. pass 2 (parallel): every chunk translates its symbols into bytes through a 64 Ki-entry table (literals + its window).
//
// A search that settled on something that is not a block start is noticed by the chunk in front of it: its decoder arrives at
// another bit position.  It then simply keeps inflating through the impostor's territory (whose output is dropped) until it does
// meet a later chunk's start.  Input that is not text, a header the search cannot get past, corrupt data: the stream ends where
// zlib's would, or the caller falls back to gzread (usable() == false) -- nothing is guessed.  The gzip trailer's CRC-32 is not
// recomputed (zlib reports a mismatch only after it has delivered every byte, and the reference's reader ignores that error:
// kseq.h:94-110 treats a negative gzread as end of file).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace shk {
namespace gzp {

constexpr uint32_t WIN = 32768;

// ---- bit input over a byte array (LSB first, deflate order) ----------------------------------------------------------------------
struct Bits {
  const uint8_t *p = nullptr;
  size_t n = 0;          // bytes
  uint64_t pos = 0;      // absolute bit position of the next unread bit
  uint64_t buf = 0;      // the next bits of the input, LSB = bit `pos`; only the low `cnt` are promised
  unsigned cnt = 0;
  size_t bp = 0;         // the next byte to load
  bool over = false;     // something was read beyond the end of the input

  void seek(uint64_t bitpos)
  {
    pos = bitpos;
    bp = (size_t)(bitpos >> 3);
    buf = 0;
    cnt = 0;
    over = false;
    const unsigned skip = (unsigned)(bitpos & 7);
    if (skip && bp < n) {
      buf = (uint64_t)p[bp] >> skip;
      cnt = 8 - skip;
      ++bp;
    }
  }
  // at least 56 bits in the buffer (fewer only at the end of the input)
  inline void refill()
  {
    if (bp + 8 <= n) {
      uint64_t w;
      memcpy(&w, p + bp, 8);
      buf |= w << cnt;
      const unsigned add = (63u - cnt) >> 3;
      bp += add;
      cnt += add * 8;
    } else {
      while (cnt <= 56 && bp < n) {
        buf |= (uint64_t)p[bp++] << cnt;
        cnt += 8;
      }
    }
  }
  inline uint32_t peek(unsigned k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
  inline void drop(unsigned k)
  {
    if (k > cnt) { over = true; k = cnt; }
    buf >>= k;
    cnt -= k;
    pos += k;
  }
  inline uint32_t take(unsigned k)
  {
    if (cnt < k) refill();
    const uint32_t v = peek(k);
    drop(k);
    return v;
  }
  void align_byte() { drop((unsigned)((8 - (pos & 7)) & 7)); }
  bool at_end() const { return bp >= n; }
};

// ---- canonical Huffman decoding tables: PRIMARY bits direct, longer codes through second-level tables ------------------------------
// entry: bits [15:0] symbol (or the index of a sub-table), [19:16] code length consumed at this level, [20] sub-table link
struct Huff {
  static constexpr unsigned MAXBITS = 15;
  unsigned primary = 0;
  std::vector<uint32_t> tab;

  // lens[0..n): code lengths (0 = unused).  false: over-subscribed, or incomplete (a single code of length 1 is accepted, as zlib does)
  bool build(const uint8_t *lens, unsigned n, unsigned primary_bits)
  {
    unsigned count[MAXBITS + 1] = {0};
    for (unsigned i = 0; i < n; ++i) count[lens[i]]++;
    if (count[0] == n) return false;
    unsigned maxlen = MAXBITS;
    while (maxlen > 1 && count[maxlen] == 0) --maxlen;
    int left = 1;
    for (unsigned l = 1; l <= MAXBITS; ++l) {
      left <<= 1;
      left -= (int)count[l];
      if (left < 0) return false;
    }
    const bool single = (n - count[0]) == 1 && count[1] == 1;
    if (left > 0 && !single) return false;
    primary = std::min(primary_bits, maxlen);
    unsigned next[MAXBITS + 2];
    {
      unsigned code = 0;
      count[0] = 0;
      for (unsigned l = 1; l <= MAXBITS; ++l) {
        code = (code + count[l - 1]) << 1;
        next[l] = code;
      }
    }
    // size: primary table + one sub-table per distinct primary prefix of a longer code
    tab.assign((size_t)1 << primary, 0);
    // first pass: codes up to `primary` bits
    struct Long { uint32_t rev; uint8_t len; uint16_t sym; };
    std::vector<Long> longs;
    for (unsigned s = 0; s < n; ++s) {
      const unsigned l = lens[s];
      if (!l) continue;
      const unsigned code = next[l]++;
      // bit-reversed code (deflate sends Huffman codes MSB first into an LSB-first stream)
      uint32_t rev = 0;
      for (unsigned b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
      if (l <= primary) {
        const uint32_t e = s | (l << 16);
        for (uint32_t i = rev; i < (1u << primary); i += 1u << l) tab[i] = e;
      } else {
        longs.push_back({rev, (uint8_t)l, (uint16_t)s});
      }
    }
    if (!longs.empty()) {
      // sub-tables of (maxlen - primary) bits, one per primary prefix in use
      const unsigned sub = maxlen - primary;
      std::vector<int32_t> where((size_t)1 << primary, -1);
      for (const Long &g : longs) {
        const uint32_t pre = g.rev & ((1u << primary) - 1);
        if (where[pre] < 0) {
          where[pre] = (int32_t)tab.size();
          tab.resize(tab.size() + ((size_t)1 << sub), 0);
          tab[pre] = (uint32_t)where[pre] | (sub << 16) | (1u << 20);
        }
        const uint32_t hi = g.rev >> primary;
        const unsigned hl = g.len - primary;
        const uint32_t e = g.sym | ((uint32_t)hl << 16);
        for (uint32_t i = hi; i < (1u << sub); i += 1u << hl) tab[(size_t)where[pre] + i] = e;
      }
    }
    return true;
  }
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// ---- the inflater: deflate blocks into 16-bit symbols ----------------------------------------------------------------------------
// out[] holds WIN marker symbols followed by the decoded text; `floor` is the lowest index a back-reference may reach (WIN for a
// stream whose start this decoder has seen: nothing lies in front of a member's first byte).
struct Inflater {
  Bits in;
  // symbols: WIN markers, then the text; grown without initialisation (a std::vector would zero tens of megabytes per chunk)
  struct SymBuf {
    uint16_t *p = nullptr;
    size_t cap = 0;
    ~SymBuf() { free(p); }
    SymBuf() = default;
    SymBuf(const SymBuf &) = delete;
    SymBuf &operator=(const SymBuf &) = delete;
    size_t size() const { return cap; }
    uint16_t *data() { return p; }
    uint16_t &operator[](size_t i) { return p[i]; }
    void resize(size_t want)
    {
      if (want <= cap) return;
      uint16_t *q = static_cast<uint16_t *>(realloc(p, want * sizeof(uint16_t)));
      if (!q) throw std::bad_alloc();
      p = q;
      cap = want;
    }
  } out;
  size_t n = 0;
  size_t floor = 0;
  bool text_only = false;      // reject literals that are not text (the search's validation)
  Huff lit, dist;
  bool have_dist = false;

  enum Rc { OK = 0, END_OF_BLOCK_FINAL = 1, BAD = -1, NEED_INPUT = -2 };

  void reset_output()
  {
    if (out.size() < WIN + (1u << 16)) out.resize(WIN + (1u << 20));
    for (uint32_t i = 0; i < WIN; ++i) out[i] = (uint16_t)(0x8000u | i);
    n = WIN;
  }
  inline void need(size_t more)
  {
    if (n + more + 320 > out.size()) out.resize(std::max(out.size() * 2, n + more + 320));
  }

  // header of the block at the current position; *final_block, *type.  For type 2 the tables are built.
  Rc read_header(bool &final_block, unsigned &type)
  {
    in.refill();
    if (in.cnt < 3) return NEED_INPUT;
    final_block = in.take(1) != 0;
    type = in.take(2);
    if (type == 3) return BAD;
    if (type == 0) return OK;
    if (type == 1) {
      uint8_t l[288];
      for (int i = 0; i < 144; ++i) l[i] = 8;
      for (int i = 144; i < 256; ++i) l[i] = 9;
      for (int i = 256; i < 280; ++i) l[i] = 7;
      for (int i = 280; i < 288; ++i) l[i] = 8;
      uint8_t d[30];
      for (int i = 0; i < 30; ++i) d[i] = 5;
      if (!lit.build(l, 288, 10)) return BAD;
      // (the fixed distance code has 30 of 32 codes: incomplete by the rule above, complete for decoding purposes)
      uint8_t d32[32];
      for (int i = 0; i < 32; ++i) d32[i] = 5;
      (void)d;
      if (!dist.build(d32, 32, 8)) return BAD;
      have_dist = true;
      if (!text_only) build_multi();
      return OK;
    }
    in.refill();
    if (in.cnt < 14) return NEED_INPUT;
    const unsigned hlit = in.take(5) + 257, hdist = in.take(5) + 1, hclen = in.take(4) + 4;
    if (hlit > 286 || hdist > 30) return BAD;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (unsigned i = 0; i < hclen; ++i) {
      if (in.cnt < 3) { in.refill(); if (in.cnt < 3) return NEED_INPUT; }
      cl[order[i]] = (uint8_t)in.take(3);
    }
    Huff clh;
    if (!clh.build(cl, 19, 7)) return BAD;
    uint8_t lens[286 + 30 + 16];
    unsigned i = 0;
    while (i < hlit + hdist) {
      in.refill();
      if (in.cnt == 0) return NEED_INPUT;
      const uint32_t e = clh.tab[in.peek(clh.primary)];
      const unsigned l = (e >> 16) & 15;
      if (!l) return BAD;
      if (l > in.cnt) return NEED_INPUT;
      in.drop(l);
      const unsigned sym = e & 0xFFFF;
      if (sym < 16) {
        lens[i++] = (uint8_t)sym;
      } else {
        unsigned rep, val = 0;
        if (sym == 16) {
          if (i == 0) return BAD;
          val = lens[i - 1];
          rep = 3 + in.take(2);
        } else if (sym == 17) {
          rep = 3 + in.take(3);
        } else {
          rep = 11 + in.take(7);
        }
        if (i + rep > hlit + hdist) return BAD;
        while (rep--) lens[i++] = (uint8_t)val;
      }
      if (in.over) return NEED_INPUT;
    }
    if (lens[256] == 0) return BAD;                       // no end-of-block code
    if (!lit.build(lens, hlit, 10)) return BAD;
    // (a block without matches may carry one unused distance code of length 0/1; zlib accepts an incomplete code of one symbol)
    unsigned nz = 0;
    for (unsigned k = 0; k < hdist; ++k) nz += lens[hlit + k] != 0;
    have_dist = nz != 0;
    if (have_dist && !dist.build(lens + hlit, hdist, 8)) return BAD;
    if (!text_only) build_multi();
    return OK;
  }

  // Runs of literals -- a FASTQ record's bases are 150 of them, at two to three bits each --: for every MULTI_BITS-bit pattern of
  // the input, the literals it holds whole (up to four: one 8-byte store of 16-bit symbols), their count and their bits.  Built per
  // block from the literal table (1 024 patterns, a few microseconds against the 16 384 symbols of a gzip -1 block); the fast loop
  // of huffman_body asks it first.
  static constexpr unsigned MULTI_BITS = 10;
  uint64_t multi_syms[1u << MULTI_BITS];
  uint8_t multi_info[1u << MULTI_BITS];      // count (low 3 bits) | bits << 3; 0: the pattern does not start with a whole literal
  // ... and the two tables with what the fast loop would otherwise look up behind them folded in: an entry of a length code
  // carries the base length and its number of extra bits, one of a distance code the base distance and its extra bits
  //   [15:0] literal / base length / base distance (sub-table link: its index)   [19:16] code bits at this level   [20] link
  //   [24:21] extra bits   [26:25] kind: 0 literal, 1 length (distance table: a distance), 2 end of block, 3 not a code of a valid stream
  std::vector<uint32_t> lit_fast, dist_fast;
  void build_fast_tables()
  {
    lit_fast = lit.tab;
    for (uint32_t &e : lit_fast) {
      if (e & (1u << 20)) continue;
      const unsigned sym = e & 0xFFFF;
      if (!((e >> 16) & 15) || sym < 256) continue;          // (unused slot: stays 0, "no code"; literal: as it is)
      if (sym == 256) e = (e & 0xF0000u) | (2u << 25);
      else if (sym > 285) e = (e & 0xF0000u) | (3u << 25);
      else e = (e & 0xF0000u) | LEN_BASE[sym - 257] | ((uint32_t)LEN_EXTRA[sym - 257] << 21) | (1u << 25);
    }
    dist_fast.clear();
    if (have_dist) {
      dist_fast = dist.tab;
      for (uint32_t &e : dist_fast) {
        if (e & (1u << 20)) continue;
        const unsigned sym = e & 0xFFFF;
        if (!((e >> 16) & 15)) continue;
        if (sym > 29) e = (e & 0xF0000u) | (3u << 25);
        else e = (e & 0xF0000u) | DIST_BASE[sym] | ((uint32_t)DIST_EXTRA[sym] << 21) | (1u << 25);
      }
    }
  }

  void build_multi()
  {
    build_fast_tables();
    const uint32_t *lt = lit.tab.data();
    const unsigned lp = lit.primary;
    for (unsigned idx = 0; idx < (1u << MULTI_BITS); ++idx) {
      unsigned bits = 0, k = 0;
      uint64_t syms = 0;
      while (k < 4 && bits < MULTI_BITS) {
        const unsigned rem = MULTI_BITS - bits;
        // (fewer known bits than the table's index: an entry whose code fits into them is the same under every extension)
        const uint32_t e = lt[(idx >> bits) & ((1u << (rem < lp ? rem : lp)) - 1)];
        const unsigned l = (e >> 16) & 15;
        if ((e & (1u << 20)) || !l || l > rem || (e & 0xFFFF) >= 256) break;
        syms |= (uint64_t)(e & 0xFFFF) << (16 * k);
        ++k;
        bits += l;
      }
      multi_syms[idx] = syms;
      multi_info[idx] = (uint8_t)(k ? (k | (bits << 3)) : 0);
    }
  }

  // the body of a stored block (the header has been read)
  Rc stored_body()
  {
    in.align_byte();
    in.refill();
    if (in.cnt < 32) return NEED_INPUT;
    const uint32_t len = in.take(16), nlen = in.take(16);
    if ((len ^ 0xFFFFu) != nlen) return BAD;
    const uint64_t byte = in.pos >> 3;
    if (byte + len > in.n) return NEED_INPUT;
    need(len);
    const uint8_t *s = in.p + byte;
    if (text_only)
      for (uint32_t i = 0; i < len; ++i)
        if (!is_text(s[i])) return BAD;
    for (uint32_t i = 0; i < len; ++i) out[n + i] = s[i];
    n += len;
    in.seek(in.pos + 8ull * len);
    return OK;
  }

  static inline bool is_text(unsigned c) { return (c >= 0x20 && c < 0x7f) || c == '\n' || c == '\t' || c == '\r'; }

  // the body of a Huffman block, up to and including its end-of-block symbol
  Rc huffman_body()
  {
    const uint32_t *lt = lit.tab.data(), *dt = have_dist ? dist.tab.data() : nullptr;
    const unsigned lp = lit.primary, dp = have_dist ? dist.primary : 0;
    const uint32_t lmask = (1u << lp) - 1, dmask = (1u << dp) - 1;
    for (;;) {
      need(600);
      uint16_t *o = out.data();
      size_t w = n;
      const size_t w_stop = out.size() - 300;
      // the fast loop (decoding proper, away from the end of the input): the bit buffer in locals, no bookkeeping per symbol -- the
      // position is 8 bp - cnt at any time --, one refill (56 bits) per run of literals or per length and distance with their
      // extra bits (48 at most).  A code the table does not know ends it with BAD as below.
      if (!text_only && in.bp + 16 <= in.n) {
        uint64_t buf = in.buf;
        unsigned cnt = in.cnt;
        size_t bp = in.bp;
        const uint8_t *const ip = in.p;
        const size_t bp_stop = in.n - 16;
        const uint32_t *const ltf = lit_fast.data(), *const dtf = have_dist ? dist_fast.data() : nullptr;
        Rc rc = OK;
        bool ended = false;
        while (w < w_stop && bp <= bp_stop) {
          {
            uint64_t x;
            memcpy(&x, ip + bp, 8);
            buf |= x << cnt;
            const unsigned add = (63u - cnt) >> 3;
            bp += add;
            cnt += add * 8;
          }
          // literals first, up to four per look (five looks fit the 56 bits: up to 20 symbols per refill; the capacity margin of
          // 300 symbols covers them and the 8-byte stores)
          unsigned mi = multi_info[buf & ((1u << MULTI_BITS) - 1)];
          if (mi) {
            for (int look = 0; look < 5 && mi; ++look) {
              memcpy(o + w, &multi_syms[buf & ((1u << MULTI_BITS) - 1)], 8);
              w += mi & 7u;
              buf >>= mi >> 3;
              cnt -= mi >> 3;
              mi = multi_info[buf & ((1u << MULTI_BITS) - 1)];
            }
            continue;
          }
          uint32_t e = ltf[buf & lmask];
          if (__builtin_expect(e & (1u << 20), 0)) {
            const unsigned sub = (e >> 16) & 15;
            e = ltf[(e & 0xFFFF) + ((buf >> lp) & ((1u << sub) - 1))];
            buf >>= lp;
            cnt -= lp;
          }
          unsigned l = (e >> 16) & 15;
          if (__builtin_expect(!l, 0)) { rc = BAD; ended = true; break; }
          buf >>= l;
          cnt -= l;
          const unsigned kind = (e >> 25) & 3u;
          if (kind == 0) {       // (a literal with a code longer than MULTI_BITS)
            o[w++] = (uint16_t)(e & 0xFFFF);
            continue;
          }
          if (__builtin_expect(kind != 1, 0)) {      // end of block, or a length code no valid stream uses
            if (kind == 3) rc = BAD;
            ended = true;
            break;
          }
          if (__builtin_expect(!dtf, 0)) { rc = BAD; ended = true; break; }
          const unsigned le = (e >> 21) & 15u;
          const unsigned len = (e & 0xFFFF) + (unsigned)(buf & ((1u << le) - 1));
          buf >>= le;
          cnt -= le;
          uint32_t d = dtf[buf & dmask];
          if (__builtin_expect(d & (1u << 20), 0)) {
            const unsigned sub = (d >> 16) & 15;
            d = dtf[(d & 0xFFFF) + ((buf >> dp) & ((1u << sub) - 1))];
            buf >>= dp;
            cnt -= dp;
          }
          l = (d >> 16) & 15;
          if (__builtin_expect(!l || ((d >> 25) & 3u) != 1u, 0)) { rc = BAD; ended = true; break; }
          buf >>= l;
          cnt -= l;
          const unsigned de = (d >> 21) & 15u;
          const unsigned dd = (d & 0xFFFF) + (unsigned)(buf & ((1u << de) - 1));
          buf >>= de;
          cnt -= de;
          if (__builtin_expect(dd > w - floor, 0)) { rc = BAD; ended = true; break; }
          const uint16_t *s = o + w - dd;
          uint16_t *t = o + w;
          if (__builtin_expect(dd >= 8, 1)) {
            // sixteen symbols without asking (the bases of a FASTQ record come as matches of 3 to 16 symbols: a trip count that
            // follows len is a misprediction per match), in two steps of eight so that a distance of 8 to 15 copies what the
            // first step wrote; the rest, rarely, eight at a time.  (Up to 15 symbols past len: the capacity margin covers them.)
            memcpy(t, s, 16);
            memcpy(t + 8, s + 8, 16);
            for (unsigned i = 16; i < len; i += 8) memcpy(t + i, s + i, 16);
          } else {
            for (unsigned i = 0; i < len; ++i) t[i] = s[i];
          }
          w += len;
        }
        in.buf = buf;
        in.cnt = cnt;
        in.bp = bp;
        in.pos = 8ull * bp - cnt;
        if (ended) { n = w; return rc; }
        if (w >= w_stop) { n = w; continue; }      // (more room, then on)
      }
      // the careful loop: the end of the input, and the block search's trial decoding
      while (w < w_stop) {
        in.refill();
        uint32_t e = lt[in.buf & lmask];
        if (e & (1u << 20)) {                     // a code longer than the primary table's bits: second level
          const unsigned sub = (e >> 16) & 15;
          e = lt[(e & 0xFFFF) + ((in.buf >> lp) & ((1u << sub) - 1))];
          in.drop(lp);
        }
        unsigned l = (e >> 16) & 15;
        if (!l || l > in.cnt) { n = w; return (in.at_end() && in.cnt < 48) ? NEED_INPUT : BAD; }
        in.drop(l);
        const unsigned sym = e & 0xFFFF;
        if (sym < 256) {
          if (text_only && !is_text(sym)) { n = w; return BAD; }
          o[w++] = (uint16_t)sym;
          continue;
        }
        if (sym == 256) { n = w; return OK; }
        if (sym > 285 || !dt) { n = w; return BAD; }
        const unsigned li = sym - 257;
        unsigned len = LEN_BASE[li];
        if (LEN_EXTRA[li]) { len += in.peek(LEN_EXTRA[li]); in.drop(LEN_EXTRA[li]); }
        if (in.cnt < 32) in.refill();
        uint32_t d = dt[in.buf & dmask];
        if (d & (1u << 20)) {
          const unsigned sub = (d >> 16) & 15;
          d = dt[(d & 0xFFFF) + ((in.buf >> dp) & ((1u << sub) - 1))];
          in.drop(dp);
        }
        l = (d >> 16) & 15;
        if (!l || l > in.cnt) { n = w; return (in.at_end() && in.cnt < 48) ? NEED_INPUT : BAD; }
        in.drop(l);
        const unsigned ds = d & 0xFFFF;
        if (ds > 29) { n = w; return BAD; }
        unsigned dd = DIST_BASE[ds];
        if (DIST_EXTRA[ds]) { dd += in.peek(DIST_EXTRA[ds]); in.drop(DIST_EXTRA[ds]); }
        if (in.over) { n = w; return NEED_INPUT; }
        if (dd > w - floor) { n = w; return BAD; }          // reaches in front of what may be referenced
        const uint16_t *s = o + w - dd;
        uint16_t *t = o + w;
        if (dd >= 8) {
          // (a copy may run up to 7 symbols past len: the capacity margin covers it)
          for (unsigned i = 0; i < len; i += 8) memcpy(t + i, s + i, 16);
        } else {
          for (unsigned i = 0; i < len; ++i) t[i] = s[i];
        }
        w += len;
      }
      n = w;
    }
  }

  // one whole block at the current position
  Rc block(bool &final_block)
  {
    unsigned type = 0;
    Rc rc = read_header(final_block, type);
    if (rc != OK) return rc;
    if (in.over) return NEED_INPUT;
    rc = type == 0 ? stored_body() : huffman_body();
    if (rc == OK && in.over) return NEED_INPUT;
    return rc;
  }
};

// a gzip member header at byte `at`: returns the byte offset of its deflate data, 0 when there is none
inline size_t gzip_header_end(const uint8_t *p, size_t n, size_t at)
{
  if (at + 18 > n || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8) return 0;
  const unsigned flg = p[at + 3];
  if (flg & 0xE0) return 0;
  size_t q = at + 10;
  if (flg & 4) {
    if (q + 2 > n) return 0;
    q += 2 + (size_t)(p[q] | (p[q + 1] << 8));
  }
  for (int bit : {8, 16})
    if (flg & bit) {
      while (q < n && p[q]) ++q;
      ++q;
    }
  if (flg & 2) q += 2;
  return q < n ? q : 0;
}

}  // namespace gzp

// set when any ParallelGunzip of the process ran out of memory (its stream then ended early): whoever drives readers that sit
// on top of one -- the CLI -- asks once, at the end, and fails instead of presenting a prefix of the sample as the sample
inline std::atomic<bool> &parallel_gunzip_out_of_memory()
{
  static std::atomic<bool> flag{false};
  return flag;
}

// ---------------------------------------------------------------------------------------------------------------------------------
class ParallelGunzip {
 public:
  // chunk: compressed bytes per chunk (SHARK_GZ_CHUNK in the environment overrides it: the tests cut small files into many chunks)
  ParallelGunzip(const std::string &path, unsigned threads, size_t chunk = 4u << 20) : threads_(std::max(1u, threads)), CHUNK(chunk)
  {
    if (const char *e = getenv("SHARK_GZ_CHUNK")) {
      const long v = atol(e);
      if (v >= 4096) CHUNK = (size_t)v;
    }
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return;
    struct stat st;
    if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 64) return;
    n_ = (size_t)st.st_size;
    void *m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) return;
    p_ = static_cast<const uint8_t *>(m);
    madvise(m, n_, MADV_SEQUENTIAL);
    const size_t data = gzp::gzip_header_end(p_, n_, 0);
    if (!data) return;
    // worth it only when there is more than a couple of chunks, and only for text: the first block must decode as text
    if (n_ < 3 * CHUNK || threads_ < 2) return;
    {
      gzp::Inflater probe;
      probe.in.p = p_;
      probe.in.n = n_;
      probe.in.seek(8ull * data);
      probe.reset_output();
      probe.floor = gzp::WIN;
      probe.text_only = true;
      bool fin = false;
      if (probe.block(fin) != gzp::Inflater::OK) return;
    }
    n_chunks_ = (n_ + CHUNK - 1) / CHUNK;
    ch_.resize(n_chunks_);
    for (size_t i = 0; i < n_chunks_; ++i) ch_[i].reset(new Chunk());
    ch_[0]->start_bit = 8ull * data;
    ch_[0]->start_known = 1;
    ch_[0]->clean_start = true;
    piece_syms_ = std::max<size_t>((size_t)64 << 20, 16 * CHUNK);   // (far above what a chunk of a compressed sample inflates to)
    if (const char *e = getenv("SHARK_GZ_PIECE")) {
      const long v = atol(e);
      if (v >= (long)(2 * gzp::WIN)) piece_syms_ = (size_t)v;
    }
    // A stream the search finds no block start in -- fixed-Huffman or stored blocks only, many small members that are not BGZF
    // (every block is a final one) -- would be inflated by its first chunk alone while every other chunk searches its territory
    // bit by bit in vain: slower than gzread, for nothing.  The second chunk's search is made here: no start there, no parallel
    // inflate (the caller falls back to gzread, constant memory, as the reference reads it).  A stream that only turns unfriendly
    // further on stays correct and bounded: a chunk that inflates far beyond its territory hands its text out in pieces.
    // (ONE chunk without a start -- a stored or fixed-Huffman stretch near the head of an otherwise ordinary file -- does
    //  not decide for the whole file: the search goes on through the third and fourth chunk, and the first start found anywhere
    //  takes the file on; a chunk without a start is inflated by the chunk in front of it, as everywhere else in the file)
    {
      bool any = false;
      for (size_t i = 1; i < std::min<size_t>(n_chunks_, 4) && !any; ++i) {
        gzp::Inflater t;
        uint64_t bit = 0;
        const bool found = find_block(t, i * CHUNK, std::min(n_, (i + 1) * CHUNK), bit);
        ch_[i]->start_bit = bit;
        ch_[i]->search_state.store(1);
        ch_[i]->start_known.store(found ? 1 : -1, std::memory_order_release);
        any = found;
      }
      if (!any && !getenv("SHARK_GZ_FORCE_PARALLEL")) return;
    }
    usable_ = true;
    for (unsigned t = 0; t < threads_; ++t) th_.emplace_back([this] { worker(); });
  }
  ~ParallelGunzip()
  {
    {
      std::lock_guard<std::mutex> l(m_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
    for (auto &b : pool_) free(b.first);
    free(handed_buf_);
    if (p_) munmap(const_cast<uint8_t *>(p_), n_);
    if (fd_ >= 0) ::close(fd_);
  }
  ParallelGunzip(const ParallelGunzip &) = delete;
  ParallelGunzip &operator=(const ParallelGunzip &) = delete;
  bool usable() const { return usable_; }
  // a worker ran out of memory: next() has returned false early and the text delivered so far is NOT the whole stream
  bool failed() const { return failed_.load(); }
  // CPU seconds the workers spent: looking for block starts, decoding (pass 1), translating symbols to bytes (pass 2)
  void cpu_seconds(double &search, double &pass1, double &pass2) const
  {
    search = 1e-9 * (double)ns_search_.load();
    pass1 = 1e-9 * (double)ns_pass1_.load();
    pass2 = 1e-9 * (double)ns_pass2_.load();
  }

  // the next piece of uncompressed text, in file order (valid until the next call); false at the end of the stream (or when a
  // worker failed: failed())
  bool next(const char *&data, size_t &len)
  {
    for (;;) {
      std::unique_lock<std::mutex> l(m_);
      if (handed_buf_) {   // the buffer handed out last time goes back to the pool (pages already touched)
        if (pool_.size() < threads_ + 6) pool_.push_back({handed_buf_, handed_cap_});
        else free(handed_buf_);
        handed_buf_ = nullptr;
        cv_.notify_all();
      }
      if (consume_ >= n_chunks_) return false;
      Chunk &c = *ch_[consume_];
      cv_.wait(l, [&] { return c.done || !c.pieces.empty() || quit_; });
      if (quit_) return false;
      if (!c.pieces.empty()) {
        // text of a chunk that is still inflating (it went far beyond its territory: a stretch without block starts)
        const Piece pc = c.pieces.front();
        c.pieces.pop_front();
        handed_buf_ = pc.p;
        handed_cap_ = pc.cap;
        data = pc.p;
        len = pc.n;
        cv_.notify_all();
        return true;
      }
      ++consume_;
      cv_.notify_all();
      drop_pages((consume_ - 1) * CHUNK, consume_ * CHUNK);     // (nobody reads this territory any more)
      if (c.absorbed || c.n_bytes == 0) {
        if (c.stream_ends) { consume_ = n_chunks_; return false; }
        continue;                                  // (its territory was inflated by the chunk in front of it)
      }
      data = c.bytes;
      len = c.n_bytes;
      handed_buf_ = c.bytes;                        // (remembered by itself: consume_ may jump to the end below)
      handed_cap_ = c.cap_bytes;
      c.bytes = nullptr;
      if (c.stream_ends) consume_ = n_chunks_;      // the stream ended (or broke) inside this chunk: nothing behind it is delivered
      return true;
    }
  }

 private:
  struct Piece { char *p; size_t n, cap; };
  struct Chunk {
    std::atomic<int> start_known{0};     // 0 = still searching, 1 = start_bit valid, -1 = no block start found in its territory
    std::atomic<int> search_state{0};    // 0 = nobody has searched its block start yet, 1 = somebody has or is at it
    uint64_t start_bit = 0;
    bool clean_start = false;            // starts at a member's first block: no window in front of it
    bool absorbed = false;               // the chunk in front of it did not arrive at start_bit: it inflated this territory itself
    bool pass1 = false;                  // symbols decoded (guarded by m_)
    bool have_window = false;            // window_in valid (guarded by m_)
    bool done = false;                   // bytes ready, or absorbed (guarded by m_)
    bool stream_ends = false;            // the gzip stream ends, or is corrupt, inside this chunk
    std::vector<uint8_t> window_in;      // the WIN bytes in front of the chunk's text
    char *bytes = nullptr;               // malloc'ed text of the chunk
    size_t n_bytes = 0, cap_bytes = 0;
    std::deque<Piece> pieces;            // text handed out BEFORE the chunk is done, in order (guarded by m_)
    ~Chunk()
    {
      free(bytes);
      for (const Piece &pc : pieces) free(pc.p);
    }
  };

  // ---- search: a deflate block that starts at or behind byte `from` (and before byte `to`) ----
  bool find_block(gzp::Inflater &t, size_t from, size_t to, uint64_t &bit)
  {
    t.in.p = p_;
    t.in.n = n_;
    t.text_only = true;
    t.reset_output();
    const uint64_t lo = 8ull * from, hi = 8ull * std::min(to, n_ > 8 ? n_ - 8 : 0);
    for (uint64_t b = lo; b < hi; ++b) {
      // non-final (0), dynamic (01 as read LSB first: BTYPE = 2): bits b..b+2 = 0, 0, 1
      const uint64_t byte = b >> 3;
      uint32_t w;
      memcpy(&w, p_ + byte, 4);
      const uint32_t h = w >> (b & 7);
      if ((h & 7u) != 4u) continue;
      if (((h >> 3) & 31u) > 29u || ((h >> 8) & 31u) > 29u) continue;      // HLIT, HDIST
      t.in.seek(b);
      t.n = gzp::WIN;          // (the markers in front stay as they are)
      t.floor = 0;
      bool fin = false;
      if (t.block(fin) != gzp::Inflater::OK || fin) continue;
      if (t.n - gzp::WIN < 1024) continue;                                 // (a real block of a compressed sample is not this short)
      // a valid header must follow
      gzp::Inflater u;
      u.text_only = true;      // (tables only)
      u.in.p = p_;
      u.in.n = n_;
      u.in.seek(t.in.pos);
      unsigned type = 0;
      bool f2 = false;
      const gzp::Inflater::Rc rc = u.read_header(f2, type);
      if (rc == gzp::Inflater::BAD) continue;
      bit = b;
      return true;
    }
    return false;
  }

  void worker()
  {
    gzp::Inflater z, t;      // this thread's decoder and its search decoder: their buffers are reused from chunk to chunk
    for (;;) {
      size_t i;
      bool inflate;
      {
        std::unique_lock<std::mutex> l(m_);
        // Inflating runs not too far ahead of the consumer: every chunk in flight holds tens of megabytes of symbols.  SEARCHING a
        // chunk's block start costs no memory and has no such limit: a worker the limit keeps from inflating searches ahead -- and it
        // has to: a chunk that inflates through a stretch without block starts (a member of fixed-Huffman or stored blocks) waits
        // for the verdict of every chunk on its way, however far ahead of the consumer that is (with the searches tied to the
        // inflating, 24 such chunks of 64 KiB were a deadlock).
        cv_.wait(l, [&] { return quit_ || next_chunk_ >= n_chunks_ || next_chunk_ < consume_ + threads_ + 4 || next_search_ < n_chunks_; });
        if (quit_ || next_chunk_ >= n_chunks_) return;
        inflate = next_chunk_ < consume_ + threads_ + 4;
        if (inflate) i = next_chunk_++;
        else i = next_search_++;
      }
      try {
        if (inflate) run_chunk(i, z, t);
        else search_chunk(i, t);
      } catch (const std::bad_alloc &) {
        // out of memory: the stream cannot be delivered; every thread stops, next() returns false, failed() says why
        failed_.store(true);
        parallel_gunzip_out_of_memory().store(true);
        std::lock_guard<std::mutex> l(m_);
        quit_ = true;
        cv_.notify_all();
        return;
      }
    }
  }

  // the block start of chunk i, by whoever comes first (a worker about to inflate it, or one searching ahead)
  void search_chunk(size_t i, gzp::Inflater &t)
  {
    Chunk &c = *ch_[i];
    int expected = 0;
    if (i == 0 || !c.search_state.compare_exchange_strong(expected, 1)) return;
    const uint64_t t0 = thread_ns();
    uint64_t bit = 0;
    const bool found = find_block(t, i * CHUNK, std::min(n_, (i + 1) * CHUNK), bit);
    c.start_bit = bit;
    c.start_known.store(found ? 1 : -1, std::memory_order_release);
    ns_search_ += thread_ns() - t0;
    std::lock_guard<std::mutex> l(m_);
    cv_.notify_all();
  }

  void finish(Chunk &c)
  {
    std::lock_guard<std::mutex> l(m_);
    c.done = true;
    cv_.notify_all();
  }

  // pass 2: symbols to bytes through the table (literals map to themselves, markers to the window's bytes)
  static void translate(const uint16_t *s, char *o, size_t total, const uint8_t *L)
  {
    size_t k = 0;
    for (; k + 8 <= total; k += 8) {
      o[k] = (char)L[s[k]]; o[k + 1] = (char)L[s[k + 1]]; o[k + 2] = (char)L[s[k + 2]]; o[k + 3] = (char)L[s[k + 3]];
      o[k + 4] = (char)L[s[k + 4]]; o[k + 5] = (char)L[s[k + 5]]; o[k + 6] = (char)L[s[k + 6]]; o[k + 7] = (char)L[s[k + 7]];
    }
    for (; k < total; ++k) o[k] = (char)L[s[k]];
  }
  // ... 32 symbols at a time where none of them is a marker (one pack instead of 32 look-ups)
  __attribute__((target("avx2"))) static void translate_avx2(const uint16_t *s, char *o, size_t total, const uint8_t *L)
  {
    size_t k = 0;
    for (; k + 32 <= total; k += 32) {
      const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k));
      const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k + 16));
      if (_mm256_testz_si256(_mm256_or_si256(a, b), _mm256_set1_epi16((short)0xFF00))) {
        const __m256i p = _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8);
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(o + k), p);
      } else {
        for (size_t j = k; j < k + 32; ++j) o[j] = (char)L[s[j]];
      }
    }
    for (; k < total; ++k) o[k] = (char)L[s[k]];
  }

  static uint64_t thread_ns()
  {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
  }

  void run_chunk(size_t i, gzp::Inflater &z, gzp::Inflater &t)
  {
    Chunk &c = *ch_[i];
    uint64_t t0 = thread_ns();
    // 1. where it starts (searched here, unless a worker searching ahead has done it or is at it)
    if (i > 0) {
      search_chunk(i, t);
      int k;
      while ((k = c.start_known.load(std::memory_order_acquire)) == 0) {
        std::unique_lock<std::mutex> l(m_);
        if (quit_) return;
        cv_.wait_for(l, std::chrono::milliseconds(1));
      }
      const bool found = k == 1;
      if (!found) {
        // nothing to start from in its territory (a long stored block, the tail of the file): the chunk in front inflates through it
        std::unique_lock<std::mutex> l(m_);
        c.absorbed = true;
        c.done = true;
        cv_.notify_all();
        return;
      }
    }
    // 2. pass 1: symbols, up to the start of the next chunk that has one
    t0 = thread_ns();
    z.in.p = p_;
    z.in.n = n_;
    z.in.seek(c.start_bit);
    z.reset_output();
    z.text_only = false;
    z.floor = c.clean_start ? gzp::WIN : 0;
    // What this chunk decides about the chunks behind it (their territory is inflated here: `absorbed`) it may only say once it is
    // AUTHORITATIVE -- once it knows that it starts at a real block start: a member's first block, or the chunk in front of it has
    // arrived exactly at its start and handed it its window.  A chunk decoding from an impostor's start marks nothing: its own
    // output is dropped, and so must be its opinion about others.
    bool authoritative = c.clean_start;
    std::vector<uint8_t> lut(65536, 0);
    for (unsigned k = 0; k < 256; ++k) lut[k] = (uint8_t)k;
    size_t nxt = i + 1;            // the chunk whose start this one is heading for
    size_t marked = i + 1;         // chunks [i + 1, marked) have been told that they are absorbed
    auto mark_absorbed = [&] {     // (m_ held)
      for (; marked < nxt; ++marked) ch_[marked]->absorbed = true;
      cv_.notify_all();
    };
    // the window in front of this chunk's text (the chunk in front of it publishes it); false: absorbed after all, or quitting
    auto become_authoritative = [&]() -> bool {
      std::unique_lock<std::mutex> l(m_);
      cv_.wait(l, [&] { return c.have_window || c.absorbed || quit_; });
      if (quit_) return false;
      if (!c.have_window) { c.done = true; cv_.notify_all(); return false; }   // (the chunk in front went through this territory itself)
      memcpy(lut.data() + 0x8000, c.window_in.data(), gzp::WIN);
      authoritative = true;
      return true;
    };
    bool ends = false;
    for (;;) {
      // arrived at the next chunk's start?
      bool stop = false;
      while (nxt < n_chunks_) {
        Chunk &d = *ch_[nxt];
        if (z.in.pos < 8ull * nxt * CHUNK) break;               // not in its territory yet
        int k;
        // (nobody has looked for its start yet: this thread does -- every other worker may be waiting for a window this chunk is
        //  on its way to deliver)
        if (d.start_known.load(std::memory_order_acquire) == 0) search_chunk(nxt, t);
        while ((k = d.start_known.load(std::memory_order_acquire)) == 0) {
          std::unique_lock<std::mutex> l(m_);
          if (quit_) return;
          cv_.wait_for(l, std::chrono::milliseconds(1));
        }
        if (k == 1 && d.start_bit == z.in.pos) { stop = true; break; }
        if (k == 1 && d.start_bit > z.in.pos) break;            // still in front of it
        // behind its start without having met it (an impostor), or it has none: this chunk inflates its territory too
        ++nxt;
      }
      if (stop) break;
      // far beyond its own territory (a stretch without block starts): the text goes out in pieces, so that what a chunk holds stays
      // bounded however long the stretch is -- gzread needs constant memory for the same stream
      if (z.n - gzp::WIN >= piece_syms_) {
        ns_pass1_ += thread_ns() - t0;
        if (!authoritative && !become_authoritative()) return;
        t0 = thread_ns();
        const size_t total = z.n - gzp::WIN;
        Piece pc{nullptr, total, 0};
        take_buffer(pc.p, pc.cap, total);
        to_bytes(z.out.data() + gzp::WIN, pc.p, total, lut.data());
        // the last WIN symbols, as the bytes they stand for, are the window of what follows: from here on this decoder has seen
        // everything it may refer to (no markers any more)
        {
          const uint16_t *sy = z.out.data() + z.n - gzp::WIN;
          uint16_t *w = z.out.data();
          for (uint32_t k = 0; k < gzp::WIN; ++k) w[k] = lut[sy[k]];
        }
        z.floor = z.floor > total ? z.floor - total : 0;
        z.n = gzp::WIN;
        drop_pages(i * CHUNK, (size_t)(z.in.pos >> 3) > (1u << 16) ? (size_t)(z.in.pos >> 3) - (1u << 16) : 0);
        ns_pass2_ += thread_ns() - t0;
        {
          std::unique_lock<std::mutex> l(m_);
          mark_absorbed();
          cv_.wait(l, [&] { return c.pieces.size() < 2 || quit_; });      // (the consumer takes them in order; two in flight)
          if (quit_) { free(pc.p); return; }
          c.pieces.push_back(pc);
          cv_.notify_all();
        }
        t0 = thread_ns();
      }
      bool fin = false;
      const gzp::Inflater::Rc rc = z.block(fin);
      if (rc != gzp::Inflater::OK) { ends = true; break; }       // corrupt or truncated: the stream ends here, as zlib's would
      if (fin) {
        // end of a member: trailer (CRC-32, ISIZE), then perhaps another member -- with nothing in front of it
        z.in.align_byte();
        size_t at = (size_t)(z.in.pos >> 3) + 8;
        const size_t data = at < n_ ? gzp::gzip_header_end(p_, n_, at) : 0;
        if (!data) { ends = true; break; }                      // (trailing garbage is ignored, as gzread ignores it)
        z.in.seek(8ull * data);
        z.floor = z.n;
      }
    }
    ns_pass1_ += thread_ns() - t0;
    // 3. the window in front of this chunk's text
    if (!authoritative && !become_authoritative()) return;
    // 4. the window for the next chunk, then this chunk's bytes
    t0 = thread_ns();
    const size_t total = z.n - gzp::WIN;
    {
      std::vector<uint8_t> wout;
      if (!ends && nxt < n_chunks_) {
        // the last WIN symbols (markers in front of the text included: a short chunk hands part of its own window on)
        wout.resize(gzp::WIN);
        const uint16_t *sy = z.out.data() + z.n - gzp::WIN;
        for (uint32_t k = 0; k < gzp::WIN; ++k) wout[k] = lut[sy[k]];
      }
      std::lock_guard<std::mutex> l(m_);
      mark_absorbed();
      if (!wout.empty()) {
        Chunk &d = *ch_[nxt];
        d.window_in.swap(wout);
        d.have_window = true;
        cv_.notify_all();
      }
    }
    take_buffer(c.bytes, c.cap_bytes, total);
    c.n_bytes = total;
    to_bytes(z.out.data() + gzp::WIN, c.bytes, total, lut.data());
    c.stream_ends = ends || nxt >= n_chunks_;
    ns_pass2_ += thread_ns() - t0;
    finish(c);
  }

  // the mapped input between two byte offsets is not needed any more: its pages leave the resident set (a long sample would
  // otherwise count its whole compressed file as resident by the end; they come back from the page cache if touched again)
  void drop_pages(size_t from, size_t to)
  {
    const size_t pg = 4096;
    from = (from + pg - 1) & ~(pg - 1);
    to = std::min(to, n_) & ~(pg - 1);
    if (to > from) madvise(const_cast<uint8_t *>(p_) + from, to - from, MADV_DONTNEED);
  }

  // a buffer for `total` bytes of text: one the consumer has given back, if one is large enough (a fresh 20 MB allocation is 5 000
  // page faults), else a new one
  void take_buffer(char *&bytes, size_t &cap, size_t total)
  {
    bytes = nullptr;
    {
      std::lock_guard<std::mutex> l(m_);
      for (size_t k = 0; k < pool_.size(); ++k)
        if (pool_[k].second >= total) {
          bytes = pool_[k].first;
          cap = pool_[k].second;
          pool_[k] = pool_.back();
          pool_.pop_back();
          break;
        }
    }
    if (!bytes) {
      cap = (total ? total : 1) + (total >> 3);
      bytes = static_cast<char *>(malloc(cap));
      if (!bytes) throw std::bad_alloc();
    }
  }
  static void to_bytes(const uint16_t *sy, char *o, size_t total, const uint8_t *L)
  {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) translate_avx2(sy, o, total, L);
    else translate(sy, o, total, L);
  }

  unsigned threads_;
  size_t CHUNK;
  int fd_ = -1;
  const uint8_t *p_ = nullptr;
  size_t n_ = 0, n_chunks_ = 0;
  bool usable_ = false;
  std::vector<std::unique_ptr<Chunk>> ch_;
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  size_t next_chunk_ = 0, consume_ = 0, next_search_ = 1;
  bool quit_ = false;
  char *handed_buf_ = nullptr;                        // the text buffer next() handed out last (guarded by m_)
  size_t handed_cap_ = 0;
  size_t piece_syms_ = (size_t)64 << 20;              // a chunk holding this many symbols hands its text out as a piece and goes on
  std::atomic<bool> failed_{false};
  std::vector<std::pair<char *, size_t>> pool_;       // byte buffers given back by the consumer (guarded by m_)
  std::atomic<uint64_t> ns_search_{0}, ns_pass1_{0}, ns_pass2_{0};
};

}  // namespace shk
