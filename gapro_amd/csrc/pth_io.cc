// Native reader / writer for the reference's on-disk formats (host C++, no Python objects, no GIL).
//
// Replaces, in the gen_ps driver, the `torch.load` of reference gapro/gen_ps.py:45-46 (the scene tuple written by
// ISBNet/dataset/scannetv2/prepare_data_inst.py:104 and the superpoint ids written by prepare_superpoint.py:27) and
// the `torch.save` of gapro/gen_ps.py:132 (the 5-tuple of pseudo-labels).
//
// What those files are: `torch.save(obj)` of a NumPy array or a tuple of NumPy arrays is a STORED (uncompressed) zip
// archive whose first member `<stem>/data.pkl` is a pickle; torch's default pickle protocol is 2, which has no bytes
// type, so every array buffer is written as `_codecs.encode(<text>, 'latin1')` with the buffer as a latin-1 string
// RE-ENCODED AS UTF-8 (a byte >= 0x80 becomes two): a 150k-point scene is 9.6 MB of data in a 13.6 MB pickle, and
// `pickle.load` spends ~25 ms of a core on it while holding the GIL -- which capped the whole host at 270 .. 350
// scenes/s whatever the number of loader processes, below what ONE MI355X generates.  Here the archive
// is mapped, the pickle's opcodes are walked by a small interpreter that understands exactly the objects NumPy's
// __reduce__ emits (numpy.core / numpy._core multiarray._reconstruct, numpy.dtype, _codecs.encode), and the UTF-8
// payload is transcoded back to bytes straight into a caller-given buffer (e.g. pinned staging memory): AVX-512 VBMI2
// (vpcompressb) where the CPU has it, a branch-free scalar loop elsewhere.  Anything unexpected -- a compressed
// member, tensor storages (BINPERSID), object arrays, big-endian data, Fortran order, an opcode outside the set
// below -- returns GAPRO_ERR_UNSUPPORTED and the Python caller falls back to torch.load.
//
// The writer produces the same container (stored zip: <stem>/data.pkl, byteorder, version) with a protocol-2 pickle of
// the same structure, naming `numpy.core.multiarray` (importable under NumPy 1.x AND 2.x -- a file pickled by NumPy 2
// names numpy._core, which the reference's own NumPy 1.x environment cannot import).
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <memory>
#include <utility>
#include <string>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/gapro_hip.h"

namespace {

thread_local std::string t_err;
// resize() without the zero fill: the payload area is overwritten by the encoder right away (a label file is ~3 MB of it)
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  template <class U, class... A> void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
int fail(int code, const std::string& msg) {
  t_err = msg;
  return code;
}

// ---- UTF-8 <-> latin-1 ------------------------------------------------------------------------------------------
// decode: every character is U+0000 .. U+00FF, i.e. a byte < 0x80, or 0xC2 / 0xC3 followed by 0x80 .. 0xBF.
// Returns the number of bytes written, or -1 on malformed input / if it would write more than dst_cap.
typedef long long (*decode_fn)(const unsigned char* src, long long n, unsigned char* dst, long long dst_cap);

long long decode_scalar(const unsigned char* src, long long n, unsigned char* dst, long long dst_cap) {
  const unsigned char* end = src + n;
  unsigned char* d = dst;
  unsigned char* dend = dst + dst_cap;
  unsigned bad = 0;
  // eight ASCII bytes at a time where there are any (label arrays are mostly zeros); otherwise one character per
  // iteration without a data-dependent branch (random mantissa bytes mispredict every other one)
  while (src + 9 <= end && d + 8 <= dend) {
    unsigned long long w;
    memcpy(&w, src, 8);
    if (!(w & 0x8080808080808080ULL)) {
      memcpy(d, &w, 8);
      src += 8;
      d += 8;
      continue;
    }
    for (int i = 0; i < 4 && src + 2 <= end; ++i) {
      const unsigned c = src[0], c1 = src[1];
      const unsigned hi = c >> 7;
      bad |= hi & (((c & 0xFE) != 0xC2) | ((c1 & 0xC0) != 0x80));
      *d++ = (unsigned char)(hi ? ((c << 6) | (c1 & 0x3F)) : c);
      src += 1 + hi;
    }
  }
  while (src < end) {
    const unsigned c = *src;
    if (d >= dend) return -1;
    if (c < 0x80) {
      *d++ = (unsigned char)c;
      ++src;
    } else {
      if (src + 1 >= end) return -1;
      const unsigned c1 = src[1];
      bad |= ((c & 0xFE) != 0xC2) | ((c1 & 0xC0) != 0x80);
      *d++ = (unsigned char)((c << 6) | (c1 & 0x3F));
      src += 2;
    }
  }
  return bad ? -1 : (long long)(d - dst);
}

#if defined(__x86_64__)
// 64 input bytes per iteration: lead bytes (>= 0xC0) are dropped with vpcompressb, a byte that follows 0xC3 gets
// + 0x40 (0xC2 leaves its continuation byte as it is), the last byte's "was a lead / was 0xC3" carries into the next
// block.  Validation in mask arithmetic: continuation bytes are exactly the successors of lead bytes, and the only
// lead bytes are 0xC2 and 0xC3 (>= 0xC4: beyond latin-1; 0xC0 / 0xC1: overlong forms, which decode_scalar, decode_bmi2
// and Python's UTF-8 decoder reject -- ADVICE r04: this tier read 0xC0 0x80 as the byte 0x80).
__attribute__((target("avx512f,avx512bw,avx512vbmi2,popcnt"))) long long decode_avx512(const unsigned char* src,
                                                                                      long long n, unsigned char* dst,
                                                                                      long long dst_cap) {
  const __m512i vC0 = _mm512_set1_epi8((char)0xC0), vC3 = _mm512_set1_epi8((char)0xC3);
  const __m512i vC4 = _mm512_set1_epi8((char)0xC4), v80 = _mm512_set1_epi8((char)0x80);
  const __m512i vC2 = _mm512_set1_epi8((char)0xC2);
  const __m512i v40 = _mm512_set1_epi8(0x40);
  unsigned char* d = dst;
  unsigned char* dend = dst + dst_cap;
  unsigned long long carry_lead = 0, carry_c3 = 0, bad = 0;
  long long i = 0;
  for (; i + 64 <= n && d + 64 <= dend; i += 64) {
    const __m512i v = _mm512_loadu_si512((const void*)(src + i));
    const unsigned long long lead = _mm512_cmpge_epu8_mask(v, vC0);
    const unsigned long long hi = _mm512_cmpge_epu8_mask(v, v80);
    const unsigned long long cont = hi & ~lead;
    const unsigned long long c3 = _mm512_cmpeq_epi8_mask(v, vC3);
    const unsigned long long after_lead = (lead << 1) | carry_lead;
    bad |= (cont ^ after_lead) | _mm512_cmpge_epu8_mask(v, vC4) | (lead & ~_mm512_cmpge_epu8_mask(v, vC2));
    const unsigned long long after_c3 = (c3 << 1) | carry_c3;
    const __m512i fixed = _mm512_mask_add_epi8(v, after_c3, v, v40);
    const unsigned long long keep = ~lead;
    const __m512i packed = _mm512_maskz_compress_epi8(keep, fixed);
    _mm512_storeu_si512((void*)d, packed);
    d += _mm_popcnt_u64(keep);
    carry_lead = lead >> 63;
    carry_c3 = c3 >> 63;
  }
  if (bad) return -1;
  // tail (and the last blocks when dst is nearly full): scalar, starting on a character boundary
  long long start = i;
  if (carry_lead) {  // the block ended on a lead byte: finish that character here
    if (start >= n || d >= dend) return -1;
    const unsigned c1 = src[start];
    if ((c1 & 0xC0) != 0x80) return -1;
    *d++ = (unsigned char)((carry_c3 ? 0xC0 : 0x80) | (c1 & 0x3F));
    ++start;
  }
  const long long r = decode_scalar(src + start, n - start, d, dend - d);
  return r < 0 ? -1 : (long long)(d - dst) + r;
}
#endif

#if defined(__x86_64__)
// The same in 64-bit words for CPUs without AVX-512 VBMI2: byte classes by SWAR arithmetic, the lead bytes squeezed out
// with pext (BMI2).  ~1 input byte per cycle against ~7 cycles per character of the scalar loop.
__attribute__((target("bmi2,popcnt"))) long long decode_bmi2(const unsigned char* src, long long n, unsigned char* dst,
                                                             long long dst_cap) {
  const unsigned long long H = 0x8080808080808080ULL, L7 = 0x7F7F7F7F7F7F7F7FULL;
  unsigned char* d = dst;
  unsigned char* dend = dst + dst_cap;
  unsigned long long carry_lead = 0, carry_c3 = 0, bad = 0;  // carries: 0x80 in byte 0
  long long i = 0;
  for (; i + 8 <= n && d + 8 <= dend; i += 8) {
    unsigned long long w;
    memcpy(&w, src + i, 8);
    const unsigned long long hi = w & H;
    if (!(hi | carry_lead)) {  // eight ASCII bytes
      memcpy(d, &w, 8);
      d += 8;
      continue;
    }
    const unsigned long long lead = hi & (w << 1);  // 0x80 where the byte is >= 0xC0
    const unsigned long long cont = hi & ~lead;
    const unsigned long long after_lead = (lead << 8) | carry_lead;
    const unsigned long long x3 = w ^ 0xC3C3C3C3C3C3C3C3ULL;  // zero bytes <=> 0xC3
    const unsigned long long c3 = ~(((x3 & L7) + L7) | x3 | L7);
    const unsigned long long t = (w & 0x3E3E3E3E3E3E3E3EULL) ^ 0x0202020202020202ULL;  // zero for 0xC2 / 0xC3
    const unsigned long long not_c2c3 = (((t & L7) + L7) | t) & H;
    bad |= (cont ^ after_lead) | (lead & not_c2c3);
    const unsigned long long after_c3 = (c3 << 8) | carry_c3;
    const unsigned long long fixed = w + (after_c3 >> 1);  // + 0x40 in the bytes that follow 0xC3 (no carry: <= 0xFF)
    const unsigned long long keep = ~((lead >> 7) * 0xFFULL);
    const unsigned long long packed = _pext_u64(fixed, keep);
    memcpy(d, &packed, 8);
    d += 8 - _mm_popcnt_u64(lead);
    carry_lead = lead >> 56;
    carry_c3 = c3 >> 56;
  }
  if (bad) return -1;
  long long start = i;
  if (carry_lead) {
    if (start >= n || d >= dend) return -1;
    const unsigned c1 = src[start];
    if ((c1 & 0xC0) != 0x80) return -1;
    *d++ = (unsigned char)((carry_c3 ? 0xC0 : 0x80) | (c1 & 0x3F));
    ++start;
  }
  const long long r = decode_scalar(src + start, n - start, d, dend - d);
  return r < 0 ? -1 : (long long)(d - dst) + r;
}
#endif

const char* g_decoder_name = "scalar";
decode_fn pick_decoder() {
  // GAPRO_PTH_DECODER = scalar | bmi2 | avx512 pins a tier (tests run every tier the CPU has)
  const char* force = getenv("GAPRO_PTH_DECODER");
  const std::string want = force ? force : "";
  if (want == "scalar") {
    g_decoder_name = "scalar";
    return decode_scalar;
  }
#if defined(__x86_64__)
  __builtin_cpu_init();
  const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                      __builtin_cpu_supports("avx512vbmi2");
  const bool hasbmi = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt");
  if (has512 && want != "bmi2") {
    g_decoder_name = "avx512";
    return decode_avx512;
  }
  if (hasbmi) {
    g_decoder_name = "bmi2";
    return decode_bmi2;
  }
#endif
  g_decoder_name = "scalar";
  return decode_scalar;
}

// ---- latin-1 -> UTF-8 (the writer's half of the transcoding) ---------------------------------------------------------
// Returns the encoded length; dst needs 2 n + 64 bytes of room (the vector tiers store whole registers).  Round 5's
// byte loop took 4.6 ms of a core per 150 k-point label file -- half of the host work per scene, which is what caps
// any number of workers under the pool's CPU quota -- while the reader had three SIMD tiers.
typedef long long (*encode_fn)(const unsigned char* src, long long n, unsigned char* dst);

long long encode_scalar(const unsigned char* src, long long n, unsigned char* dst) {
  unsigned char* d = dst;
  for (long long i = 0; i < n; ++i) {
    const unsigned b = src[i];
    const unsigned hi = b >> 7;
    d[0] = (unsigned char)(hi ? (0xC0 | (b >> 6)) : b);
    d[1] = (unsigned char)(0x80 | (b & 0x3F));
    d += 1 + hi;
  }
  return (long long)(d - dst);
}

#if defined(__x86_64__)
// 32 input bytes per iteration, the mirror of decode_avx512: every byte b becomes the 16-bit lane
// [b < 0x80 ? b : 0xC0 | b >> 6,  0x80 | (b & 0x3F)], and vpcompressb keeps the first byte of every lane and the second
// one where b >= 0x80.  64 ASCII bytes (label arrays are mostly small integers) are copied as they are.
__attribute__((target("avx512f,avx512bw,avx512vbmi2,bmi2,popcnt"))) long long encode_avx512(const unsigned char* src,
                                                                                           long long n,
                                                                                           unsigned char* dst) {
  unsigned char* d = dst;
  long long i = 0;
  const __m512i v3f = _mm512_set1_epi16(0x3F), v80hi = _mm512_set1_epi16((short)0x8000), vc0 = _mm512_set1_epi16(0xC0);
  const __m512i v7f = _mm512_set1_epi16(0x7F);
  while (i + 64 <= n) {
    const __m512i v = _mm512_loadu_si512((const void*)(src + i));
    const unsigned long long hi = (unsigned long long)_mm512_movepi8_mask(v);
    if (!hi) {
      _mm512_storeu_si512((void*)d, v);
      d += 64;
      i += 64;
      continue;
    }
    for (int half = 0; half < 2; ++half) {
      const __m256i h = _mm256_loadu_si256((const __m256i*)(src + i + 32 * half));
      const unsigned hm = (unsigned)(hi >> (32 * half));
      const __m512i x = _mm512_cvtepu8_epi16(h);                         // [b, 0] per lane
      const __mmask32 big = _mm512_cmpgt_epu16_mask(x, v7f);
      const __m512i lead = _mm512_or_si512(vc0, _mm512_srli_epi16(x, 6));
      const __m512i lo = _mm512_mask_blend_epi16(big, x, lead);
      const __m512i second = _mm512_or_si512(v80hi, _mm512_slli_epi16(_mm512_and_si512(x, v3f), 8));
      const __m512i lanes = _mm512_or_si512(lo, second);
      const unsigned long long keep = 0x5555555555555555ULL | _pdep_u64((unsigned long long)hm, 0xAAAAAAAAAAAAAAAAULL);
      _mm512_storeu_si512((void*)d, _mm512_maskz_compress_epi8(keep, lanes));
      d += 32 + _mm_popcnt_u32(hm);
    }
    i += 64;
  }
  return (long long)(d - dst) + encode_scalar(src + i, n - i, d);
}

// The same four bytes at a time in a 64-bit word for CPUs without AVX-512 VBMI2: pdep spreads the bytes into 16-bit
// lanes, the two output bytes of every lane are built by SWAR arithmetic, pext drops the second byte of ASCII lanes.
__attribute__((target("bmi2,popcnt"))) long long encode_bmi2(const unsigned char* src, long long n, unsigned char* dst) {
  const unsigned long long LO = 0x00FF00FF00FF00FFULL;
  unsigned char* d = dst;
  long long i = 0;
  for (; i + 8 <= n; i += 8) {
    unsigned long long w;
    memcpy(&w, src + i, 8);
    if (!(w & 0x8080808080808080ULL)) {
      memcpy(d, &w, 8);
      d += 8;
      continue;
    }
    for (int half = 0; half < 2; ++half) {
      const unsigned long long x = _pdep_u64((w >> (32 * half)) & 0xFFFFFFFFULL, LO);
      const unsigned long long h = x & 0x0080008000800080ULL;
      const unsigned long long hm = (h >> 7) * 0xFFULL;  // 0x00FF in the lanes of bytes >= 0x80
      const unsigned long long lead = ((x >> 6) & 0x0003000300030003ULL) | 0x00C000C000C000C0ULL;
      const unsigned long long first = (x & ~hm) | (lead & hm);
      const unsigned long long second = ((x & 0x003F003F003F003FULL) | 0x0080008000800080ULL) << 8;
      const unsigned long long packed = _pext_u64(first | second, LO | (hm << 8));
      memcpy(d, &packed, 8);
      d += 4 + _mm_popcnt_u64(h);
    }
  }
  return (long long)(d - dst) + encode_scalar(src + i, n - i, d);
}
#endif

const char* g_encoder_name = "scalar";
encode_fn pick_encoder() {
  // GAPRO_PTH_ENCODER = scalar | bmi2 | avx512 pins a tier (tests run every tier the CPU has)
  const char* force = getenv("GAPRO_PTH_ENCODER");
  const std::string want = force ? force : "";
  if (want != "scalar") {
#if defined(__x86_64__)
    __builtin_cpu_init();
    const bool hasbmi = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt");
    const bool has512 = hasbmi && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                        __builtin_cpu_supports("avx512vbmi2");
    if (has512 && want != "bmi2") {
      g_encoder_name = "avx512";
      return encode_avx512;
    }
    if (hasbmi) {
      g_encoder_name = "bmi2";
      return encode_bmi2;
    }
#endif
  }
  g_encoder_name = "scalar";
  return encode_scalar;
}
long long encode_utf8(const unsigned char* src, long long n, unsigned char* dst) {
  static const encode_fn enc = pick_encoder();
  return (getenv("GAPRO_PTH_ENCODER") ? pick_encoder() : enc)(src, n, dst);
}

// ---- CRC-32 (zip) ---------------------------------------------------------------------------------------------------
// slice-by-8 tables (the portable tier and the tail of the other one)
struct CrcTable {
  unsigned t[8][256];
  CrcTable() {
    for (unsigned i = 0; i < 256; ++i) {
      unsigned c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (unsigned i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
unsigned crc32_table(const unsigned char* p, size_t n, unsigned c) {  // c: running state (already inverted)
  static const CrcTable T;
  while (n >= 8) {
    unsigned long long w;
    memcpy(&w, p, 8);
    w ^= c;
    c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
        T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][(w >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c;
}

#if defined(__x86_64__)
// Carry-less-multiply folding (Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ", the
// bit-reflected form for the zip polynomial 0xEDB88320): four 128-bit lanes folded across 64 bytes per iteration with
// x^(512+-32) mod P, then 4 -> 1 lanes with x^(128+-32), 128 -> 64 -> 32 bits by one more fold and a Barrett
// reduction.  n >= 64 and a multiple of 16; returns the running state.  Held to the table tier on every length by
// tests/test_pth_io.py.
__attribute__((target("pclmul,sse4.1"))) unsigned crc32_clmul(const unsigned char* buf, size_t len, unsigned crc) {
  const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596LL, 0x0154442bd4LL);
  const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009eLL, 0x01751997d0LL);
  const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124LL);
  const __m128i poly = _mm_set_epi64x(0x01f7011641LL, 0x01db710641LL);
  __m128i x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00)), x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
  __m128i x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20)), x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  buf += 64;
  len -= 64;
  while (len >= 64) {
    const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
    const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
    x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
    x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
    x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i*)(buf + 0x00)));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i*)(buf + 0x10)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i*)(buf + 0x20)));
    x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i*)(buf + 0x30)));
    buf += 64;
    len -= 64;
  }
  // (a macro, not a lambda: a lambda does not inherit this function's target attribute)
#define GAPRO_CRC_FOLD(acc, next) \
  _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k3k4, 0x11), _mm_clmulepi64_si128(acc, k3k4, 0x00)), next)
  x1 = GAPRO_CRC_FOLD(x1, x2);
  x1 = GAPRO_CRC_FOLD(x1, x3);
  x1 = GAPRO_CRC_FOLD(x1, x4);
  while (len >= 16) {
    x1 = GAPRO_CRC_FOLD(x1, _mm_loadu_si128((const __m128i*)buf));
    buf += 16;
    len -= 16;
  }
#undef GAPRO_CRC_FOLD
  // 128 -> 64 bits
  const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
  __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
  t = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, mask32);
  x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), t);
  // Barrett reduction to 32 bits
  t = _mm_and_si128(x1, mask32);
  t = _mm_clmulepi64_si128(t, poly, 0x10);
  t = _mm_and_si128(t, mask32);
  t = _mm_clmulepi64_si128(t, poly, 0x00);
  x1 = _mm_xor_si128(x1, t);
  return (unsigned)_mm_extract_epi32(x1, 1);
}
#endif

const char* g_crc_name = "table";
bool pick_crc_clmul() {
  // GAPRO_PTH_CRC = table | clmul pins a tier
  const char* force = getenv("GAPRO_PTH_CRC");
  if (force && std::string(force) == "table") {
    g_crc_name = "table";
    return false;
  }
#if defined(__x86_64__)
  __builtin_cpu_init();
  if (__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1")) {
    g_crc_name = "clmul";
    return true;
  }
#endif
  g_crc_name = "table";
  return false;
}
unsigned crc32_buf(const unsigned char* p, size_t n) {
  static const bool clmul_default = pick_crc_clmul();
  const bool clmul = getenv("GAPRO_PTH_CRC") ? pick_crc_clmul() : clmul_default;
  unsigned c = 0xFFFFFFFFu;
#if defined(__x86_64__)
  if (clmul && n >= 64) {
    const size_t m = n & ~(size_t)15;
    c = crc32_clmul(p, m, c);
    p += m;
    n -= m;
  }
#else
  (void)clmul;
#endif
  return crc32_table(p, n, c) ^ 0xFFFFFFFFu;
}

struct Val;
typedef std::shared_ptr<Val> VP;
enum Kind { V_NONE, V_BOOL, V_INT, V_STR, V_BYTES, V_ENCBYTES, V_GLOBAL, V_TUPLE, V_LIST, V_MARK, V_DTYPE, V_NDARRAY };
struct Val {
  Kind k = V_NONE;
  long long i = 0;                  // V_BOOL / V_INT
  const unsigned char* p = nullptr; // V_STR (UTF-8), V_BYTES (raw), V_ENCBYTES (UTF-8 of latin-1 text)
  long long n = 0;
  std::string mod, name;            // V_GLOBAL
  std::vector<VP> items;            // V_TUPLE / V_LIST
  // V_DTYPE
  char dkind = 0;
  int itemsize = 0;
  bool dtype_built = false;
  // V_NDARRAY
  std::vector<long long> shape;
  VP dtype, data;
  bool built = false;
};
VP mk(Kind k) {
  VP v = std::make_shared<Val>();
  v->k = k;
  return v;
}
bool str_is(const VP& v, const char* s) {
  return v && v->k == V_STR && (long long)strlen(s) == v->n && memcmp(v->p, s, (size_t)v->n) == 0;
}

struct Array {
  char kind;
  int itemsize;
  std::vector<long long> shape;
  const unsigned char* payload;
  long long payload_len;
  bool encoded;
  long long nbytes;
};

int unsupported(const std::string& what) { return fail(GAPRO_ERR_UNSUPPORTED, "pth: " + what); }

// numpy.dtype(str, False, True)
int make_dtype(const VP& args, VP* out) {
  if (args->k != V_TUPLE || args->items.size() != 3 || args->items[0]->k != V_STR)
    return unsupported("numpy.dtype with unexpected arguments");
  const VP& s = args->items[0];
  if (s->n < 2 || s->n > 3) return unsupported("dtype string");
  const char kind = (char)s->p[0];
  int size = 0;
  for (long long j = 1; j < s->n; ++j) {
    if (s->p[j] < '0' || s->p[j] > '9') return unsupported("dtype string");
    size = size * 10 + (s->p[j] - '0');
  }
  if (!(kind == 'f' || kind == 'i' || kind == 'u' || kind == 'b') || size < 1 || size > 8 || (size & (size - 1)))
    return unsupported(std::string("dtype kind '") + kind + "'");
  VP d = mk(V_DTYPE);
  d->dkind = kind;
  d->itemsize = size;
  *out = d;
  return GAPRO_OK;
}

int build_dtype(const VP& d, const VP& st) {
  // (3, '<' | '|', None, None, None, -1, -1, 0)
  if (st->k != V_TUPLE || st->items.size() != 8 || st->items[0]->k != V_INT || st->items[0]->i != 3)
    return unsupported("dtype state");
  const VP& bo = st->items[1];
  if (!(str_is(bo, "<") || str_is(bo, "|"))) return unsupported("byte order (only little-endian data is read natively)");
  for (int j = 2; j <= 4; ++j)
    if (st->items[j]->k != V_NONE) return unsupported("structured / sub-array dtype");
  d->dtype_built = true;
  return GAPRO_OK;
}

int build_ndarray(const VP& a, const VP& st) {
  // (1, shape, dtype, is_fortran, data)
  if (st->k != V_TUPLE || st->items.size() != 5 || st->items[0]->k != V_INT || st->items[0]->i != 1)
    return unsupported("ndarray state");
  const VP& sh = st->items[1];
  if (sh->k != V_TUPLE || sh->items.size() > 4) return unsupported("ndarray shape");
  a->shape.clear();
  for (const VP& e : sh->items) {
    if (e->k != V_INT || e->i < 0) return unsupported("ndarray shape");
    a->shape.push_back(e->i);
  }
  if (st->items[2]->k != V_DTYPE || !st->items[2]->dtype_built) return unsupported("ndarray dtype");
  if (st->items[3]->k != V_BOOL) return unsupported("ndarray order flag");
  long long count = 1;
  for (long long s : a->shape) count *= s;
  if (st->items[3]->i && a->shape.size() > 1 && count > 0) return unsupported("Fortran-ordered array");
  const VP& d = st->items[4];
  if (d->k != V_BYTES && d->k != V_ENCBYTES) return unsupported("ndarray payload (object array?)");
  a->dtype = st->items[2];
  a->data = d;
  a->built = true;
  return GAPRO_OK;
}

// Walk the pickle; on success `top` is the unpickled value (arrays reference the payload in place).
int unpickle(const unsigned char* p, long long n, VP* top) {
  std::vector<VP> stack, memo;
  long long pos = 0;
  auto need = [&](long long k) { return pos + k <= n; };
  auto rd_u = [&](int bytes) {
    unsigned long long v = 0;
    for (int b = 0; b < bytes; ++b) v |= (unsigned long long)p[pos + b] << (8 * b);
    pos += bytes;
    return v;
  };
  auto memo_put = [&](size_t idx) {
    if (idx >= memo.size()) memo.resize(idx + 1);
    memo[idx] = stack.back();
  };
  auto pop_mark = [&](std::vector<VP>* out) -> bool {
    size_t m = stack.size();
    while (m > 0 && stack[m - 1]->k != V_MARK) --m;
    if (m == 0) return false;
    out->assign(stack.begin() + m, stack.end());
    stack.resize(m - 1);
    return true;
  };
  VP vnone = mk(V_NONE);
  while (pos < n) {
    const unsigned char op = p[pos++];
    switch (op) {
      case 0x80:  // PROTO
        if (!need(1)) return unsupported("truncated pickle");
        if (p[pos] < 2 || p[pos] > 5) return unsupported("pickle protocol " + std::to_string(p[pos]));
        ++pos;
        break;
      case 0x95:  // FRAME
        if (!need(8)) return unsupported("truncated pickle");
        pos += 8;
        break;
      case '(': stack.push_back(mk(V_MARK)); break;
      case 'N': stack.push_back(vnone); break;
      case 0x88: case 0x89: {
        VP v = mk(V_BOOL);
        v->i = op == 0x88;
        stack.push_back(v);
        break;
      }
      case 'K': case 'M': case 'J': {
        const int nb = op == 'K' ? 1 : op == 'M' ? 2 : 4;
        if (!need(nb)) return unsupported("truncated pickle");
        VP v = mk(V_INT);
        const unsigned long long u = rd_u(nb);
        v->i = op == 'J' ? (long long)(int)(unsigned)u : (long long)u;
        stack.push_back(v);
        break;
      }
      case 0x8a: {  // LONG1
        if (!need(1)) return unsupported("truncated pickle");
        const int nb = p[pos++];
        if (nb > 8 || !need(nb)) return unsupported("LONG1 beyond 64 bits");
        unsigned long long u = rd_u(nb);
        if (nb > 0 && nb < 8 && (u >> (8 * nb - 1)) & 1) u |= ~0ULL << (8 * nb);  // sign-extend
        VP v = mk(V_INT);
        v->i = (long long)u;
        stack.push_back(v);
        break;
      }
      case 'X': case 0x8c: case 0x8d: case 'B': case 'C': case 0x8e: {
        const bool is_str = op == 'X' || op == 0x8c || op == 0x8d;
        const int nb = (op == 0x8c || op == 'C') ? 1 : (op == 'X' || op == 'B') ? 4 : 8;
        if (!need(nb)) return unsupported("truncated pickle");
        const unsigned long long len = rd_u(nb);
        if (len > (unsigned long long)(n - pos)) return unsupported("truncated pickle");
        VP v = mk(is_str ? V_STR : V_BYTES);
        v->p = p + pos;
        v->n = (long long)len;
        pos += (long long)len;
        stack.push_back(v);
        break;
      }
      case 'c': {  // GLOBAL module\nname\n
        VP v = mk(V_GLOBAL);
        for (int part = 0; part < 2; ++part) {
          const long long s0 = pos;
          while (pos < n && p[pos] != '\n') ++pos;
          if (pos >= n) return unsupported("truncated pickle");
          (part == 0 ? v->mod : v->name).assign((const char*)p + s0, (size_t)(pos - s0));
          ++pos;
        }
        stack.push_back(v);
        break;
      }
      case 0x93: {  // STACK_GLOBAL
        if (stack.size() < 2 || stack[stack.size() - 1]->k != V_STR || stack[stack.size() - 2]->k != V_STR)
          return unsupported("STACK_GLOBAL operands");
        VP v = mk(V_GLOBAL);
        v->name.assign((const char*)stack.back()->p, (size_t)stack.back()->n);
        stack.pop_back();
        v->mod.assign((const char*)stack.back()->p, (size_t)stack.back()->n);
        stack.pop_back();
        stack.push_back(v);
        break;
      }
      case 'q': case 'r': {  // BINPUT / LONG_BINPUT
        const int nb = op == 'q' ? 1 : 4;
        if (!need(nb) || stack.empty()) return unsupported("BINPUT");
        memo_put((size_t)rd_u(nb));
        break;
      }
      case 0x94:  // MEMOIZE
        if (stack.empty()) return unsupported("MEMOIZE");
        memo.push_back(stack.back());
        break;
      case 'h': case 'j': {  // BINGET / LONG_BINGET
        const int nb = op == 'h' ? 1 : 4;
        if (!need(nb)) return unsupported("truncated pickle");
        const size_t idx = (size_t)rd_u(nb);
        if (idx >= memo.size() || !memo[idx]) return unsupported("BINGET of an unset memo slot");
        stack.push_back(memo[idx]);
        break;
      }
      case ')': stack.push_back(mk(V_TUPLE)); break;
      case ']': stack.push_back(mk(V_LIST)); break;
      case 0x85: case 0x86: case 0x87: {
        const size_t cnt = op - 0x84;
        if (stack.size() < cnt) return unsupported("TUPLEn");
        VP t = mk(V_TUPLE);
        t->items.assign(stack.end() - cnt, stack.end());
        stack.resize(stack.size() - cnt);
        stack.push_back(t);
        break;
      }
      case 't': case 'l': {
        VP t = mk(op == 't' ? V_TUPLE : V_LIST);
        if (!pop_mark(&t->items)) return unsupported("TUPLE without MARK");
        stack.push_back(t);
        break;
      }
      case 'a': {  // APPEND
        if (stack.size() < 2 || stack[stack.size() - 2]->k != V_LIST) return unsupported("APPEND");
        VP v = stack.back();
        stack.pop_back();
        stack.back()->items.push_back(v);
        break;
      }
      case 'e': {  // APPENDS
        std::vector<VP> its;
        if (!pop_mark(&its) || stack.empty() || stack.back()->k != V_LIST) return unsupported("APPENDS");
        for (VP& v : its) stack.back()->items.push_back(v);
        break;
      }
      case 'R': {  // REDUCE
        if (stack.size() < 2) return unsupported("REDUCE");
        VP args = stack.back();
        stack.pop_back();
        VP fn = stack.back();
        stack.pop_back();
        if (fn->k != V_GLOBAL || args->k != V_TUPLE) return unsupported("REDUCE of a non-global");
        if (fn->mod == "_codecs" && fn->name == "encode") {
          if (args->items.size() != 2 || args->items[0]->k != V_STR || !str_is(args->items[1], "latin1"))
            return unsupported("_codecs.encode with unexpected arguments");
          VP v = mk(V_ENCBYTES);
          v->p = args->items[0]->p;
          v->n = args->items[0]->n;
          stack.push_back(v);
        } else if (fn->mod == "numpy" && fn->name == "dtype") {
          VP d;
          const int rc = make_dtype(args, &d);
          if (rc != GAPRO_OK) return rc;
          stack.push_back(d);
        } else if ((fn->mod == "numpy.core.multiarray" || fn->mod == "numpy._core.multiarray") &&
                   fn->name == "_reconstruct") {
          if (args->items.size() != 3 || args->items[0]->k != V_GLOBAL || args->items[0]->mod != "numpy" ||
              args->items[0]->name != "ndarray")
            return unsupported("_reconstruct of a ndarray subclass");
          stack.push_back(mk(V_NDARRAY));
        } else {
          return unsupported("callable " + fn->mod + "." + fn->name);
        }
        break;
      }
      case 'b': {  // BUILD
        if (stack.size() < 2) return unsupported("BUILD");
        VP st = stack.back();
        stack.pop_back();
        VP obj = stack.back();
        int rc;
        if (obj->k == V_DTYPE) rc = build_dtype(obj, st);
        else if (obj->k == V_NDARRAY) rc = build_ndarray(obj, st);
        else return unsupported("BUILD of an unexpected object");
        if (rc != GAPRO_OK) return rc;
        break;
      }
      case 'Q': return unsupported("persistent id (a torch tensor storage): not a NumPy payload");
      case '.':
        if (stack.size() != 1) return unsupported("pickle ended with a stack of " + std::to_string(stack.size()));
        *top = stack.back();
        return GAPRO_OK;
      default: {
        char buf[64];
        snprintf(buf, sizeof(buf), "pickle opcode 0x%02x at %lld", op, pos - 1);
        return unsupported(buf);
      }
    }
  }
  return unsupported("pickle without STOP");
}

int to_array(const VP& v, Array* a) {
  if (v->k != V_NDARRAY || !v->built) return unsupported("a member that is not a NumPy array");
  a->kind = v->dtype->dkind;
  a->itemsize = v->dtype->itemsize;
  a->shape = v->shape;
  long long cnt = 1;
  for (long long s : a->shape) cnt *= s;
  a->nbytes = cnt * a->itemsize;
  a->payload = v->data->p;
  a->payload_len = v->data->n;
  a->encoded = v->data->k == V_ENCBYTES;
  if (!a->encoded && a->payload_len != a->nbytes) return unsupported("payload length does not match the shape");
  if (a->encoded && (a->payload_len < a->nbytes || a->payload_len > 2 * a->nbytes))
    return unsupported("payload length does not match the shape");
  return GAPRO_OK;
}

unsigned rd16(const unsigned char* p) { return p[0] | (p[1] << 8); }
unsigned rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); }
unsigned long long rd64(const unsigned char* p) { return rd32(p) | ((unsigned long long)rd32(p + 4) << 32); }

}  // namespace

struct gapro_pth_file {
  int fd = -1;
  const unsigned char* map = nullptr;
  size_t size = 0;
  bool owned = false;        // map is a malloc'd copy of the file (read), not a mapping
  bool is_sequence = false;  // tuple / list of arrays (false: one bare array)
  std::vector<Array> arrays;
  decode_fn decode = nullptr;
};

extern "C" {

const char* gapro_pth_last_error(void) { return t_err.c_str(); }

const char* gapro_pth_decoder(void) {
  (void)pick_decoder();
  return g_decoder_name;
}

const char* gapro_pth_encoder(void) {
  (void)pick_encoder();
  return g_encoder_name;
}

const char* gapro_pth_crc(void) {
  (void)pick_crc_clmul();
  return g_crc_name;
}

uint32_t gapro_pth_crc32(const void* data, int64_t n) { return data && n >= 0 ? crc32_buf((const unsigned char*)data, (size_t)n) : 0; }

int64_t gapro_pth_encode_latin1(const void* src, int64_t n, void* dst, int64_t dst_cap) {
  if (!src || !dst || n < 0 || dst_cap < 2 * n + 64) return GAPRO_ERR_BAD_ARG;
  return encode_utf8((const unsigned char*)src, n, (unsigned char*)dst);
}

void gapro_pth_close(gapro_pth_file* f) {
  if (!f) return;
  if (f->map && f->owned) free((void*)f->map);
  else if (f->map) munmap((void*)f->map, f->size);
  if (f->fd >= 0) close(f->fd);
  delete f;
}

int gapro_pth_open(const char* path, gapro_pth_file** out) {
  if (!path || !out) return fail(GAPRO_ERR_BAD_ARG, "gapro_pth_open: bad argument");
  *out = nullptr;
  std::unique_ptr<gapro_pth_file, void (*)(gapro_pth_file*)> f(new (std::nothrow) gapro_pth_file(), gapro_pth_close);
  if (!f) return fail(GAPRO_ERR_OOM, "gapro_pth_open: out of memory");
  f->fd = open(path, O_RDONLY | O_CLOEXEC);
  if (f->fd < 0) return fail(GAPRO_ERR_IO, std::string("gapro_pth_open: ") + path + ": " + strerror(errno));
  struct stat st;
  if (fstat(f->fd, &st) != 0 || st.st_size < 22)
    return fail(GAPRO_ERR_IO, std::string("gapro_pth_open: ") + path + ": not a zip archive (too short)");
  f->size = (size_t)st.st_size;
  // The file is READ into a heap buffer by default, not mapped: mmap / munmap take the address-space lock of the whole
  // process and every first touch of a mapped page is a fault under it -- with 8 .. 16 loader threads per worker
  // process that is what the threads queue on (GAPRO_PTH_MAP=1 maps instead: one thread, or files beyond memory).
  static const bool use_map = [] { const char* e = getenv("GAPRO_PTH_MAP"); return e && e[0] == '1'; }();
  if (use_map) {
    void* m = mmap(nullptr, f->size, PROT_READ, MAP_PRIVATE, f->fd, 0);
    if (m == MAP_FAILED) {
      f->map = nullptr;
      return fail(GAPRO_ERR_IO, std::string("gapro_pth_open: mmap: ") + strerror(errno));
    }
    f->map = (const unsigned char*)m;
    (void)madvise(m, f->size, MADV_SEQUENTIAL);
  } else {
    unsigned char* buf = (unsigned char*)malloc(f->size);
    if (!buf) return fail(GAPRO_ERR_OOM, "gapro_pth_open: out of memory");
    f->map = buf;
    f->owned = true;
    size_t got = 0;
    while (got < f->size) {
      const ssize_t r = pread(f->fd, buf + got, f->size - got, (off_t)got);
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) return fail(GAPRO_ERR_IO, std::string("gapro_pth_open: read: ") + (r < 0 ? strerror(errno) : "short file"));
      got += (size_t)r;
    }
    close(f->fd);
    f->fd = -1;
  }
  const unsigned char* z = f->map;
  const size_t n = f->size;
  // end of central directory: scan back over a possible archive comment
  long long eocd = -1;
  for (long long o = (long long)n - 22; o >= 0 && o >= (long long)n - 22 - 65535; --o)
    if (rd32(z + o) == 0x06054b50u) {
      eocd = o;
      break;
    }
  if (eocd < 0) return unsupported("no zip end-of-central-directory record (a legacy torch.save file?)");
  unsigned long long cd_off = rd32(z + eocd + 16), cd_entries = rd16(z + eocd + 10);
  if (cd_off == 0xFFFFFFFFu || cd_entries == 0xFFFF) {  // zip64
    if (eocd < 20 || rd32(z + eocd - 20) != 0x07064b50u) return unsupported("zip64 locator missing");
    const unsigned long long e64 = rd64(z + eocd - 20 + 8);
    if (e64 + 56 > n || rd32(z + e64) != 0x06064b50u) return unsupported("zip64 end record");
    cd_entries = rd64(z + e64 + 32);
    cd_off = rd64(z + e64 + 48);
  }
  // the member `<stem>/data.pkl`
  unsigned long long o = cd_off, pkl_off = 0, pkl_size = 0;
  bool found = false;
  for (unsigned long long e = 0; e < cd_entries; ++e) {
    if (o + 46 > n || rd32(z + o) != 0x02014b50u) return unsupported("zip central directory");
    const unsigned method = rd16(z + o + 10), nlen = rd16(z + o + 28), xlen = rd16(z + o + 30), clen = rd16(z + o + 32);
    unsigned long long csize = rd32(z + o + 20), usize = rd32(z + o + 24), lho = rd32(z + o + 42);
    if (o + 46 + nlen + xlen + clen > n) return unsupported("zip central directory");
    const char* name = (const char*)z + o + 46;
    const bool is_pkl = (nlen == 8 && memcmp(name, "data.pkl", 8) == 0) ||
                        (nlen > 9 && memcmp(name + nlen - 9, "/data.pkl", 9) == 0);
    if (is_pkl) {
      if (csize == 0xFFFFFFFFu || usize == 0xFFFFFFFFu || lho == 0xFFFFFFFFu) {  // zip64 extra field 0x0001
        const unsigned char* x = z + o + 46 + nlen;
        unsigned xo = 0;
        bool ok = false;
        while (xo + 4 <= xlen) {
          const unsigned id = rd16(x + xo), sz = rd16(x + xo + 2);
          if (id == 1) {
            unsigned q = xo + 4;
            if (usize == 0xFFFFFFFFu) { usize = rd64(x + q); q += 8; }
            if (csize == 0xFFFFFFFFu) { csize = rd64(x + q); q += 8; }
            if (lho == 0xFFFFFFFFu) { lho = rd64(x + q); q += 8; }
            ok = true;
            break;
          }
          xo += 4 + sz;
        }
        if (!ok) return unsupported("zip64 extra field");
      }
      if (method != 0 || csize != usize) return unsupported("compressed data.pkl");
      if (lho + 30 > n || rd32(z + lho) != 0x04034b50u) return unsupported("zip local header");
      const unsigned long long data = lho + 30 + rd16(z + lho + 26) + rd16(z + lho + 28);
      if (data + usize > n) return unsupported("zip member beyond the end of the file");
      pkl_off = data;
      pkl_size = usize;
      found = true;
      break;
    }
    o += 46 + nlen + xlen + clen;
  }
  if (!found) return unsupported("no data.pkl in the archive");
  VP top;
  int rc = unpickle(z + pkl_off, (long long)pkl_size, &top);
  if (rc != GAPRO_OK) return rc;
  if (top->k == V_NDARRAY) {
    Array a;
    if ((rc = to_array(top, &a)) != GAPRO_OK) return rc;
    f->arrays.push_back(a);
  } else if (top->k == V_TUPLE || top->k == V_LIST) {
    f->is_sequence = true;
    for (const VP& v : top->items) {
      Array a;
      if ((rc = to_array(v, &a)) != GAPRO_OK) return rc;
      f->arrays.push_back(a);
    }
  } else {
    return unsupported("top-level object is neither an array nor a sequence of arrays");
  }
  static const decode_fn dec = pick_decoder();
  f->decode = getenv("GAPRO_PTH_DECODER") ? pick_decoder() : dec;
  *out = f.release();
  return GAPRO_OK;
}

int gapro_pth_count(const gapro_pth_file* f) { return f ? (int)f->arrays.size() : GAPRO_ERR_BAD_ARG; }

int gapro_pth_is_sequence(const gapro_pth_file* f) { return f ? (f->is_sequence ? 1 : 0) : GAPRO_ERR_BAD_ARG; }

int gapro_pth_info(const gapro_pth_file* f, int32_t index, gapro_pth_array* out) {
  if (!f || !out || index < 0 || index >= (int)f->arrays.size()) return fail(GAPRO_ERR_BAD_ARG, "gapro_pth_info: bad argument");
  const Array& a = f->arrays[index];
  out->kind = a.kind;
  out->itemsize = a.itemsize;
  out->ndim = (int32_t)a.shape.size();
  out->encoded = a.encoded ? 1 : 0;
  for (int d = 0; d < 4; ++d) out->shape[d] = d < (int)a.shape.size() ? a.shape[d] : 1;
  out->nbytes = a.nbytes;
  return GAPRO_OK;
}

int gapro_pth_read(const gapro_pth_file* f, int32_t index, void* h_dst, int64_t dst_bytes) {
  if (!f || index < 0 || index >= (int)f->arrays.size() || (!h_dst && dst_bytes > 0))
    return fail(GAPRO_ERR_BAD_ARG, "gapro_pth_read: bad argument");
  const Array& a = f->arrays[index];
  if (dst_bytes != a.nbytes) return fail(GAPRO_ERR_BAD_ARG, "gapro_pth_read: the buffer must have exactly nbytes");
  if (a.nbytes == 0) return GAPRO_OK;
  if (!a.encoded) {
    memcpy(h_dst, a.payload, (size_t)a.nbytes);
    return GAPRO_OK;
  }
  const long long got = f->decode(a.payload, a.payload_len, (unsigned char*)h_dst, a.nbytes);
  if (got != a.nbytes) return fail(GAPRO_ERR_IO, "gapro_pth_read: malformed UTF-8 payload (or length mismatch)");
  return GAPRO_OK;
}

// ---- default features (gen_ps.py:55) ---------------------------------------------------------------------------
// feats = np.concatenate([xyz, rgb], -1) cast to float32 (gen_ps.py:55 builds the float64 concatenation, :84 uploads it
// as a float tensor): one pass, each element rounded float64 -> float32 exactly as the cast does.
int gapro_scene_default_feats(const double* h_xyz, const double* h_rgb, int64_t n_points, float* h_feats) {
  if (n_points < 0 || (n_points > 0 && (!h_xyz || !h_rgb || !h_feats)))
    return fail(GAPRO_ERR_BAD_ARG, "gapro_scene_default_feats: bad argument");
  for (int64_t i = 0; i < n_points; ++i) {
    float* o = h_feats + 6 * i;
    const double* a = h_xyz + 3 * i;
    const double* b = h_rgb + 3 * i;
    o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)a[2];
    o[3] = (float)b[0]; o[4] = (float)b[1]; o[5] = (float)b[2];
  }
  return GAPRO_OK;
}

// ---- GT instance boxes on the host (gen_ps_utils.py:195-239 getInstanceInfo, without the corner labels) ---------------
// One pass over the points.  The loader threads of the gen_ps driver call this before the upload: the device form
// (gapro_instance_info) is a kernel, and a short kernel is not dispatched beside a running fit launch -- every loader
// thread waited for the launch to drain once per scene, which starved the pipeline (round 4: 115 scenes/s of loading
// beside a running launch against 850 without).  Box index = rank among the non-empty instance ids; class = semantic
// label of the instance's first point (ScanNet: shifted by -2 unless -100); volume = prod(clip(max - min, 0)).
// Returns GAPRO_OK with *n_boxes (0: no instance), or GAPRO_ERR_BAD_ARG with *instance_num set when cap is too small.
int gapro_scene_instance_boxes(const double* h_xyz, const double* h_inst, const double* h_sem, int64_t n_points,
                               int32_t scannet, int32_t cap, double* h_box, double* h_cls, double* h_vol,
                               int32_t* n_boxes, int32_t* instance_num) {
  if (n_points < 0 || cap < 0 || !n_boxes || !instance_num || (n_points > 0 && (!h_xyz || !h_inst || !h_sem)) ||
      (cap > 0 && (!h_box || !h_cls || !h_vol)))
    return fail(GAPRO_ERR_BAD_ARG, "gapro_scene_instance_boxes: bad argument");
  *n_boxes = 0;
  *instance_num = 0;
  long long mx = -1;
  for (int64_t i = 0; i < n_points; ++i) {
    const long long l = (long long)h_inst[i];
    mx = l > mx ? l : mx;
  }
  if (mx < 0) return GAPRO_OK;
  *instance_num = (int32_t)(mx + 1);
  if (mx + 1 > cap) return fail(GAPRO_ERR_BAD_ARG, "gapro_scene_instance_boxes: more instance ids than cap");
  const int I = (int)(mx + 1);
  std::vector<double> mn(3 * (size_t)I), mxv(3 * (size_t)I), cl((size_t)I);
  std::vector<char> seen((size_t)I, 0);
  for (int64_t i = 0; i < n_points; ++i) {
    const long long l = (long long)h_inst[i];
    if (l < 0) continue;
    const double* p = h_xyz + 3 * i;
    double* a = mn.data() + 3 * l;
    double* b = mxv.data() + 3 * l;
    if (!seen[l]) {
      seen[l] = 1;
      cl[l] = h_sem[i];
      for (int d = 0; d < 3; ++d) a[d] = b[d] = p[d];
    } else {
      for (int d = 0; d < 3; ++d) {  // NaN propagates, as np.minimum / np.maximum do
        a[d] = (p[d] < a[d] || p[d] != p[d]) ? p[d] : a[d];
        b[d] = (p[d] > b[d] || p[d] != p[d]) ? p[d] : b[d];
      }
    }
  }
  int nb = 0;
  for (int l = 0; l < I; ++l) {
    if (!seen[l]) continue;
    double v = 1.0;
    for (int d = 0; d < 3; ++d) {
      h_box[6 * nb + d] = mn[3 * l + d];
      h_box[6 * nb + 3 + d] = mxv[3 * l + d];
      const double e = mxv[3 * l + d] - mn[3 * l + d];
      v *= e > 0.0 ? e : (e != e ? e : 0.0);
    }
    h_vol[nb] = v;
    h_cls[nb] = (scannet && cl[l] != -100.0) ? cl[l] - 2.0 : cl[l];
    ++nb;
  }
  *n_boxes = nb;
  return GAPRO_OK;
}

// ---- writer -----------------------------------------------------------------------------------------------------
namespace {
struct Buf {
  std::vector<unsigned char, NoInitAlloc<unsigned char>> v;
  void b(unsigned x) { v.push_back((unsigned char)x); }
  void u16(unsigned x) { b(x & 0xFF); b((x >> 8) & 0xFF); }
  void u32(unsigned long long x) { for (int i = 0; i < 4; ++i) b((unsigned)(x >> (8 * i)) & 0xFF); }
  void u64(unsigned long long x) { for (int i = 0; i < 8; ++i) b((unsigned)(x >> (8 * i)) & 0xFF); }
  void s(const char* t) { v.insert(v.end(), (const unsigned char*)t, (const unsigned char*)t + strlen(t)); }
  void unicode(const char* t) { b('X'); u32(strlen(t)); s(t); }
  void pint(long long x) {
    if (x >= 0 && x < 256) { b('K'); b((unsigned)x); }
    else if (x >= 0 && x < 65536) { b('M'); u16((unsigned)x); }
    else if (x >= -2147483647LL - 1 && x <= 2147483647LL) { b('J'); u32((unsigned long long)(unsigned)(int)x); }
    else { b(0x8a); b(8); u64((unsigned long long)x); }
  }
};
}  // namespace

int gapro_pth_write(const char* path, int32_t n_arrays, const gapro_pth_array* descs, const void* const* h_data,
                    int32_t as_tuple) {
  if (!path || n_arrays <= 0 || !descs || !h_data || (!as_tuple && n_arrays != 1))
    return fail(GAPRO_ERR_BAD_ARG, "gapro_pth_write: bad argument");
  // One pickle buffer per writer thread, kept at its capacity between files: a fresh 3 .. 6 MB vector per file is an
  // mmap, ~1500 first-touch page faults and (growing array by array) three reallocations with copies -- 3 of the 4 ms
  // a label file took once the transcoder and the CRC were vectorised.
  static thread_local Buf pk_tls;
  Buf& pk = pk_tls;
  pk.v.clear();
  {
    size_t bound = 4096;
    for (int a = 0; a < n_arrays; ++a) bound += 2 * (size_t)(descs[a].nbytes > 0 ? descs[a].nbytes : 0) + 512;
    pk.v.reserve(bound);
  }
  pk.b(0x80); pk.b(2);
  if (as_tuple) pk.b('(');
  for (int a = 0; a < n_arrays; ++a) {
    const gapro_pth_array& d = descs[a];
    const char kind = (char)d.kind;
    if (!(kind == 'f' || kind == 'i' || kind == 'u' || kind == 'b') || d.itemsize < 1 || d.itemsize > 8 ||
        (d.itemsize & (d.itemsize - 1)) || d.ndim < 0 || d.ndim > 4)
      return fail(GAPRO_ERR_BAD_ARG, "gapro_pth_write: unsupported dtype / rank");
    long long cnt = 1;
    for (int k = 0; k < d.ndim; ++k) {
      if (d.shape[k] < 0) return fail(GAPRO_ERR_BAD_ARG, "gapro_pth_write: negative extent");
      cnt *= d.shape[k];
    }
    if (cnt * d.itemsize != d.nbytes || (d.nbytes > 0 && !h_data[a]))
      return fail(GAPRO_ERR_BAD_ARG, "gapro_pth_write: nbytes does not match the shape");
    if (d.nbytes > 0x7FFFFFFFLL) return fail(GAPRO_ERR_UNSUPPORTED, "gapro_pth_write: array beyond 2 GiB");
    if (d.nbytes == 0) return fail(GAPRO_ERR_UNSUPPORTED, "gapro_pth_write: empty array (pickled differently)");
    // numpy.core.multiarray._reconstruct(numpy.ndarray, (0,), b'b')
    pk.s("cnumpy.core.multiarray\n_reconstruct\n");
    pk.s("cnumpy\nndarray\n");
    pk.pint(0); pk.b(0x85);
    pk.s("c_codecs\nencode\n"); pk.unicode("b"); pk.unicode("latin1"); pk.b(0x86); pk.b('R');
    pk.b(0x87); pk.b('R');
    // state: (1, shape, dtype, False, data)
    pk.b('(');
    pk.pint(1);
    if (d.ndim == 0) pk.b(')');
    else if (d.ndim <= 3) { for (int k = 0; k < d.ndim; ++k) pk.pint(d.shape[k]); pk.b(0x84 + d.ndim); }
    else { pk.b('('); for (int k = 0; k < d.ndim; ++k) pk.pint(d.shape[k]); pk.b('t'); }
    char ds[8];
    snprintf(ds, sizeof(ds), "%c%d", kind, (int)d.itemsize);
    pk.s("cnumpy\ndtype\n"); pk.unicode(ds); pk.b(0x89); pk.b(0x88); pk.b(0x87); pk.b('R');
    pk.b('('); pk.pint(3); pk.unicode(d.itemsize == 1 ? "|" : "<"); pk.b('N'); pk.b('N'); pk.b('N'); pk.pint(-1); pk.pint(-1);
    pk.pint(0); pk.b('t'); pk.b('b');
    pk.b(0x89);
    pk.s("c_codecs\nencode\n");
    pk.b('X');
    const size_t len_at = pk.v.size();
    pk.u32(0);
    const size_t at = pk.v.size();
    pk.v.resize(at + 2 * (size_t)d.nbytes + 64);  // (+ 64: the vector tiers store whole registers)
    const long long enc = encode_utf8((const unsigned char*)h_data[a], d.nbytes, pk.v.data() + at);
    pk.v.resize(at + (size_t)enc);
    if (enc > 0xFFFFFFFFLL) return fail(GAPRO_ERR_UNSUPPORTED, "gapro_pth_write: payload beyond 4 GiB");
    for (int i = 0; i < 4; ++i) pk.v[len_at + i] = (unsigned char)((unsigned long long)enc >> (8 * i));
    pk.unicode("latin1"); pk.b(0x86); pk.b('R');
    pk.b('t'); pk.b('b');
  }
  if (as_tuple) pk.b('t');
  pk.b('.');
  // archive stem = the file's base name without its extension, as torch.save names it
  std::string stem(path);
  const size_t sl = stem.find_last_of('/');
  if (sl != std::string::npos) stem = stem.substr(sl + 1);
  const size_t dot = stem.find_last_of('.');
  if (dot != std::string::npos && dot > 0) stem = stem.substr(0, dot);
  if (stem.empty()) stem = "archive";
  struct Member { std::string name; const unsigned char* p; size_t n; unsigned crc; unsigned long long off; };
  static const unsigned char k_little[] = "little", k_ver[] = "3\n";
  std::vector<Member> mem = {{stem + "/data.pkl", pk.v.data(), pk.v.size(), 0, 0},
                             {stem + "/byteorder", k_little, 6, 0, 0},
                             {stem + "/version", k_ver, 2, 0, 0}};
  if (pk.v.size() >= 0xFFFFFFFFULL) return fail(GAPRO_ERR_UNSUPPORTED, "gapro_pth_write: pickle beyond 4 GiB");
  Buf hdr;  // everything but the pickle payload is assembled here and written with writev-like sequencing
  char tmp[4096];
  snprintf(tmp, sizeof(tmp), "%s.tmp.%d.%lx", path, (int)getpid(), (unsigned long)(uintptr_t)&tmp);
  const int fd = open(tmp, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
  if (fd < 0) return fail(GAPRO_ERR_IO, std::string("gapro_pth_write: ") + tmp + ": " + strerror(errno));
  auto wr = [&](const void* p, size_t n) -> bool {
    const unsigned char* q = (const unsigned char*)p;
    while (n) {
      const ssize_t w = write(fd, q, n);
      if (w < 0) {
        if (errno == EINTR) continue;
        return false;
      }
      q += w;
      n -= (size_t)w;
    }
    return true;
  };
  unsigned long long off = 0;
  bool ok = true;
  for (Member& m : mem) {
    m.crc = crc32_buf(m.p, m.n);
    m.off = off;
    Buf lh;
    lh.u32(0x04034b50u); lh.u16(20); lh.u16(0); lh.u16(0); lh.u16(0); lh.u16(0x21);  // stored; 1980-01-01
    lh.u32(m.crc); lh.u32(m.n); lh.u32(m.n); lh.u16((unsigned)m.name.size()); lh.u16(0);
    lh.s(m.name.c_str());
    ok = ok && wr(lh.v.data(), lh.v.size()) && wr(m.p, m.n);
    off += lh.v.size() + m.n;
  }
  Buf cd;
  for (const Member& m : mem) {
    cd.u32(0x02014b50u); cd.u16(20); cd.u16(20); cd.u16(0); cd.u16(0); cd.u16(0); cd.u16(0x21);
    cd.u32(m.crc); cd.u32(m.n); cd.u32(m.n); cd.u16((unsigned)m.name.size()); cd.u16(0); cd.u16(0); cd.u16(0); cd.u16(0);
    cd.u32(0); cd.u32(m.off);
    cd.s(m.name.c_str());
  }
  const size_t cd_size = cd.v.size();
  cd.u32(0x06054b50u); cd.u16(0); cd.u16(0); cd.u16((unsigned)mem.size()); cd.u16((unsigned)mem.size());
  cd.u32(cd_size); cd.u32(off); cd.u16(0);
  ok = ok && wr(cd.v.data(), cd.v.size());
  const int e = errno;
  if (close(fd) != 0) ok = false;
  if (!ok) {
    unlink(tmp);
    return fail(GAPRO_ERR_IO, std::string("gapro_pth_write: write failed: ") + strerror(e));
  }
  if (rename(tmp, path) != 0) {
    const int e2 = errno;
    unlink(tmp);
    return fail(GAPRO_ERR_IO, std::string("gapro_pth_write: rename: ") + strerror(e2));
  }
  return GAPRO_OK;
}

}  // extern "C"
