// mw_host.hpp — host-only helpers: exact decimal <-> multi-word float conversion.
// The SDP input format and sdpb's outputs carry numbers as decimal strings
// (Json_Block_Data_Parser.hxx:26-36, print_iteration.cxx:91-104), so the library
// converts at the C-ABI boundary.  Conversion is exact-then-truncated, computed
// with a small arbitrary-size integer; no GMP dependency.
#pragma once
#include <cstring>
#include "mw.hpp"

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace mw
{
struct BigNat // little-endian base 2^32 natural number
{
  std::vector<uint32_t> w;
  void trim()
  {
    while(!w.empty() && w.back() == 0)
      w.pop_back();
  }
  bool is_zero() const { return w.empty(); }
  void mul_small(uint32_t f, uint32_t addend = 0)
  {
    uint64_t carry = addend;
    for(auto &x : w)
      {
        const uint64_t t = (uint64_t)x * f + carry;
        x = (uint32_t)t;
        carry = t >> 32;
      }
    if(carry)
      w.push_back((uint32_t)carry);
  }
  uint32_t div_small(uint32_t d) // returns remainder
  {
    uint64_t rem = 0;
    for(size_t i = w.size(); i-- > 0;)
      {
        const uint64_t t = (rem << 32) | w[i];
        w[i] = (uint32_t)(t / d);
        rem = t % d;
      }
    trim();
    return (uint32_t)rem;
  }
  size_t bit_length() const
  {
    if(w.empty())
      return 0;
    return 32 * (w.size() - 1) + (32 - clz32(w.back()));
  }
  void shl(size_t bits)
  {
    if(w.empty() || bits == 0)
      return;
    const size_t q = bits / 32, r = bits % 32;
    std::vector<uint32_t> o(w.size() + q + 1, 0);
    for(size_t i = 0; i < w.size(); ++i)
      {
        o[i + q] |= r ? (w[i] << r) : w[i];
        if(r)
          o[i + q + 1] |= w[i] >> (32 - r);
      }
    w.swap(o);
    trim();
  }
  void shr(size_t bits)
  {
    const size_t q = bits / 32, r = bits % 32;
    if(q >= w.size())
      {
        w.clear();
        return;
      }
    std::vector<uint32_t> o(w.size() - q, 0);
    for(size_t i = 0; i < o.size(); ++i)
      {
        o[i] = r ? (w[i + q] >> r) : w[i + q];
        if(r && i + q + 1 < w.size())
          o[i] |= w[i + q + 1] << (32 - r);
      }
    w.swap(o);
    trim();
  }
};

// Parse "[+-]ddd[.ddd][(e|E)[+-]ddd]" exactly, truncate to NL limbs.
template <int NL> Mw<NL> from_decimal(const char *s, const char *end = nullptr)
{
  const char *p = s;
  auto at_end = [&]() { return end ? p >= end : *p == 0; };
  Mw<NL> r = zero<NL>();
  uint32_t negative = 0;
  if(!at_end() && (*p == '+' || *p == '-'))
    {
      negative = (*p == '-');
      ++p;
    }
  BigNat D;
  long e10 = 0;
  bool any = false, after_point = false;
  uint32_t chunk = 0, chunk_mul = 1;
  auto flush = [&]() {
    if(chunk_mul > 1)
      {
        if(D.w.empty())
          {
            if(chunk)
              D.w.push_back(chunk);
          }
        else
          D.mul_small(chunk_mul, chunk);
      }
    chunk = 0;
    chunk_mul = 1;
  };
  for(; !at_end(); ++p)
    {
      const char c = *p;
      if(c >= '0' && c <= '9')
        {
          any = true;
          chunk = chunk * 10 + (uint32_t)(c - '0');
          chunk_mul *= 10;
          if(after_point)
            --e10;
          if(chunk_mul == 1000000000u)
            flush();
        }
      else if(c == '.' && !after_point)
        after_point = true;
      else
        break;
    }
  flush();
  if(!any)
    throw std::runtime_error(std::string("bad number: '") + std::string(s, end ? (size_t)(end - s) : strnlen(s, 32)) + "'");
  if(!at_end() && (*p == 'e' || *p == 'E'))
    {
      ++p;
      char *q;
      e10 += std::strtol(p, &q, 10);
      p = q;
    }
  if(!at_end())
    throw std::runtime_error(std::string("trailing characters in number: '") + std::string(s, end ? (size_t)(end - s) : strnlen(s, 32)) + "'");
  D.trim();
  if(D.is_zero())
    return r;
  long e2 = 0; // value = D * 2^e2 * 10^e10
  if(e10 >= 0)
    {
      for(long k = 0; k < e10 / 9; ++k)
        D.mul_small(1000000000u);
      uint32_t f = 1;
      for(long k = 0; k < e10 % 9; ++k)
        f *= 10;
      if(f > 1)
        D.mul_small(f);
    }
  else
    {
      // make the quotient at least 32*(NL+2) bits: 10^n < 2^(3.33 n)
      const long n = -e10;
      const size_t need = 32 * (NL + 2) + (size_t)(n * 3.33) + 8;
      if(D.bit_length() < need)
        {
          const size_t sh = need - D.bit_length();
          D.shl(sh);
          e2 -= (long)sh;
        }
      // floor(floor(x/a)/b) == floor(x/(ab)) for positive integers
      for(long k = 0; k < n / 9; ++k)
        D.div_small(1000000000u);
      uint32_t f = 1;
      for(long k = 0; k < n % 9; ++k)
        f *= 10;
      if(f > 1)
        D.div_small(f);
    }
  const size_t bl = D.bit_length();
  // take the top 32*NL bits
  long e = (long)bl + e2;
  if(bl > 32u * NL)
    D.shr(bl - 32u * NL);
  else
    D.shl(32u * NL - bl);
  for(int i = 0; i < NL; ++i)
    r.m[i] = i < (int)D.w.size() ? D.w[i] : 0u;
  r.e = (int32_t)e;
  r.neg = negative;
  return r;
}

// Decimal string "[-]0.ddddde[-]k" with `digits` significant digits (truncated);
// digits = 0 -> enough to round-trip (32*NL*log10(2) + 3).
template <int NL> std::string to_decimal(const Mw<NL> &a, int digits = 0)
{
  if(a.e == EZERO)
    return "0";
  if(digits <= 0)
    digits = (int)(32 * NL * 0.30103) + 3;
  BigNat M;
  M.w.assign(a.m, a.m + NL);
  M.trim();
  // value = M * 2^s
  long s = (long)a.e - 32L * NL;
  long e10 = 0;
  if(s >= 0)
    M.shl((size_t)s);
  else
    {
      // N = floor(M * 10^t / 2^-s) with t s.t. N has >= digits+1 digits
      const long bits_int = (long)M.bit_length() + s; // log2(value) ~ bits_int
      long t = digits + 2 - (long)(bits_int * 0.30103);
      if(t < 0)
        t = 0;
      for(long k = 0; k < t / 9; ++k)
        M.mul_small(1000000000u);
      uint32_t f = 1;
      for(long k = 0; k < t % 9; ++k)
        f *= 10;
      if(f > 1)
        M.mul_small(f);
      M.shr((size_t)(-s));
      e10 = -t;
    }
  // integer M -> decimal digits
  std::string dig;
  while(!M.is_zero())
    {
      uint32_t rem = M.div_small(1000000000u);
      for(int k = 0; k < 9; ++k)
        {
          dig.push_back((char)('0' + rem % 10));
          rem /= 10;
        }
    }
  while(!dig.empty() && dig.back() == '0')
    dig.pop_back();
  if(dig.empty())
    return "0";
  std::string msd(dig.rbegin(), dig.rend());
  const long point = (long)msd.size() + e10; // value = 0.msd * 10^point
  if((int)msd.size() > digits)
    msd.resize(digits);
  while(msd.size() > 1 && msd.back() == '0')
    msd.pop_back();
  std::string out = a.neg ? "-0." : "0.";
  out += msd;
  out += "e";
  out += std::to_string(point);
  return out;
}

// ---- binary records in GMP's mpf_t layout ------------------------------------------
// The binary side of the C ABI (include/sdpb_hip.h: *_mpf entry points): one number =
// 2 + L 64-bit words, word 0 = (int64) _mp_size (signed count of used limbs, 0 = zero),
// word 1 = (int64) _mp_exp (exponent in 64-bit limbs), words 2.. = _mp_d[0..L), least
// significant limb first; value = sign * (sum_i d[i] 2^(64 i)) * 2^(64 (exp - |size|)),
// i.e. exactly what a caller holding an mpf_t (El::BigFloat::gmp_float) can memcpy.
template <int NL> Mw<NL> from_mpf_record(const uint64_t *rec, int limbs64)
{
  const long long size = (long long)rec[0], expl = (long long)rec[1];
  long long n = size < 0 ? -size : size;
  if(n > limbs64)
    throw std::runtime_error("mpf record: |_mp_size| exceeds the record's limb count");
  const uint64_t *d = rec + 2;
  while(n > 0 && d[n - 1] == 0)
    --n;
  Mw<NL> r = zero<NL>();
  if(n == 0)
    return r;
  const int z = __builtin_clzll(d[n - 1]);
  // bit string d[n-1] .. d[0], shifted left by z, top 32 NL bits kept (truncation toward zero)
  auto limb = [&](long long i) -> uint64_t { return (i >= 0 && i < n) ? d[i] : 0; };
  for(int k = 0; k < NL; ++k)
    {
      // 32-bit word k counted from the top: bits [64n - 32(k+1) - z, +32) of the integer
      const long long bit = 64 * n - 32LL * (k + 1) - z;
      uint32_t w = 0;
      if(bit > -32)
        {
          const long long q = bit >= 0 ? bit / 64 : -1, rbit = bit >= 0 ? bit % 64 : 0;
          if(bit >= 0)
            {
              uint64_t v = limb(q) >> rbit;
              if(rbit > 32)
                v |= limb(q + 1) << (64 - rbit);
              w = (uint32_t)v;
            }
          else
            w = (uint32_t)(limb(0) << (-bit)); // the lowest word hangs over the end
        }
      r.m[NL - 1 - k] = w;
    }
  const long long e = 64 * expl - z + 64 * (n - (size < 0 ? -size : size));
  if(e > (1LL << 29) || e < -(1LL << 29))
    throw std::runtime_error("mpf record: exponent out of range");
  r.e = (int32_t)e;
  r.neg = size < 0 ? 1u : 0u;
  return r;
}
// Exact when limbs64 >= NL/2 + 1 (GMP's own allocation _mp_prec + 1 for the same precision);
// fewer limbs truncate the low bits.
template <int NL> void to_mpf_record(const Mw<NL> &v, uint64_t *rec, int limbs64)
{
  for(int i = 0; i < limbs64 + 2; ++i)
    rec[i] = 0;
  if(v.e == EZERO)
    return;
  // value = 0.m * 2^e;  exp = ceil(e / 64), the mantissa is shifted right by s = 64 exp - e bits
  const long long e = v.e;
  const long long expl = e >= 0 ? (e + 63) / 64 : -((-e) / 64);
  const int s = (int)(64 * expl - e); // 0..63
  // fraction bits: s zeros, then the 32 NL bits of m; limb k from the top (k = 0 most significant)
  auto frac_bit_word = [&](long long bit) -> uint64_t { // 64 bits starting `bit` bits below the binary point
    uint64_t out = 0;
    for(int b = 0; b < 64; b += 32)
      {
        // 32-bit chunk covering fraction bits [bit + b, bit + b + 32)
        const long long pos = bit + b - s; // position inside m's bit string (0 = top bit)
        uint32_t w = 0;
        if(pos > -32 && pos < 32LL * NL)
          {
            const long long k = pos >= 0 ? pos / 32 : -1, r = pos >= 0 ? pos % 32 : 0;
            if(pos >= 0)
              {
                const uint32_t hi = v.m[NL - 1 - k], lo = (k + 1 < NL) ? v.m[NL - 2 - k] : 0u;
                w = r ? ((hi << r) | (lo >> (32 - r))) : hi;
              }
            else
              w = v.m[NL - 1] >> (-pos);
          }
        out |= (uint64_t)w << (32 - b);
      }
    return out;
  };
  int used = 0;
  for(int k = 0; k < limbs64; ++k)
    {
      const uint64_t w = frac_bit_word(64LL * k);
      rec[2 + (limbs64 - 1 - k)] = w;
      if(w)
        used = k + 1;
    }
  // drop low zero limbs like mpf does: keep `used` limbs at the bottom of d[]
  if(used < limbs64)
    {
      for(int k = 0; k < used; ++k)
        rec[2 + k] = rec[2 + (limbs64 - used) + k];
      for(int k = used; k < limbs64; ++k)
        rec[2 + k] = 0;
    }
  rec[0] = (uint64_t)(long long)(v.neg ? -used : used);
  rec[1] = (uint64_t)expl;
}
} // namespace mw
