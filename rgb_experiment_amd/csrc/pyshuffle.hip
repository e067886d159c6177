// Host-side helper (no kernel): Python's `random.seed(int); random.shuffle(list(range(n)))`, bit for bit, in C++.
// The reference's split masks (rgb_experiment/utils/mask.py:66-102, called from itexperiments.py:210-215 on every run with a
// supplied Data and from rd2pd.py) are a seeded `random.shuffle` of the labelled node positions; they select the loss rows, so
// they must be reproduced exactly (golden G4). CPython's shuffle over 2 M positions takes 1.5-4 s — more than twenty epochs
// of the HIP path at that size; this restatement of the same generator takes ~15 ms.
//   random.seed(a), int a      : MT19937 init_by_array over the 32-bit words of |a|, little-endian (Modules/_randommodule.c)
//   random.shuffle(x)          : for i = len-1 .. 1: j = _randbelow(i + 1); swap x[i], x[j]            (Lib/random.py)
//   _randbelow(n)              : k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
//   getrandbits(k), k <= 32    : genrand_uint32() >> (32 - k)
#include "rgbx_common.h"

namespace rgbx {
namespace {

struct MT19937 {
  uint32_t mt[624];
  int idx;
  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void init_by_array(const uint32_t* key, int len) {
    init_genrand(19650218u);
    int i = 1, j = 0;
    for (int k = 624 > len ? 624 : len; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
      if (++j >= len) j = 0;
    }
    for (int k = 623; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
};

}  // namespace
}  // namespace rgbx

// perm[0..n) = list(range(n)) after random.seed(seed); random.shuffle(it). HOST pointer (this entry point runs on the CPU,
// synchronously; it is the one exception to the header's "device pointers" convention). n < 2^32.
extern "C" int rgbx_py_random_shuffle_i64(int64_t seed, int64_t n, int64_t* perm) {
  using namespace rgbx;
  if (n < 0 || (n > 0 && !perm)) return fail(RGBX_E_ARG, "py_random_shuffle: bad argument");
  if (n >= ((int64_t)1 << 32)) return fail(RGBX_E_RANGE, "py_random_shuffle: n = %lld needs more than 32 random bits", (long long)n);
  const uint64_t a = seed < 0 ? (uint64_t)0 - (uint64_t)seed : (uint64_t)seed;  // |seed|, also for INT64_MIN
  uint32_t key[2] = {(uint32_t)(a & 0xffffffffu), (uint32_t)(a >> 32)};
  MT19937 g;
  g.init_by_array(key, key[1] ? 2 : 1);
  for (int64_t i = 0; i < n; ++i) perm[i] = i;
  for (int64_t i = n - 1; i >= 1; --i) {
    const uint32_t bound = (uint32_t)(i + 1);
    const int k = 32 - __builtin_clz(bound);  // bound.bit_length()
    uint32_t r;
    do { r = g.next() >> (32 - k); } while (r >= bound);
    const int64_t t = perm[i];
    perm[i] = perm[r];
    perm[r] = t;
  }
  return RGBX_OK;
}
