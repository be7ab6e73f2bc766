// k_serial.hip — reference-order kernels: one workgroup per independent chain (component stream or
// byte plane), lane 0 walks the recurrence exactly as the format defines it.  These are the
// always-correct device path for every stream type and parameter combination (generic table
// exponents on decode, both LZ4 table variants); the throughput kernels (k_fpc32_encode.hip,
// k_lz4_encode.hip, ...) take over where they apply.  Device code only — this is not a CPU fallback.
//
// Reference loops restated: fpsc.c:128-204 / 617-794 (encode), 245-414 / 837-1160 (decode),
// lz4.c:863-1172 (greedy block compressor), lz4.c:1657-2072 (safe block decoder).
#include "common.hpp"

namespace trico {

template <int W> struct word;
template <> struct word<32> { typedef uint32_t type; };
template <> struct word<64> { typedef uint64_t type; };

template <typename T> __device__ __forceinline__ unsigned byte_len(T x)
  {
  if (sizeof(T) == 4)
    return x ? (unsigned)(39 - __builtin_clz((uint32_t)x)) >> 3 : 0u;
  return x ? (unsigned)(71 - __builtin_clzll((uint64_t)x)) >> 3 : 0u;
  }

template <typename T> __device__ __forceinline__ uint8_t* put_be(uint8_t* o, T x, unsigned nb)
  {
  while (nb)
    {
    --nb;
    *o++ = (uint8_t)(x >> (8 * nb));
    }
  return o;
  }

// ---- fp encode ------------------------------------------------------------------------------------
// grid.x = arity; component c reads src[i*arity + c].  W=32: tables in LDS (16+1024 entries);
// W=64: tables in global scratch (2 x 2^20 u64 per component, zeroed by the host).
template <int W>
__global__ void __launch_bounds__(64) k_fpc_encode_serial(const typename word<W>::type* __restrict__ src, uint32_t n, int arity,
                                                          uint8_t* out_base, size_t out_stride, uint32_t* sizes,
                                                          uint64_t* gtables, unsigned E1, unsigned E2)
  {
  // E1 / E2: table size exponents as the reference normalises them (even, 0..30: fpsc.c:88-93, 578-583).  W = 32 with at most
  // (4, 10): tables in LDS; otherwise in the zeroed global scratch, component c at c * (2^E1 + 2^E2) entries.
  typedef typename word<W>::type T;
  constexpr unsigned G = (W == 32) ? 8 : 2, WB = W / 8;
  __shared__ T lds_tab[(W == 32) ? (16 + 1024) : 1];
  const int c = blockIdx.x;
  T* T1;
  T* T2;
  if (W == 32 && E1 <= 4 && E2 <= 10)
    {
    for (unsigned i = threadIdx.x; i < 16 + 1024; i += blockDim.x)
      lds_tab[i] = 0;
    T1 = lds_tab;
    T2 = lds_tab + 16;
    }
  else
    {
    T1 = (T*)gtables + (size_t)c * (((size_t)1 << E1) + ((size_t)1 << E2));
    T2 = T1 + ((size_t)1 << E1);
    }
  __syncthreads();
  if (threadIdx.x != 0)
    return;
  uint8_t* out = out_base + (size_t)c * out_stride;
  uint8_t* o = out;
  *o++ = (uint8_t)(((E1 >> 1) << 4) | (E2 >> 1));
  o = put_be<uint32_t>(o, n, 4);
  T h1 = 0, h2 = 0, p1 = 0, p2 = 0, last = 0;
  T x[G];
  unsigned code[G];
  unsigned j = 0;
  for (uint32_t i = 0; i < n; ++i)
    {
    j = i % G;
    const T v = src[(size_t)i * arity + c];
    const T x1 = v ^ p1;
    T1[h1] = v;
    h1 = E1 ? (T)(v >> (W - E1)) : (T)0;           // the shifted-in old hash is masked away entirely; exponent 0: one entry
    p1 = T1[h1];
    const T s = v - last;
    const T x2 = v ^ (T)(last + p2);
    last = v;
    T2[h2] = s;
    h2 = E2 ? (T)(((h2 << (E2 / 2)) ^ (s >> (W - E2))) & (((T)1 << E2) - 1)) : (T)0;
    p2 = T2[h2];
    const unsigned n1 = byte_len(x1);
    unsigned n2 = byte_len(x2);
    n2 = n2 ? n2 : 1u;
    if (n1 <= 1u) { code[j] = n1; x[j] = x1; }
    else if (n2 < n1) { code[j] = WB + n2; x[j] = x2; }
    else { code[j] = n1; x[j] = x1; }
    if (j == G - 1 || i == n - 1)
      {
      for (unsigned l = j + 1; l < G; ++l) { code[l] = 1; x[l] = 0; }     // tail padding
      if (W == 32)
        {
        uint32_t bc = 0;
        for (unsigned k = 0; k < G; ++k) bc |= code[k] << (3 * k);
        o = put_be<uint32_t>(o, bc, 3);
        }
      else
        *o++ = (uint8_t)((code[1 % G] << 4) | code[0]);
      for (unsigned k = 0; k < G; ++k)
        o = put_be<T>(o, x[k], code[k] <= WB ? code[k] : code[k] - WB);
      }
    }
  if (n == 0)
    {
    // undefined in the reference; defined as one full pad group
    if (W == 32) { *o++ = 0x24; *o++ = 0x92; *o++ = 0x49; for (int k = 0; k < 8; ++k) *o++ = 0; }
    else { *o++ = 0x11; *o++ = 0; *o++ = 0; }
    }
  sizes[c] = (uint32_t)(o - out);
  }

// ---- fp decode ------------------------------------------------------------------------------------
struct DecodeArgs
  {
  const uint8_t* pay[3];
  uint32_t size[3];
  };

template <int W>
__global__ void __launch_bounds__(64) k_fpc_decode_serial(DecodeArgs a, int arity, uint32_t n, typename word<W>::type* dst,
                                                          uint64_t* gtables, size_t gstride, uint32_t* status)
  {
  typedef typename word<W>::type T;
  constexpr unsigned G = (W == 32) ? 8 : 2, WB = W / 8;
  constexpr unsigned LDS_E1 = 4, LDS_E2 = 10;
  __shared__ T lds_tab[(W == 32) ? (16 + 1024) : 1];
  const int c = blockIdx.x;
  const uint8_t* in = a.pay[c];
  const uint32_t len = a.size[c];
  if (len < 5)
    {
    if (threadIdx.x == 0) atomicOr(status, 1u);
    return;
    }
  const unsigned e1 = (unsigned)(in[0] >> 4) << 1, e2 = (unsigned)(in[0] & 15) << 1;
  const uint32_t cnt = ((uint32_t)in[1] << 24) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 8) | in[4];
  // any table shape the reference can write (fpsc.c:214-217, 580-583: even exponents up to 30): floats up to (4,10) in LDS,
  // everything else in the zeroed global scratch, component c at c * gstride entries (the launcher sized it from the
  // header bytes; a payload whose header asks for more than that is refused)
  bool ok = cnt == n && e1 <= 30 && e2 <= 30;
  const bool in_lds = W == 32 && e1 <= LDS_E1 && e2 <= LDS_E2;
  if (!in_lds)
    ok = ok && gtables != nullptr && (((size_t)1 << e1) + ((size_t)1 << e2)) <= gstride;
  if (!ok)
    {
    if (threadIdx.x == 0) atomicOr(status, 2u);
    return;
    }
  T* T1;
  T* T2;
  if (in_lds)
    {
    for (unsigned i = threadIdx.x; i < 16 + 1024; i += blockDim.x)
      lds_tab[i] = 0;
    T1 = lds_tab;
    T2 = lds_tab + 16;
    }
  else
    {
    T1 = (T*)gtables + (size_t)c * gstride;
    T2 = T1 + ((size_t)1 << e1);
    }
  __syncthreads();
  if (threadIdx.x != 0)
    return;
  const T m1 = (T)(((T)1 << e1) - 1), m2 = (T)(((T)1 << e2) - 1);
  T h1 = 0, h2 = 0, p1 = 0, p2 = 0, last = 0;
  uint32_t pos = 5;
  for (uint32_t i = 0; i < n; i += G)
    {
    unsigned code[G];
    if (W == 32)
      {
      if (pos + 3 > len) { atomicOr(status, 4u); return; }
      const uint32_t bc = ((uint32_t)in[pos] << 16) | ((uint32_t)in[pos + 1] << 8) | in[pos + 2];
      pos += 3;
      for (unsigned k = 0; k < G; ++k) code[k] = (bc >> (3 * k)) & 7u;
      }
    else
      {
      if (pos + 1 > len) { atomicOr(status, 4u); return; }
      const unsigned bc = in[pos++];
      code[0] = bc & 15u;
      code[1 % G] = bc >> 4;
      }
    const unsigned m = (n - i < G) ? (n - i) : G;
    for (unsigned k = 0; k < m; ++k)
      {
      const unsigned nb = code[k] <= WB ? code[k] : code[k] - WB;
      if (pos + nb > len) { atomicOr(status, 4u); return; }
      T xr = 0;
      for (unsigned b = 0; b < nb; ++b) xr = (T)(xr << 8) | in[pos++];
      if (code[k] > WB) p1 = p2;
      const T v = xr ^ p1;
      T1[h1] = v;
      h1 = e1 ? (T)(((h1 << e1) ^ (v >> (W - e1))) & m1) : (T)0;      // exponent 0: a one-entry table (the reference masks with 0)
      p1 = T1[h1];
      const T s = v - last;
      T2[h2] = s;
      h2 = e2 ? (T)(((h2 << (e2 / 2)) ^ (s >> (W - e2))) & m2) : (T)0;
      p2 = (T)(v + T2[h2]);
      last = v;
      dst[(size_t)(i + k) * arity + c] = v;
      }
    }
  }

// ---- LZ4 block compress (greedy, LZ4 1.9.2 parse) -------------------------------------------------
__device__ __forceinline__ uint32_t rd32(const uint8_t* p)
  {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
  }
__device__ __forceinline__ uint64_t rd64(const uint8_t* p)
  {
  return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32);
  }
__device__ __forceinline__ uint8_t* put_len(uint8_t* op, uint32_t len)
  {
  for (; len >= 255; len -= 255) *op++ = 255;
  *op++ = (uint8_t)len;
  return op;
  }

__global__ void __launch_bounds__(64) k_lz4_encode_serial(const uint8_t* planes, size_t plane_stride, uint32_t n, uint8_t* out_base,
                                                          size_t out_stride, uint32_t* sizes)
  {
  __shared__ uint32_t tab[4096];   // u32[4096] or, for n < 65547, u16[8192] in the same 16 KiB
  for (unsigned i = threadIdx.x; i < 4096; i += blockDim.x)
    tab[i] = 0;
  __syncthreads();
  if (threadIdx.x != 0)
    return;
  const uint8_t* src = planes + (size_t)blockIdx.x * plane_stride;
  uint8_t* dst = out_base + (size_t)blockIdx.x * out_stride;
  uint16_t* tab16 = (uint16_t*)tab;
  const bool small = n < 65547u;
#define LZ_HASH(p) (small ? ((rd32(p) * 2654435761u) >> 19) : (uint32_t)(((rd64(p) << 24) * 889523592379ull) >> 52))
#define LZ_GET(h) (small ? (uint32_t)tab16[h] : tab[h])
#define LZ_SET(h, v) do { if (small) tab16[h] = (uint16_t)(v); else tab[h] = (v); } while (0)
  uint8_t* op = dst;
  uint32_t anchor = 0;
  if (n >= 13u)
    {
    const uint32_t mfl1 = n - 11u, mlim = n - 5u;
    LZ_SET(LZ_HASH(src), 0u);
    uint32_t ip = 1;
    uint32_t fh = LZ_HASH(src + 1);
    bool done = false;
    while (!done)
      {
      uint32_t cand = 0, fwd = ip, step = 1, nb = 64;
      for (;;)
        {
        const uint32_t h = fh, cur = fwd;
        cand = LZ_GET(h);
        ip = fwd;
        fwd += step;
        step = nb++ >> 6;
        if (fwd > mfl1) { done = true; break; }
        fh = LZ_HASH(src + fwd);
        LZ_SET(h, cur);
        if (!small && cand + 65535u < cur) continue;
        if (rd32(src + cand) == rd32(src + ip)) break;
        }
      if (done) break;
      while (ip > anchor && cand > 0 && src[ip - 1] == src[cand - 1]) { --ip; --cand; }
      const uint32_t lit = ip - anchor;
      uint8_t* token = op++;
      if (lit >= 15u) { *token = 0xf0; op = put_len(op, lit - 15u); }
      else *token = (uint8_t)(lit << 4);
      for (uint32_t k = 0; k < lit; ++k) op[k] = src[anchor + k];
      op += lit;
      for (;;)
        {
        *op++ = (uint8_t)(ip - cand);
        *op++ = (uint8_t)((ip - cand) >> 8);
        uint32_t m = 0;
        while (ip + 4 + m < mlim && src[ip + 4 + m] == src[cand + 4 + m]) ++m;
        ip += m + 4;
        if (m >= 15u) { *token += 15; op = put_len(op, m - 15u); }
        else *token += (uint8_t)m;
        anchor = ip;
        if (ip >= mfl1) { done = true; break; }
        LZ_SET(LZ_HASH(src + ip - 2), ip - 2);
        const uint32_t h = LZ_HASH(src + ip);
        cand = LZ_GET(h);
        LZ_SET(h, ip);
        if ((small || cand + 65535u >= ip) && rd32(src + cand) == rd32(src + ip))
          {
          token = op++;
          *token = 0;
          continue;
          }
        break;
        }
      if (done) break;
      fh = LZ_HASH(src + (++ip));
      }
    }
  const uint32_t run = n - anchor;
  if (run >= 15u) { *op++ = 0xf0; op = put_len(op, run - 15u); }
  else *op++ = (uint8_t)(run << 4);
  for (uint32_t k = 0; k < run; ++k) op[k] = src[anchor + k];
  op += run;
  sizes[blockIdx.x] = (uint32_t)(op - dst);
#undef LZ_HASH
#undef LZ_GET
#undef LZ_SET
  }

// ---- LZ4 block decode (safe) -----------------------------------------------------------------------
struct Lz4DecArgs
  {
  const uint8_t* pay[8];
  uint32_t size[8];
  };

__global__ void __launch_bounds__(64) k_lz4_decode_serial(Lz4DecArgs a, uint8_t* planes, size_t plane_stride, uint32_t cap, uint32_t* status)
  {
  if (threadIdx.x != 0)
    return;
  const uint8_t* src = a.pay[blockIdx.x];
  const uint32_t n = a.size[blockIdx.x];
  uint8_t* dst = planes + (size_t)blockIdx.x * plane_stride;
  uint32_t ip = 0, op = 0;
  bool bad = (n == 0);
  while (!bad)
    {
    if (ip >= n) { bad = true; break; }
    const unsigned tok = src[ip++];
    uint32_t lit = tok >> 4;
    if (lit == 15u)
      {
      unsigned b;
      do { if (ip >= n) { bad = true; break; } b = src[ip++]; lit += b; } while (b == 255u);
      if (bad) break;
      }
    if (lit > n - ip || lit > cap - op) { bad = true; break; }
    for (uint32_t k = 0; k < lit; ++k) dst[op + k] = src[ip + k];
    ip += lit; op += lit;
    if (ip == n) break;
    if (n - ip < 2u) { bad = true; break; }
    const uint32_t off = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8);
    ip += 2;
    if (off == 0u || off > op) { bad = true; break; }
    uint32_t ml = tok & 15u;
    if (ml == 15u)
      {
      unsigned b;
      do { if (ip >= n) { bad = true; break; } b = src[ip++]; ml += b; } while (b == 255u);
      if (bad) break;
      }
    ml += 4u;
    if (ml > cap - op) { bad = true; break; }
    for (uint32_t k = 0; k < ml; ++k) dst[op + k] = dst[op + k - off];
    op += ml;
    }
  if (bad || op != cap)
    atomicOr(status, 8u);
  }

// ---- launchers --------------------------------------------------------------------------------------

int launch_fpc_encode_serial(const void* d_src, uint32_t n, int arity, int width, uint8_t* d_out, size_t out_stride,
                             uint32_t* d_sizes, uint64_t* d_tables, unsigned e1, unsigned e2)
  {
  if (width == 4)
    hipLaunchKernelGGL(k_fpc_encode_serial<32>, dim3(arity), dim3(64), 0, current_stream(),
                       (const uint32_t*)d_src, n, arity, d_out, out_stride, d_sizes, d_tables, e1, e2);
  else
    hipLaunchKernelGGL(k_fpc_encode_serial<64>, dim3(arity), dim3(64), 0, current_stream(),
                       (const uint64_t*)d_src, n, arity, d_out, out_stride, d_sizes, d_tables, e1, e2);
  return hip_ok(hipGetLastError(), "k_fpc_encode_serial") ? 1 : 0;
  }

int launch_fpc_decode_serial(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, int width,
                             uint32_t n, void* d_dst, uint64_t* d_tables, size_t table_stride, uint32_t* d_status)
  {
  DecodeArgs a;
  for (int c = 0; c < 3; ++c)
    {
    a.pay[c] = c < arity ? d_payloads[c] : nullptr;
    a.size[c] = c < arity ? sizes[c] : 0;
    }
  if (width == 4)
    hipLaunchKernelGGL(k_fpc_decode_serial<32>, dim3(arity), dim3(64), 0, current_stream(),
                       a, arity, n, (uint32_t*)d_dst, d_tables, table_stride, d_status);
  else
    hipLaunchKernelGGL(k_fpc_decode_serial<64>, dim3(arity), dim3(64), 0, current_stream(),
                       a, arity, n, (uint64_t*)d_dst, d_tables, table_stride, d_status);
  return hip_ok(hipGetLastError(), "k_fpc_decode_serial") ? 1 : 0;
  }

int launch_lz4_encode_serial(const uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, int nplanes, uint8_t* d_out,
                             size_t out_stride, uint32_t* d_sizes)
  {
  hipLaunchKernelGGL(k_lz4_encode_serial, dim3(nplanes), dim3(64), 0, current_stream(),
                     d_planes, plane_stride, plane_bytes, d_out, out_stride, d_sizes);
  return hip_ok(hipGetLastError(), "k_lz4_encode_serial") ? 1 : 0;
  }

int launch_lz4_decode_serial(const uint8_t* const d_payloads[8], const uint32_t sizes[8], int nplanes,
                             uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, uint32_t* d_status)
  {
  Lz4DecArgs a;
  for (int c = 0; c < 8; ++c)
    {
    a.pay[c] = c < nplanes ? d_payloads[c] : nullptr;
    a.size[c] = c < nplanes ? sizes[c] : 0;
    }
  hipLaunchKernelGGL(k_lz4_decode_serial, dim3(nplanes), dim3(64), 0, current_stream(),
                     a, d_planes, plane_stride, plane_bytes, d_status);
  return hip_ok(hipGetLastError(), "k_lz4_decode_serial") ? 1 : 0;
  }

} // namespace trico
