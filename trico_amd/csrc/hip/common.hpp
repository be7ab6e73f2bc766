// common.hpp — internal declarations shared by the HIP translation units (gfx950 only).
// Nothing here crosses the C-ABI; see include/trico/trico_hip.h for the exported surface.
#pragma once
#include <stdlib.h>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "trico/trico_hip.h"

namespace trico {

// Switches that exist for measurements and for the tests' A/B runs (which coder, how many waves, self-check off ...) are read in the
// test-hooks build only (tests/_build/libtrico_testhooks.so, -DTRICO_HIP_TEST_HOOKS); the product library does not look at them.
#ifdef TRICO_HIP_TEST_HOOKS
inline const char* tune_env(const char* name) { return getenv(name); }
#else
inline const char* tune_env(const char*) { return nullptr; }
#endif


// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* msg);
bool hip_ok(hipError_t e, const char* what);
#define TRICO_HIP_TRY(expr) do { if (!::trico::hip_ok((expr), #expr)) return 0; } while (0)

hipStream_t current_stream();
void set_current_stream(hipStream_t s);   // stream of this thread's later launches (what trico_hip_set_stream sets)
bool device_ready();                      // false (and an error message) without a HIP device
void stats_count_repeat();                // trico_hip_last_stats word 2

// ---- growable device workspace --------------------------------------------------------------
struct DevBuf
  {
  uint8_t* p = nullptr;
  size_t cap = 0;
  int kind = 0;                 // buffers are recycled only among buffers of the same kind (1: scratch of the chain decoders)
  bool reserve(size_t bytes);   // grow-only, contents NOT preserved
  void release();               // returns the memory to a process-wide pool
  };
void trim_pool();               // hipFree everything the pool holds
// stage `bytes` from src (host or device) so that kernels can read it: device sources are returned as they are, host sources
// are copied to buf.p + offset on current_stream()
const void* stage_in(DevBuf& buf, const void* src, size_t bytes, size_t offset = 0);

// ---- copies between host and device (staging.hip) -------------------------------------------
// Large transfers from / to PAGEABLE host memory go through a ring of pinned chunks filled by a few host threads; everything else is
// one hipMemcpyAsync on `st`.  upload: the source may be reused on return, kernels launched on st afterwards see the bytes.
// download: what st has produced so far; with wait = true the bytes are in h_dst on return (a staged download always is).
bool upload_bytes(void* d_dst, const void* h_src, size_t bytes, hipStream_t st);
bool download_bytes(void* h_dst, const void* d_src, size_t bytes, hipStream_t st, bool wait);

} // namespace trico

// Per-archive workspace.  One context must not be used from two threads at once (same rule as
// the reference's archive handle, SURVEY.md §8(b) "threading").
struct trico_hip_ctx
  {
  trico::DevBuf in;        // staged host input / staged host payloads
  trico::DevBuf out;       // encoded payloads (component c at out.p + c * out_stride) / staged decode output
  trico::DevBuf tmp;       // planes, SoA intermediates, predictor tables
  trico::DevBuf aux;       // small: sizes, status words, segment summaries
  trico::DevBuf ws;        // large kernel workspaces (chunked LZ4 descriptors / tables)
  trico::DevBuf unit;      // de-interleaved components / planes of the unit encoders (dist.hip)
  trico::DevBuf vws;       // self-check of the chain decoders: workspace and payloads of the re-encode (shim.hip)
  trico::DevBuf chain;     // tables / rings of the chain decoders (scalar-cache resident): never recycled into anything else
  trico_hip_ctx() { chain.kind = 1; }
  // the decode the self-check belongs to (it may have to be repeated)
  bool chk_active = false;
  bool other_writer_seen = false;      // a stream of this context decoded to values that do not code back even in reference order
  uint32_t other_writer_streams = 0;   // how many (trico_hip_ctx_other_writer_streams)
  const uint8_t* chk_pay[3] = { nullptr, nullptr, nullptr };
  uint32_t chk_sizes[3] = { 0, 0, 0 };
  int chk_arity = 0, chk_width = 0;
  uint32_t chk_n = 0;
  void* chk_dst = nullptr;
  size_t out_stride = 0;
  uint32_t out_sizes[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  uint32_t out_sizes_raw[24] = { 0 };
  int out_count = 0;
  // float encoder: payloads stay in segment slots (tmp) until they are gathered to their destination
  bool out_in_slots = false;
  uint32_t slots_n = 0;
  int slots_arity = 0;
  bool out_materialized[3] = { false, false, false };
  uint32_t* h_pinned = nullptr;   // 64 words of pinned host memory for size/status read-back
  };

namespace trico {

// ---- profiling spans ---------------------------------------------------------------------------
struct ProfSpan
  {
  int k;
  bool active, counted;
  hipEvent_t e0, e1;
  explicit ProfSpan(int kernel_id, bool count_launch = true);
  void stop();                    // the span ends here (before the host waits for the stream), not at the end of the scope
  ~ProfSpan();
  };

// ---- payload size bounds -------------------------------------------------------------------------
inline size_t fpc_bound(uint32_t n, int width)
  {
  const size_t g = (width == 4) ? 8 : 2, hdr = (width == 4) ? 3 : 1;
  return 5 + (size_t)width * n + hdr * (((size_t)n + g - 1) / g + 1) + g;
  }
inline size_t lz4_bound(uint32_t n) { return (size_t)n + n / 255 + 16; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- kernel launchers (k_*.hip) ----------------------------------------------------------------
// All launch on current_stream(); device pointers only.  `status` is a device word that kernels
// set non-zero on malformed input.

// serial reference-order kernels (k_serial.hip): one workgroup per component stream / plane.
int launch_fpc_encode_serial(const void* d_src, uint32_t n, int arity, int width, uint8_t* d_out, size_t out_stride,
                             uint32_t* d_sizes, uint64_t* d_tables, unsigned e1, unsigned e2);
// d_tables: zeroed scratch, component c at c * table_stride entries of `width` bytes (tables of 2^e1 + 2^e2 entries each); encode
// side: c * (2^e1 + 2^e2).  Float streams with exponents up to (4,10) keep their tables in LDS and need none.
int launch_fpc_decode_serial(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, int width,
                             uint32_t n, void* d_dst, uint64_t* d_tables, size_t table_stride, uint32_t* d_status);
int launch_lz4_encode_serial(const uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, int nplanes, uint8_t* d_out,
                             size_t out_stride, uint32_t* d_sizes);
int launch_lz4_decode_serial(const uint8_t* const d_payloads[8], const uint32_t sizes[8], int nplanes,
                             uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, uint32_t* d_status);

// throughput float encoder: the one-sweep coder (k_fpc32_sweep.hip) and the two-sweep coder with ballots (k_fpc32_encode.hip)
size_t fpc32_encode_workspace(uint32_t n, int arity);
// d_sizes: 6 words - payload bytes of components 0..2, then the flags their waves raised (FPC32_FLAG_*): a non-zero flag means
// the payloads are not to be used - code the stream again with FPC32_CODER_BALLOT, which raises none.
// FPC32_CODER_BALLOT: always the two-sweep coder with ballots (what the decoders' self-check uses).
constexpr int FPC32_CODER_AUTO = 0, FPC32_CODER_BALLOT = 1;
constexpr uint32_t FPC32_FLAG_ORDER = 1u;      // a sampled step found the LDS exchange out of lane order (-> fpc32_distrust_lane_order())
constexpr uint32_t FPC32_FLAG_SCAN = 4u;       // a bounded wait between the workgroups of the one-sweep coder's scan kernel ran out (k_fpc32_scanfix)
constexpr uint32_t FPC32_FLAG_SENTINEL = 2u;   // a payload equal to the one-sweep coder's "never written" mark was written to a table
int launch_fpc32_encode(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                        uint8_t* d_ws, size_t ws_bytes, int coder = FPC32_CODER_AUTO, uint32_t* h_sizes = nullptr);
// (h_sizes: six words of pinned host memory that get the same words as d_sizes, valid when the stream is done - saves the copy launch)
void fpc32_distrust_lane_order();    // this device's exchange is not used again in this process
bool lds_lane_order_ok();            // the device applies the lanes of one LDS exchange in lane order (tested once per device)
int fpc32_code_sweep_mode();          // what FPC32_CODER_AUTO runs: 0 two sweeps + ballots, 2 two sweeps + exchange, 3 one sweep + exchange
int launch_fpc32_gather(uint32_t n, int arity, int c, const uint8_t* d_ws, uint8_t* d_dst);
int launch_fpc32_gather_all(uint32_t n, int arity, const uint8_t* d_ws, uint8_t* const d_dst[3]);
int launch_fpc32_gather_framed(uint32_t n, int arity, const uint8_t* d_ws, uint8_t* d_first, const uint32_t* d_sizes);
int launch_fpc32_compare(uint32_t n, int arity, const uint8_t* d_ws, const uint32_t* d_sizes, const uint8_t* const d_pay[3],
                         const uint32_t sizes[3], uint32_t* d_status, uint32_t flag);
// generic: sets bit `flag` of *d_status if *d_size != n_expected or the first n_expected bytes of a and b differ
int launch_bytes_compare(const uint8_t* d_a, const uint8_t* d_b, uint32_t n_expected, const uint32_t* d_size, uint32_t* d_status, uint32_t flag);
// the self-check of the chain decoders (shim.hip): re-encode + compare, bit 0x100 << c of *d_status where component c differs
bool decode_check_enabled();
int fpc_selfcheck_launch(const void* d_vals, uint32_t n, int arity, int width, const uint8_t* const d_pay[3], const uint32_t sizes[3],
                         DevBuf& vws, uint32_t* d_vsizes, uint32_t* d_status);
int decode_sabotage(int attempt, void* d_vals, uint32_t n, int arity, int width);   // test hook (no-op in the product library)
void set_first_attempt(int a);    // the single-stream ladder of this thread starts at attempt a (engine.hip hands streams over at 1)
bool force_serial();              // TRICO_HIP_SERIAL: see shim.hip
bool force_serial_stage(int bit);

// status bits of the chain decoders (k_fpc32_decode.hip, k_fpc64.hip), or-ed into a job's status word
constexpr uint32_t FPC_STATUS_SHORT = 1u;        // payload shorter than its 5-byte header
constexpr uint32_t FPC_STATUS_HEADER = 2u;       // value count or table exponents not what the launch was told
constexpr uint32_t FPC_STATUS_MALFORMED = 4u;    // the groups run past the end of the payload
constexpr uint32_t FPC_STATUS_TIMEOUT = 0x10u;   // the kernel's two waves lost each other (bounded wait): repeat the stream
constexpr uint32_t FPC_STATUS_CHECK = 0x700u;    // 0x100 << c: component c of the re-encode differs from the payload (shim.hip)

// One chain of a chain decoder: a component stream and where its values go.  `dst` points at the FIRST value of this component
// inside the interleaved output, `stride` is the number of components of the output (distance between values, in elements).
struct Fpc32ChainJob { const uint8_t* pay; uint32_t* dst; uint32_t* status; uint32_t size, n, stride, pad; };
struct Fpc64ChainJob { const uint8_t* pay; uint64_t* dst; uint32_t* status; uint32_t size, n, stride, pad; };

// latency-optimised float decoder (k_fpc32_decode.hip): one pair of waves per component stream
// d_tables: 8 KiB of scratch per component (the predictor tables, reached through the scalar data cache)
constexpr size_t FPC32_DECODE_TABLE_BYTES = 8192;
int launch_fpc32_decode(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, uint32_t n, void* d_dst,
                        uint32_t* d_status, uint32_t* d_tables);
// every chain of a batch in one launch: d_jobs = device table, d_scratch = FPC32_DECODE_TABLE_BYTES per chain
int launch_fpc32_decode_batch(const Fpc32ChainJob* d_jobs, uint32_t njobs, uint32_t* d_scratch, int chains_per_group);
// the same chains with tables, ring and counters in LDS (nothing behind the scalar cache): ~2 x the time per value, survives a
// save / restore of its workgroup; third rung of the repeat ladder, first choice under TRICO_HIP_DECODE_ROBUST=1
int launch_fpc32_decode_robust(const Fpc32ChainJob* d_jobs, uint32_t njobs);
bool decode_robust_first();       // TRICO_HIP_DECODE_ROBUST=1

// double-precision coder (k_fpc64.hip): one wave per component stream, 2 x 2^20-entry tables per stream in d_tables (zeroed)
int launch_fpc64_encode(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes, uint64_t* d_tables);
// d_tables: 2 x 2^20 entries per component, followed by FPC64_DECODE_SCRATCH_BYTES per component (rings of the kernel's two waves)
constexpr size_t FPC64_DECODE_SCRATCH_BYTES = 8192;
int launch_fpc64_decode(const uint8_t* const d_payloads[3], const uint32_t sizes[3], int arity, uint32_t n, void* d_dst,
                        uint64_t* d_tables, uint32_t* d_status);
// every chain of a batch in one launch: d_tables = FPC64_DECODE_CHAIN_BYTES per chain (2 x 2^20 entries, then the ring)
constexpr size_t FPC64_DECODE_CHAIN_BYTES = 2 * ((size_t)1 << 20) * 8 + FPC64_DECODE_SCRATCH_BYTES;
int launch_fpc64_decode_batch(const Fpc64ChainJob* d_jobs, uint32_t njobs, uint8_t* d_tables);

// sort-based throughput encoder for doubles (k_fpc64_sort.hip): table lookups as stable sorts by hash
uint32_t fpc64_sorted_threshold();
size_t fpc64_sorted_workspace(uint32_t n, int arity);
int launch_fpc64_encode_sorted(const void* d_src, uint32_t n, int arity, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                               uint8_t* d_ws, size_t ws_bytes);

// stable LSD radix sort of (u32 key, u32 value) pairs and exclusive u32 scan (k_sort.hip); workspaces in bytes
size_t sort_workspace(uint32_t n);
int radix_sort_pairs(const uint32_t* d_keys_in, const uint32_t* d_vals_in, uint32_t* d_keys_out, uint32_t* d_vals_out, uint32_t n,
                     int key_bits, uint8_t* d_ws, size_t ws_bytes);
size_t scan_workspace(uint32_t n);
int exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, uint32_t n, uint8_t* d_ws, size_t ws_bytes);

// vertex welding for the STL reader (k_weld.hip): sort + unique of corner positions
size_t weld_workspace(uint32_t n);
int launch_weld(const uint32_t* d_pos, uint32_t n, uint32_t* d_out_pos, uint32_t* d_out_tri, uint8_t* d_ws, size_t ws_bytes, uint32_t* d_result);

// workgroup-per-plane LZ4 compressor for small planes (k_lz4.hip)
int launch_lz4_encode_wave(const uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, int nplanes, uint8_t* d_out,
                           size_t out_stride, uint32_t* d_sizes);


// LDS-window LZ4 decompressor (k_lz4_decode.hip): one workgroup per plane
int launch_lz4_measure(const uint8_t* d_payload, uint32_t size, uint32_t capacity, uint32_t* d_out);      // decoded size of one block (k_lz4_decode.hip)
int launch_lz4_decode_lds(const uint8_t* const d_payloads[8], const uint32_t sizes[8], int nplanes,
                          uint8_t* d_planes, size_t plane_stride, uint32_t plane_bytes, uint32_t* d_status);

// data-parallel LZ4 decompressor (k_lz4_pdecode.hip): planes of at least lz4_pdecode_threshold() bytes
uint32_t lz4_pdecode_threshold();
size_t lz4_pdecode_workspace(uint32_t plane_bytes, const uint32_t* sizes, int nplanes);
int launch_lz4_decode_parallel(const uint8_t* const d_payloads[8], const uint32_t sizes[8], int nplanes, uint8_t* d_planes, size_t plane_stride,
                               uint32_t plane_bytes, uint32_t* d_status, uint8_t* d_ws, size_t ws_bytes);

// chunk-speculative exact LZ4 compressor for large planes (k_lz4_chunked.hip)
uint32_t lz4_chunked_threshold();
size_t lz4_chunked_workspace(uint32_t n, int nplanes, size_t plane_stride);
int launch_lz4_encode_chunked(const uint8_t* d_planes, size_t plane_stride, uint32_t n, int nplanes, uint8_t* d_out, size_t out_stride,
                              uint32_t* d_sizes, uint8_t* d_ws, size_t ws_bytes, uint32_t* d_status);

// component split / merge of interleaved reals (k_planes.hip): comp c of element i <-> d_soa + c * comp_stride bytes
int launch_deinterleave(const void* d_aos, uint32_t n, int arity, int width, uint8_t* d_soa, size_t comp_stride);
int launch_interleave(const uint8_t* d_soa, size_t comp_stride, uint32_t n, int arity, int width, void* d_aos);

// byte-plane split / merge (k_planes.hip)
int launch_planes_split(const void* d_src, uint32_t count, int width, uint8_t* d_planes, size_t plane_stride);
int launch_planes_merge(const uint8_t* d_planes, size_t plane_stride, uint32_t count, int width, void* d_dst);

} // namespace trico
