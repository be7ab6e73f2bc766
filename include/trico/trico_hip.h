/*
 * trico_hip.h — thin C-ABI shim between the host C container code (trico_amd/csrc/host/archive.c)
 * and the hand-written HIP kernels for gfx950 (trico_amd/csrc/hip/).  Plain C types only: no HIP,
 * torch or C++ types cross this boundary.  This surface has no reference counterpart; each entry
 * names the reference loops it replaces.
 *
 * Pointers named `src`/`dst`/`payloads[]` may be host or HIP device pointers (detected with
 * hipPointerGetAttributes); host data is staged through the context's device workspace.
 * All functions return 1 on success and 0 on failure unless stated; on failure
 * trico_hip_last_error() returns a static description.  Nothing aborts.
 */
#ifndef TRICO_TRICO_HIP_H
#define TRICO_TRICO_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#include "trico_api.h"

typedef struct trico_hip_ctx trico_hip_ctx;   /* per-archive device workspace (not thread-shared) */

/* ---- device / context ------------------------------------------------------------------- */
TRICO_API int  trico_hip_available(void);                 /* 1 if a HIP device can be used */
TRICO_API const char* trico_hip_last_error(void);         /* thread-local, never NULL */
TRICO_API trico_hip_ctx* trico_hip_ctx_create(void);      /* NULL if no device */
TRICO_API void trico_hip_ctx_destroy(trico_hip_ctx* ctx);
TRICO_API void trico_hip_set_stream(void* hip_stream);    /* stream for all later launches of this thread (default: null stream) */
TRICO_API int  trico_hip_synchronize(void);

/* ---- memory helpers ---------------------------------------------------------------------- */
TRICO_API int   trico_hip_pointer_is_device(const void* p);
TRICO_API void* trico_hip_device_alloc(size_t bytes);
TRICO_API void  trico_hip_device_free(void* p);
TRICO_API int   trico_hip_copy(void* dst, const void* src, size_t bytes);   /* any direction, complete on return */
TRICO_API uint64_t trico_hip_device_free_bytes(void);     /* free device memory right now (0 without a device) */

/* ---- floating-point streams ---------------------------------------------------------------
 * Replaces trico_transpose_*_aos_to_soa (transpose_aos_to_soa.c:8-82) fused with
 * trico_compress / trico_compress_double_precision (fpsc.c:86-210 / 576-800), called per
 * component with exponents (4,10) for width 4 and (20,20) for width 8 (trico.c:231,396).
 * `src`: n elements of `arity` interleaved components (arity 1..3), `width` 4 or 8 bytes.
 * On success sizes[c] is the payload size of component c; payloads stay in the context until
 * the next encode and are copied out with trico_hip_fetch_payload. */
TRICO_API int trico_hip_fpc_encode(trico_hip_ctx* ctx, const void* src, uint32_t n, int arity, int width, uint32_t sizes[3]);

/* Same with explicit table size exponents (the arguments of trico_compress, fpsc.c:86), normalised like the reference
 * does (odd values rounded down, at most 30).  Shapes other than the archive API's run in reference order on the device. */
TRICO_API int trico_hip_fpc_encode_ex(trico_hip_ctx* ctx, const void* src, uint32_t n, int arity, int width,
                                      uint32_t e1, uint32_t e2, uint32_t sizes[3]);

/* Stand-alone transposes (transpose_aos_to_soa.c:8-147), for the low-level API in
 * include/trico/transpose_aos_to_soa.h.  width 4 / 8: `arity` components of interleaved reals; width 1: the `arity`
 * byte planes of n `arity`-byte integers.  comps[c] and aos may be host or device pointers. */
TRICO_API int trico_hip_split_components(trico_hip_ctx* ctx, const void* aos, uint32_t n, int arity, int width, void* const* comps);
TRICO_API int trico_hip_merge_components(trico_hip_ctx* ctx, const void* const* comps, uint32_t n, int arity, int width, void* aos);

/* Inverse: trico_decompress / trico_decompress_double_precision (fpsc.c:212-417 / 803-1164) per
 * component, then trico_transpose_*_soa_to_aos.  Every payload must announce exactly `n` values.
 * dst == NULL decodes and discards. */
TRICO_API int trico_hip_fpc_decode(trico_hip_ctx* ctx, const uint8_t* const payloads[3], const uint32_t sizes[3],
                                   int arity, int width, uint32_t n, void* dst);

/* ---- integer streams ------------------------------------------------------------------------
 * Replaces trico_transpose_uint{16,32,64}_aos_to_soa (transpose_aos_to_soa.c:84-147) and one
 * LZ4_compress_default per byte plane (trico.c:343-368; lz4.c:1271 -> 793-1181), byte-exact with
 * LZ4 1.9.2.  `width` 1, 2, 4 or 8; width 1 is a single unsplit block (trico.c:630-656). */
TRICO_API int trico_hip_int_encode(trico_hip_ctx* ctx, const void* src, uint32_t count, int width, uint32_t sizes[8]);

/* Inverse: LZ4_decompress_safe per plane (trico.c:1100-1129) + trico_transpose_*_soa_to_aos. */
TRICO_API int trico_hip_int_decode(trico_hip_ctx* ctx, const uint8_t* const payloads[8], const uint32_t sizes[8],
                                   int width, uint32_t count, void* dst);

/* Decoded size of ONE LZ4 block whose size the caller only knows an upper bound of (what LZ4_decompress_safe's dstCapacity is,
 * lz4/lz4.h:153-158): the sequences are walked on the device without moving a byte.  0: malformed, or more than `capacity` bytes. */
TRICO_API int trico_hip_lz4_decoded_size(trico_hip_ctx* ctx, const void* payload, uint32_t size, uint32_t capacity, uint32_t* out_size);

/* ---- batched decode: one launch for all chains -------------------------------------------------
 * The format leaves ONE serial chain per floating-point component (fpsc.c:308-326), so decode throughput is the number of
 * chains in flight.  A job is one stream (what one trico_read_* call decodes, trico.c:943-1668); a batch may hold the streams
 * of one archive or of many.  trico_hip_decode_jobs decodes them all: every float chain of the batch in ONE kernel launch,
 * every double chain in one, the integer streams beside them, the chain decoders' self-checks behind them; it returns when
 * everything is in place.  `payloads[c]` / `dst` may be host or device pointers; dst == NULL skips the job.
 * fp: arity 1..3 components of `width` 4 / 8 bytes, n values per component, payloads[0..arity);
 * int: `width` 1, 2, 4 or 8 byte planes, n integers, payloads[0..width).
 * Returns 1 if every job succeeded; jobs[i].ok says which did: 1 decoded, 0 the stream is malformed or its decode failed (a
 * malformed stream fails alone), -1 not attempted - the batch as a whole could not run (workspaces, a launch or a copy failed):
 * nothing is known about the stream, decoding it in a smaller batch or by itself may succeed. */
typedef struct trico_hip_decode_job
  {
  int32_t is_int, arity, width;
  uint32_t n;
  const uint8_t* payloads[8];
  uint32_t sizes[8];
  void* dst;
  int32_t ok;              /* out: 1, 0 or -1, see above */
  int32_t other_writer;    /* out: 1 if the stream was delivered although the reference's encoder would not have written its payload (trico_hip_set_strict) */
  } trico_hip_decode_job;
TRICO_API int trico_hip_decode_jobs(trico_hip_decode_job* jobs, int count);
/* allocates (and keeps) the device workspaces a batch of this shape needs, without decoding: takes the allocation out of the
 * latency of the first trico_hip_decode_jobs call.  Later batches of the same or a smaller shape allocate nothing either way. */
TRICO_API int trico_hip_decode_jobs_reserve(const trico_hip_decode_job* jobs, int count);
/* The engine's workspaces and the library's pool of recycled device buffers are kept for the next call (grow-only); this gives
 * all of that memory back to the device (buffers owned by live archive handles stay).  Not to be called while another thread decodes. */
TRICO_API void trico_hip_release_workspaces(void);

/* ---- whole archives at once --------------------------------------------------------------------
 * trico_hip_list_streams: describes the streams from the cursor of a read archive to its end, without consuming them
 * (at most `cap` entries; returns how many there are, -1 if the framing is broken before the end).
 * trico_hip_read_archives: decodes the remaining streams of `count` read archives as ONE batch (trico_hip_decode_jobs).
 * dsts[a][s] receives stream s (counted from the cursor) of archive a: caller-allocated like the `*ptr` of the trico_read_*
 * calls (host or device), NULL skips the stream; nstreams[a] entries are given, streams beyond them stay unread.  On return
 * every archive's cursor is behind the last stream that was decoded (or skipped) successfully, exactly as if the matching
 * trico_read_* calls had been made one by one.  Returns 1 if all streams asked for were decoded. */
typedef struct trico_hip_stream_info
  {
  int32_t type;            /* enum trico_stream_type */
  int32_t is_int, arity, width;
  uint32_t count;          /* the count field of the stream (what the trico_get_number_of_* peek returns) */
  uint32_t n;              /* values per component / integers */
  uint64_t decoded_bytes;  /* size of the buffer a trico_read_* call fills */
  uint64_t payload_bytes;  /* compressed bytes of all its components */
  } trico_hip_stream_info;
TRICO_API int trico_hip_list_streams(void* archive, trico_hip_stream_info* out, int cap);
TRICO_API int trico_hip_read_archives(void* const* archives, int count, void* const* const* dsts, const int* nstreams);

/* ---- framing of a device-resident archive --------------------------------------------------------
 * trico_open_archive_for_reading accepts a device pointer; its readers need the few framing bytes of every stream on the
 * host (type byte, count, one size field per component: trico.c:100-124, 943-957 read them from host memory).  This walks up
 * to `cap` streams on the device, starting at the type byte at position `pos`, and brings their framing bytes back with one
 * copy.  ncomp_of_type[t] = components of stream type t (0: unknown, the walk stops).  head8 receives the 8 bytes of the file
 * header.  Returns the number of streams described (a stream cut off by the end of the archive is described as far as it
 * goes and ends the walk), -1 on a HIP error. */
typedef struct trico_hip_frame_bytes
  {
  uint64_t tpos;            /* position of the stream's type byte */
  uint64_t size_pos[8];     /* position of the size field of component c */
  uint32_t ncomp;           /* size fields found */
  uint32_t nbytes_head;     /* how many of the 5 bytes at tpos (type, count) lie inside the archive */
  uint8_t  size_valid[8];
  uint8_t  bytes[40];       /* type, count, then 4 bytes per size field */
  } trico_hip_frame_bytes;
TRICO_API int trico_hip_walk_frames(const uint8_t* d_data, uint64_t size, uint64_t pos, const uint8_t ncomp_of_type[21],
                                    trico_hip_frame_bytes* out, int cap, uint8_t head8[8]);

/* copy payload `c` of the last encode on this context to dst (host or device) */
TRICO_API int trico_hip_fetch_payload(trico_hip_ctx* ctx, int c, void* dst);
/* Vertex welding for the binary STL reader (trico_io/iostl.c:69-134): `corners` holds 3 * ntri positions (xyz floats, host
 * or device).  1: `vertices` (capacity 3 * ntri positions) receives the *nr_of_vertices unique positions in (x, y, z)
 * order and `triangles` the 3 * ntri re-indexed corners; 2: the input contains -0.0 or NaN, where the reference's result
 * depends on its quicksort's tie order — nothing was written, weld on the host; 0: error. */
TRICO_API int trico_hip_weld_vertices(trico_hip_ctx* ctx, const float* corners, uint32_t ntri, float* vertices, uint32_t* triangles,
                                      uint32_t* nr_of_vertices);

/* all `count` payloads of the last encode, payload c to dsts[c]; float payloads going to device memory take one
 * fused gather launch (what the archive writers use) */
TRICO_API int trico_hip_fetch_payloads(trico_hip_ctx* ctx, int count, void* const* dsts);
/* A float stream (width 4: trico_compress with the API's table sizes) coded AND framed in one queue of launches: from d_first on, in
 * device memory, `u32 bytes, payload` for each of the `arity` components (the body of a stream behind its type and count,
 * trico.c:215-262), placed by the sizes the device computed - the host waits once, at the end, and gets them in sizes[].  d_first needs
 * room for arity * (4 + 5 + 4 n + 3 (n / 8 + 2) + 8) bytes.  1: done; 0: error; -1: not this way (width, n == 0, destination not in
 * device memory, full verification switched on, or the encoder raised a flag) - nothing to rely on was written, call
 * trico_hip_fpc_encode + trico_hip_fetch_payloads, which cover every case.  What the archive writers do for device-resident archives. */
TRICO_API int trico_hip_fpc_encode_place(trico_hip_ctx* ctx, const void* src, uint32_t n, int arity, int width, void* d_first, uint32_t sizes[3]);
/* device address of payload `c` of the last encode (valid until the next encode on ctx) */
TRICO_API const uint8_t* trico_hip_payload_device_pointer(trico_hip_ctx* ctx, int c);

/* ---- device-resident archives ---------------------------------------------------------------
 * Same handle type and semantics as trico_open_archive_for_writing (trico.c:126-156) but the
 * archive buffer lives in HBM: trico_get_buffer_pointer then returns a device pointer.  Used when
 * the .trc bytes are consumed on the GPU (RCCL gather, device-side decode) so nothing crosses PCIe. */
TRICO_API void* trico_hip_open_archive_for_writing_device(uint64_t initial_buffer_size);


/* ---- stream-sharded encoding (one big mesh over several GPUs) --------------------
 * The independent units of a stream are its components (trico.c:229-260: x, y, z are compressed one after the other)
 * and, for integer streams, its byte planes (trico.c:346-368).  A rank encodes the units it owns, the payloads travel
 * to the root (trico_hip_comm_gather below), and the root frames them exactly as the writers do (type byte, count,
 * then per unit a 4-byte size and the payload).
 *
 * _encode_component: component `comp` of n interleaved `arity`-vectors of `width`-byte reals; _encode_plane: byte plane
 * `plane` of `count` integers of `width` bytes.  The payload is payload 0 of the context afterwards
 * (trico_hip_fetch_payload / trico_hip_payload_device_pointer). */
TRICO_API int trico_hip_fpc_encode_component(trico_hip_ctx* ctx, const void* src, uint32_t n, int arity, int width, int comp, uint32_t* size);
TRICO_API int trico_hip_int_encode_plane(trico_hip_ctx* ctx, const void* src, uint32_t count, int width, int plane, uint32_t* size);
/* Appends one stream to a writable archive from already encoded units: payloads[c] (host or device) of sizes[c] bytes,
 * c < nunits (3 / 2 / 1 components, or `width` planes).  `count_field` is the count the matching trico_write_* stores
 * (trico.c:215-858).  The archive bytes are those the writer itself would have produced. */
TRICO_API int trico_hip_append_encoded_stream(void* archive, int stream_type, uint32_t count_field, int nunits,
                                              const void* const* payloads, const uint32_t* sizes);

/* ---- RCCL exchange (one process per GPU; xGMI inside a node) -----------------------------------------
 * Thin wrapper over RCCL (loaded with dlopen when first used, so libtrico.so has no link-time dependency on it).
 * Rank 0 obtains an id and hands its 128 bytes to the other ranks by any means (MPI, a file, torch.distributed...);
 * every rank then creates its communicator with the GPU it will use made current. */
typedef struct trico_hip_comm trico_hip_comm;
TRICO_API int trico_hip_comm_unique_id(uint8_t id[128]);
TRICO_API trico_hip_comm* trico_hip_comm_create(const uint8_t id[128], int rank, int world);
TRICO_API void trico_hip_comm_destroy(trico_hip_comm* comm);
/* Every rank contributes `local_bytes` device bytes at d_local.  sizes[r] (host, `world` entries, filled on every rank)
 * receives the byte counts; on `root`, d_root receives the contributions back to back in rank order (capacity
 * root_capacity bytes; too small -> 0 on every rank after the size exchange, nothing moved).  One all-gather of the
 * sizes, then point-to-point transfers straight to their final offsets. */
TRICO_API int trico_hip_comm_gather(trico_hip_comm* comm, const void* d_local, uint64_t local_bytes, int root, void* d_root,
                                    uint64_t root_capacity, uint64_t* sizes);

/* ---- kernel timing (bench instrumentation) ---------------------------------------------------
 * When enabled, every launch of a hot kernel is bracketed by hipEvents on the launch stream and
 * the durations are accumulated per kernel id (TRICO_HIP_K_*).  Costs a sync per query only. */
enum trico_hip_kernel_id
  {
  TRICO_HIP_K_FPC32_ENCODE = 0,   /* float coder, all passes (AoS load -> payload bytes) */
  TRICO_HIP_K_FPC64_ENCODE = 1,
  TRICO_HIP_K_FPC32_DECODE = 2,
  TRICO_HIP_K_FPC64_DECODE = 3,
  TRICO_HIP_K_PLANES_SPLIT = 4,
  TRICO_HIP_K_PLANES_MERGE = 5,
  TRICO_HIP_K_LZ4_ENCODE   = 6,
  TRICO_HIP_K_LZ4_DECODE   = 7,
  TRICO_HIP_K_COUNT        = 8
  };
/* diagnostics: out[0] (this thread) = LZ4 chunks accepted by the stitch pass of the last trico_hip_int_encode,
 * out[1] = of those re-parsed serially (speculation not provably equivalent); 0,0 for small planes;
 * out[2] (process-wide) = float / double stream decodes that had to be repeated because the decoded values did not code
 * back to the payload (the decoders check themselves, see shim.hip; counts up, never reset);
 * out[3] (process-wide) = streams whose values, decoded in reference order at the end of the repeat ladder, still do not code back to their
 * payload: payloads the reference's encoder would not have written (decoded all the same; counts up, never reset) */
TRICO_API void trico_hip_last_stats(uint32_t out[4]);
/* which float encoder the library uses on the current device: 3 = ONE sweep, run starts resolved with one lane-ordered LDS exchange
 * per predictor and step (k_fpc32_sweep.hip; the choice once the device has passed the order test, which the first call runs:
 * ~0.3 ms), 0 = two sweeps, run starts resolved with ballots (k_fpc32_encode.hip; a device that fails the test, TRICO_FPC32_XCHG=0,
 * and every stream the one-sweep coder raised a flag on), 2 = two sweeps with the exchange (TRICO_FPC32_SWEEPS=2, measurements).
 * All write the same bytes. */
TRICO_API int trico_hip_fpc32_code_sweep(void);
/* write-side guard of the float encoder, process-wide, counting up: out[0] = streams coded again with the ballot coder because a
 * sampled step of the one-sweep coder found the LDS exchange out of lane order (the device is not asked again afterwards),
 * out[1] = because a value or stride equal to the coder's "never written" table mark was stored (2^-32 per value on random bits). */
TRICO_API void trico_hip_encode_stats(uint32_t out[2]);
/* ... and because a bounded wait between the workgroups of the encoder's scan kernel ran out (k_fpc32_scanfix; never seen outside the
 * test that provokes it) */
TRICO_API uint32_t trico_hip_encode_scan_recodes(void);
/* Opt-in FULL verification of the float encoder (process-wide; -1 returns to what TRICO_HIP_ENCODE_VERIFY=1 says, default off).
 * Every float stream of the API's table sizes is coded a second time by the two-sweep coder with ballots and the payloads are
 * compared byte for byte on the device before the call returns (about +1 ms per 50 M vertices of device time, and the payload is
 * gathered to the context first).  A difference is counted, printed on stderr, and the ballot coder's payload is used.
 * out[0] = streams verified, out[1] = values verified, out[2] = streams that differed (never seen so far). */
TRICO_API void trico_hip_set_encode_verify(int on);
TRICO_API void trico_hip_encode_verify_stats(uint64_t out[3]);
/* Reader policy for a float / double payload that decodes but that the reference's encoder would not have written (its values,
 * decoded in reference order, do not code back to it: another writer chose other, equally decodable codes - or the archive is
 * corrupt in a way that still parses).  Default (0, or -1 = what TRICO_HIP_STRICT says): the values are delivered and counted
 * (trico_hip_last_stats word 3, trico_hip_archive_other_writer_streams, trico_hip_decode_job::other_writer).  1: the read fails.
 * A caller that wrote its archives with this library or with the reference can treat every such stream as corruption. */
TRICO_API void trico_hip_set_strict(int on);
/* streams of this context / of this read archive handle that were delivered under the "another writer" rule above */
TRICO_API uint32_t trico_hip_ctx_other_writer_streams(const trico_hip_ctx* ctx);
TRICO_API uint32_t trico_hip_archive_other_writer_streams(void* archive);
TRICO_API void trico_hip_profile_enable(int on);
TRICO_API void trico_hip_profile_reset(void);
/* returns accumulated milliseconds and number of timed spans for kernel id `k` (syncs first) */
TRICO_API double trico_hip_profile_ms(int k, uint64_t* spans);

#if defined(__cplusplus)
}
#endif

#endif /* TRICO_TRICO_HIP_H */
