// liodom_hip.hip — host side of libliodom_hip.so: handle, HBM layout, launch sequencing and the
// C-ABI declared in include/liodom_hip.h.  No torch types, no CPU fallback: every entry point
// drives the HIP kernels of liodom_kernels.h or fails with an error code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "liodom_kernels.h"

using namespace liodom_dev;

namespace {

thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      char _buf[512];                                                                         \
      snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
               __FILE__, __LINE__);                                                           \
      g_last_error = _buf;                                                                    \
      return LIODOM_ERR_HIP;                                                                  \
    }                                                                                         \
  } while (0)

enum KernelId {
  KID_CLASSIFY = 0, KID_RING_EXTRACT, KID_COMPACT, KID_KNN, KID_LM, KID_HASH_CLEAR,
  KID_WINDOW_INSERT, KID_HASH_ALLOC, KID_HASH_SCATTER, KID_RING_SCATTER, KID_HASH_BUILD, KID_OTHER
};
const char* kKernelNames[LIODOM_NUM_KERNELS] = {
    "k_classify", "k_ring_extract", "k_compact_edges", "k_knn", "k_lm_solve",
    "k_hash_clear", "k_window_insert", "k_hash_alloc", "k_hash_scatter", "k_ring_scatter", "k_hash_build", "other"};

struct EventPair { hipEvent_t a, b; int kid; };

}  // namespace

#include "liodom_map_host.h"

std::atomic<int> g_live_handles{0};      // handles alive in this process (the overlapped second kNN pass is for a GPU one handle has to itself)

struct liodom_handle {
  liodom_handle() {
    for (int b = 0; b < kEdgePipeBufs; b++) { ev_free_valid[b] = false; eb_reader[b] = 0u; tk_seq[b] = 0u; }
  }
  liodom_params_t params;
  liodom_config_t config;
  DevView v{};
  DevView* d_view = nullptr;         // device copy: kernels take a pointer (8-byte kernarg)
  // Two sides, as in the reference's liodom_node (src/liodom_node.cc:89-91: one FeatureExtractor thread,
  // one LaserOdometer thread).  The EXTRACTION side owns stream_x, the ring-split scratch, stage_in and
  // edge buffer kEdgeBufX; the ODOMETRY side owns `stream`, edge buffers 0 / 1 / 2, the window, the hash and
  // the result records.  mx_x / mx_o serialise callers of each side; an entry point that needs both
  // (process_scan, the resident replay, reset, ...) takes mx_o first, then mx_x.  liodom_extract_edges
  // (mx_x only) and liodom_odometry_step (mx_o only) can therefore run concurrently from two threads.
  std::mutex mx_x, mx_o;
  hipStream_t stream = nullptr;      // odometry side
  hipStream_t stream_k = nullptr;    // overlapped second kNN pass of a scan (kernels_sync.h "Overlapped second kNN pass"): beside the first solve
  bool counted_live = false;         // this handle is part of g_live_handles
  bool ov_ok = false;                // the handle qualifies for it (one stream, streamed rebuild, the pass leaves 2/3 of the wave slots free)
  unsigned int ov_seq = 0;           // launch sequence number its flags carry
  hipEvent_t ev_ov = nullptr;        // recorded on the odometry stream in front of the first overlapped scan after scans that were not
  bool stream_c_shared = false;      // stream_c is stream_k
  bool ov_suppress = false;          // host-fed replay: the host's enqueue work per scan is the limit there, and the overlapped pass costs two more launches
  bool ov_prev = false;              // the previous scan of this handle was overlapped
  // Chain mode (kernels_sync.h "Chain mode"): the scan's kNN passes and the rebuild on stream_k, the two solves — launches of the
  // solving workgroups alone — on `stream`; the first solve's launch is resident beside the first pass.
  double replay_enq_ns = 0.0, replay_wait_ns = 0.0;      // depth-1 resident replay: host time per scan spent enqueueing / waiting for the previous pose
  long long replay_timed = 0;
  bool chain_ok = false;             // the handle qualifies (one stream, streamed rebuild, flags, no IMU override; the passes' waiting workgroups
                                     // may take up to half of the wave slots: the rebuild's workgroups are light there and the solve is resident
                                     // before the second pass is dispatched; LIODOM_CHAIN=0 switches it off)
  bool chain_prev = false;           // the previous scan was enqueued in chain mode
  unsigned int chain_count = 0;      // first-pass workgroups launched in chain mode since the last reset (what knn_done0 counts up to)
  int verdict_scan = -1;             // scans completed when the host last collected stream 0's pose, and whether that scan's speculative
  bool verdict_confirmed = false;    // hand-over was confirmed (HostOut::pad, written by finalize_scan)
  bool chain_fix_pending = false;    // the last chain-mode scan's APPEND may need the repair of k_chain_redo0 (speculative hand-over)
  bool chain_used = false;           // any scan was: the odometry side's results are complete when stream AND stream_k have drained
  hipEvent_t ev_ch = nullptr;        // at a switch out of chain mode: the odometry stream waits for stream_k
  int ov_warm = 0;                   // scans enqueued so far, up to kOvWarmScans (the first ones run every kernel of the chain for the first time)
  hipStream_t stream_x = nullptr;    // extraction side (liodom_extract_edges, and the next scan's extraction in the pipelined replay)
  hipStream_t stream_c = nullptr;    // host-fed replay: uploads (a copy engine works beside the extraction kernels of the previous scan)
  hipEvent_t ev_up[3] = {nullptr, nullptr, nullptr};      // staging slot uploaded
  hipEvent_t ev_xdone[3] = {nullptr, nullptr, nullptr};   // extraction that read the staging slot has been issued (recorded on the extraction stream)
  bool ev_xdone_valid[3] = {false, false, false};
  hipEvent_t ev_edges[kEdgePipeBufs] = {nullptr, nullptr, nullptr};   // edge buffer b written
  hipEvent_t ev_free[kEdgePipeBufs] = {nullptr, nullptr, nullptr};    // odometry finished reading edge buffer b
  std::atomic<bool> ev_free_valid[kEdgePipeBufs];      // (written by the odometry side, read by the extraction side)
  int parity = 0;                    // edge buffer of the next scan to enter odometry
  // pipelined replay without cross-stream events (pipe_flags in DevView; events remain for handles with >= 16 streams and as the fallback):
  unsigned int ext_seq = 0, odo_seq = 0;                 // extractions issued / odometries enqueued through the pipelined replay
  unsigned int eb_seq[kEdgePipeBufs] = {0, 0, 0};        // sequence number of the extraction last issued into buffer b
  std::atomic<unsigned int> eb_reader[kEdgePipeBufs];    // number of the odometry that last read buffer b (0: none to wait for); written by the odometry side
  // Device-resident hand-off of the two-thread binding (liodom_extract_edges_device on the extraction side fills pipeline buffer
  // x_next and returns a ticket; liodom_odometry_step_device on the odometry side consumes it): the element of the reference's
  // feature queue (shared_data.cc:64-89) without the cloud leaving HBM.
  std::atomic<unsigned int> tk_seq[kEdgePipeBufs];       // 0: slot free, else the sequence number of the extraction it holds; freed by the odometry side
                                                         // when the pose of that scan has been collected (its odometry has completed)
  int x_next = 0;                                        // slot of the next liodom_extract_edges_device (extraction side)
  int odo_fifo[2] = {0, 0}, odo_pending = 0;             // odometry side: slots of the submitted, not yet collected scans (oldest first)
  float4* host_edges = nullptr;      // host-mapped mirror of the dense edges of pipeline buffers 0..2 (one-stream handles; DevView::host_edges)
  int4* host_edges_meta = nullptr;
  unsigned int* host_edges_hdr = nullptr;
  float4* pin_ring = nullptr;        // page-locked scan staging ring [kEdgePipeBufs][max_points]: liodom_scan_buffer hands slots out, pageable scans are copied through it
  float4* stage_ring = nullptr;      // device side of the hand-off's uploads [kEdgePipeBufs][max_points] when they run on the copy stream
  std::vector<double> replay_stamps;  // (debug) host time, us since the call began, at which every pose of the last liodom_replay_resident call was collected
  bool fold_publish = true;          // k_compact_edges publishes the extraction itself (LIODOM_FOLD_PUBLISH=0: k_set_flag / k_publish_edges in a launch behind it)
  bool tk_copy_stream = false;       // ... (LIODOM_COPY_STREAM, default on): the upload of scan k+1 runs beside the extraction of scan k
  hipEvent_t ev_cp[kEdgePipeBufs] = {nullptr, nullptr, nullptr};      // the upload into device staging slot r has completed (copy stream)
  hipEvent_t ev_sdone[kEdgePipeBufs] = {nullptr, nullptr, nullptr};   // the extraction that read device staging slot r has been issued (recorded on the extraction stream)
  bool ev_sdone_valid[kEdgePipeBufs] = {false, false, false};
  int stage_next = 0;
  hipEvent_t ev_pin[kEdgePipeBufs] = {nullptr, nullptr, nullptr};     // the upload out of staging slot r has completed
  bool ev_pin_valid[kEdgePipeBufs] = {false, false, false};
  int pin_next = 0;
  const float4* replay_host_dev = nullptr;   // host-fed replay in progress with zero-copy input: device-visible address of the caller's buffer ...
  const float4* replay_host_base = nullptr;  // ... whose host address is this
  bool zero_copy = false;            // LIODOM_ZERO_COPY=1: page-locked scans are read over PCIe by the extraction's first kernel instead of being
                                     // uploaded by a copy call.  Measured slower (shader loads reach the host as 64-byte PCIe reads: two-thread
                                     // binding 10.3k -> 9.0k scans/s, host-fed replay 11.5k -> 8.9k): off by default
  bool safe_mode = false;            // no in-kernel waits at all: events between the streams, one workgroup per solve, three-kernel hash rebuild
  bool ring_split_lb = false;        // lock-step batches: k_ring_split_lb (one pass, rings at a fixed pitch, predecessors' counts summed as they appear); LIODOM_RING_SPLIT_LB=0: k_classify + k_ring_scatter
  unsigned int lb_tag = 0;           // launch tag its count words carry
  bool ring_split = true;            // ring split in one pass (k_ring_split) where every workgroup of the launch is resident at once; LIODOM_RING_SPLIT=0: always k_classify + k_ring_scatter
  int ring_split_max_wgs = 0;        // ... i.e. launches of at most this many workgroups (liodom_create: occupancy of k_ring_split x CUs, with headroom for the odometry chain's kernels)
  bool streams_concurrent = true;    // liodom_create's probe: kernels of two streams of this handle ran side by side
  std::atomic<bool> ov_off_for_copies{false};  // the overlapped pass's stream carries the hand-off's uploads (LIODOM_COPY_STREAM=2)
  std::atomic<bool> pipe_active{false};        // scans went through the pipeline edge buffers by ticket since the last drain
  std::atomic<bool> replay_live{false};        // scans went through them by the pipelined replay since the last drain (their odometries may not have been collected)
  std::atomic<bool> fallback_pending{false};   // a kernel of this handle gave up an in-kernel wait (LIODOM_STATUS_PIPE_TIMEOUT): liodom_reset() switches to events
  int pf_slot = -1;                  // resident slot whose extraction has been issued ahead
  int last_eb = 0;                   // edge buffer of the most recent scan that entered odometry (inspection)
  hipEvent_t pose_event = nullptr;
  int S = 1, H = 0, P = 0;
  size_t ring_lds_bytes = 0;
  // staging
  float4* stage_in = nullptr;        // [S][max_points]  (host-provided scans / edges)
  float4* resident = nullptr;        // [S][n_slots][max_points]
  int n_slots = 0;
  HostOut* host_out = nullptr;       // host-mapped pinned result records, one per stream
  std::vector<int> scans_enqueued;   // per stream: scans launched so far (expected HostOut.seq)
  std::vector<void*> allocs;
  // profiling
  std::atomic<bool> profiling{false};   // read without a lock by SideLocks / extract_queue, written under both mutexes
  std::vector<liodom_map*> mappers;   // per stream: attached device map (mapping replay) or null
  std::vector<int> mapper_cells_xy, mapper_cells_z;
  int hb_since = -1;            // hash_incr: scans since the last k_hash_build (-1: none yet)
  int knn8_grid = 1;            // k_knn8 workgroups per stream (each walks the blocks b, b + grid, ... of 32 queries)
  bool knn8 = false;            // handles with >= 16 streams: k_knn8 (eight lanes per query) instead of k_knn<128>; LIODOM_KNN8=0 keeps the latter
  bool lds_hash_build = false;  // k_hash_build (one workgroup per stream, LDS) instead of the 3 global-atomic kernels
  bool use_flags = false;       // pipelined replay: dependencies between the two streams through flags in device memory instead of events
  bool flag_gate = false;       // ... polled by a one-wave gate launch in front of the scan's first k_knn launch instead of by that launch itself
  std::vector<EventPair> ev_pool;
  size_t ev_used = 0;
  double k_ms[LIODOM_NUM_KERNELS] = {0};
  long long k_count[LIODOM_NUM_KERNELS] = {0};
};

namespace {

template <typename T>
int dev_alloc(liodom_handle* h, T** p, size_t count, int memset_value = 0) {
  void* raw = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIP_TRY(hipMalloc(&raw, bytes));
  h->allocs.push_back(raw);
  HIP_TRY(hipMemsetAsync(raw, memset_value, bytes, h->stream));
  *p = static_cast<T*>(raw);
  return LIODOM_OK;
}

// The odometry side's device results (window, correspondences, pose log, state) are complete when its streams have drained: in
// chain mode the kNN passes and the rebuild run on stream_k.
// Speculative hand-over of a chain-mode scan's result (kernels_sync.h): the repair of an APPEND that started from a pose the solve
// did not end with rides behind the NEXT scan's first pass (k_chain_redo0).  When no such pass follows — the handle leaves chain
// mode, or the host is about to read the odometry side's results — the repair is enqueued on its own (it waits for the scan's
// verdict; normally it finds nothing to do).
void chain_flush(liodom_handle* h) {
  if (!h->chain_fix_pending) return;
  h->chain_fix_pending = false;
  // (the last chain-mode scan's pose has been collected and its record says "confirmed": nothing to repair, no launch — the usual
  //  case when the host synchronises after a replay)
  if (h->verdict_scan == h->scans_enqueued[0] && h->verdict_confirmed) return;
  const int nA = (h->v.edge_cap + 255) / 256;
  hipLaunchKernelGGL(k_chain_redo0<256>, dim3(nA, 1), dim3(256), 0, h->stream_k, h->v, 0, 0, 0u, 0u, 0u, h->scans_enqueued[0], nA, 0, 1);
}
hipError_t sync_odometry(liodom_handle* h) {
  chain_flush(h);
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess && h->chain_used && h->stream_k) e = hipStreamSynchronize(h->stream_k);
  return e;
}

int drain_events(liodom_handle* h) {
  if (h->ev_used == 0) return LIODOM_OK;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->stream_x) HIP_TRY(hipStreamSynchronize(h->stream_x));
  for (size_t i = 0; i < h->ev_used; i++) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b));
    h->k_ms[h->ev_pool[i].kid] += ms;
    h->k_count[h->ev_pool[i].kid] += 1;
  }
  h->ev_used = 0;
  return LIODOM_OK;
}

struct ProfScope {
  liodom_handle* h;
  EventPair* ep = nullptr;
  hipStream_t st;
  ProfScope(liodom_handle* hh, int kid, hipStream_t stream = nullptr) : h(hh), st(stream ? stream : hh->stream) {
    if (!h->profiling) return;
    if (h->ev_used == h->ev_pool.size()) {
      if (h->ev_pool.size() < 4096) {
        EventPair e; e.kid = kid;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
        h->ev_pool.push_back(e);
      } else {
        drain_events(h);
      }
    }
    ep = &h->ev_pool[h->ev_used++];
    ep->kid = kid;
    hipEventRecord(ep->a, st);
  }
  ~ProfScope() { if (ep) hipEventRecord(ep->b, st); }
};

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- launch sequences -----------------------------------------------------------------------
// Feature extraction of `count` streams starting at s0; input scan for stream s0+i at in + i*stride.
// host_in (optional, device-visible pointer to page-locked HOST memory, stride host_stride per stream): the scan is read from
// there by the first kernel of the chain — no upload call — and `in` is the device buffer that kernel leaves a copy in.
int launch_extract(liodom_handle* h, hipStream_t q, int eb, int s0, int count, const float4* in, size_t in_stride,
                   int n, int height, int width, unsigned int wait_odo = 0, int mirror = 0,
                   const float4* host_in = nullptr, size_t host_stride = 0,
                   unsigned int* pub_flag = nullptr, unsigned int* pub_host = nullptr, unsigned int pub_value = 0u) {
  const DevView& v = h->v;
  const int tiles = std::max(1, cdiv(n, kTilePts));
  if (v.lidar_type == 1 && width > 0 && (long long)h->H * width <= (long long)v.max_points) {
    // organised cloud: ring = row, the split is a per-row compaction (no classify / scatter passes): one pass over the scan
    ProfScope ps(h, KID_RING_SCATTER, q);
    hipLaunchKernelGGL(k_row_compact, dim3(h->H, count), dim3(kRowThreads), 0, q, v, s0, host_in ? host_in : in, host_in ? host_stride : in_stride, n, height, width);
  } else {
    if (h->ring_split_lb && !host_in && !((long long)tiles * count <= h->ring_split_max_wgs && h->ring_split)) {
      // lock-step batches: one pass, rings at a fixed pitch, tiles sum their predecessors' counts (booked as the scatter)
      ProfScope ps(h, KID_RING_SCATTER, q);
      if (++h->lb_tag == 0u) h->lb_tag = 1u;
      hipLaunchKernelGGL(k_ring_split_lb, dim3(tiles * count), dim3(kTileThreads), ring_split_lb_lds_bytes(h->H), q, v, s0, in, in_stride, n, height, width, tiles, h->lb_tag);
      hipLaunchKernelGGL(k_ring_split_fix, dim3(1, count), dim3(kTileThreads), ring_split_lb_lds_bytes(h->H), q, v, s0, in, in_stride, n, height, width, tiles);
    } else if (h->ring_split && !host_in && (long long)tiles * count <= h->ring_split_max_wgs) {      // (every workgroup resident at once: k_ring_split waits inside the launch)
      // one pass: classification and scatter in one kernel (booked as the scatter)
      ProfScope ps(h, KID_RING_SCATTER, q);
      hipLaunchKernelGGL(k_ring_split, dim3(tiles, count), dim3(kTileThreads), ring_scatter_lds_bytes(h->H), q, v, s0, in, in_stride, n, height, width);
    } else {
    {
      ProfScope ps(h, KID_CLASSIFY, q);
      if (host_in) hipLaunchKernelGGL(k_classify, dim3(tiles, count), dim3(kTileThreads), 0, q, v, s0, host_in, host_stride, n, height, width, const_cast<float4*>(in), in_stride);
      else hipLaunchKernelGGL(k_classify, dim3(tiles, count), dim3(kTileThreads), 0, q, v, s0, in, in_stride, n, height, width, (float4*)nullptr, (size_t)0);
    }
    {
      ProfScope ps(h, KID_RING_SCATTER, q);
      hipLaunchKernelGGL(k_ring_scatter, dim3(tiles, count), dim3(kTileThreads), ring_scatter_lds_bytes(h->H), q, v, s0, in, in_stride, n);
    }
    }
  }
  {
    ProfScope ps(h, KID_RING_EXTRACT, q);
    const int ext = ring_extract_threads(v.scan_regions);
    // instance by the longest region the expected ring width gives (the last region takes the split's remainder)
    const int w = h->config.max_width > 0 ? h->config.max_width : std::max(1, h->config.max_points / std::max(1, h->H));
    const int total = std::max(0, w - 10), sector = total / std::max(1, v.scan_regions);
    const bool big = total - sector * (v.scan_regions - 1) > kExLPR * kExIPL;
    const dim3 grid(h->H, count), block(ext);
    if (ext <= 256) {
      if (big) hipLaunchKernelGGL((k_ring_extract<256, kExIPLBig>), grid, block, h->ring_lds_bytes, q, v, s0);
      else hipLaunchKernelGGL((k_ring_extract<256, kExIPL>), grid, block, h->ring_lds_bytes, q, v, s0);
    } else {
      if (big) hipLaunchKernelGGL((k_ring_extract<1024, kExIPLBig>), grid, block, h->ring_lds_bytes, q, v, s0);
      else hipLaunchKernelGGL((k_ring_extract<1024, kExIPL>), grid, block, h->ring_lds_bytes, q, v, s0);
    }
  }
  {
    ProfScope ps(h, KID_COMPACT, q);
    hipLaunchKernelGGL(k_compact_edges, dim3(kCompactBlocks, count), dim3(256), 0, q, v, s0, eb, wait_odo, mirror, pub_flag, pub_host, pub_value);
  }
  HIP_TRY(hipGetLastError());
  return LIODOM_OK;
}

// Odometry on the dense edges already on the device.  If pose_dst != nullptr the poses + infos
// of the streams are copied to pinned memory right after the solve and pose_event is recorded,
// so the host can pick them up while the window / hash rebuild still runs.
int enqueue_odometry(liodom_handle* h, int eb, int s0, int count, unsigned int wait_edges, unsigned int signal_odo);

int launch_odometry(liodom_handle* h, int eb, int s0, int count, unsigned int wait_edges = 0, unsigned int signal_odo = 0) {
  // (a hipGraph replay of this launch sequence was measured slower than the eager launches in rounds 1-2 — the chain is
  //  bound by its kernels, the host enqueue is hidden behind them, hipGraphLaunch adds start latency — and removed in round 3)
  const int rc = enqueue_odometry(h, eb, s0, count, wait_edges, signal_odo);
  if (rc) return rc;
  h->last_eb = eb;       // results are published by k_lm_solve into host-mapped memory (HostOut)
  for (int i = 0; i < count; i++) h->scans_enqueued[s0 + i]++;
  return LIODOM_OK;
}

int enqueue_odometry(liodom_handle* h, int eb, int s0, int count, unsigned int wait_edges, unsigned int signal_odo) {
  const DevView& v = h->v;
  const bool knn_small = h->S >= 16;            // many streams: 4 queries per workgroup, else 8
  if (v.use_imu) {
    ProfScope ps(h, KID_OTHER);
    hipLaunchKernelGGL(k_imu_override, dim3(cdiv(count, 64)), dim3(64), 0, h->stream, v, s0, count);
  }
  // early rebuild ("streamed rebuild", kernels_rebuild.h): the four launches of a scan carry extra workgroups that build
  // the next scan's cell hash in the second table; nothing follows the finalising solve
  const int map_blocks = cdiv(h->v.map_cap, 256);
  const bool early = v.early_rebuild != 0;
  const int nC = cdiv(h->v.edge_cap * std::max(1, h->P - 1), kLmThreads), nP = cdiv(h->v.edge_cap, kLmThreads);
  // Overlapped second kNN pass (kernels_sync.h): the pass goes to stream_k behind the first solve's launch and waits inside
  // the kernel; it needs kernels of different streams to run side by side (as the flags of the pipelined replay do) and the
  // GPU mostly to itself: not while a second handle lives in this process (its waiting workgroups and ours could end up
  // behind each other in a shared hardware queue), not under per-kernel profiling.
  // (chain mode — only for scans whose edges come from the extraction stream by flag, see below — also overlaps the pass on shapes
  //  where the four-launch chain cannot: Ouster-128's 704 waiting workgroups beside full-CU rebuild workgroups cost 9 %, in chain
  //  mode the overlapped pass gains 19 % there)
  const bool chain_cand = h->chain_ok && wait_edges != 0u && !h->flag_gate;
  const bool overlap_ok = early && (h->ov_ok || chain_cand) && h->use_flags && !h->profiling && !h->ov_suppress && !h->ov_off_for_copies.load() && count == 1 && g_live_handles.load() <= 1;
  // The first scans of a handle are not overlapped: their launches are the first of every kernel of the chain on this queue
  // (scratch set-up, code upload), which can hold the odometry stream back for longer than a waiting kernel is willing to
  // poll.  At a switch to overlapped scans stream_k waits (event) for the odometry stream to have drained, so that its
  // first polling launch cannot start before everything it depends on has run once.
  constexpr int kOvWarmScans = 2;
  const bool overlap = overlap_ok && h->ov_warm >= kOvWarmScans;
  if (h->ov_warm < kOvWarmScans) h->ov_warm++;
  unsigned int seq_k = 0u;                           // (0 means "not overlapped" to the kernels)
  if (overlap) {
    if (++h->ov_seq == 0u) h->ov_seq = 1u;
    seq_k = h->ov_seq;
    if (!h->ov_prev) {
      HIP_TRY(hipEventRecord(h->ev_ov, h->stream));
      HIP_TRY(hipStreamWaitEvent(h->stream_k, h->ev_ov, 0));
    }
  }
  // Chain mode: only for scans whose edges come from the extraction stream by flag (the pipelined replay, the ticket API): the
  // first pass then runs on stream_k, where nothing orders it behind an extraction enqueued on the odometry stream itself.
  const bool chain = overlap && chain_cand;
  if (chain != h->chain_prev) {
    if (chain) {
      // (stream_k waits for the odometry stream: recorded above — ev_ov — unless the previous scan was overlapped without the chain)
      if (h->ov_prev) { HIP_TRY(hipEventRecord(h->ev_ov, h->stream)); HIP_TRY(hipStreamWaitEvent(h->stream_k, h->ev_ov, 0)); }
    } else {
      chain_flush(h);
      HIP_TRY(hipEventRecord(h->ev_ch, h->stream_k));
      HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_ch, 0));
    }
  }
  h->chain_prev = chain;
  h->ov_prev = overlap;
  if (chain) {
    h->chain_used = true;
    const int scan_no = h->scans_enqueued[s0];
    h->chain_count += (unsigned int)v.knn_grid;                        // (wraps with the device counter: the kernels compare differences)
    const unsigned int done_target = h->chain_count;
    const int gx = (h->v.lm_groups - 1) * 8 + 1;                       // solvers on blocks 0, 8, 16, ... (one XCD); nothing else in the launch
    const size_t lds = lm_lds_bytes(h->v.edge_cap);
    // stream_k: first pass | gate (first solve's launch has started; + repair of the previous scan's hand-over: k_chain_redo0) | second pass + COUNT + PAD |
    //           ALLOC (+ repeat of unconfirmed second-pass workgroups: k_knn_redo) | APPEND + CLEAR + SCATTER
    // stream:   first solve (resident beside the first pass: waits for its done flags) | finalising solve (waits for the second pass's)
    const int nCP = cdiv(h->v.edge_cap * std::max(1, h->P - 1), 256) + cdiv(h->v.edge_cap, 256);      // COUNT + PAD workgroups of 256 threads
    hipLaunchKernelGGL((k_knn<256, false, true>), dim3(v.knn_grid, 1), dim3(256), 0, h->stream_k, v, s0, 0, eb, wait_edges, signal_odo, seq_k, scan_no);
    if (v.speculate) {      // (speculative hand-over of the previous scan's result not confirmed: rare)
      const int nA = cdiv(h->v.edge_cap, 256);
      // (+ 1: the gate in front of the second pass, k_ov_gate's job otherwise)
      hipLaunchKernelGGL(k_chain_redo0<256>, dim3(nA + v.knn_grid + 1, 1), dim3(256), 0, h->stream_k, v, s0, eb, wait_edges, signal_odo, seq_k, scan_no, nA, 1, h->chain_fix_pending ? 1 : 0);
    }
    h->chain_fix_pending = v.speculate != 0;
    hipLaunchKernelGGL((k_lm_solve<0, true>), dim3(gx, 1), dim3(kLmThreads), lds, h->stream, v, s0, eb, seq_k, done_target);
    if (!v.speculate) hipLaunchKernelGGL(k_ov_gate, dim3(1), dim3(64), 0, h->stream_k, v, s0, seq_k);
    hipLaunchKernelGGL((k_knn<256, true>), dim3(v.knn_grid + nCP, 1), dim3(256), 0, h->stream_k, v, s0, 1, eb, 0u, 0u, seq_k, scan_no);
    // (speculative hand-over not confirmed — rare —: the pass's workgroups once more; the launch's first workgroups are ALLOC)
    if (v.speculate) hipLaunchKernelGGL(k_knn_redo<256>, dim3(kRebuildAllocBlocks + v.knn_grid, 1), dim3(256), 0, h->stream_k, v, s0, eb, seq_k, scan_no, kRebuildAllocBlocks);
    hipLaunchKernelGGL((k_lm_solve<1, true>), dim3(gx, 1), dim3(kLmThreads), lds, h->stream, v, s0, eb, seq_k, done_target);
    if (!v.speculate) hipLaunchKernelGGL(k_rebuild_alloc, dim3(kRebuildAllocBlocks, 1), dim3(256), 0, h->stream_k, v, s0);
    const int nCf = cdiv(h->v.edge_cap * std::max(1, h->P - 1), kRebFinThreads), nPf = cdiv(h->v.edge_cap, kRebFinThreads);
    hipLaunchKernelGGL(k_rebuild_fin, dim3(nPf + kRebuildAuxBlocks + nCf, 1), dim3(kRebFinThreads), 0, h->stream_k, v, s0, eb);
    HIP_TRY(hipGetLastError());
    return LIODOM_OK;
  }
  for (int it = 0; it < 2; it++) {
    {
      ProfScope ps(h, KID_KNN);
      const int kx = v.knn_grid + ((early && it == 1 && !seq_k) ? kRebuildAuxBlocks : 0);     // it 1: + ALLOC (overlapped pass: k_rebuild_alloc below)
      if (knn_small && h->knn8) {
        // lock-step batches: eight lanes per query (kernels_knn8.h); the workgroups of a stream walk its blocks of 32 queries
        const dim3 g8(h->knn8_grid, count);
        if (it == 0) hipLaunchKernelGGL(k_knn8<0>, g8, dim3(kKnn8Threads), 0, h->stream, v, s0, eb);
        else hipLaunchKernelGGL(k_knn8<1>, g8, dim3(kKnn8Threads), 0, h->stream, v, s0, eb);
        hipLaunchKernelGGL(k_knn8_exact, dim3(kKnn8ExactBlocks, count), dim3(kKnn8Threads), 0, h->stream, v, s0, it, eb);      // (the ~1 % of the queries the fast path cannot certify)
        hipLaunchKernelGGL(k_line_gate, dim3(cdiv(h->v.knn_blocks * h->v.knn_queries, 256), count), dim3(256), 0, h->stream, v, s0, it, eb);
      } else if (knn_small) {
        hipLaunchKernelGGL(k_knn<128>, dim3(kx, count), dim3(128), 0, h->stream, v, s0, it, eb, wait_edges, signal_odo, 0u, 0);
        if (v.knn_nn) hipLaunchKernelGGL(k_line_gate, dim3(cdiv(h->v.knn_blocks * h->v.knn_queries, 256), count), dim3(256), 0, h->stream, v, s0, it, eb);
      } else if (it == 1 && seq_k) {
        hipLaunchKernelGGL(k_ov_gate, dim3(1), dim3(64), 0, h->stream_k, v, s0, seq_k);
        hipLaunchKernelGGL((k_knn<256, true>), dim3(kx, count), dim3(256), 0, h->stream_k, v, s0, it, eb, 0u, 0u, seq_k, -1);
        if (v.speculate) hipLaunchKernelGGL(k_knn_redo<256>, dim3(v.knn_grid, count), dim3(256), 0, h->stream_k, v, s0, eb, seq_k, -1, 0);      // (speculative hand-over not confirmed: rare)
        // ALLOC between the two solve launches, beside the pass's tail
        hipLaunchKernelGGL(k_rebuild_alloc, dim3(kRebuildAllocBlocks, count), dim3(256), 0, h->stream, v, s0);
      } else {
        hipLaunchKernelGGL(k_knn<256>, dim3(kx, count), dim3(256), 0, h->stream, v, s0, it, eb, wait_edges, signal_odo, 0u, 0);
      }
    }
    {
      ProfScope ps(h, KID_LM);
      // it 0: + COUNT, PAD; it 1: + APPEND, CLEAR, SCATTER
      const int extra = !early ? 0 : (it == 0 ? nC + nP : nP + kRebuildAuxBlocks + nC);
      const int gx = std::max(h->v.lm_groups + extra, (h->v.lm_groups - 1) * 8 + 1);      // solvers on blocks 0, 8, 16, ... (one XCD)
      if (it == 0) hipLaunchKernelGGL((k_lm_solve<0, false>), dim3(gx, count), dim3(kLmThreads), lm_lds_bytes(h->v.edge_cap), h->stream, v, s0, eb, seq_k, 0u);
      else hipLaunchKernelGGL((k_lm_solve<1, false>), dim3(gx, count), dim3(kLmThreads), lm_lds_bytes(h->v.edge_cap), h->stream, v, s0, eb, seq_k, 0u);
    }
  }
  if (v.mapping) {
    // synchronous replay of the mapping node for the streams with an attached map: updateMap(edges_k,
    // pose_k), then getLocalMap(pose_k) straight into the stream's received-map buffer
    for (int s = s0; s < s0 + count; s++) {
      liodom_map* mp = h->mappers[s];
      if (!mp) continue;
      ProfScope ps(h, KID_OTHER);
      StreamState* st = v.state + s;
      int rc = map_enqueue_update(mp, v.edges + ((size_t)eb * v.n_streams + s) * v.edge_cap, &st->n_edges_buf[eb], st->final_odom, h->stream);
      if (rc) return rc;
      rc = map_enqueue_local(mp, st->final_odom, h->mapper_cells_xy[s], h->mapper_cells_z[s], v.recv_pts + (size_t)s * v.recv_cap,
                             v.recv_cap, &st->n_recv, h->stream, 1);
      if (rc) return rc;
    }
  }
  if (early) {
    // (nothing: the next cell hash is complete when the finalising solve launch ends)
  } else if (h->lds_hash_build) {
    ProfScope ps(h, KID_HASH_BUILD);        // window append + LDS-built cell hash, one workgroup per stream
    // (hash_incr: the new frame is appended to the table of the last rebuild; k_hash_build only works when that says so)
    // (hash_incr: k_hash_build every kHbPeriod-th scan, k_hash_append — the new frame into the cells of the last rebuild — in between)
    const bool rebuild = !v.hash_incr || h->hb_since < 0 || h->hb_since >= kHbPeriod - 1;
    h->hb_since = rebuild ? 0 : h->hb_since + 1;
    if (!rebuild) hipLaunchKernelGGL(k_hash_append, dim3(count), dim3(kBuildThreads), 0, h->stream, v, s0, eb);
    else hipLaunchKernelGGL(k_hash_build, dim3(count), dim3(kBuildThreads), hash_build_lds_bytes(), h->stream, v, s0, eb);
  } else {
    {
      ProfScope ps(h, KID_WINDOW_INSERT);   // window append + cell hash in global memory, map_blocks workgroups per stream
      hipLaunchKernelGGL(k_window_insert, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0, eb);
    }
    {
      ProfScope ps(h, KID_HASH_ALLOC);
      hipLaunchKernelGGL(k_hash_alloc, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0);
    }
    {
      ProfScope ps(h, KID_HASH_SCATTER);
      hipLaunchKernelGGL(k_hash_scatter, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0);
    }
  }
  if (v.filter_local_map) {     // VoxelGrid(0.4) of the full window (every kernel exits unless the window is full)
    {
      ProfScope ps(h, KID_OTHER);
      hipLaunchKernelGGL(k_voxel_bbox, dim3(count), dim3(1024), 0, h->stream, v, s0);
      hipLaunchKernelGGL(k_voxel_insert, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0);
      hipLaunchKernelGGL(k_voxel_alloc, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0);
      hipLaunchKernelGGL(k_voxel_scatter, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0);
      hipLaunchKernelGGL(k_voxel_centroid, dim3(cdiv(h->v.map_cap, 8), count), dim3(256), 0, h->stream, v, s0);
      hipLaunchKernelGGL(k_filt_insert, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0);
    }
    {
      ProfScope ps(h, KID_HASH_ALLOC);
      hipLaunchKernelGGL(k_filt_alloc, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0);
    }
    {
      ProfScope ps(h, KID_HASH_SCATTER);
      hipLaunchKernelGGL(k_filt_scatter, dim3(map_blocks, count), dim3(256), 0, h->stream, v, s0);
    }
  }
  HIP_TRY(hipGetLastError());
  return LIODOM_OK;
}

// Every entry point starts here: the calling thread's current device becomes the handle's (other
// handles / maps of the process may live on other GPUs, and HIP's current device is per thread).
int enter(liodom_handle* h) {
  if (!h) return LIODOM_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->config.device));
  return LIODOM_OK;
}
int check_stream(liodom_handle* h, int stream) {
  int rc = enter(h);
  if (rc) return rc;
  if (stream < 0 || stream >= h->S) { g_last_error = "stream index out of range"; return LIODOM_ERR_INVALID_ARG; }
  return LIODOM_OK;
}
// entry points that enqueue scans: not after an in-kernel wait of this handle gave up (see wait_pose)
int check_usable(liodom_handle* h) {
  if (h->fallback_pending.load()) {
    g_last_error = "the handle had a LIODOM_STATUS_PIPE_TIMEOUT: call liodom_reset() before processing further scans";
    return LIODOM_ERR_HIP;
  }
  return LIODOM_OK;
}
// Lock guards of the two sides.  While per-kernel profiling is on, extraction runs on the odometry
// stream (so that HIP-event durations are not inflated by the other side's kernels) and shares the
// event pool: then every entry point takes both locks and the two sides are serialised.
struct SideLocks {
  std::unique_lock<std::mutex> lo, lx;
  SideLocks(liodom_handle* h, bool odo, bool ext) {
    if (odo || h->profiling) lo = std::unique_lock<std::mutex>(h->mx_o);
    if (ext || h->profiling) lx = std::unique_lock<std::mutex>(h->mx_x);
    // profiling was switched on between the test above and the locks (set_profiling holds both): take the rest
    if (h->profiling && !(lo.owns_lock() && lx.owns_lock())) {
      if (lx.owns_lock()) lx.unlock();
      if (!lo.owns_lock()) lo = std::unique_lock<std::mutex>(h->mx_o);
      lx = std::unique_lock<std::mutex>(h->mx_x);
    }
  }
};
hipStream_t extract_queue(liodom_handle* h) { return h->profiling ? h->stream : h->stream_x; }

int round_up(int x, int m) { return (x + m - 1) / m * m; }

// The plain (non-pipelined) entry points run everything on h->stream with edge buffer 0; make
// sure no extraction issued ahead by the pipelined replay is still in flight.
int tickets_idle(liodom_handle* h) {      // the plain entry points use pipeline edge buffer 0 themselves
  for (int b = 0; b < kEdgePipeBufs; b++) {
    if (h->tk_seq[b].load() != 0u) { g_last_error = "edge tickets of liodom_extract_edges_device are outstanding: consume them with liodom_odometry_step_device first"; return LIODOM_ERR_BUSY; }
  }
  return LIODOM_OK;
}
int drain_pipeline(liodom_handle* h) {
  const bool replayed = h->replay_live.exchange(false);
  if (h->pf_slot >= 0 || h->parity != 0 || h->pipe_active.exchange(false) || replayed) {
    HIP_TRY(hipStreamSynchronize(h->stream_x));
    HIP_TRY(sync_odometry(h));
    h->pf_slot = -1; h->parity = 0; h->ev_free_valid[0] = h->ev_free_valid[1] = h->ev_free_valid[2] = false;
    for (int b = 0; b < kEdgePipeBufs; b++) h->eb_reader[b] = 0;      // (everything has completed: nothing to wait for)
  }
  return LIODOM_OK;
}

// The odometry of the scan in pipeline edge buffer eb behind whatever produces that buffer (extraction number wait_seq with
// flags, ev_edges[eb] with events), as the pipelined replay and the device-resident hand-off enqueue it.  Odometry side.
int enqueue_pipeline_odometry(liodom_handle* h, int eb, unsigned int wait_seq) {
  int rc;
  if (h->use_flags) {
    // no cross-stream events (they cost ~11 us of idle odometry stream per scan, with the host far ahead as well): the
    // first kNN launch waits for the extraction's flag and signals that the previous odometry has completed
    const unsigned int m = ++h->odo_seq == 0 ? ++h->odo_seq : h->odo_seq;
    if (h->flag_gate) {
      hipLaunchKernelGGL(k_pipe_gate, dim3(1), dim3(64), 0, h->stream, h->v, 0, eb, wait_seq, m - 1u);
      rc = launch_odometry(h, eb, 0, h->S);
    } else {
      rc = launch_odometry(h, eb, 0, h->S, wait_seq, m - 1u);
    }
    if (rc) return rc;
    h->eb_reader[eb] = m;
  } else {
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_edges[eb], 0));
    rc = launch_odometry(h, eb, 0, h->S);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev_free[eb], h->stream));
    h->ev_free_valid[eb] = true;
  }
  return LIODOM_OK;
}

// Extraction of resident slot `slot` into edge buffer `eb` on the extraction stream.
int issue_extract(liodom_handle* h, int slot, int eb, int n, int height, int width, const float4* host_dev = nullptr, size_t host_stride = 0) {
  // while per-kernel profiling is on, everything runs on one stream so that the HIP-event
  // durations are not inflated by kernels of the other stream sharing the GPU
  hipStream_t q = extract_queue(h);
  const float4* in = h->resident + (size_t)slot * h->S * (size_t)h->v.max_points;
  if (h->use_flags) {
    // dependencies through flags in device memory (pipe_wait / k_set_flag): the buffer's last reader must have
    // completed before k_compact_edges rewrites it; the flag of this extraction is set by a launch that follows it
    h->eb_seq[eb] = ++h->ext_seq;
    if (h->ext_seq == 0) h->eb_seq[eb] = ++h->ext_seq;      // (0 means "nothing to wait for")
    // (fold_publish: the last workgroup of k_compact_edges sets the flag; else a launch of its own behind it)
    int rc = launch_extract(h, q, eb, 0, h->S, in, (size_t)h->v.max_points, n, height, width, h->eb_reader[eb], 0, host_dev, host_stride,
                            h->fold_publish ? h->v.pipe_flags + eb : nullptr, nullptr, h->fold_publish ? h->eb_seq[eb] : 0u);
    if (rc) return rc;
    if (!h->fold_publish) hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, q, h->v.pipe_flags + eb, h->eb_seq[eb]);
    HIP_TRY(hipGetLastError());
    return LIODOM_OK;
  }
  if (h->ev_free_valid[eb]) HIP_TRY(hipStreamWaitEvent(q, h->ev_free[eb], 0));
  int rc = launch_extract(h, q, eb, 0, h->S, in, (size_t)h->v.max_points, n, height, width, 0u, 0, host_dev, host_stride);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h->ev_edges[eb], q));
  return LIODOM_OK;
}

// Safe mode: every dependency that a kernel of this handle would wait for INSIDE a kernel is replaced by one the runtime orders.
// In-kernel waits need the producer to run beside the waiter; a GPU saturated by another process (or a tool that serialises
// kernels) breaks that, the bounded waits give up (LIODOM_STATUS_PIPE_TIMEOUT / LM_SYNC_TIMEOUT) and the scan is lost.  Afterwards:
//   stream dependencies   flags polled by kernels            -> hipEvent pairs
//   second kNN pass       beside the first solve, polling     -> behind it in stream order
//   pose solve            G workgroups exchanging partial sums in the launch -> one workgroup (sums in a different order: poses
//                         agree with the G-workgroup solve to rounding, not to the bit)
//   ring split            one pass whose tiles wait for each other's histograms -> k_classify + k_ring_scatter (bit-identical)
//   hash rebuild          workgroups inside the solve launches waiting for its pose -> k_window_insert / k_hash_alloc /
//                         k_hash_scatter behind the solve (bit-identical: test_early_rebuild_equals_three_kernel_rebuild)
// Entered by liodom_reset() after a timeout, or at creation with LIODOM_SAFE_MODE=1.
void enter_safe_mode(liodom_handle* h) {
  h->safe_mode = true;
  h->use_flags = false;
  h->chain_ok = false;
  h->v.lm_groups = 1;
  h->v.early_rebuild = 0;      // (the second table, the padding and the overflow list stay allocated and unused)
  h->ring_split = false;       // k_ring_split's workgroups wait for each other inside the launch: k_classify + k_ring_scatter instead
  h->ring_split_lb = false;    // (k_ring_split_lb's tiles wait for their predecessors' counts; the pitched buffers stay allocated)
}

int reset_state(liodom_handle* h) {
  // nothing of an earlier scan may still be in flight: its finalize would publish into the records
  // zeroed below, and an extraction issued ahead would write into scratch that is being reset
  if (h->stream_x) HIP_TRY(hipStreamSynchronize(h->stream_x));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->stream_k) HIP_TRY(hipStreamSynchronize(h->stream_k));
  if (h->fallback_pending.exchange(false)) enter_safe_mode(h);      // an in-kernel wait gave up (wait_pose)
  std::vector<StreamState> init((size_t)h->S);
  for (auto& st : init) {
    std::memset(&st, 0, sizeof(st));
    iso_identity(st.odom); iso_identity(st.prev_odom); iso_identity(st.final_odom); iso_identity(st.pred_odom[0]); iso_identity(st.pred_odom[1]);
    st.param_q[3] = 1.0;
    st.table_mask = (uint32_t)h->v.table_size - 1u;
  }
  HIP_TRY(hipMemcpyAsync(h->v.state, init.data(), sizeof(StreamState) * init.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemsetAsync(h->v.pub_counter, 0, sizeof(unsigned int) * (size_t)kEdgeBufs, h->stream));
  if (h->v.lb_ticket) { HIP_TRY(hipMemsetAsync(h->v.lb_ticket, 0, sizeof(unsigned int), h->stream)); HIP_TRY(hipMemsetAsync(h->v.lb_ovf, 0, sizeof(unsigned int) * (size_t)h->S, h->stream)); }
  {
    std::vector<double> qid((size_t)h->S * 4, 0.0);          // IMU orientation: identity until imuClb delivers one
    for (int s = 0; s < h->S; s++) qid[(size_t)s * 4 + 3] = 1.0;
    HIP_TRY(hipMemcpy(h->v.imu_q, qid.data(), sizeof(double) * qid.size(), hipMemcpyHostToDevice));
  }
  {
    const size_t total = (size_t)h->S * h->v.table_size * (h->v.early_rebuild ? 2 : 1);
    hipLaunchKernelGGL(k_init_cells, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->v);
    HIP_TRY(hipGetLastError());
  }
  h->pf_slot = -1; h->parity = 0; h->last_eb = 0; h->ev_free_valid[0] = h->ev_free_valid[1] = h->ev_free_valid[2] = false;
  h->replay_live.store(false);
  h->hb_since = -1;
  h->ext_seq = h->odo_seq = 0;
  // the first scans after a reset are not overlapped (as after liodom_create): the first one runs with st.initialized == 0, where
  // no first solve publishes the pose an overlapped second kNN pass would wait for
  h->ov_warm = 0; h->ov_prev = false;
  for (int b = 0; b < kEdgePipeBufs; b++) { h->tk_seq[b] = 0u; h->ev_pin_valid[b] = false; h->ev_sdone_valid[b] = false; }      // outstanding edge tickets are void
  if (h->stream_c && !h->stream_c_shared) HIP_TRY(hipStreamSynchronize(h->stream_c));
  h->x_next = 0; h->odo_pending = 0;
  if (h->host_edges_hdr) std::memset(h->host_edges_hdr, 0, sizeof(unsigned int) * 2 * kEdgePipeBufs);
  for (int b = 0; b < kEdgePipeBufs; b++) { h->eb_seq[b] = 0; h->eb_reader[b] = 0; }
  HIP_TRY(hipMemsetAsync(h->v.pipe_flags, 0, sizeof(unsigned int) * (kEdgePipeBufs + 1), h->stream));
  HIP_TRY(hipMemsetAsync(h->v.lm_xch, 0, sizeof(unsigned long long) * (size_t)h->S * 2 * kLmGroupsMax * 64, h->stream));
  HIP_TRY(hipMemsetAsync(h->v.pose_xch, 0, sizeof(unsigned long long) * (size_t)h->S * 64, h->stream));
  HIP_TRY(hipMemsetAsync(h->v.redo_sync, 0, sizeof(unsigned int) * 64, h->stream));
  if (h->v.pred_xch) HIP_TRY(hipMemsetAsync(h->v.pred_xch, 0, sizeof(unsigned long long) * (size_t)h->S * kOvReplicas * 512, h->stream));
  HIP_TRY(hipMemsetAsync(h->v.knn_done0, 0, sizeof(unsigned int) * ((size_t)h->S + 64), h->stream));
  h->chain_prev = false; h->chain_count = 0; h->chain_fix_pending = false; h->verdict_scan = -1; h->replay_enq_ns = 0.0; h->replay_wait_ns = 0.0; h->replay_timed = 0;
  std::memset(h->host_out, 0, sizeof(HostOut) * 2 * (size_t)h->S);
  std::fill(h->scans_enqueued.begin(), h->scans_enqueued.end(), 0);
  HIP_TRY(hipMemsetAsync(h->v.win_n, 0, sizeof(int) * (size_t)h->S * h->P, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return LIODOM_OK;
}

}  // namespace

extern "C" {

const char* liodom_last_error(void) { return g_last_error.c_str(); }

void liodom_params_default(liodom_params_t* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->min_range = 3.0;            // params.cc:40
  p->max_range = 75.0;           // params.cc:44
  p->lidar_type = 0;             // params.cc:48
  p->scan_lines = 64;            // params.cc:52
  p->scan_regions = 8;           // params.cc:56
  p->edges_per_region = 10;      // params.cc:60
  p->min_points_per_scan = 8 * 10 + 10;  // params.cc:63
  p->local_map_size = 5;         // params.cc:90-93
  p->save_results = 0;           // params.cc:66
  std::strcpy(p->results_dir, "~/");        // params.cc:70
  std::strcpy(p->fixed_frame, "odom");      // params.cc:74
  std::strcpy(p->base_frame, "base_link");  // params.cc:78
  p->laser_frame[0] = 0;                    // params.cc:82
  p->use_imu = 0; p->filter_local_map = 0; p->mapping = 0;   // params.cc:96,100,104
  p->publish_tf = 1;                                         // params.cc:108
}

void liodom_config_default(liodom_config_t* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->device = 0; c->n_streams = 1; c->max_points = 64 * 1800; c->max_width = 1800;
  c->reserved1 = 0; c->lm_apply_step_on_ftol = 0; c->pose_log_capacity = 1024; c->debug_buffers = 0;
  c->lm_workgroups = 0;
  c->pose_rotation_mode = 1;    // Eigen 3.3.x Transform::rotation()
}

int liodom_create(const liodom_params_t* params, const liodom_config_t* config, liodom_handle_t** out) {
  if (!params || !config || !out) return LIODOM_ERR_INVALID_ARG;
  *out = nullptr;
  if (params->scan_lines < 1 || params->scan_lines > 254 || params->scan_regions < 1 ||
      params->edges_per_region < 0 || params->local_map_size < 1 || params->local_map_size > (uint64_t)kMaxFrames ||
      config->n_streams < 1 || config->max_points < 1 || !(params->max_range > params->min_range)) {
    g_last_error = "liodom_create: parameter out of range";
    return LIODOM_ERR_INVALID_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_last_error = "no HIP device available (libliodom_hip has no CPU fallback)";
    return LIODOM_ERR_NO_DEVICE;
  }
  if (config->device < 0 || config->device >= ndev) { g_last_error = "bad device ordinal"; return LIODOM_ERR_INVALID_ARG; }
  HIP_TRY(hipSetDevice(config->device));
  liodom_handle* h = new liodom_handle();
  h->params = *params;
  h->config = *config;
  h->S = config->n_streams; h->H = params->scan_lines; h->P = (int)params->local_map_size;
  int rc = LIODOM_OK;
  auto fail = [&](int code) { liodom_destroy(h); return code; };
  // The odometry chain is the critical path; the extraction of the next scan only has to finish
  // before that chain ends.  Stream priorities let the chain's kernels win the CUs when both want them.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  const bool use_prio = std::getenv("LIODOM_NO_STREAM_PRIORITY") == nullptr && prio_least != prio_greatest;
  // (a CU split — hipExtStreamCreateWithCUMask: the extraction stream on 32 / 64 / 96 CUs, the chain's streams on the rest —
  //  was measured: -0.1 / -1.5 / -2.7 %; the wave priority of the chain's kernels, LIODOM_CHAIN_PRIO, is what helps)
  auto make_stream = [&](hipStream_t* st, int prio) {
    return use_prio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  };
  if (make_stream(&h->stream, prio_greatest) != hipSuccess) { g_last_error = "hipStreamCreate failed"; return fail(LIODOM_ERR_HIP); }
  if (hipEventCreateWithFlags(&h->ev_ov, hipEventDisableTiming) != hipSuccess) { g_last_error = "hipEventCreate failed"; return fail(LIODOM_ERR_HIP); }
  if (hipEventCreateWithFlags(&h->pose_event, hipEventDisableTiming) != hipSuccess) { g_last_error = "hipEventCreate failed"; return fail(LIODOM_ERR_HIP); }
  // (extraction: one level below the odometry stream, not the lowest: kernels of the odometry stream may wait in-kernel for it)
  const int prio_x = (prio_least - prio_greatest >= 2) ? prio_greatest + 1 : prio_least;
  if (make_stream(&h->stream_x, prio_x) != hipSuccess) { g_last_error = "hipStreamCreate failed"; return fail(LIODOM_ERR_HIP); }
  for (int b = 0; b < kEdgePipeBufs; b++) {
    if (hipEventCreateWithFlags(&h->ev_edges[b], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_free[b], hipEventDisableTiming) != hipSuccess) { g_last_error = "hipEventCreate failed"; return fail(LIODOM_ERR_HIP); }
  }

  DevView& v = h->v;
  v.min_range = params->min_range; v.max_range = params->max_range;
  v.lidar_type = params->lidar_type; v.scan_lines = params->scan_lines;
  v.scan_regions = params->scan_regions; v.edges_per_region = params->edges_per_region;
  v.min_points_per_scan = (long long)params->min_points_per_scan;
  v.prev_frames = h->P;
  v.apply_on_ftol = config->lm_apply_step_on_ftol;
  v.rotation_mode = config->pose_rotation_mode != 0 ? 1 : 0;
  v.filter_local_map = (params->filter_local_map && !params->mapping) ? 1 : 0;   // laser_odometry.cc:286
  // auto: several CUs per solve pay off only when one CU would spend >> the ~4.5 us in-launch
  // exchange on an evaluation (measured: ~2000 edges -> no gain; Ouster-128 shape -> yes)
  // Measured on MI355X (headline shape): one stream rebuilds its hash in 28 us with the three
  // global-atomic kernels (many workgroups) but needs 86 us as a single LDS workgroup; 64 lock-step
  // streams need 247 us (L2-atomic bound) against 103 us with one LDS workgroup each.
  h->lds_hash_build = config->n_streams >= 16;
  {
    // Flags instead of events between the extraction and the odometry stream: the first kNN launch of a scan polls the
    // extraction's flag in every workgroup, so all its workgroups must fit on the GPU with ample room left for the
    // extraction kernels they may be waiting for (512-thread workgroups): one stream only, and at most 12 of the 24
    // wave slots per CU the kernel's 74 VGPRs allow — HDL-64 (2 816 waves of 3 072) qualifies, Ouster-128 (5 632) does
    // not: with 24 it starved the extraction in the non-pipelined replay until the bounded waits gave up.
    // (Lock-step batches are throughput-bound: 11 us per multi-millisecond step do not matter there.)
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, config->device);
    const int ecap = round_up(std::max(1, params->scan_lines * params->scan_regions * (params->edges_per_region + 1)), 64);
    h->use_flags = config->n_streams < 16;
    h->flag_gate = !(config->n_streams == 1 && cdiv(cdiv(ecap, 8), 2) * 4 <= cus * 12);     // (larger launches: a one-wave gate launch polls instead)
    // kernels of different streams never run side by side under these: the in-kernel waits could only time out
    for (const char* name : {"AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING", "ROCPROFILER_PMC", "ROCPROF_COUNTERS"}) {
      const char* e = std::getenv(name);
      if (e && e[0] && std::strcmp(e, "0") != 0) h->use_flags = false;
    }
    if (const char* e = std::getenv("LIODOM_PIPE_FLAGS")) { if (std::atoi(e) == 0) h->use_flags = false; }
  }
  if (const char* e = std::getenv("LIODOM_HASH_BUILD")) h->lds_hash_build = std::strcmp(e, "global") != 0;
  v.lds_cells_max = kLdsCellsMax;
  if (const char* e = std::getenv("LIODOM_LDS_CELLS_MAX")) v.lds_cells_max = std::max(1, std::min(kLdsCellsMax, std::atoi(e)));
  // solve split over G workgroups (partial sums exchanged inside the launch, ~3 us per evaluation under load): pays once an
  // evaluation is long enough.  Measured (scans/s, G = 1 / 4 / 8): HDL-64 10.1k / 10.35k / 10.34k, Ouster-128 7.3k / 8.1k / 8.3k,
  // VLP-16 12.2k / 12.1k / -.
  {
    const int ecap = params->scan_lines * params->scan_regions * (params->edges_per_region + 1);
    // (round 3, A/B on one box, scans/s: HDL-64 G = 2 / 4 / 8: 11.4k / 11.9k / 12.1k; VLP-16 G = 1 / 2 / 4 / 8: 13.2k / 13.5k / 14.0k / 14.0k;
    //  16 workgroups — a build with kLmGroupsMax = 16 — lose: the exchange's fan-in grows, HDL-64 11.4k, Ouster-128 8.8k vs 9.05k)
    const int auto_g = config->n_streams > 4 ? 1 : (ecap >= 2048 ? kLmGroupsMax : (ecap >= 512 ? 4 : 1));
    v.lm_groups = config->lm_workgroups == 0 ? auto_g
                                             : (config->lm_workgroups >= kLmGroupsMax ? kLmGroupsMax : (config->lm_workgroups < 1 ? 1 : config->lm_workgroups));
  }
  if (const char* e = getenv("LIODOM_LM_GROUPS")) { const int gq = atoi(e); if (gq >= 1 && gq <= kLmGroupsMax) v.lm_groups = gq; }
  v.vox_inv = 1.0f / 0.4f;                                                          // setLeafSize(0.4) :290
  v.n_streams = h->S;
  v.max_points = config->max_points;
  // (k_ring_extract stages nothing per point in LDS, so there is no per-ring capacity — a ring may hold up to max_points
  // points; config.max_width only picks the kernel instance, launch_extract)
  v.ring_cap = config->max_points;
  v.slots_per_ring = params->scan_regions * (params->edges_per_region + 1);
  h->ring_lds_bytes = ring_extract_lds_bytes(v.slots_per_ring, params->scan_regions);
  if (h->ring_lds_bytes > 160 * 1024) { g_last_error = "pick lists (scan_regions * (edges_per_region + 1)) exceed 160 KiB of LDS"; return fail(LIODOM_ERR_CAPACITY); }
  v.edge_cap = round_up(std::max(1, h->H * v.slots_per_ring), 64);
  v.use_imu = params->use_imu ? 1 : 0;
  iso_identity(v.laser_to_base);
  v.mapping = params->mapping ? 1 : 0;
  // (after lds_hash_build and filter_local_map are known)
  // (measured, scans/s aggregate, streamed / three-kernel rebuild: 4 streams 27.6k / 27.3k, 8 streams 40.4k / 41.1k, 12 streams
  //  47.3k / 51.3k — with many streams the waiting workgroups of one stream hold the CUs the next stream's solve needs)
  v.early_rebuild = (!h->lds_hash_build && !v.filter_local_map && !params->mapping && config->n_streams <= 4) ? 1 : 0;
  if (const char* e = std::getenv("LIODOM_EARLY_REBUILD")) { if (std::atoi(e) == 0) v.early_rebuild = 0; }
  if (const char* e = std::getenv("LIODOM_SAFE_MODE")) { if (std::atoi(e) != 0) enter_safe_mode(h); }
  if (const char* e = std::getenv("LIODOM_ZERO_COPY")) h->zero_copy = std::atoi(e) != 0;
  if (const char* e = std::getenv("LIODOM_FOLD_PUBLISH")) h->fold_publish = std::atoi(e) != 0;
  v.recv_cap = v.mapping ? (config->recv_capacity > 0 ? config->recv_capacity : 262144) : 0;
  v.map_cap = v.edge_cap * h->P + v.recv_cap;
  int ts = 1024;
  while (ts < 2 * (v.map_cap + (v.early_rebuild ? 8 * v.edge_cap : 0))) ts <<= 1;    // (early rebuild: cells that only the padding touches)
  if (const char* e = std::getenv("LIODOM_TABLE_SIZE")) { const int t = std::atoi(e); if (t >= 1024 && (t & (t - 1)) == 0) ts = t; }   // (experiments; a table that is too small raises LIODOM_STATUS_HASH_FULL)
  v.table_size = ts;
  v.pose_log_cap = std::max(1, config->pose_log_capacity);
  v.debug = config->debug_buffers & 1;
  // in-kernel phase timestamps (tools/gpu_debug.py clocks): they change no result.  The result-changing ablation bits
  // of earlier rounds (LIODOM_ABLATE) are gone from the product build.
  // instrumented builds only (-DLIODOM_INSTRUMENT, tools/variant_build.sh): 1: stamps, 65: + histograms (shared-counter atomics: they perturb the timing)
  if (kInstrument) { if (const char* e = getenv("LIODOM_DEBUG_CLOCKS")) { if (atoi(e) != 0) v.debug |= ((atoi(e) & (128 | 256)) ? (atoi(e) & 32) : 32) | (atoi(e) & (64 | 128)) | ((atoi(e) >> 8) << 8); } }
  v.ring_id_stride = (size_t)round_up(config->max_points + 512, 256);

  const size_t S = (size_t)h->S;
#define ALLOC(ptr, count, fill) do { rc = dev_alloc(h, &(ptr), (count), (fill)); if (rc != LIODOM_OK) return fail(rc); } while (0)
  ALLOC(v.state, S, 0);
  ALLOC(v.ring_id, S * v.ring_id_stride, 0xFF);
  v.tile_cap = std::max(1, cdiv(config->max_points, kTilePts));
  ALLOC(v.tile_hist, S * (size_t)v.tile_cap * h->H, 0);
  // + padding: region_keys_load reads unconditionally up to 16 * IPL + 10 points past the start of a ring's last region,
  // i.e. up to kExLPR * kExIPLBig + 10 points past the end of the last ring of the last stream (values never used)
  // k_ring_split_lb (lock-step batches of Velodyne-type clouds): rings at a fixed pitch of 9/8 of the nominal ring length
  h->ring_split_lb = config->n_streams >= 16 && params->lidar_type == 0 && !h->safe_mode;
  if (const char* e = std::getenv("LIODOM_RING_SPLIT_LB")) h->ring_split_lb = h->ring_split_lb && std::atoi(e) != 0;
  v.ring_pitch = round_up(cdiv((long long)config->max_points * 9, (long long)std::max(1, h->H) * 8), 8);
  if (const char* e = std::getenv("LIODOM_RING_PITCH")) v.ring_pitch = std::max(8, std::atoi(e));      // (tests: a pitch that real rings outgrow)
  v.ring_stride = h->ring_split_lb ? std::max((size_t)config->max_points, (size_t)h->H * (size_t)v.ring_pitch) : (size_t)config->max_points;
  v.lb_hpad = round_up(h->H, 64);
  if (h->ring_split_lb) {
    ALLOC(v.lb_desc, S * (size_t)v.tile_cap * v.lb_hpad, 0); ALLOC(v.lb_ticket, 1, 0); ALLOC(v.lb_ovf, S, 0);
    const size_t lds = ring_split_lb_lds_bytes(h->H);
    if (lds > 48 * 1024 && (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_split_lb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
                            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_split_fix), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)) {
      (void)hipGetLastError(); h->ring_split_lb = false;
    }
  } else { v.lb_desc = nullptr; v.lb_ticket = nullptr; v.lb_ovf = nullptr; }
  ALLOC(v.ring_pts, S * v.ring_stride + kExLPR * kExIPLBig + 64, 0);
  ALLOC(v.ring_src, S * v.ring_stride, 0);
  ALLOC(v.ring_start, S * (size_t)(h->H + 1), 0);
  ALLOC(v.ring_len, S * (size_t)h->H, 0);
  ALLOC(v.edges_pad, S * h->H * v.slots_per_ring, 0);
  ALLOC(v.edges_pad_meta, S * h->H * v.slots_per_ring, 0);
  ALLOC(v.ring_nedges, S * h->H, 0);
  {
    if (const char* e = std::getenv("LIODOM_RING_SPLIT")) h->ring_split = std::atoi(e) != 0;
    h->ring_split = h->ring_split && !h->safe_mode;      // (safe mode = no in-kernel waits at all: it overrides the switch, whatever the order of the variables)
    if (h->ring_split) {
      // k_ring_split's tiles wait for each other inside the launch, so EVERY workgroup of a launch must be resident at once.  How
      // many fit is a property of the device (CUs, LDS per CU: a tile holds ~50 KB), not a constant: occupancy query x CU count,
      // half of it left to the odometry chain's kernels that run beside the extraction.  Launches above the budget — and devices
      // or partitions where a single scan's tiles do not fit (CPX partitions, CU-masked runs) — take k_classify + k_ring_scatter.
      int cus = 0, per_cu = 0;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, config->device);
      const size_t lds = ring_scatter_lds_bytes(h->H);
      if (lds > 48 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_split), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError(); per_cu = 0;
      } else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_ring_split), kTileThreads, lds) != hipSuccess) {
        (void)hipGetLastError(); per_cu = 0;
      }
      long long budget = (long long)std::max(0, per_cu) * std::max(0, cus) / 2;
      if (budget > 256) budget = 256;                    // (measured: above ~4 HDL-64 streams per launch the waiting tiles lose to the two-kernel split anyway)
      if (const char* e = std::getenv("LIODOM_RING_SPLIT_MAX_WGS")) budget = std::max(0, std::atoi(e));      // (tests)
      h->ring_split_max_wgs = (int)budget;
      const int tiles_one = std::max(1, cdiv(config->max_points, kTilePts));
      if (tiles_one > h->ring_split_max_wgs) h->ring_split = false;      // not even one stream's scan fits: never use it
    }
    v.split_ctr = nullptr;
    v.split_hist = nullptr; v.split_pad = round_up(v.tile_cap, 8);
    if (h->ring_split) { ALLOC(v.split_ctr, 2 * S, 0); ALLOC(v.split_hist, S * (size_t)h->H * v.split_pad + 64, 0); }      // (+ one batch of 64 tiles: k_ring_split reads whole batches)
  }
  ALLOC(v.ring_npoints, S * h->H, 0);
  ALLOC(v.ring_c, S * v.ring_stride, 0);
  ALLOC(v.ring_picked, S * v.ring_stride, 0);
  ALLOC(v.edges, kEdgeBufs * S * v.edge_cap, 0);
  ALLOC(v.edges_meta, kEdgeBufs * S * v.edge_cap, 0);
  ALLOC(v.corr_a, S * 2 * v.edge_cap, 0);
  ALLOC(v.corr_b, S * 2 * v.edge_cap, 0);
  ALLOC(v.corr_idx, S * 2 * v.edge_cap, 0xFF);
  if (v.debug & 1) ALLOC(v.knn_q, S * 2 * v.edge_cap, 0); else v.knn_q = nullptr;
  ALLOC(v.win_pts, S * h->P * v.edge_cap, 0);
  ALLOC(v.win_n, S * h->P, 0);
  ALLOC(v.win_base, S * (h->P + 1), 0);
  ALLOC(v.win_slot, S * h->P, 0);
  const size_t ntab = v.early_rebuild ? 2 : 1;     // early_rebuild: two cell hashes per stream (index s + parity * S)
  ALLOC(v.cells, ntab * S * v.table_size, 0);
  ALLOC(v.pt_rank, S * v.map_cap, 0);
  ALLOC(v.cell_bits, ntab * S * (size_t)(v.table_size / 32), 0);
  v.used_cap = v.early_rebuild ? v.map_cap + 8 * v.edge_cap : v.map_cap;
  ALLOC(v.used_cells, ntab * S * (size_t)v.used_cap, 0);
  ALLOC(v.pt_cell, S * v.map_cap, 0xFF);
  if (v.recv_cap) ALLOC(v.recv_pts, S * v.recv_cap, 0);
  ALLOC(v.imu_q, S * 4, 0);
  v.ovf_base = v.early_rebuild ? v.map_cap + 8 * v.edge_cap : v.map_cap;
  v.sorted_cap = v.early_rebuild ? v.ovf_base + v.edge_cap : v.map_cap;
  {
    // incremental cell hash (k_hash_append; decided for good below, once the kNN instance is known): every cell keeps room for the
    // points of the frames that arrive before the next rebuild — twice the window + 64k places per stream
    bool want = config->n_streams >= 16 && h->lds_hash_build && !params->mapping && !params->filter_local_map;
    if (const char* e = std::getenv("LIODOM_HASH_INCR")) { if (std::atoi(e) == 0) want = false; }
    if (want && !v.early_rebuild) v.sorted_cap = 2 * v.map_cap + 65536 + (kHbPeriod - 1) * v.edge_cap;      // (+ the spill list)
  }
  ALLOC(v.sorted_pts, (v.early_rebuild ? 2 : 1) * S * (size_t)v.sorted_cap, 0);
  if (v.early_rebuild) ALLOC(v.cell_pad, 2 * S * (size_t)v.table_size, 0); else v.cell_pad = nullptr;
  v.rebuild_delta = 0.25f;
  if (const char* e = std::getenv("LIODOM_REBUILD_DELTA")) { const float d = (float)std::atof(e); if (d > 0.0f && d <= 0.45f) v.rebuild_delta = d; }
  if (v.filter_local_map) {
    ALLOC(v.vox_cells, S * v.table_size, 0);
    ALLOC(v.vox_fill, S * v.table_size, 0);
    ALLOC(v.vox_used_list, S * v.map_cap, 0);
    ALLOC(v.pt_vox, S * v.map_cap, 0xFF);
    ALLOC(v.vox_pts, S * v.map_cap, 0);
    ALLOC(v.filt_pts, S * v.map_cap, 0);
    ALLOC(v.filt_int, S * v.map_cap, 0);
  }
  ALLOC(v.pose_log, S * v.pose_log_cap * 7, 0);
  ALLOC(v.info_log, S * v.pose_log_cap, 0);
  ALLOC(h->stage_in, S * (size_t)config->max_points, 0);
  ALLOC(v.dbg_clk, 16 * 32, 0);
  if (v.debug & 32) ALLOC(v.dbg_q, 2 * (size_t)v.edge_cap * 12, 0); else v.dbg_q = nullptr;
  ALLOC(v.lm_xch, S * 2 * kLmGroupsMax * 64, 0);
  ALLOC(v.pose_xch, S * 64, 0);
  ALLOC(v.redo_sync, 64, 0);
  ALLOC(v.pipe_flags, kEdgePipeBufs + 1, 0);
  v.host_edges = nullptr; v.host_edges_meta = nullptr; v.host_edges_hdr = nullptr;
  if (S == 1) {
    // device-resident hand-off (liodom_extract_edges_device): host-mapped mirror of the dense edges of the three pipeline buffers
    const size_t ne = (size_t)kEdgePipeBufs * v.edge_cap;
    void *he = nullptr, *hm = nullptr, *hh = nullptr;
    if (hipHostMalloc(&he, sizeof(float4) * ne, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostMalloc(&hm, sizeof(int4) * ne, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostMalloc(&hh, sizeof(unsigned int) * 2 * kEdgePipeBufs, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      if (he) hipHostFree(he);
      if (hm) hipHostFree(hm);
      g_last_error = "hipHostMalloc (edge mirror) failed"; return fail(LIODOM_ERR_HIP);
    }
    h->host_edges = static_cast<float4*>(he); h->host_edges_meta = static_cast<int4*>(hm); h->host_edges_hdr = static_cast<unsigned int*>(hh);
    std::memset(hh, 0, sizeof(unsigned int) * 2 * kEdgePipeBufs);
    void *de = nullptr, *dm = nullptr, *dh = nullptr;
    if (hipHostGetDevicePointer(&de, he, 0) != hipSuccess || hipHostGetDevicePointer(&dm, hm, 0) != hipSuccess ||
        hipHostGetDevicePointer(&dh, hh, 0) != hipSuccess) { g_last_error = "hipHostGetDevicePointer failed"; return fail(LIODOM_ERR_HIP); }
    v.host_edges = static_cast<float4*>(de); v.host_edges_meta = static_cast<int4*>(dm); v.host_edges_hdr = static_cast<unsigned int*>(dh);
  }
  if (h->use_flags) {
    // Flags need kernels of the handle's streams to run side by side.  Known serialisers are caught by name above; this probe
    // catches the rest (a profiler collecting counters, a debugger): one wave on the odometry stream waits up to ~2 ms (k_probe_wait) for a flag
    // that a launch on the extraction stream sets.  If it gives up, the handle uses events from the start.
    unsigned int* probe = nullptr;
    ALLOC(probe, 4, 0);
    hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, h->stream, probe + 2, 1u);        // (first launches on both streams: code upload)
    hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, h->stream_x, probe + 3, 1u);
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipStreamSynchronize(h->stream_x) != hipSuccess) { g_last_error = "stream probe failed"; return fail(LIODOM_ERR_HIP); }
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, h->stream, probe, probe + 1);
    hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, h->stream_x, probe, 1u);
    unsigned int res = 0;
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipStreamSynchronize(h->stream_x) != hipSuccess ||
        hipMemcpy(&res, probe + 1, sizeof(res), hipMemcpyDeviceToHost) != hipSuccess) { g_last_error = "stream probe failed"; return fail(LIODOM_ERR_HIP); }
    h->streams_concurrent = res == 1u;
    if (!h->streams_concurrent) h->use_flags = false;
  }
  {
    bool gate_kernel = config->n_streams >= 16;         // lock-step batches: line gates in their own launch (k_line_gate)
    if (const char* e = std::getenv("LIODOM_GATE_KERNEL")) gate_kernel = gate_kernel && std::atoi(e) != 0;
    if (gate_kernel) ALLOC(v.knn_nn, S * (size_t)v.edge_cap * 5, 0); else v.knn_nn = nullptr;
  }
  h->knn8 = config->n_streams >= 16 && v.knn_nn != nullptr;
  if (const char* e = std::getenv("LIODOM_KNN8")) { if (std::atoi(e) == 0) h->knn8 = false; }
  h->knn8_grid = std::max(1, cdiv(cdiv(v.edge_cap, kKnn8Queries), kKnnGridDiv));
  if (const char* e = std::getenv("LIODOM_KNN8_GRID")) h->knn8_grid = std::max(1, std::min(65535, std::atoi(e)));      // (experiments)
  // incremental cell hash: lock-step batches that search with k_knn8 (it skips evicted points) on the LDS-built table, window only
  v.hash_incr = (h->knn8 && h->lds_hash_build && !params->mapping && !params->filter_local_map && h->P > kHbPeriod &&
                 v.sorted_cap >= 2 * v.map_cap + (kHbPeriod - 1) * v.edge_cap) ? 1 : 0;      // (windows of more frames than a period: the evicted frames are frames the rebuild knew)
  v.hb_spill_base = v.sorted_cap - (kHbPeriod - 1) * v.edge_cap;
  if (const char* e = std::getenv("LIODOM_HASH_INCR")) { if (std::atoi(e) == 0) v.hash_incr = 0; }
  if (v.hash_incr) ALLOC(v.cell_cap, S * (size_t)v.table_size, 0); else v.cell_cap = nullptr;
  v.hb_slack_min = kHbSlackMin; v.hb_new_room = kHbNewRoom;
  if (const char* e = std::getenv("LIODOM_HB_SLACK")) v.hb_slack_min = std::max(0, std::atoi(e));            // (tests: cells that run out of room)
  if (const char* e = std::getenv("LIODOM_HB_NEW_ROOM")) v.hb_new_room = std::max(1, std::atoi(e));
  if (h->knn8) { ALLOC(v.knn8_cnt, S, 0); ALLOC(v.knn8_list, S * (size_t)v.edge_cap, 0); } else { v.knn8_cnt = nullptr; v.knn8_list = nullptr; }
  v.knn_queries = config->n_streams >= 16 ? 4 : 8;          // must match the k_knn instance launch_odometry picks (k_knn8 leaves k_line_gate the same layout)
  v.knn_partials = config->n_streams >= 16 ? 0 : 1;         // measured: +37 % on the VALU-bound 256-stream kNN pass, -2 us per solve on one stream
  v.knn_blocks = round_up(cdiv(v.edge_cap, v.knn_queries), 4);
  v.knn_grid = std::max(1, cdiv(v.knn_blocks, kKnnGridDiv));   // sized for the usual edge count (~1/3 of the capacity): a workgroup takes a second block if there are more
  {
    // LIODOM_KNN_SAVE: 2 (default) second pass re-ranks the first pass's kept candidates and prunes with its fifth distance;
    // 1: pruning bound only; 0: the second pass searches like the first (all three give the same results)
    const int save = std::getenv("LIODOM_KNN_SAVE") ? std::atoi(std::getenv("LIODOM_KNN_SAVE")) : 2;
    if (save >= 1) ALLOC(v.knn_save_q, S * (size_t)v.edge_cap, 0); else v.knn_save_q = nullptr;
    if (save >= 2) { ALLOC(v.knn_save_pos, S * (size_t)v.edge_cap * kKnnGroup, 0xFF); ALLOC(v.knn_save_g, S * (size_t)v.edge_cap, 0); }
    else { v.knn_save_pos = nullptr; v.knn_save_g = nullptr; }
  }
  if (const char* e = std::getenv("LIODOM_KNN_EXACT_ONLY")) v.knn_exact_only = std::atoi(e) != 0 ? 1 : 0;
  ALLOC(v.knn_part, S * 2 * (size_t)v.knn_blocks * 32, 0);
  ALLOC(v.ov_flags, S, 0);
  ALLOC(v.pose_xch0, S * (size_t)kOvReplicas * 512, 0);
  ALLOC(v.knn_done, S * (size_t)v.knn_grid, 0);
  ALLOC(v.knn_done0, S + 64, 0);
  if (v.early_rebuild) ALLOC(v.pred_xch, S * (size_t)kOvReplicas * 512, 0); else v.pred_xch = nullptr;
  ALLOC(v.edge_cnt, (size_t)kEdgeBufs * 32, 0);
  ALLOC(v.pub_counter, (size_t)kEdgeBufs, 0);
  if (v.early_rebuild) ALLOC(v.edges_keep, S * (size_t)v.edge_cap, 0); else v.edges_keep = nullptr;
  // (the two passes' validity bytes never share a 128-byte line: the overlapped second pass writes its half while the finalising
  //  solve's launch — which must not read it before ov_wait_knn_done — may hold the first pass's half in its caches)
  v.mask_stride = round_up(v.knn_blocks, 128);
  ALLOC(v.corr_mask, S * 2 * (size_t)v.mask_stride, 0);
  {
    void* hp = nullptr;
    if (hipHostMalloc(&hp, sizeof(HostOut) * 2 * S, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { g_last_error = "hipHostMalloc failed"; return fail(LIODOM_ERR_HIP); }
    h->host_out = static_cast<HostOut*>(hp);
    std::memset(hp, 0, sizeof(HostOut) * 2 * S);
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { g_last_error = "hipHostGetDevicePointer failed"; return fail(LIODOM_ERR_HIP); }
    v.host_out = static_cast<HostOut*>(dp);
    h->scans_enqueued.assign(S, 0);
    h->mappers.assign(S, nullptr); h->mapper_cells_xy.assign(S, 2); h->mapper_cells_z.assign(S, 1);
  }
  if (ring_scatter_lds_bytes(h->H) > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ring_scatter_lds_bytes(h->H)) != hipSuccess) {
      g_last_error = "hipFuncSetAttribute(max dynamic LDS) failed"; return fail(LIODOM_ERR_HIP);
    }
  }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hash_build), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)hash_build_lds_bytes()) != hipSuccess) {
    g_last_error = "hipFuncSetAttribute(max dynamic LDS) failed"; return fail(LIODOM_ERR_HIP);
  }
  v.lm_lds_reduce = lm_lds_reduce_fits(v.edge_cap) ? 1 : 0;
  if (lm_lds_bytes(v.edge_cap) + 8192 > 160 * 1024) { g_last_error = "liodom_create: edge capacity too large for the solve's LDS tile"; return fail(LIODOM_ERR_INVALID_ARG); }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lm_solve<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm_lds_bytes(h->v.edge_cap)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lm_solve<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm_lds_bytes(h->v.edge_cap)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lm_solve<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm_lds_bytes(h->v.edge_cap)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lm_solve<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm_lds_bytes(h->v.edge_cap)) != hipSuccess) {
    g_last_error = "hipFuncSetAttribute(max dynamic LDS) failed"; return fail(LIODOM_ERR_HIP);
  }
  if (h->ring_lds_bytes > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_extract<256, kExIPL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->ring_lds_bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_extract<256, kExIPLBig>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->ring_lds_bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_extract<1024, kExIPL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->ring_lds_bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_extract<1024, kExIPLBig>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->ring_lds_bytes) != hipSuccess) {
      g_last_error = "hipFuncSetAttribute(max dynamic LDS) failed"; return fail(LIODOM_ERR_HIP);
    }
  }
  {
    // Overlapped second kNN pass: its workgroups wait inside the kernel for the first solve, so they must leave most of the GPU
    // to the launches they wait for (and to the next scan's extraction): one stream, at most a third of the wave slots
    // (HDL-64: 352 workgroups x 4 waves = 1 408 of 6 144).  LIODOM_KNN_OVERLAP=0 keeps the pass on the odometry stream.
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, config->device);
    // (measured with half of the slots allowed: Ouster-128, 704 workgroups = 2 816 waves, loses — 9.0k -> 8.4k scans/s)
    h->ov_ok = v.early_rebuild && S == 1 && v.knn_partials && (long long)v.knn_grid * 4 * 3 <= (long long)cus * 24;
    if (const char* e = std::getenv("LIODOM_KNN_OVERLAP")) { if (std::atoi(e) == 0) h->ov_ok = false; if (std::atoi(e) == 2) h->ov_ok = v.early_rebuild && S == 1 && v.knn_partials; }      // (2: also where the pass takes more than a third of the wave slots)
    // (the stream exists only on handles that use it: HIP multiplexes its streams onto a few hardware queues, and one more
    //  stream made the host-fed replay's copy stream share a queue — 11.3k -> 7.5k scans/s on every workload)
    // chain mode (kernels_sync.h): one-stream handles with the streamed rebuild whose passes leave at least half of the wave slots
    // free; the IMU override rewrites the prediction between two scans on the odometry stream (k_imu_override), which the first
    // pass on stream_k would not be ordered behind
    h->chain_ok = v.early_rebuild && S == 1 && v.knn_partials && !v.use_imu && (long long)v.knn_grid * 4 * 2 <= (long long)cus * 24;
    if (const char* e = std::getenv("LIODOM_KNN_OVERLAP")) { if (std::atoi(e) == 0) h->chain_ok = false; }
    if (const char* e = std::getenv("LIODOM_CHAIN")) { if (std::atoi(e) == 0) h->chain_ok = false; }
    // speculative hand-over of the solves' results (kernels_sync.h): LIODOM_SPECULATE=0 off, 1 by the model's predicted cost change
    // (default), 2 (tests): as early as possible, i.e. practically always wrong — every receiver is then repeated from the confirmed
    // result; 4 / 5 (debugging): only the first / only the finalising solve's hand-over.  LIODOM_SPEC_THETA: the predictor's threshold
    // (fraction of the function tolerance, default 0.8)
    v.speculate = (h->ov_ok || h->chain_ok) ? 1 : 0;
    if (const char* e = std::getenv("LIODOM_SPECULATE")) { if (v.speculate) v.speculate = std::max(0, std::min(7, std::atoi(e))); }
    v.spec_backoff = 16;
    if (const char* e = std::getenv("LIODOM_SPEC_BACKOFF")) v.spec_backoff = std::max(0, std::atoi(e));
    v.spec_theta = 0.8;
    if (const char* e = std::getenv("LIODOM_SPEC_THETA")) v.spec_theta = std::atof(e);
    if ((h->ov_ok || h->chain_ok) && make_stream(&h->stream_k, prio_greatest) != hipSuccess) { g_last_error = "hipStreamCreate failed"; return fail(LIODOM_ERR_HIP); }
    if (h->chain_ok && hipEventCreateWithFlags(&h->ev_ch, hipEventDisableTiming) != hipSuccess) { g_last_error = "hipEventCreate failed"; return fail(LIODOM_ERR_HIP); }
  }
  {
    // wall-clock bound of every in-kernel wait (g_wait_ticks, 100 MHz ticks): LIODOM_WAIT_MS, default 50 ms
    double ms = 50.0;
    if (const char* e = std::getenv("LIODOM_WAIT_MS")) { const double x = std::atof(e); if (x >= 1.0 && x <= 10000.0) ms = x; }
    const unsigned long long ticks = (unsigned long long)(ms * 1.0e5);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wait_ticks), &ticks, sizeof(ticks)) != hipSuccess) { g_last_error = "hipMemcpyToSymbol(g_wait_ticks) failed"; return fail(LIODOM_ERR_HIP); }
  }
  ALLOC(h->d_view, 1, 0);
  if (hipMemcpy(h->d_view, &h->v, sizeof(DevView), hipMemcpyHostToDevice) != hipSuccess) { g_last_error = "DevView upload failed"; return fail(LIODOM_ERR_HIP); }
  rc = reset_state(h);
  if (rc != LIODOM_OK) return fail(rc);
  g_live_handles.fetch_add(1);
  h->counted_live = true;
  *out = h;
  return LIODOM_OK;
#undef ALLOC
}

void liodom_destroy(liodom_handle_t* h) {
  if (!h) return;
  if (h->counted_live) g_live_handles.fetch_sub(1);
  if (h->stream_x) hipStreamSynchronize(h->stream_x);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->stream_k) { hipStreamSynchronize(h->stream_k); hipStreamDestroy(h->stream_k); }
  if (h->ev_ov) hipEventDestroy(h->ev_ov);
  if (h->ev_ch) hipEventDestroy(h->ev_ch);
  for (liodom_map* mp : h->mappers) {          // attached maps outlive the handle: give them a stream of their own again
    if (!mp) continue;
    mp->stream = nullptr; mp->own_stream = false;
    if (hipStreamCreateWithFlags(&mp->stream, hipStreamNonBlocking) == hipSuccess) mp->own_stream = true;
  }
  for (void* p : h->allocs) hipFree(p);
  if (h->resident) hipFree(h->resident);
  if (h->host_out) hipHostFree(h->host_out);
  if (h->host_edges) hipHostFree(h->host_edges);
  if (h->host_edges_meta) hipHostFree(h->host_edges_meta);
  if (h->host_edges_hdr) hipHostFree(h->host_edges_hdr);
  if (h->pin_ring) hipHostFree(h->pin_ring);
  for (int b = 0; b < kEdgePipeBufs; b++) { if (h->ev_pin[b]) hipEventDestroy(h->ev_pin[b]); if (h->ev_sdone[b]) hipEventDestroy(h->ev_sdone[b]); if (h->ev_cp[b]) hipEventDestroy(h->ev_cp[b]); }
  for (auto& e : h->ev_pool) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  if (h->pose_event) hipEventDestroy(h->pose_event);
  for (int b = 0; b < kEdgePipeBufs; b++) { if (h->ev_edges[b]) hipEventDestroy(h->ev_edges[b]); if (h->ev_free[b]) hipEventDestroy(h->ev_free[b]); }
  if (h->stream_c && !h->stream_c_shared) { hipStreamSynchronize(h->stream_c); hipStreamDestroy(h->stream_c); }
  for (int b = 0; b < 3; b++) { if (h->ev_up[b]) hipEventDestroy(h->ev_up[b]); if (h->ev_xdone[b]) hipEventDestroy(h->ev_xdone[b]); }
  if (h->stream_x) hipStreamDestroy(h->stream_x);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int liodom_reset(liodom_handle_t* h) {
  int rc = enter(h);
  if (rc) return rc;
  SideLocks lk(h, true, true);
  return reset_state(h);
}

static int copy_edges_out(liodom_handle_t* h, int stream, int eb, hipStream_t q, float* edges_xyzi, int32_t* edge_ring,
                          int32_t* edge_idx, int32_t* edge_src, int cap, int* n_edges) {
  int E = 0;
  HIP_TRY(hipMemcpyAsync(&E, &h->v.state[stream].n_edges_buf[eb], sizeof(int), hipMemcpyDeviceToHost, q));
  HIP_TRY(hipStreamSynchronize(q));
  if (n_edges) *n_edges = E;
  if (E > cap) { g_last_error = "edge buffer too small"; return LIODOM_ERR_CAPACITY; }
  if (E == 0) return LIODOM_OK;
  if (edges_xyzi)
    HIP_TRY(hipMemcpyAsync(edges_xyzi, h->v.edges + ((size_t)eb * h->S + stream) * h->v.edge_cap, sizeof(float4) * (size_t)E, hipMemcpyDeviceToHost, q));
  std::vector<int4> meta;
  if (edge_ring || edge_idx || edge_src) {
    meta.resize((size_t)E);
    HIP_TRY(hipMemcpyAsync(meta.data(), h->v.edges_meta + ((size_t)eb * h->S + stream) * h->v.edge_cap, sizeof(int4) * (size_t)E, hipMemcpyDeviceToHost, q));
  }
  HIP_TRY(hipStreamSynchronize(q));
  for (int i = 0; i < E && !meta.empty(); i++) {
    if (edge_ring) edge_ring[i] = meta[i].x;
    if (edge_idx) edge_idx[i] = meta[i].y;
    if (edge_src) edge_src[i] = meta[i].z;
  }
  return LIODOM_OK;
}

int liodom_extract_edges(liodom_handle_t* h, int stream, const float* xyzi, int64_t n, int height,
                         int width, float* edges_xyzi, int32_t* edge_ring, int32_t* edge_idx,
                         int32_t* edge_src, int cap, int* n_edges) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if ((rc = check_usable(h))) return rc;
  if (n < 0 || n > h->v.max_points || (n > 0 && !xyzi)) { g_last_error = "bad point count"; return LIODOM_ERR_CAPACITY; }
  // Extraction side only (mx_x, stream_x, edge buffer kEdgeBufX): safe beside a concurrent
  // liodom_odometry_step of another thread.  An extraction the pipelined replay issued ahead is on the
  // same HIP stream, so the shared ring-split scratch is used in stream order.
  SideLocks lk(h, false, true);
  hipStream_t q = extract_queue(h);
  float4* in = h->stage_in + (size_t)stream * h->v.max_points;
  if (n) HIP_TRY(hipMemcpyAsync(in, xyzi, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, q));
  rc = launch_extract(h, q, kEdgeBufX, stream, 1, in, 0, (int)n, height, width);
  if (rc) return rc;
  return copy_edges_out(h, stream, kEdgeBufX, q, edges_xyzi, edge_ring, edge_idx, edge_src, cap, n_edges);
}

int liodom_get_edges(liodom_handle_t* h, int stream, float* edges_xyzi, int32_t* edge_ring,
                     int32_t* edge_idx, int32_t* edge_src, int cap, int* n_edges) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  SideLocks lk(h, true, false);
  return copy_edges_out(h, stream, h->last_eb, h->stream, edges_xyzi, edge_ring, edge_idx, edge_src, cap, n_edges);
}

static int wait_pose(liodom_handle_t* h, int s0, int count, double* pose_out, liodom_step_info_t* info, int lag = 0) {
  // zero-copy: k_lm_solve's finalize writes pose + diagnostics into host-mapped memory and
  // releases HostOut.seq; spin on it (an event / memcpy round trip costs ~15 us on this stack)
  bool timed_out = false;
  for (int i = 0; i < count; i++) {
    const int expect = h->scans_enqueued[s0 + i] - lag;      // lag 1: the scan before the one enqueued last
    volatile HostOut* ho = h->host_out + (size_t)(s0 + i) * 2 + ((expect - 1) & 1);     // (scan k = expect - 1 publishes into record k & 1)
    unsigned long long spins = 0;
    while (__atomic_load_n(&ho->seq, __ATOMIC_ACQUIRE) != expect) {
      if ((++spins & 0xFFFFull) == 0) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q != hipErrorNotReady && q != hipSuccess) { g_last_error = std::string("stream error while waiting: ") + hipGetErrorString(q); return LIODOM_ERR_HIP; }
        if (q == hipSuccess && __atomic_load_n(&ho->seq, __ATOMIC_ACQUIRE) != expect) { g_last_error = "stream drained without publishing the scan result"; return LIODOM_ERR_HIP; }
      }
    }
    const HostOut* r = h->host_out + (size_t)(s0 + i) * 2 + ((expect - 1) & 1);
    if (pose_out) std::memcpy(pose_out + 7 * i, r->pose, sizeof(double) * 7);
    if (info) info[i] = r->info;
    if (r->info.status & (LIODOM_STATUS_PIPE_TIMEOUT | LIODOM_STATUS_LM_SYNC_TIMEOUT)) timed_out = true;
    // (speculative hand-over: the record carries the scan's verdict — a confirmed scan needs no repair, chain_flush)
    if (s0 + i == 0) { h->verdict_scan = expect; h->verdict_confirmed = r->pad == 1; }
  }
  if (timed_out) {
    // A kernel gave up waiting for another HIP stream of the handle (pipe_wait / ov_wait_*): its workgroups skipped the scan, the
    // published pose is the prediction.  Fail loudly; the switch to event-based dependencies needs both sides of the handle
    // (this caller may hold the odometry side only while another thread extracts): it is applied by liodom_reset(), and every
    // entry point that enqueues work refuses until then (check_usable).
    h->fallback_pending.store(true);
    g_last_error = "a kernel timed out waiting for another kernel of the handle (LIODOM_STATUS_PIPE_TIMEOUT / LM_SYNC_TIMEOUT): kernels are "
                   "serialised across streams (profiler with --pmc, AMD_SERIALIZE_KERNEL, debugger) or the GPU is saturated by another "
                   "process; the scan's result is invalid; call liodom_reset(): the handle then continues in safe mode (no in-kernel waits)";
    return LIODOM_ERR_HIP;
  }
  return LIODOM_OK;
}

int liodom_odometry_step(liodom_handle_t* h, int stream, const float* edges_xyzi, int n_edges,
                         double stamp, double* pose_out, liodom_step_info_t* info) {
  (void)stamp;
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if ((rc = check_usable(h))) return rc;
  if (n_edges < 0 || n_edges > h->v.edge_cap || (n_edges > 0 && !edges_xyzi)) { g_last_error = "edge count exceeds capacity"; return LIODOM_ERR_CAPACITY; }
  // Odometry side only (mx_o, h->stream, edge buffer 0): safe beside a concurrent liodom_extract_edges.
  SideLocks lk(h, true, false);
  if ((rc = tickets_idle(h))) return rc;
  rc = drain_pipeline(h);
  if (rc) return rc;
  if (n_edges)
    HIP_TRY(hipMemcpyAsync(h->v.edges + (size_t)stream * h->v.edge_cap, edges_xyzi, sizeof(float4) * (size_t)n_edges, hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, KID_OTHER);
    hipLaunchKernelGGL(k_set_edges, dim3(1), dim3(64), 0, h->stream, h->v, stream, n_edges, 0);
  }
  rc = launch_odometry(h, 0, stream, 1);
  if (rc) return rc;
  return wait_pose(h, stream, 1, pose_out, info);
}

// ---- device-resident hand-off between the two sides (the reference's feature queue without the cloud leaving HBM) ----
static int ensure_pin_ring(liodom_handle* h) {
  if (h->pin_ring) return LIODOM_OK;
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, sizeof(float4) * (size_t)kEdgePipeBufs * (size_t)h->v.max_points, hipHostMallocDefault));
  h->pin_ring = static_cast<float4*>(p);
  for (int b = 0; b < kEdgePipeBufs; b++) HIP_TRY(hipEventCreateWithFlags(&h->ev_pin[b], hipEventDisableTiming));
  // LIODOM_COPY_STREAM=1: uploads on a copy stream of their own, into a ring of device staging slots, so that the DMA of scan k+1
  // (1.84 MB, ~37 us) runs beside the extraction kernels of scan k instead of in front of its own on the extraction stream.
  // Measured (MI355X, HDL-64 shape, two C++ threads): 10.3k scans/s WITHOUT it, 7.5k with it, with GPU_MAX_HW_QUEUES=8 as well —
  // the same loss the host-fed replay saw with a fourth stream per handle (the two cross-stream event edges per scan cost more
  // than the overlap gains).  Off by default; kept for runtimes where a fourth stream is cheap.
  // LIODOM_COPY_STREAM=2: the uploads take the stream of the overlapped second kNN pass (as the host-fed replay does), which is
  // then not overlapped on this handle any more: three streams per handle, the upload beside the previous extraction.
  int want = 0;
  if (const char* e = std::getenv("LIODOM_COPY_STREAM")) want = std::atoi(e);
  if (want != 0 && !h->profiling) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(float4) * (size_t)kEdgePipeBufs * (size_t)h->v.max_points));
    h->stage_ring = static_cast<float4*>(d);
    h->allocs.push_back(d);
    if (!h->stream_c) {
      if (want == 2 && h->stream_k) {
        // (the odometry side may be enqueueing an overlapped pass right now: from here on it does not — ov_off_for_copies is read
        //  by enqueue_odometry — and what is already in that stream simply runs ahead of the first copy)
        h->ov_off_for_copies.store(true);
        h->stream_c = h->stream_k; h->stream_c_shared = true;
      } else {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream_c, hipStreamNonBlocking));
      }
    }
    for (int b = 0; b < kEdgePipeBufs; b++) {
      HIP_TRY(hipEventCreateWithFlags(&h->ev_sdone[b], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_cp[b], hipEventDisableTiming));
    }
    h->tk_copy_stream = true;
  }
  return LIODOM_OK;
}

int liodom_scan_buffer(liodom_handle_t* h, int stream, float** xyzi, int64_t* capacity_points) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (!xyzi) return LIODOM_ERR_INVALID_ARG;
  if (h->S != 1) { g_last_error = "liodom_scan_buffer: one-stream handles only"; return LIODOM_ERR_UNSUPPORTED; }
  SideLocks lk(h, false, true);
  if ((rc = ensure_pin_ring(h))) return rc;
  const int r = h->pin_next;
  // the upload that last read this slot (three scans ago) must have left it
  if (h->ev_pin_valid[r]) { HIP_TRY(hipEventSynchronize(h->ev_pin[r])); h->ev_pin_valid[r] = false; }
  *xyzi = reinterpret_cast<float*>(h->pin_ring + (size_t)r * h->v.max_points);
  if (capacity_points) *capacity_points = h->v.max_points;
  return LIODOM_OK;
}

int liodom_extract_edges_device(liodom_handle_t* h, int stream, const float* xyzi, int64_t n, int height, int width,
                                liodom_edge_ticket_t* ticket) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if ((rc = check_usable(h))) return rc;
  if (!ticket) return LIODOM_ERR_INVALID_ARG;
  if (h->S != 1) { g_last_error = "liodom_extract_edges_device: one-stream handles only (lock-step handles advance all streams together: liodom_process_resident)"; return LIODOM_ERR_UNSUPPORTED; }
  if (n < 0 || n > h->v.max_points || (n > 0 && !xyzi)) { g_last_error = "bad point count"; return LIODOM_ERR_CAPACITY; }
  SideLocks lk(h, false, true);                      // extraction side only: safe beside a concurrent liodom_odometry_step_device
  if (h->pf_slot >= 0) { g_last_error = "liodom_extract_edges_device: a pipelined replay of this handle has an extraction issued ahead: call liodom_sync() first"; return LIODOM_ERR_NEEDS_SYNC; }
  // The pipelined replay uses the same three edge buffers (h->parity) and may have odometries in flight that nobody has collected
  // (liodom_process_resident_pipelined without read-back): this extraction would rewrite a buffer under them (wait_odo = 0 below
  // relies on the ticket discipline: a slot is refilled only after its pose has been collected).
  if (h->replay_live.load()) { g_last_error = "liodom_extract_edges_device: scans of a pipelined replay (liodom_process_resident / liodom_replay_*) are still in the edge buffers: call liodom_sync() first"; return LIODOM_ERR_NEEDS_SYNC; }      // (not BUSY: a caller that keeps the cloud and retries would spin forever)
  const int eb = h->x_next;
  if (h->tk_seq[eb].load() != 0u) {
    g_last_error = "liodom_extract_edges_device: all hand-off slots hold edge clouds no liodom_odometry_step_device has taken yet";
    return LIODOM_ERR_BUSY;
  }
  hipStream_t q = extract_queue(h);
  float4* in = h->stage_in + (size_t)stream * h->v.max_points;
  const float4* host_dev = nullptr;                  // device-visible address of the page-locked scan (zero-copy)
  int pin_slot_used = -1;
  if (n) {
    // The scan's upload is asynchronous when it starts from page-locked memory: a slot of the handle's own ring
    // (liodom_scan_buffer), or a buffer the caller registered (liodom_pin_host_buffer) — which must stay untouched until
    // liodom_wait_edges / liodom_odometry_step_device of this ticket has returned.  A pageable scan is copied through the ring.
    if ((rc = ensure_pin_ring(h))) return rc;
    const char* lo = reinterpret_cast<const char*>(h->pin_ring);
    const char* hi = lo + sizeof(float4) * (size_t)kEdgePipeBufs * (size_t)h->v.max_points;
    const char* src = reinterpret_cast<const char*>(xyzi);
    bool own = src >= lo && src < hi, pinned = own;
    if (!own) {
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, xyzi) == hipSuccess) pinned = attr.type == hipMemoryTypeHost;
      else (void)hipGetLastError();                  // (unregistered pageable memory: not an error)
    }
    const int r = own ? (int)((src - lo) / (sizeof(float4) * (size_t)h->v.max_points)) : h->pin_next;
    if (!pinned) {
      if (h->ev_pin_valid[r]) { HIP_TRY(hipEventSynchronize(h->ev_pin[r])); h->ev_pin_valid[r] = false; }
      std::memcpy(h->pin_ring + (size_t)r * h->v.max_points, xyzi, sizeof(float4) * (size_t)n);
      xyzi = reinterpret_cast<const float*>(h->pin_ring + (size_t)r * h->v.max_points);
    }
    // Zero-copy (LIODOM_ZERO_COPY=1; default is hipMemcpyAsync): the extraction's first kernel reads the page-locked scan over
    // PCIe itself — no copy call, no second stream, no event.  Measured slower than the DMA upload (see zero_copy).
    if (h->zero_copy && (reinterpret_cast<uintptr_t>(xyzi) & 15u) == 0) {
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, const_cast<float*>(xyzi), 0) == hipSuccess && dp) host_dev = static_cast<const float4*>(dp);
      else (void)hipGetLastError();
    }
    hipStream_t qc = q;
    int sr = -1;
    if (host_dev) {
      if (own || !pinned) { pin_slot_used = r; h->pin_next = (r + 1) % kEdgePipeBufs; }
    } else
    if (h->tk_copy_stream && !h->profiling) {
      // device staging slot sr: free once the extraction that last read it has run (ev_sdone, recorded on the extraction stream)
      sr = h->stage_next;
      h->stage_next = (sr + 1) % kEdgePipeBufs;
      qc = h->stream_c;
      in = h->stage_ring + (size_t)sr * h->v.max_points;
      if (h->ev_sdone_valid[sr]) HIP_TRY(hipStreamWaitEvent(qc, h->ev_sdone[sr], 0));
    }
    if (!host_dev) HIP_TRY(hipMemcpyAsync(in, xyzi, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, qc));
    if (!host_dev && (own || !pinned)) {             // the page-locked ring slot may be refilled once this upload has left it
      HIP_TRY(hipEventRecord(h->ev_pin[r], qc));
      h->ev_pin_valid[r] = true;
      h->pin_next = (r + 1) % kEdgePipeBufs;
    }
    if (sr >= 0) {                                   // the extraction starts when the upload into its staging slot has completed
      HIP_TRY(hipEventRecord(h->ev_cp[sr], qc));
      HIP_TRY(hipStreamWaitEvent(q, h->ev_cp[sr], 0));
    }
  }
  const bool staged_on_ring = n > 0 && !host_dev && h->tk_copy_stream && !h->profiling;
  unsigned int seq = ++h->ext_seq;
  if (seq == 0u) seq = ++h->ext_seq;                 // (0 means "nothing to wait for")
  unsigned int* host_seq = h->v.host_edges_hdr ? h->v.host_edges_hdr + eb : nullptr;
  // (the slot is free: the odometry that last read buffer eb has been collected, i.e. has completed — no wait on the device,
  //  which would depend on when the other thread submits its next scan)
  unsigned int* const dev_flag = h->use_flags ? h->v.pipe_flags + eb : (unsigned int*)nullptr;
  rc = launch_extract(h, q, eb, stream, 1, in, 0, (int)n, height, width, 0u, 1, host_dev, 0,
                      h->fold_publish ? dev_flag : nullptr, h->fold_publish ? host_seq : nullptr, h->fold_publish ? seq : 0u);
  if (rc) return rc;
  if (!h->fold_publish) hipLaunchKernelGGL(k_publish_edges, dim3(1), dim3(1), 0, q, dev_flag, host_seq, seq);
  if (!h->use_flags) HIP_TRY(hipEventRecord(h->ev_edges[eb], q));
  HIP_TRY(hipGetLastError());
  if (pin_slot_used >= 0) {                          // (zero-copy) the ring slot may be refilled once the extraction's first kernel has read it
    HIP_TRY(hipEventRecord(h->ev_pin[pin_slot_used], q));
    h->ev_pin_valid[pin_slot_used] = true;
  }
  if (staged_on_ring) {
    const int sr = (h->stage_next + kEdgePipeBufs - 1) % kEdgePipeBufs;
    HIP_TRY(hipEventRecord(h->ev_sdone[sr], q));
    h->ev_sdone_valid[sr] = true;
  }
  h->eb_seq[eb] = seq;
  h->pipe_active.store(true);
  h->tk_seq[eb].store(seq);
  h->x_next = (eb + 1) % kEdgePipeBufs;
  ticket->seq = seq; ticket->slot = eb; ticket->stream = stream; ticket->reserved = 0;
  return LIODOM_OK;
}

int liodom_wait_edges(liodom_handle_t* h, const liodom_edge_ticket_t* ticket, float* edges_xyzi, int32_t* edge_ring,
                      int32_t* edge_idx, int32_t* edge_src, int cap, int* n_edges) {
  if (!h || !ticket) return LIODOM_ERR_INVALID_ARG;
  int rc = check_stream(h, ticket->stream);
  if (rc) return rc;
  const int eb = ticket->slot;
  if (eb < 0 || eb >= kEdgePipeBufs || !h->host_edges_hdr) { g_last_error = "liodom_wait_edges: bad ticket"; return LIODOM_ERR_INVALID_ARG; }
  // No lock: the slot's mirror is rewritten only by an extraction issued after the ticket has been consumed
  // (liodom_odometry_step_device), which the caller orders behind this call.
  volatile unsigned int* hs = h->host_edges_hdr + eb;
  unsigned long long spins = 0;
  while ((int)(__atomic_load_n(hs, __ATOMIC_ACQUIRE) - ticket->seq) < 0) {
    if ((++spins & 0xFFFFull) == 0) {
      const hipError_t q = hipStreamQuery(extract_queue(h));
      if (q != hipErrorNotReady && q != hipSuccess) { g_last_error = std::string("stream error while waiting for edges: ") + hipGetErrorString(q); return LIODOM_ERR_HIP; }
      if (q == hipSuccess && (int)(__atomic_load_n(hs, __ATOMIC_ACQUIRE) - ticket->seq) < 0) { g_last_error = "extraction stream drained without publishing the ticket's edges"; return LIODOM_ERR_HIP; }
    }
  }
  if (__atomic_load_n(hs, __ATOMIC_ACQUIRE) != ticket->seq) { g_last_error = "liodom_wait_edges: the ticket's slot has been reused (stale ticket)"; return LIODOM_ERR_INVALID_ARG; }
  const int E = (int)h->host_edges_hdr[kEdgePipeBufs + eb];
  if (n_edges) *n_edges = E;
  if (E > cap && (edges_xyzi || edge_ring || edge_idx || edge_src)) { g_last_error = "edge buffer too small"; return LIODOM_ERR_CAPACITY; }
  const float4* he = h->host_edges + (size_t)eb * h->v.edge_cap;
  const int4* hm = h->host_edges_meta + (size_t)eb * h->v.edge_cap;
  if (edges_xyzi && E) std::memcpy(edges_xyzi, he, sizeof(float4) * (size_t)E);
  if (edge_ring || edge_idx || edge_src) {
    for (int i = 0; i < E; i++) {
      if (edge_ring) edge_ring[i] = hm[i].x;
      if (edge_idx) edge_idx[i] = hm[i].y;
      if (edge_src) edge_src[i] = hm[i].z;
    }
  }
  return LIODOM_OK;
}

int liodom_odometry_submit_device(liodom_handle_t* h, const liodom_edge_ticket_t* ticket, double stamp) {
  (void)stamp;
  if (!h || !ticket) return LIODOM_ERR_INVALID_ARG;
  int rc = check_stream(h, ticket->stream);
  if (rc) return rc;
  if ((rc = check_usable(h))) return rc;
  const int eb = ticket->slot;
  if (h->S != 1 || eb < 0 || eb >= kEdgePipeBufs) { g_last_error = "liodom_odometry_submit_device: bad ticket"; return LIODOM_ERR_INVALID_ARG; }
  SideLocks lk(h, true, false);                      // odometry side only: safe beside a concurrent liodom_extract_edges_device
  if (ticket->seq == 0u || h->tk_seq[eb].load() != ticket->seq) {
    g_last_error = "liodom_odometry_submit_device: stale or unknown ticket (already consumed, or voided by liodom_reset)";
    return LIODOM_ERR_INVALID_ARG;
  }
  for (int i = 0; i < h->odo_pending; i++) if (h->odo_fifo[i] == eb) { g_last_error = "liodom_odometry_submit_device: ticket already submitted"; return LIODOM_ERR_INVALID_ARG; }
  if (h->odo_pending >= 2) { g_last_error = "liodom_odometry_submit_device: two scans are in flight: collect a pose first (liodom_odometry_collect)"; return LIODOM_ERR_BUSY; }
  rc = enqueue_pipeline_odometry(h, eb, ticket->seq);
  if (rc) return rc;
  h->pipe_active.store(true);
  h->odo_fifo[h->odo_pending++] = eb;
  return LIODOM_OK;
}

int liodom_odometry_collect(liodom_handle_t* h, int stream, double* pose_out, liodom_step_info_t* info) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  SideLocks lk(h, true, false);
  if (h->odo_pending <= 0) { g_last_error = "liodom_odometry_collect: no scan has been submitted"; return LIODOM_ERR_INVALID_ARG; }
  rc = wait_pose(h, stream, 1, pose_out, info, h->odo_pending - 1);      // the oldest of the (at most two) scans in flight
  const int eb = h->odo_fifo[0];
  h->odo_fifo[0] = h->odo_fifo[1];
  h->odo_pending--;
  h->tk_seq[eb].store(0u);                           // its odometry has completed: the slot may be refilled
  return rc;
}

int liodom_odometry_step_device(liodom_handle_t* h, const liodom_edge_ticket_t* ticket, double stamp, double* pose_out,
                                liodom_step_info_t* info) {
  if (!h || !ticket) return LIODOM_ERR_INVALID_ARG;
  if (h->odo_pending != 0) { g_last_error = "liodom_odometry_step_device: a submitted scan has not been collected (liodom_odometry_collect)"; return LIODOM_ERR_BUSY; }
  const int rc = liodom_odometry_submit_device(h, ticket, stamp);
  if (rc) return rc;
  return liodom_odometry_collect(h, ticket->stream, pose_out, info);
}

int liodom_process_scan(liodom_handle_t* h, int stream, const float* xyzi, int64_t n, int height,
                        int width, double stamp, double* pose_out, liodom_step_info_t* info) {
  (void)stamp;
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if ((rc = check_usable(h))) return rc;
  if (n < 0 || n > h->v.max_points || (n > 0 && !xyzi)) { g_last_error = "bad point count"; return LIODOM_ERR_CAPACITY; }
  SideLocks lk(h, true, true);        // uses the extraction scratch on the odometry stream
  if ((rc = tickets_idle(h))) return rc;
  rc = drain_pipeline(h);             // nothing issued ahead on the extraction stream may still use that scratch
  if (rc) return rc;
  float4* in = h->stage_in + (size_t)stream * h->v.max_points;
  if (n) HIP_TRY(hipMemcpyAsync(in, xyzi, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  rc = launch_extract(h, h->stream, 0, stream, 1, in, 0, (int)n, height, width);
  if (rc) return rc;
  rc = launch_odometry(h, 0, stream, 1);
  if (rc) return rc;
  return wait_pose(h, stream, 1, pose_out, info);
}

// Rebuilds the kNN structure of one stream from window ++ received map without appending a frame.
static int rebuild_search_structure(liodom_handle* h, int stream) {
  const DevView& v = h->v;
  const int map_blocks = cdiv(v.map_cap, 256);
  hipLaunchKernelGGL(k_hash_reset, dim3(64), dim3(256), 0, h->stream, v, stream);
  hipLaunchKernelGGL(k_hash_reset_done, dim3(1), dim3(1), 0, h->stream, v, stream);
  if (h->lds_hash_build) {
    hipLaunchKernelGGL(k_hash_build, dim3(1), dim3(kBuildThreads), hash_build_lds_bytes(), h->stream, v, stream, -1);
  } else {
    hipLaunchKernelGGL(k_window_insert, dim3(map_blocks, 1), dim3(256), 0, h->stream, v, stream, -1);
    hipLaunchKernelGGL(k_hash_alloc, dim3(map_blocks, 1), dim3(256), 0, h->stream, v, stream);
    hipLaunchKernelGGL(k_hash_scatter, dim3(map_blocks, 1), dim3(256), 0, h->stream, v, stream);
  }
  HIP_TRY(hipGetLastError());
  return LIODOM_OK;
}

int liodom_set_received_map(liodom_handle_t* h, int stream, const float* xyzi, int64_t n) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (!h->v.mapping) { g_last_error = "liodom_set_received_map: the handle was created with mapping = 0"; return LIODOM_ERR_UNSUPPORTED; }
  if (n < 0 || (n > 0 && !xyzi)) return LIODOM_ERR_INVALID_ARG;
  if (n > h->v.recv_cap) { g_last_error = "liodom_set_received_map: cloud larger than recv_capacity"; return LIODOM_ERR_CAPACITY; }
  SideLocks lk(h, true, false);
  const int ni = (int)n;
  if (n) HIP_TRY(hipMemcpyAsync(h->v.recv_pts + (size_t)stream * h->v.recv_cap, xyzi, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(&h->v.state[stream].n_recv, &ni, sizeof(int), hipMemcpyHostToDevice, h->stream));
  rc = rebuild_search_structure(h, stream);
  if (rc) return rc;
  HIP_TRY(sync_odometry(h));        // xyzi / ni are the caller's and this frame's memory
  return LIODOM_OK;
}

int liodom_set_imu_orientation(liodom_handle_t* h, int stream, const double* q_xyzw) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (!q_xyzw) return LIODOM_ERR_INVALID_ARG;
  SideLocks lk(h, true, false);
  // pageable-memory async copies are staged by the runtime before the call returns
  HIP_TRY(hipMemcpyAsync(h->v.imu_q + (size_t)stream * 4, q_xyzw, sizeof(double) * 4, hipMemcpyHostToDevice, h->stream));
  return LIODOM_OK;
}

int liodom_set_laser_to_base(liodom_handle_t* h, const double* T) {
  if (!h || !T) return LIODOM_ERR_INVALID_ARG;
  if (int rc = enter(h)) return rc;
  SideLocks lk(h, true, true);
  for (int k = 0; k < 12; k++) h->v.laser_to_base[k] = T[k];      // kernels take the view by value
  return LIODOM_OK;
}

int liodom_get_received_map(liodom_handle_t* h, int stream, float* xyzi, int64_t cap, int64_t* n_points) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (!h->v.mapping) { if (n_points) *n_points = 0; return LIODOM_OK; }
  SideLocks lk(h, true, false);
  HIP_TRY(sync_odometry(h));
  int n = 0;
  HIP_TRY(hipMemcpy(&n, &h->v.state[stream].n_recv, sizeof(int), hipMemcpyDeviceToHost));
  if (n_points) *n_points = n;
  if (n > cap) { g_last_error = "received-map buffer too small"; return LIODOM_ERR_CAPACITY; }
  if (n && xyzi) HIP_TRY(hipMemcpy(xyzi, h->v.recv_pts + (size_t)stream * h->v.recv_cap, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost));
  return LIODOM_OK;
}

int liodom_attach_mapper(liodom_handle_t* h, int stream, liodom_map_t* m, int cells_xy, int cells_z) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (!h->v.mapping) { g_last_error = "liodom_attach_mapper: the handle was created with mapping = 0"; return LIODOM_ERR_UNSUPPORTED; }
  SideLocks lk(h, true, false);
  HIP_TRY(sync_odometry(h));
  if (liodom_map* old = h->mappers[stream]) {       // detach: the map gets a stream of its own again
    h->mappers[stream] = nullptr;
    old->stream = nullptr; old->own_stream = false;
    HIP_TRY(hipStreamCreateWithFlags(&old->stream, hipStreamNonBlocking));
    old->own_stream = true;
  }
  if (!m) return LIODOM_OK;
  if (m->device != h->config.device) { g_last_error = "liodom_attach_mapper: map and handle live on different devices"; return LIODOM_ERR_INVALID_ARG; }
  if (cells_xy < 0 || cells_z < 0) return LIODOM_ERR_INVALID_ARG;
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (m->own_stream) { (void)hipStreamDestroy(m->stream); m->own_stream = false; }
  m->stream = h->stream;
  h->mappers[stream] = m;
  h->mapper_cells_xy[stream] = cells_xy;
  h->mapper_cells_z[stream] = cells_z;
  return LIODOM_OK;
}

int liodom_alloc_resident(liodom_handle_t* h, int n_slots) {
  if (!h || n_slots < 1) return LIODOM_ERR_INVALID_ARG;
  int rc0 = enter(h);
  if (rc0) return rc0;
  SideLocks lk(h, true, true);
  HIP_TRY(hipStreamSynchronize(h->stream_x));       // an extraction issued ahead may still read the old buffer
  HIP_TRY(sync_odometry(h));
  h->pf_slot = -1;
  if (h->resident) { hipFree(h->resident); h->resident = nullptr; h->n_slots = 0; }
  const size_t bytes = sizeof(float4) * (size_t)h->S * (size_t)n_slots * (size_t)h->v.max_points;
  void* raw = nullptr;
  HIP_TRY(hipMalloc(&raw, bytes));
  h->resident = static_cast<float4*>(raw);
  h->n_slots = n_slots;
  return LIODOM_OK;
}

int liodom_upload_scan(liodom_handle_t* h, int stream, int slot, const float* xyzi, int64_t n) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (!h->resident || slot < 0 || slot >= h->n_slots || n < 0 || n > h->v.max_points) { g_last_error = "bad resident slot"; return LIODOM_ERR_INVALID_ARG; }
  SideLocks lk(h, false, true);
  float4* dst = h->resident + ((size_t)slot * h->S + stream) * (size_t)h->v.max_points;
  if (n) HIP_TRY(hipMemcpy(dst, xyzi, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice));
  return LIODOM_OK;
}

int liodom_process_resident(liodom_handle_t* h, int slot, int64_t n, int height, int width,
                            double* poses_out, liodom_step_info_t* infos_out) {
  return liodom_process_resident_pipelined(h, slot, -1, n, height, width, poses_out, infos_out);
}

// One scan of the pipelined replay (both sides locked by the caller).  wait: read the poses back before returning.
static int replay_one(liodom_handle_t* h, int slot, int next_slot, int64_t n, int height, int width, bool wait,
                      double* poses_out, liodom_step_info_t* infos_out, const float* next_host = nullptr, int64_t host_stride = 0);

int liodom_process_resident_pipelined(liodom_handle_t* h, int slot, int next_slot, int64_t n, int height,
                                      int width, double* poses_out, liodom_step_info_t* infos_out) {
  int rc0 = enter(h);
  if (rc0) return rc0;
  if ((rc0 = check_usable(h))) return rc0;
  SideLocks lk(h, true, true);
  if ((rc0 = tickets_idle(h))) return rc0;        // (the replay fills the same pipeline edge buffers)
  return replay_one(h, slot, next_slot, n, height, width, poses_out != nullptr || infos_out != nullptr, poses_out, infos_out);
}

int liodom_replay_resident(liodom_handle_t* h, int first_slot, int count, int ahead, int depth, int64_t n, int height, int width,
                           double* poses_out, liodom_step_info_t* infos_out) {
  int rc0 = enter(h);
  if (rc0) return rc0;
  if ((rc0 = check_usable(h))) return rc0;
  if (count < 0 || first_slot < 0 || depth < 0 || depth > 1) { g_last_error = "bad resident range / depth"; return LIODOM_ERR_INVALID_ARG; }
  SideLocks lk(h, true, true);
  if ((rc0 = tickets_idle(h))) return rc0;
  auto out_p = [&](int i) { return poses_out ? poses_out + (size_t)i * h->S * 7 : nullptr; };
  auto out_i = [&](int i) { return infos_out ? infos_out + (size_t)i * h->S : nullptr; };
  const auto t_call = std::chrono::steady_clock::now();
  h->replay_stamps.clear();
  h->replay_stamps.reserve((size_t)count);
  auto stamp = [&]() { h->replay_stamps.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count()); };
  for (int i = 0; i < count; i++) {
    const int slot = first_slot + i;
    const int next = (i + 1 < count || ahead) ? slot + 1 : -1;
    if (depth == 0) {
      const int rc = replay_one(h, slot, next, n, height, width, true, out_p(i), out_i(i));
      if (rc) return rc;
      continue;
    }
    // depth 1: enqueue scan i, then collect the pose of scan i - 1 (its sequence number is one behind the enqueue count)
    const auto t_a = std::chrono::steady_clock::now();
    int rc = replay_one(h, slot, next, n, height, width, false, nullptr, nullptr);
    if (rc) return rc;
    const auto t_b = std::chrono::steady_clock::now();
    if (i > 0) { rc = wait_pose(h, 0, h->S, out_p(i - 1), out_i(i - 1), 1); if (rc) return rc; stamp(); }
    // (where the host's time goes in this loop: liodom_get_modes reports the two averages — the host has one scan's duration to
    //  enqueue the next scan's launches; if the first number approaches the scan period the GPU starves)
    h->replay_enq_ns += std::chrono::duration<double, std::nano>(t_b - t_a).count();
    h->replay_wait_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t_b).count();
    h->replay_timed++;
  }
  if (depth == 1 && count > 0) { const int rc = wait_pose(h, 0, h->S, out_p(count - 1), out_i(count - 1), 0); if (rc) return rc; stamp(); }
  return LIODOM_OK;
}

// Host-fed replay: the scan of every stream for resident slot `slot` is copied from host memory on the extraction stream
// (ordered behind the extraction that last read the slot).
static int upload_slot_async(liodom_handle_t* h, int slot, const float* host, int64_t stride, int64_t n) {
  // copies on their own stream (copy engine) so that they run beside the extraction kernels of the previous scan; the
  // slot is free once the extraction that last read it has completed, and its extraction waits for the upload
  const int r = slot % 3;
  if (!h->stream_c) {
    // HIP multiplexes its streams onto a few hardware queues: a handle that owns stream_k (overlapped second kNN pass) has
    // no queue left for a copy stream of its own — with one, the uploads of the host-fed replay ended up behind other
    // streams' launches (11.3k -> 7.5k scans/s).  That replay does not overlap the pass (ov_suppress), so stream_k is
    // idle in it and carries the uploads.
    if (h->stream_k) { h->stream_c = h->stream_k; h->stream_c_shared = true; }
    else HIP_TRY(hipStreamCreateWithFlags(&h->stream_c, hipStreamNonBlocking));
  }
  if (!h->ev_up[0]) {
    for (int b = 0; b < 3; b++) {
      HIP_TRY(hipEventCreateWithFlags(&h->ev_up[b], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_xdone[b], hipEventDisableTiming));
    }
  }
  hipStream_t q = extract_queue(h);
  if (h->ev_xdone_valid[r]) HIP_TRY(hipStreamWaitEvent(h->stream_c, h->ev_xdone[r], 0));
  for (int s = 0; s < h->S && n > 0; s++) {
    float4* dst = h->resident + ((size_t)slot * h->S + s) * (size_t)h->v.max_points;
    HIP_TRY(hipMemcpyAsync(dst, host + (size_t)s * (size_t)stride, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, h->stream_c));
  }
  HIP_TRY(hipEventRecord(h->ev_up[r], h->stream_c));
  HIP_TRY(hipStreamWaitEvent(q, h->ev_up[r], 0));
  return LIODOM_OK;
}
static int upload_slot_consumed(liodom_handle_t* h, int slot) {      // call right after the slot's extraction has been issued
  const int r = slot % 3;
  HIP_TRY(hipEventRecord(h->ev_xdone[r], extract_queue(h)));
  h->ev_xdone_valid[r] = true;
  return LIODOM_OK;
}

static int replay_one(liodom_handle_t* h, int slot, int next_slot, int64_t n, int height, int width, bool wait,
                      double* poses_out, liodom_step_info_t* infos_out, const float* next_host, int64_t host_stride) {
  if (!h->resident || slot < 0 || slot >= h->n_slots || next_slot >= h->n_slots || n < 0 || n > h->v.max_points) {
    g_last_error = "bad resident slot"; return LIODOM_ERR_INVALID_ARG;
  }
  // resident layout: [slot][stream][max_points]: one lock-step launch reads a contiguous block
  h->replay_live.store(true);
  const int eb = h->parity;
  int rc;
  if (h->pf_slot != slot) {                       // extraction not issued ahead: do it now
    rc = issue_extract(h, slot, eb, (int)n, height, width);
    if (rc) return rc;
  }
  h->pf_slot = -1;
  rc = enqueue_pipeline_odometry(h, eb, h->eb_seq[eb]);
  if (rc) return rc;
  h->parity = (eb + 1) % kEdgePipeBufs;
  if (next_slot >= 0) {                           // overlap the next scan's (upload and) extraction with this odometry
    if (next_host && !h->replay_host_dev) { rc = upload_slot_async(h, next_slot, next_host, host_stride, n); if (rc) return rc; }
    // (a gate in front of the next scan's extraction — start it when this scan's first solve starts, so that it runs beside the
    //  solves instead of beside the first kNN pass — was measured: -1.6 %; removed)
    if (next_host && h->replay_host_dev) {
      // zero-copy: the extraction's first kernel reads the page-locked scan itself (device-visible address of next_host) and
      // leaves its device copy in the resident slot; no upload, no copy stream, no events
      const float4* hd = h->replay_host_dev + (reinterpret_cast<const float4*>(next_host) - h->replay_host_base);
      rc = issue_extract(h, next_slot, h->parity, (int)n, height, width, hd, (size_t)host_stride / 4);
    } else {
      rc = issue_extract(h, next_slot, h->parity, (int)n, height, width);
    }
    if (rc) return rc;
    if (next_host && !h->replay_host_dev) { rc = upload_slot_consumed(h, next_slot); if (rc) return rc; }
    h->pf_slot = next_slot;
  }
  if (wait) return wait_pose(h, 0, h->S, poses_out, infos_out);
  return LIODOM_OK;
}

int liodom_replay_host(liodom_handle_t* h, const float* xyzi_base, int64_t scan_stride_floats, int count, int depth,
                       int64_t n, int height, int width, double* poses_out, liodom_step_info_t* infos_out) {
  int rc0 = enter(h);
  if (rc0) return rc0;
  if ((rc0 = check_usable(h))) return rc0;
  if (count < 0 || depth < 0 || depth > 1 || n < 0 || n > h->v.max_points || (count > 0 && n > 0 && !xyzi_base) || scan_stride_floats < 4 * n) {
    g_last_error = "liodom_replay_host: bad arguments"; return LIODOM_ERR_INVALID_ARG;
  }
  constexpr int kRing = 3;
  if (h->n_slots < kRing) { const int rc = liodom_alloc_resident(h, kRing); if (rc) return rc; }
  SideLocks lk(h, true, true);
  int rc = tickets_idle(h);
  if (rc) return rc;
  rc = drain_pipeline(h);
  if (rc) return rc;
  // Measured (MI355X, HDL-64 shape): with uploads the loop is bound by the host's enqueue work (an upload, three event
  // operations, six extraction and five odometry launches per scan: 88 us); the overlapped second kNN pass adds a gate and an
  // ALLOC launch on a third stream and made it 133 us.  So not here.
  struct Suppress { liodom_handle* h; ~Suppress() { h->ov_suppress = false; h->replay_host_dev = nullptr; h->replay_host_base = nullptr; } } suppress{h};
  h->ov_suppress = true;
  if (h->zero_copy && count > 0 && n > 0 && scan_stride_floats % 4 == 0 && (reinterpret_cast<uintptr_t>(xyzi_base) & 15u) == 0) {
    // page-locked AND mapped (liodom_pin_host_buffer, hipHostMalloc): the extraction reads the scans in place — the loop then
    // enqueues no upload and no event, and the overlapped second kNN pass stays on (its stream is not needed for copies)
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<float*>(xyzi_base), 0) == hipSuccess && dp) {
      h->replay_host_dev = static_cast<const float4*>(dp);
      h->replay_host_base = reinterpret_cast<const float4*>(xyzi_base);
      h->ov_suppress = false;
    } else {
      (void)hipGetLastError();
    }
  }
  auto out_p = [&](int i) { return poses_out ? poses_out + (size_t)i * h->S * 7 : nullptr; };
  auto out_i = [&](int i) { return infos_out ? infos_out + (size_t)i * h->S : nullptr; };
  auto src = [&](int i) { return xyzi_base + (size_t)i * (size_t)h->S * (size_t)scan_stride_floats; };
  for (int i = 0; i < count; i++) {
    const int slot = i % kRing, next = (i + 1 < count) ? (i + 1) % kRing : -1;
    if (i == 0) {                                 // first scan: upload + extraction now
      if (h->replay_host_dev) {
        rc = issue_extract(h, slot, h->parity, (int)n, height, width, h->replay_host_dev, (size_t)scan_stride_floats / 4);
        if (rc) return rc;
        h->pf_slot = slot;
      } else {
        rc = upload_slot_async(h, slot, src(0), scan_stride_floats, n);
        if (rc) return rc;
        rc = issue_extract(h, slot, h->parity, (int)n, height, width);
        if (rc) return rc;
        h->pf_slot = slot;
        rc = upload_slot_consumed(h, slot);
        if (rc) return rc;
      }
    }
    const float* nh = next >= 0 ? src(i + 1) : nullptr;
    if (depth == 0) {
      rc = replay_one(h, slot, next, n, height, width, true, out_p(i), out_i(i), nh, scan_stride_floats);
      if (rc) return rc;
      continue;
    }
    rc = replay_one(h, slot, next, n, height, width, false, nullptr, nullptr, nh, scan_stride_floats);
    if (rc) return rc;
    if (i > 0) { rc = wait_pose(h, 0, h->S, out_p(i - 1), out_i(i - 1), 1); if (rc) return rc; }
  }
  if (depth == 1 && count > 0) { rc = wait_pose(h, 0, h->S, out_p(count - 1), out_i(count - 1), 0); if (rc) return rc; }
  if (h->stream_c) HIP_TRY(hipStreamSynchronize(h->stream_c));
  for (int b = 0; b < 3; b++) h->ev_xdone_valid[b] = false;
  return drain_pipeline(h);                       // the caller's buffer may be reused / unpinned after the return
}

int liodom_pin_host_buffer(void* p, int64_t bytes) {
  if (!p || bytes <= 0) return LIODOM_ERR_INVALID_ARG;
  HIP_TRY(hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault));
  return LIODOM_OK;
}
int liodom_unpin_host_buffer(void* p) {
  if (!p) return LIODOM_ERR_INVALID_ARG;
  HIP_TRY(hipHostUnregister(p));
  return LIODOM_OK;
}

int liodom_sync(liodom_handle_t* h) {
  int rc0 = enter(h);
  if (rc0) return rc0;
  SideLocks lk(h, true, true);
  HIP_TRY(hipStreamSynchronize(h->stream_x));
  HIP_TRY(sync_odometry(h));
  if (h->stream_k) HIP_TRY(hipStreamSynchronize(h->stream_k));
  // everything has completed: unless an extraction has been issued ahead for the replay's next scan, the pipeline edge buffers
  // are free again (for the ticket API, or for a replay that starts over at buffer 0)
  if (h->pf_slot < 0) return drain_pipeline(h);
  return LIODOM_OK;
}

int liodom_get_pose_log(liodom_handle_t* h, int stream, int first, int count, double* poses_out,
                        liodom_step_info_t* infos_out) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (first < 0 || count < 0 || first + count > h->v.pose_log_cap) { g_last_error = "pose log range"; return LIODOM_ERR_INVALID_ARG; }
  SideLocks lk(h, true, false);
  HIP_TRY(sync_odometry(h));
  if (poses_out && count)
    HIP_TRY(hipMemcpy(poses_out, h->v.pose_log + ((size_t)stream * h->v.pose_log_cap + first) * 7, sizeof(double) * 7 * (size_t)count, hipMemcpyDeviceToHost));
  if (infos_out && count)
    HIP_TRY(hipMemcpy(infos_out, h->v.info_log + (size_t)stream * h->v.pose_log_cap + first, sizeof(liodom_step_info_t) * (size_t)count, hipMemcpyDeviceToHost));
  return LIODOM_OK;
}

static int get_window_impl(liodom_handle_t* h, int stream, float* xyzi, int64_t cap, int64_t* n_points, int* n_frames) {
  HIP_TRY(sync_odometry(h));
  StreamState st;
  HIP_TRY(hipMemcpy(&st, h->v.state + stream, sizeof(st), hipMemcpyDeviceToHost));
  const int P = h->P;
  std::vector<int> base((size_t)P + 1), slot((size_t)P);
  HIP_TRY(hipMemcpy(base.data(), h->v.win_base + (size_t)stream * (P + 1), sizeof(int) * (P + 1), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(slot.data(), h->v.win_slot + (size_t)stream * P, sizeof(int) * P, hipMemcpyDeviceToHost));
  if (n_points) *n_points = st.n_map;
  if (n_frames) *n_frames = st.n_frames;
  if (st.n_map > cap) { g_last_error = "window buffer too small"; return LIODOM_ERR_CAPACITY; }
  for (int j = 0; j < st.n_frames && xyzi; j++) {
    const int cnt = base[j + 1] - base[j];
    if (cnt > 0)
      HIP_TRY(hipMemcpy(xyzi + 4 * (size_t)base[j], h->v.win_pts + ((size_t)stream * P + slot[j]) * h->v.edge_cap,
                        sizeof(float4) * (size_t)cnt, hipMemcpyDeviceToHost));
  }
  return LIODOM_OK;
}

int liodom_get_window(liodom_handle_t* h, int stream, float* xyzi, int64_t cap, int64_t* n_points, int* n_frames) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  SideLocks lk(h, true, false);
  return get_window_impl(h, stream, xyzi, cap, n_points, n_frames);
}

int liodom_get_local_map(liodom_handle_t* h, int stream, float* xyzi, int64_t cap, int64_t* n_points, int* filtered) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  SideLocks lk(h, true, false);
  HIP_TRY(sync_odometry(h));
  StreamState st;
  HIP_TRY(hipMemcpy(&st, h->v.state + stream, sizeof(st), hipMemcpyDeviceToHost));
  if (filtered) *filtered = st.n_filt > 0 ? 1 : 0;
  if (st.n_filt == 0) {
    int nf = 0;
    int64_t nw = 0;
    rc = get_window_impl(h, stream, xyzi, cap, &nw, &nf);
    const int nr = h->v.mapping ? st.n_recv : 0;
    if (n_points) *n_points = nw + nr;
    if (rc) return rc;
    if (nw + nr > cap) { g_last_error = "local-map buffer too small"; return LIODOM_ERR_CAPACITY; }
    if (nr && xyzi) HIP_TRY(hipMemcpy(xyzi + 4 * (size_t)nw, h->v.recv_pts + (size_t)stream * h->v.recv_cap, sizeof(float4) * (size_t)nr, hipMemcpyDeviceToHost));
    return LIODOM_OK;
  }
  const int n = st.n_filt;
  if (n_points) *n_points = n;
  if (n > cap) { g_last_error = "local-map buffer too small"; return LIODOM_ERR_CAPACITY; }
  std::vector<float4> pts((size_t)n);
  std::vector<float> inten((size_t)n);
  HIP_TRY(hipMemcpy(pts.data(), h->v.filt_pts + (size_t)stream * h->v.map_cap, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(inten.data(), h->v.filt_int + (size_t)stream * h->v.map_cap, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; i++) order[i] = i;
  auto leaf = [&](int i) { unsigned int u; std::memcpy(&u, &pts[i].w, 4); return u; };
  std::sort(order.begin(), order.end(), [&](int a, int b) { return leaf(a) < leaf(b); });   // PCL output order
  for (int i = 0; i < n && xyzi; i++) {
    const float4& p = pts[order[i]];
    xyzi[4 * i] = p.x; xyzi[4 * i + 1] = p.y; xyzi[4 * i + 2] = p.z; xyzi[4 * i + 3] = inten[order[i]];
  }
  return LIODOM_OK;
}

int liodom_get_correspondences(liodom_handle_t* h, int stream, int it, int32_t* valid, int32_t* idx_a,
                               int32_t* idx_b, int cap, int* n) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (it < 0 || it > 1) return LIODOM_ERR_INVALID_ARG;
  SideLocks lk(h, true, false);
  HIP_TRY(sync_odometry(h));
  StreamState st;
  HIP_TRY(hipMemcpy(&st, h->v.state + stream, sizeof(st), hipMemcpyDeviceToHost));
  const int E = st.n_edges_buf[h->last_eb];
  if (n) *n = E;
  if (E > cap) { g_last_error = "correspondence buffer too small"; return LIODOM_ERR_CAPACITY; }
  std::vector<int2> ci((size_t)std::max(E, 1));
  if (E) HIP_TRY(hipMemcpy(ci.data(), h->v.corr_idx + ((size_t)stream * 2 + it) * h->v.edge_cap, sizeof(int2) * (size_t)E, hipMemcpyDeviceToHost));
  for (int i = 0; i < E; i++) {
    if (valid) valid[i] = ci[i].x >= 0 ? 1 : 0;
    if (idx_a) idx_a[i] = ci[i].x;
    if (idx_b) idx_b[i] = ci[i].y;
  }
  return LIODOM_OK;
}

int liodom_get_knn_queries(liodom_handle_t* h, int stream, int it, float* xyz0, int cap, int* n) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (it < 0 || it > 1) return LIODOM_ERR_INVALID_ARG;
  if (!h->v.knn_q) { g_last_error = "create the handle with debug_buffers = 1"; return LIODOM_ERR_UNSUPPORTED; }
  SideLocks lk(h, true, false);
  HIP_TRY(sync_odometry(h));
  int E = 0;
  HIP_TRY(hipMemcpy(&E, &h->v.state[stream].n_edges_buf[h->last_eb], sizeof(int), hipMemcpyDeviceToHost));
  if (n) *n = E;
  if (E > cap) { g_last_error = "query buffer too small"; return LIODOM_ERR_CAPACITY; }
  if (E && xyz0) HIP_TRY(hipMemcpy(xyz0, h->v.knn_q + ((size_t)stream * 2 + it) * h->v.edge_cap, sizeof(float4) * (size_t)E, hipMemcpyDeviceToHost));
  return LIODOM_OK;
}

int liodom_get_curvature(liodom_handle_t* h, int stream, double* curv, int64_t cap, int32_t* ring_offsets) {
  int rc = check_stream(h, stream);
  if (rc) return rc;
  if (!(h->v.debug & 1)) { g_last_error = "create the handle with debug_buffers = 1"; return LIODOM_ERR_UNSUPPORTED; }
  SideLocks lk(h, true, true);
  HIP_TRY(hipStreamSynchronize(h->stream_x));
  HIP_TRY(sync_odometry(h));
  std::vector<int> rs((size_t)h->H + 1), rl((size_t)h->H);
  HIP_TRY(hipMemcpy(rs.data(), h->v.ring_start + (size_t)stream * (h->H + 1), sizeof(int) * (h->H + 1), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(rl.data(), h->v.ring_len + (size_t)stream * h->H, sizeof(int) * h->H, hipMemcpyDeviceToHost));
  // output: the compacted rings back to back (on the device the rings of an organised cloud sit at ring * width)
  int64_t off = 0;
  for (int r = 0; r < h->H; r++) { if (ring_offsets) ring_offsets[r] = (int32_t)off; off += rl[r]; }
  if (ring_offsets) ring_offsets[h->H] = (int32_t)off;
  if (off > cap) { g_last_error = "curvature buffer too small"; return LIODOM_ERR_CAPACITY; }
  int64_t o = 0;
  for (int r = 0; r < h->H && curv; r++) {
    if (rl[r]) HIP_TRY(hipMemcpy(curv + o, h->v.ring_c + (size_t)stream * h->v.ring_stride + rs[r], sizeof(double) * (size_t)rl[r], hipMemcpyDeviceToHost));
    o += rl[r];
  }
  return LIODOM_OK;
}

int liodom_set_profiling(liodom_handle_t* h, int enable) {
  int rc = enter(h);
  if (rc) return rc;
  std::unique_lock<std::mutex> lo(h->mx_o), lx(h->mx_x);
  rc = drain_events(h);
  h->profiling = enable != 0;
  return rc;
}

int liodom_get_kernel_stats(liodom_handle_t* h, liodom_kernel_stat_t* stats) {
  if (!h || !stats) return LIODOM_ERR_INVALID_ARG;
  int rc = enter(h);
  if (rc) return rc;
  SideLocks lk(h, true, true);
  rc = drain_events(h);
  if (rc) return rc;
  for (int i = 0; i < LIODOM_NUM_KERNELS; i++) {
    std::memset(&stats[i], 0, sizeof(stats[i]));
    std::strncpy(stats[i].name, kKernelNames[i], sizeof(stats[i].name) - 1);
    stats[i].launches = h->k_count[i];
    stats[i].total_ms = h->k_ms[i];
  }
  return LIODOM_OK;
}

int liodom_reset_kernel_stats(liodom_handle_t* h) {
  int rc = enter(h);
  if (rc) return rc;
  SideLocks lk(h, true, true);
  rc = drain_events(h);
  for (int i = 0; i < LIODOM_NUM_KERNELS; i++) { h->k_ms[i] = 0; h->k_count[i] = 0; }
  return rc;
}

/* debug (schedule perturbation, tools/inject_delay.py): seed of the pseudo-random delays in front of every publication and behind
 * every successful in-kernel wait; 0 switches them off.  Only in a library built with -DLIODOM_INJECT_DELAY. */
int liodom_debug_set_inject_seed(liodom_handle_t* h, unsigned int seed) {
#if defined(LIODOM_INJECT_DELAY)
  if (!h) return LIODOM_ERR_INVALID_ARG;
  if (int rc = enter(h)) return rc;
  SideLocks lk(h, true, true);
  HIP_TRY(hipStreamSynchronize(h->stream_x));
  HIP_TRY(sync_odometry(h));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_inject_seed), &seed, sizeof(seed)));
  return LIODOM_OK;
#else
  (void)h; (void)seed;
  g_last_error = "liodom_debug_set_inject_seed: the product library carries no delay injection; build a variant with -DLIODOM_INJECT_DELAY (tools/variant_build.sh)";
  return LIODOM_ERR_UNSUPPORTED;
#endif
}

/* debug: raw phase timestamps (100 MHz) written by the kernels when LIODOM_DEBUG_CLOCKS is set */
// (debug) when the host collected every pose of the last liodom_replay_resident call (depth 1): microseconds since the call began
int liodom_debug_replay_stamps(liodom_handle_t* h, double* out_us, int cap) {
  if (!h || !out_us || cap < 0) return LIODOM_ERR_INVALID_ARG;
  const int n = (int)std::min<size_t>(h->replay_stamps.size(), (size_t)cap);
  for (int i = 0; i < n; i++) out_us[i] = h->replay_stamps[(size_t)i];
  return n;
}
int liodom_debug_clocks(liodom_handle_t* h, unsigned long long* out512) {
  if (!h || !out512) return LIODOM_ERR_INVALID_ARG;
  if (!kInstrument) { g_last_error = "liodom_debug_clocks: the product library carries no instrumentation; build a variant with -DLIODOM_INSTRUMENT (tools/variant_build.sh)"; return LIODOM_ERR_UNSUPPORTED; }
  if (int rc = enter(h)) return rc;
  SideLocks lk(h, true, true);
  HIP_TRY(hipStreamSynchronize(h->stream_x));
  HIP_TRY(sync_odometry(h));
  HIP_TRY(hipMemcpy(out512, h->v.dbg_clk, sizeof(unsigned long long) * 512, hipMemcpyDeviceToHost));
  return LIODOM_OK;
}

/* debug: per-query phase times of k_knn (stream 0, latest scan): out[2][edge_cap][8]; returns edge_cap through *cap */
int liodom_debug_knn_times(liodom_handle_t* h, unsigned int* out, int* cap) {
  if (!h || !cap) return LIODOM_ERR_INVALID_ARG;
  *cap = h->v.edge_cap;
  if (!out) return LIODOM_OK;
  if (!h->v.dbg_q) return LIODOM_ERR_UNSUPPORTED;
  if (int rc = enter(h)) return rc;
  SideLocks lk(h, true, true);
  HIP_TRY(hipStreamSynchronize(h->stream_x));
  HIP_TRY(sync_odometry(h));
  HIP_TRY(hipMemcpy(out, h->v.dbg_q, sizeof(unsigned int) * 2 * (size_t)h->v.edge_cap * 12, hipMemcpyDeviceToHost));
  return LIODOM_OK;
}

int liodom_get_modes(liodom_handle_t* h, char* buf, int cap) {
  if (!h || !buf || cap < 2) return LIODOM_ERR_INVALID_ARG;
  if (int rc = enter(h)) return rc;
  SideLocks lk(h, true, true);
  const DevView& v = h->v;
  // (speculative hand-overs of stream 0 since the last reset: the launches enqueued so far have to have run for the figures to mean
  //  anything — a caller that wants them synchronises first; this call does not)
  int spec[4] = {0, 0, 0, 0}, hbs[4] = {0, 0, 0, 0};
  unsigned int done_cnt[64] = {0};      // (chain mode: the passes' done counters of stream 0 — first pass [0], second pass [32] — against the host's target)
  if (v.knn_done0) (void)hipMemcpy(done_cnt, v.knn_done0, sizeof(done_cnt), hipMemcpyDeviceToHost);
  (void)hipMemcpy(spec, reinterpret_cast<const char*>(v.state) + offsetof(StreamState, spec_stats), sizeof(spec), hipMemcpyDeviceToHost);
  (void)hipMemcpy(hbs, reinterpret_cast<const char*>(v.state) + offsetof(StreamState, hb_stats), sizeof(hbs), hipMemcpyDeviceToHost);
  snprintf(buf, (size_t)cap,
           "n_streams=%d early_rebuild=%d hash_build=%s pipe_flags=%d flag_gate=%d lm_groups=%d knn_instance=%d knn_queries=%d "
           "knn_grid=%d/%d knn8=%d hash_incr=%d hash_rebuilds=%d hash_appends=%d hash_appends_spilled=%d hash_points_spilled=%d knn_partials=%d knn_saved_bound=%d knn_exact_only=%d line_gate_kernel=%d filter_local_map=%d mapping=%d "
           "rotation_mode=%d table_size=%d rebuild_delta=%.3f knn_overlap=%d streams_concurrent=%d safe_mode=%d ring_split=%d ring_split_max_wgs=%d ring_split_lb=%d fold_publish=%d chain=%d speculate=%d spec_early=%d/%d spec_unconfirmed=%d/%d chain_done=%u/%u/%u replay_enqueue_us=%.2f replay_wait_us=%.2f debug=%d",
           h->S, v.early_rebuild, v.early_rebuild ? "streamed" : (h->lds_hash_build ? "lds" : "global"), h->use_flags ? 1 : 0,
           (h->use_flags && h->flag_gate) ? 1 : 0, v.lm_groups, h->S >= 16 ? 128 : 256, v.knn_queries, v.knn_grid,
           v.knn_blocks, h->knn8 ? 1 : 0, v.hash_incr, hbs[0], hbs[1], hbs[2], hbs[3], v.knn_partials, v.knn_save_pos ? 2 : (v.knn_save_q ? 1 : 0), v.knn_exact_only, v.knn_nn ? 1 : 0, v.filter_local_map, v.mapping,
           v.rotation_mode, v.table_size, (double)v.rebuild_delta,
           (v.early_rebuild && (h->ov_ok || (h->chain_ok && !h->flag_gate)) && h->use_flags && g_live_handles.load() <= 1) ? 1 : 0, h->streams_concurrent ? 1 : 0, h->safe_mode ? 1 : 0, h->ring_split ? 1 : 0, h->ring_split ? h->ring_split_max_wgs : 0, h->ring_split_lb ? 1 : 0, h->fold_publish ? 1 : 0,
           (v.early_rebuild && h->chain_ok && h->use_flags && !h->flag_gate && g_live_handles.load() <= 1) ? 1 : 0, v.speculate, spec[0], spec[2], spec[1], spec[3], h->chain_count, done_cnt[0], done_cnt[32],
           h->replay_timed ? h->replay_enq_ns / (1e3 * (double)h->replay_timed) : 0.0, h->replay_timed ? h->replay_wait_ns / (1e3 * (double)h->replay_timed) : 0.0, v.debug);
  return LIODOM_OK;
}

int liodom_device_count(int* count) {
  if (!count) return LIODOM_ERR_INVALID_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return LIODOM_OK;
}

int liodom_device_pci_bus_id(int device, char* bus_id, int cap) {
  if (!bus_id || cap < 16) return LIODOM_ERR_INVALID_ARG;
  HIP_TRY(hipDeviceGetPCIBusId(bus_id, cap, device));
  return LIODOM_OK;
}

int liodom_device_info(liodom_handle_t* h, char* name, int name_cap, int* compute_units) {
  if (!h) return LIODOM_ERR_INVALID_ARG;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, h->config.device));
  if (name && name_cap > 0) {
    const char* nm = prop.name[0] ? prop.name : prop.gcnArchName;   // some driver stacks leave name empty
    std::strncpy(name, nm, (size_t)name_cap - 1); name[name_cap - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  return LIODOM_OK;
}

}  // extern "C"
