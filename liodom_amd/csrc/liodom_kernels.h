// liodom_kernels.h — hand-written HIP kernels of the LiODOM hot path for gfx950 (CDNA4).
//
// Kernel map (one launch covers all streams: blockIdx.y = stream):
//   k_classify        A1/A2  isValidPoint + ring id per point, per-tile ring histogram (feature_extractor.cc:84-179)
//   k_ring_scatter    A2     stable counting sort of the scan by ring (input order kept per ring)
//   k_ring_extract    A3-A5  11-tap curvature stencil in registers (float sums, double squares) ->
//                            greedy per-region selection with +-5 suppression (:181-313)
//   k_compact_edges          ring-padded edges -> dense edge cloud (output order of :186-252)
//   k_knn             A9     edges -> world, 27-cell voxel-hash 5-NN, FP64 line gate
//                            (laser_odometry.cc:300-366)
//   k_lm_solve        A10/A11 fused point-to-line residual/Jacobian + 6x6 normal equations,
//                            whole Ceres-style LM solve in one workgroup per stream; second call
//                            also finalises the scan (pose log, prediction :148-150, window
//                            bookkeeping :34-60)
//   streamed rebuild  A6     (handles with <= 4 streams) extra workgroups of the four launches above append the
//                            transformed edges to the sliding window (:231-235) and build the voxel hash the next
//                            scan's kNN searches, in a second table, while the scan is solved
//   k_window_insert / k_hash_alloc / k_hash_scatter
//                     A6     the same after the solve, three launches (mapping / filtered-map handles)
//   k_hash_build      A6     the same as ONE workgroup per stream with LDS atomics (handles with
//                            >= 16 streams)
//   k_voxel_* / k_filt_*  A7 filter_local_map: VoxelGrid(0.4) of the full window (:286-292)
//   k_imu_override    A8     use_imu: roll / pitch of the prediction from the IMU (:152-183)
//   (liodom_map.h)    A12-A14 the mapping node's Map: updateMap / getLocalMap / getMap
//
// All FP on the parity-critical paths is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "../../include/liodom_hip.h"
#include "liodom_math.h"
#include "wave_ops.h"

namespace liodom_dev {

// Wave priority of the kernels on a scan's dependent chain (k_knn<256>, k_lm_solve on few-stream handles): where they share a
// SIMD with the next scan's extraction kernels (other HIP stream, priority 0) the issue arbiter serves them first.  Measured
// (MI355X, HDL-64 shape, interleaved A/B of 4 rounds): 13.01k -> 13.23k scans/s at priority 1 and at 3.
#ifndef LIODOM_CHAIN_PRIO
#define LIODOM_CHAIN_PRIO 2
#endif
constexpr int kWave = 64;
constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
// Solving workgroups of 4 waves (round 6; 8 until then): one wave per SIMD with the whole register file to itself — no scratch in
// any instance of k_lm_solve (512 threads: 256 registers per lane and 108-172 B of scratch in the controller's path), half as many
// waves at every barrier.  Interleaved A/B against 512 (same box, bench.py): `value` 14 288 -> 14 577 (K = 200), 13 364 -> 13 503
// (the driver's K = 20), strict-sync +0.9 %, 256 lock-step streams 203.8k -> 207.1k; round 5 had measured +-1 % for the same switch
// (then one instance of the kernel per solve, now one per solve and mode).
#ifndef LIODOM_LM_THREADS
#define LIODOM_LM_THREADS 256
#endif
constexpr int kLmThreads = LIODOM_LM_THREADS;   // k_lm_solve: 4 waves, one per SIMD, all evaluate residual blocks
constexpr int kLmEvalThreads = kLmThreads;
constexpr int kLmCtl = kLmThreads - 64;   // lane 0 of the last wave also runs the trust-region logic; the other waves prepare (compaction, register cache) meanwhile
#ifndef LIODOM_LM_GROUPS_MAX
#define LIODOM_LM_GROUPS_MAX 8
#endif
constexpr int kLmGroupsMax = LIODOM_LM_GROUPS_MAX;
constexpr int kKnnGroup = 32;            // lanes cooperating on one query
constexpr int kMaxFrames = 256;          // window frames supported by the LDS prefix tables
constexpr int kEdgeBufs = 4;             // dense edge buffers: 0 / 1 / 2 odometry side (pipelined replay), 3 extraction side
constexpr int kEdgePipeBufs = 3;
constexpr int kEdgeBufX = 3;
constexpr int kOvReplicas = 8;           // copies of the first solve's result, 4 KiB apart, for the polling k_knn workgroups
constexpr int kOvGranules = 38;          // 19 doubles as {tag, 32 bits} granules
constexpr int kPredGranules = 38;        // 19 doubles (pred_xch): the prediction's matrix [12], its quaternion [4] and translation [3] (= the next solve's start point)

// Per-stream device state.
struct StreamState {
  double odom[12];        // pose used by the next kNN / solve (prediction or current estimate)
  double prev_odom[12];   // previous solved pose (laser_odometry.cc:149)
  double final_odom[12];  // solved pose of the scan being appended to the window
  double param_q[4];      // laser_odometry.h:98
  double param_t[3];      // laser_odometry.h:99
  int32_t initialized;    // init_ (laser_odometry.cc:108)
  int32_t append_raw;     // first frame: edges enter the window untransformed (:123)
  int32_t frame_count;    // frames ever appended
  int32_t n_frames;       // frames in the window (nframes_)
  int32_t n_edges_buf[4]; // edges in edge buffer 0 / 1 / 2 (pipelined replay: extraction of scan k+1 overlaps odometry of scan k) / 3 (liodom_extract_edges)
  int32_t reb_frame_count; // frame_count as of the scan's first solve: what the early rebuild (k_window_count_old) derives the kept frames from
  int32_t n_map;          // window points covered by the voxel hash
  int32_t n_used_tab[2];  // occupied slots of the cell hash (list used_cells); with early_rebuild one per table (the table searched
                          // while frame_count = F is table F & 1, the next build goes into the other), else only [0] is used
  int32_t cursor;         // allocation cursor into the cell-sorted point array
  int32_t scan_counter;
  uint32_t status;
  int32_t n_search;       // points covered by the kNN structure (window or filtered local map)
  uint32_t table_mask;    // slots - 1 of the cell hash currently in v.cells (LDS-built tables are smaller)
  // filter_local_map (laser_odometry.cc:286-292): VoxelGrid(0.4) of the full window
  int32_t n_filt;         // filtered points (0 when the kNN structure holds the raw window)
  int32_t vox_used;       // occupied voxels of the current voxel-grid build
  int32_t vox_cursor;
  int32_t vox_minb[3];    // PCL VoxelGrid min_b_ and div_b_
  int32_t vox_divb[3];
  int32_t n_recv;         // points of the received ~map cloud (mapping mode, SharedData::setLocalMap)
  int32_t n_ovf[2];       // early_rebuild: new-frame points kept in the overflow list of table 0 / 1 (sorted_pts[ovf_base ...])
  int32_t reb_initialized; // early_rebuild: `initialized` as of the scan's first kNN pass (the finalising solve sets it beside the builders)
  int32_t reb_pad;        // chain mode: the scan's edge count as PAD saw it (APPEND iterates over it, see edges_keep)
  int32_t spec_eval[2];   // speculative hand-over (kernels_sync.h), back-off: first / finalising solves still to sit out after a hand-over that was
                          // not confirmed
  // incremental cell hash of lock-step batches (kernels_rebuild.h, k_hash_append): between two rebuilds from the whole window the
  // new frame's points are appended to their cells and evicted frames stay where they are
  int32_t hb_main_fc;     // frame_count at the last rebuild (0: none yet)
  int32_t hb_main_old;    // absolute number of the window's oldest frame then
  int32_t hb_main_nf;     // frames in the window then
  int32_t hb_base[8];     // the window's frame offsets (win_base[0 .. 7]) then: hb_base[d] points have been evicted once d frames have
  int32_t hb_shift;       // points evicted since the rebuild: a stored window index minus hb_shift is the current one, below it: evicted
  int32_t hb_cursor;      // first free place of the point array behind everything the rebuild allocated: room for the cells k_hash_append creates
  int32_t hb_stats[4];    // since the last reset: rebuilds, appends, appends that spilled, points spilled
  int32_t hb_spill;       // appended points that found no room in their cell (or no cell): kept in the spill list at the end of the point array,
                          // which every query scans, until the next rebuild
  int32_t spec_redo[2];   // ... [0] this scan's first solve / [1] the previous scan's finalising solve handed its iterate over early and did NOT end with
                          // it: the receiving kNN pass has been repeated by other workgroups, on other XCDs — what the pass's first edition left
                          // in the L2 of the next solve's XCD is stale (k_lm_solve invalidates before it consumes the pass's results)
  int32_t spec_stats[4];  // ... how it went since the last reset: first solve's iterates handed over early, of them not confirmed; the same for the
                          // finalising solve (liodom_get_modes: spec_early / spec_unconfirmed)
  double pred_odom[2][12]; // early_rebuild: the prediction the scan started from ([frames appended so far & 1]: the repair of a speculative hand-over
                          // that was not confirmed, kernels_sync.h, needs the previous scan's while the next scan's is already there),
                          // snapshot taken by the scan's first kNN launch: st.odom moves
                          // on with the solves, and the finalising solve writes the NEXT scan's prediction while builders of this scan still run
  liodom_step_info_t info;
};

// One voxel-hash slot (16 B, 16-B aligned).
struct __attribute__((aligned(16))) CellSlot {
  unsigned long long key;   // packed cell coordinates or kEmptyKey
  unsigned int start;       // first point of the cell in sorted_pts
  unsigned int cnt;         // points in the cell
};

// early_rebuild keeps two cell hashes per stream: arrays indexed by stream (cells, cell_bits, used_cells) are indexed by
// s + parity * n_streams instead.
#define LD_TAB_PARITY(v, frame_count) ((v).early_rebuild ? ((frame_count) & 1) : 0)

// Per-stream result record in host-mapped memory.  seq is written last (system-scope release)
// with the number of scans completed; the host spins on it instead of using events / memcpy.
struct HostOut {
  double pose[7];
  liodom_step_info_t info;
  int seq;
  int pad;
};

// Everything the kernels need (passed by value).
struct DevView {
  // parameters
  double min_range, max_range;
  int lidar_type, scan_lines, scan_regions, edges_per_region;
  long long min_points_per_scan;
  int prev_frames;
  int apply_on_ftol;
  int rotation_mode;        // what Eigen's Transform::rotation() returns: 1 polar factor (Eigen 3.3.x), 0 linear() (>= 3.4)
  int filter_local_map;     // params.filter_local_map_ (and !mapping_)
  int lm_groups;            // workgroups cooperating on one stream's solve (1 or kLmGroupsMax)
  float vox_inv;            // 1.0f / 0.4f as PCL computes inverse_leaf_size_
  // capacities
  int n_streams, max_points, ring_cap, slots_per_ring, edge_cap, map_cap, table_size;
  int pose_log_cap;
  int debug;                // bit 0: debug buffers (curvature dump, kNN queries); bit 5: phase timestamps (dbg_clk).  Neither changes a result.
  // per-stream arrays (stride = capacity)
  StreamState* state;
  unsigned char* ring_id;   size_t ring_id_stride;
  unsigned short* tile_hist; // [S][tile_cap][H] points per (2048-point tile, ring)
  int tile_cap;
  float4* ring_pts;         // [S][max_points] the scan sorted by ring (stable) 
  int* ring_src;            // [S][max_points] source index of every sorted point
  int* ring_start;          // [S][H+1] offsets of the rings in ring_pts
  int* ring_len;            // [S][H] points of every ring (lidar_type 1: rings sit at ring * width, not back to back)
  float4* edges_pad;        // [S][H][slots_per_ring]
  int2* edges_pad_meta;     // (idx_in_ring, src)
  int* ring_nedges;         // [S][H]
  unsigned short* split_hist; int split_pad;   // [S][H][split_pad] k_ring_split: points per (ring, 2048-point tile); split_pad = tiles rounded up to 8
  unsigned int* split_ctr;  // [2 S] k_ring_split: tiles of stream s that have published their histogram [2 s + 1]
  // k_ring_split_lb (lock-step batches): rings at a fixed pitch, tiles sum their predecessors' tagged counts
  unsigned long long* lb_desc;   // [S][tile_cap][lb_hpad] {launch tag, points of (tile, ring)}
  unsigned int* lb_ticket;       // [1] tiles started in the current launch (k_ring_extract zeroes it)
  unsigned int* lb_ovf;          // [S] a ring of the stream outgrew its pitch: k_ring_split_fix redoes the stream
  int lb_hpad, ring_pitch;       // row of lb_desc (H rounded up to 64); points a ring may hold in the pitched layout
  size_t ring_stride;            // per-stream stride of ring_pts / ring_src / ring_c / ring_picked (>= max_points, >= H * ring_pitch)
  int* ring_npoints;        // [S][H]
  double* ring_c;           // [S][max_points] smoothness per ring-sorted point: debug dump (debug & 1) and generic-path scratch
  unsigned char* ring_picked;  // [S][max_points] picked_ marks of the generic path
  float4* edges;            // [kEdgeBufs][S][edge_cap] dense
  int4* edges_meta;         // [kEdgeBufs][S][edge_cap] (ring, idx_in_ring, src, 0)
  float4* corr_a;           // [S][2][edge_cap]  xyz of NN0, w = valid; one half per kNN pass: a pass that starts from a pose handed over early
                            // (kernels_sync.h, speculative hand-over) writes while the previous solve's last evaluation still reads
  float4* corr_b;           // [S][2][edge_cap]  xyz of NN1
  int2* corr_idx;           // [S][2][edge_cap] window indices of (NN0, NN1), debug/parity
  float4* knn_q;            // [S][2][edge_cap] world-frame float query of every edge and pass (debug_buffers only, else null)
  float4* win_pts;          // [S][P][edge_cap]
  int* win_n;               // [S][P]
  int* win_base;            // [S][P+1] logical prefix (oldest first)
  int* win_slot;            // [S][P]  logical frame -> slot
  CellSlot* cells;          // [S][table_size]  {key, start, cnt}: one 16-B load per probe
  int lm_lds_reduce;        // k_lm_solve reduces through the transposed LDS matrix (fits for edge_cap <= ~10 000)
  int use_imu;              // params.use_imu_ (laser_odometry.cc:152)
  double laser_to_base[12]; // laser_to_base_ (laser_odometry.cc:110-119), identity unless liodom_set_laser_to_base
  double* imu_q;            // [S][4] last IMU orientation [x y z w] (SharedData::last_IMU_ori_)
  int mapping;              // params.mapping_: the kNN cloud is window + received map (laser_odometry.cc:310-314)
  int recv_cap;
  float4* recv_pts;         // [S][recv_cap] last received ~map cloud (world frame)
  int lds_cells_max;   // occupied-cell limit of the LDS-built table (kLdsCellsMax; lowered by tests)
  int* pt_rank;  // [S][map_cap]  rank of each point inside its cell (old value of the count atomic)
  unsigned int* cell_bits;  // [S][table_size/32] occupancy bitmap: empty-cell probes stay in a 32 KB array
  int* used_cells;          // [S][map_cap]
  int* pt_cell;             // [S][map_cap]
  float4* sorted_pts;       // [S][map_cap]  xyz + window index bits
  double* pose_log;         // [S][pose_log_cap][7]
  liodom_step_info_t* info_log;  // [S][pose_log_cap]
  HostOut* host_out;        // [S][2] host-mapped pinned memory, polled by the host (zero-copy); record of scan k = k & 1
  // filter_local_map: voxel grouping of the window and the filtered cloud
  CellSlot* vox_cells;      // [S][table_size] key = PCL voxel index
  unsigned int* vox_fill;   // [S][table_size]
  int* vox_used_list;       // [S][map_cap]
  int* pt_vox;              // [S][map_cap] voxel slot of every window point
  int* vox_pts;             // [S][map_cap] window indices grouped by voxel, ascending inside a voxel
  float4* filt_pts;         // [S][map_cap] centroid xyz + voxel-index bits
  float* filt_int;          // [S][map_cap] centroid intensity
  double* knn_part;         // [S][2][knn_blocks][32] per-k_knn-workgroup sums of the 29-entry normal-equation accumulator at the pose the pass searched with (= the solve's first evaluation)
  unsigned char* corr_mask; // [S][2][mask_stride] bit q: query q of that k_knn workgroup has an accepted correspondence
  int mask_stride;          // knn_blocks rounded up to 128: the two passes' halves lie in different cache lines
  int knn_partials;         // k_knn also evaluates every accepted block at the solve's start pose and leaves per-workgroup sums (handles with < 16 streams)
  int knn_queries;          // queries per k_knn workgroup (8, or 4 for handles with >= 16 streams)
  unsigned int* cell_pad;   // [2 S][table_size] early_rebuild: room reserved in the cell for points of the new frame; after the allocation: end of the cell's range
  int used_cap;             // used_cells entries per table (map_cap; early_rebuild: + 8 edge_cap for cells only the padding touches)
  int sorted_cap;           // sorted_pts entries per table (early_rebuild: map_cap + 8 edge_cap of padding + edge_cap of overflow list; else map_cap)
  int ovf_base;             // first entry of the overflow list inside a table's sorted_pts
  float rebuild_delta;      // early_rebuild: a new-frame point may move this far (per axis) between the prediction and the solved pose and still land in a padded cell
  int hb_spill_base;        // hash_incr: first place of the spill list in sorted_pts (the last (kHbPeriod - 1) * edge_cap places)
  int hb_slack_min, hb_new_room;   // hash_incr: room of a cell beyond its population at a rebuild (at least this, else the population again); room of a new cell
  int hash_incr;            // lock-step batches: k_hash_append between rebuilds (LIODOM_HASH_INCR=0: k_hash_build every scan)
  unsigned int* cell_cap;   // [S][table_size] hash_incr: end of the room of every cell in sorted_pts (start + points + slack)
  int* knn8_cnt;            // [S] k_knn8: queries of the pass left to k_knn8_exact (k_line_gate resets it)
  int* knn8_list;           // [S][edge_cap] ... their numbers
  float4* knn_nn;           // [S][edge_cap][5] lock-step batches: the five neighbours of every query (w: found flag, index of NN0, NN1) for k_line_gate
  unsigned int* pipe_flags; // [kEdgePipeBufs + 1] pipelined replay without cross-stream events: [b] = sequence number of the extraction whose edges
                            // are complete in edge buffer b; [kEdgePipeBufs] = number of the last odometry (of this handle) that has completed entirely
  unsigned long long* pose_xch;   // [S][2][32] early_rebuild: solved pose handed to the workgroups that append the new frame (tagged 8-byte granules);
                                  // second copy: what the solve ended with (speculative hand-over)
  unsigned int* redo_sync;        // [2] self-resetting arrival counters of the repair path (append_fix)
  int early_rebuild;        // streamed rebuild: extra workgroups of the scan's four launches build the next scan's cell hash in the
                            // second table ("Streamed rebuild" below); no k_window_insert / k_hash_alloc / k_hash_scatter launches
  int knn_grid;             // k_knn workgroups launched per stream (each takes the query blocks b, b + knn_grid, ...)
  float4* knn_save_q;       // [S][edge_cap] first kNN pass of a scan: the query (xyz) and its fifth-nearest distance (w; inf if none): the second pass prunes with it
  int2* knn_save_pos;       // [S][edge_cap][32] first kNN pass: the two candidates every lane kept (positions in the cell-sorted array; -1: none)
  float* knn_save_g;        // [S][edge_cap] first kNN pass: guard — no map point outside the kept set was closer to that pass's query than sqrt(guard) (0: nothing saved)
  int knn_exact_only;       // (test switch) every kNN query takes the exact list path instead of the Best2 fast path: same results
  int knn_blocks;           // k_knn workgroups per stream = ceil(edge_cap / knn_queries), rounded up to a multiple of 4
  unsigned long long* lm_xch;   // [S][2][kLmGroupsMax][64] tagged granules: partial sums exchanged between the LM workgroups
  // Overlapped second kNN pass (handles with one stream, early_rebuild, flags; "Overlapped second kNN pass" below): the pass is
  // launched on its own HIP stream beside the scan's first solve and waits inside the kernel; tags = the launch sequence number.
  unsigned int* ov_flags;          // [S] sequence number of the latest scan whose first solve launch has started (its first kNN pass has completed)
  double spec_theta;               // ... by the model: the iterate leaves before an evaluation whose predicted cost change is below spec_theta x the function tolerance
  int spec_backoff;                // ... scans a solve sits out after a hand-over of its own that was not confirmed
  int speculate;                   // speculative hand-over of the first solve's result (kernels_sync.h): 0 off, 1 by the model's predicted cost change, 2 (test) as early as possible
  unsigned long long* pose_xch0;   // [S][kOvReplicas][512] the first solve's result (odom[12], q[4], t[3]) as 38 tagged granules, replicated over memory channels; granules 64 .. 101: the confirmation copy
  unsigned int* knn_done;          // [S][knn_grid] sequence number of the latest overlapped second pass workgroup b has completed
  // Chain mode (round 5; "Chain mode" in kernels_sync.h): the kNN passes and the rebuild ride on ONE HIP stream (kNN(0), gate, COUNT/PAD,
  // kNN(1), ALLOC, SCATTER/CLEAR, APPEND), the two solves on another; the first solve's launch is resident while the first pass
  // still runs and takes its results through done flags, as the finalising solve takes the second pass's.
  float4* edges_keep;              // [S][edge_cap] chain mode: the scan's edges as PAD saw them, for APPEND (the edge buffer itself may belong to a later
                                   // scan's extraction by the time APPEND's launch starts: the ticket slot is free once the pose has been collected)
  unsigned int* pub_counter;       // [kEdgeBufs] k_compact_edges: workgroups of the launch on edge buffer b that have completed their stores (the last one publishes and resets it)
  unsigned int* edge_cnt;          // [kEdgeBufs][32] write-through copy of n_edges_buf[b] of stream s (s < 32) at [b * 32 + s], a 128-byte line per buffer
  unsigned int* knn_done0;         // [S + 64] chain mode (S = 1): [s] workgroups of first passes that have completed, counted over the scans since the
                                   // last reset (one word: the first solve's workgroups poll it with one thread each); [32 + s] (a cache line of its own) the same for second passes
  unsigned long long* pred_xch;    // [S][kOvReplicas][512] the prediction the next scan starts from (st.odom, st.param_q, st.param_t after finalize_scan:
                                   // 19 doubles as 38 tagged granules, tag = scans completed), for the next scan's first kNN pass, which runs on the
                                   // other stream and may start before finalize_scan's plain stores to the state are visible
  // Device-resident hand-off of the two-thread binding (liodom_extract_edges_device -> liodom_odometry_step_device, one-stream
  // handles): k_compact_edges also leaves the dense edges of pipeline buffer b in host-mapped memory, so that the extractor
  // thread can publish ~edges (feature_extractor.cc:70-75) without a device-to-host copy call; null on other handles.
  float4* host_edges;           // [kEdgePipeBufs][edge_cap] host-mapped
  int4* host_edges_meta;        // [kEdgePipeBufs][edge_cap] (ring, idx_in_ring, src, 0)
  unsigned int* host_edges_hdr; // [2][kEdgePipeBufs]: sequence number of the extraction whose edges are complete in slot b (written by
                                // k_publish_edges, system scope); number of edges in slot b
  unsigned long long* dbg_clk;  // [16][32] phase timestamps (100 MHz) and counters, debug bit 5 only
  unsigned int* dbg_q;          // [2][edge_cap][8] per-query phase times of stream 0's latest scan (10 ns ticks since the workgroup's start), debug bit 5 only
};

// computeLocalMap's condition (laser_odometry.cc:286): filter && window full && !mapping
__device__ __forceinline__ bool filter_active(const DevView& v, const StreamState& st) {
  return v.filter_local_map != 0 && st.n_frames == v.prev_frames;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// In-kernel instrumentation (phase timestamps, per-query times, histograms; tools/gpu_debug.py) exists only in builds with
// -DLIODOM_INSTRUMENT (tools/variant_build.sh): the product library carries none of it — no debug branches in the hot kernels.
#if defined(LIODOM_INSTRUMENT)
constexpr bool kInstrument = true;
#else
constexpr bool kInstrument = false;
#endif
// phase stamps: kernel slot k, stamp index i (constant 100 MHz wall clock)
#define DBG_STAMP(v, cond, k, i) do { if (kInstrument && ((v).debug & 32) && (cond)) (v).dbg_clk[(k) * 32 + (i)] = wall_clock64(); } while (0)
// stamps of the overlapped second kNN pass (debug bit 7; entries 448.. of dbg_clk, shared with a bit-6 histogram)
#define OV_STAMP(v, cond, i) do { if (kInstrument && ((v).debug & 128) && (cond)) (v).dbg_clk[448 + (i)] = wall_clock64(); } while (0)
#define DBG_QSTAMP(i) do { if (kInstrument && (kInstrument && (v.debug & 32)) && s == 0 && hl == 0 && e < E) v.dbg_q[((size_t)outer_it * v.edge_cap + e) * 12 + (i)] = (unsigned int)(wall_clock64() - t_blk); } while (0)

// XCD-aware workgroup placement for lock-step launches (grid = blocks x streams).  Workgroups are dispatched
// round-robin over the 8 XCDs by linear id, and each XCD has its own 4 MB L2: with the natural mapping every
// XCD sees the data of ALL streams (256 x 0.7 MB for k_knn: nothing stays resident, 5.5 x the algorithmic
// traffic in round 1).  Remapped, the workgroups an XCD receives belong to one stream after the other
// (stream = 8 * (slot / blocks) + xcd), so its L2 holds one or two streams' cell-sorted points and tables at a
// time.  Identity unless the stream count is a multiple of 8.
__device__ __forceinline__ void xcd_remap(int& bx, int& by) {
  const int nbx = (int)gridDim.x, nby = (int)gridDim.y;
  if (nby < 8 || (nby & 7)) return;
  const int lin = bx + nbx * by;
  const int xcd = lin & 7, slot = lin >> 3;
  by = (slot / nbx) * 8 + xcd;
  bx = slot - (slot / nbx) * nbx;
}

__device__ __forceinline__ unsigned long long pack_cell(int cx, int cy, int cz) {
  const unsigned long long m = 0x1FFFFFull;  // 21 bits per axis; aliasing only adds far candidates
  return ((unsigned long long)(cx & m) << 42) | ((unsigned long long)(cy & m) << 21) |
         (unsigned long long)(cz & m);
}
// 32-bit hash of the three 21-bit cell coordinates (spatial-hash primes + murmur3's 32-bit finaliser): a dozen
// VALU instructions; the 64-bit finaliser used before cost ~30 (64-bit multiplies are emulated) in every probe of
// k_knn, which is VALU-issue bound on lock-step batches.
__device__ __forceinline__ unsigned int hash_cell(unsigned long long k, unsigned int mask) {
  const unsigned int x = (unsigned int)(k >> 42), y = (unsigned int)(k >> 21) & 0x1FFFFFu, z = (unsigned int)k & 0x1FFFFFu;
  unsigned int h = (x * 73856093u) ^ (y * 19349663u) ^ (z * 83492791u);
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h & mask;
}

// Every in-kernel wait is bounded twice: by an iteration count and — round 5 — by WALL-CLOCK time (g_wait_ticks, 100 MHz ticks;
// liodom_create sets it from LIODOM_WAIT_MS, default 50 ms = several hundred times the longest wait of an undisturbed scan).
// Beside a process that saturates the GPU the producers a waiter depends on get their CUs late, but often just in time for an
// iteration bound sized for ~0.3 s: the replay then crawled at tens of milliseconds per scan instead of failing over to the
// safe mode (the two-process soak took 5 minutes).  A wait that outlasts the time bound raises the same status bits.
__device__ unsigned long long g_wait_ticks = 5000000ull;
__device__ __forceinline__ bool wait_expired(unsigned int spins, unsigned long long& t0) {
  if ((spins & 31u) != 0u) return false;
  const unsigned long long now = wall_clock64();
  if (t0 == 0ull) { t0 = now; return false; }
  return now - t0 > g_wait_ticks;
}

// Schedule perturbation (tools/inject_delay.py; builds with -DLIODOM_INJECT_DELAY only — the product library carries none of it): every
// hand-off between kernels of a handle — pose / prediction granules, done counts and flags, the verdict, the pipe flags, the solve's
// exchanges, the appenders' pose — delays its publisher before the store and its waiter after a successful wait by a pseudo-random
// 0 .. 20 us (three calls in four: none), so that the orders natural timing never produces are exercised.  The faults round 5 found
// in these protocols (three races, one memory fault) were all found by waiting for natural timing to hit them.
#if defined(LIODOM_INJECT_DELAY)
__device__ unsigned int g_inject_seed = 0u;       // 0: no delays (liodom_debug_set_inject_seed)
__device__ __forceinline__ void inject_delay(unsigned int site) {
  const unsigned int seed = g_inject_seed;
  if (!seed) return;
  unsigned int h = seed ^ (site * 0x9E3779B1u) ^ ((unsigned int)blockIdx.x * 0x85EBCA6Bu) ^ ((unsigned int)blockIdx.y * 0xC2B2AE35u) ^ (unsigned int)wall_clock64();
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  h = __builtin_amdgcn_readfirstlane(h);          // (one decision per wave: s_sleep is a wave instruction)
  if ((h & 3u) != 0u) return;
  const unsigned int n = (h >> 2) % 48u;          // x s_sleep 16 (1024 cycles, ~0.43 us): 0 .. 20 us
  for (unsigned int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(16);
}
#define INJECT_DELAY(site) inject_delay(site)
#else
#define INJECT_DELAY(site) do { } while (0)
#endif

// ---- the kernels, by stage (one translation unit; the order matters: later parts use helpers of earlier ones) ----
#include "kernels_extract.h"
#include "kernels_sync.h"
#include "kernels_compact.h"
#include "kernels_knn.h"
#include "kernels_knn8.h"
#include "kernels_lm.h"
#include "kernels_rebuild.h"
#include "kernels_filter.h"

}  // namespace liodom_dev
