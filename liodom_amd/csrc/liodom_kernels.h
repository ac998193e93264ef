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

constexpr int kWave = 64;
constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr int kLmThreads = 512;           // k_lm_solve: 8 waves, two per SIMD, all evaluate residual blocks
constexpr int kLmEvalThreads = kLmThreads;
constexpr int kLmCtl = kLmThreads - 64;   // lane 0 of the last wave also runs the trust-region logic; waves 0..6 prepare (compaction, register cache) meanwhile
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
  int32_t reb_pad;
  double pred_odom[12];   // early_rebuild: the prediction the scan started from, snapshot taken by the scan's first kNN launch: st.odom moves
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
  int* ring_npoints;        // [S][H]
  double* ring_c;           // [S][max_points] smoothness per ring-sorted point: debug dump (debug & 1) and generic-path scratch
  unsigned char* ring_picked;  // [S][max_points] picked_ marks of the generic path
  float4* edges;            // [kEdgeBufs][S][edge_cap] dense
  int4* edges_meta;         // [kEdgeBufs][S][edge_cap] (ring, idx_in_ring, src, 0)
  float4* corr_a;           // [S][edge_cap]  xyz of NN0, w = valid
  float4* corr_b;           // [S][edge_cap]  xyz of NN1
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
  unsigned char* corr_mask; // [S][2][knn_blocks] bit q: query q of that k_knn workgroup has an accepted correspondence
  int knn_partials;         // k_knn also evaluates every accepted block at the solve's start pose and leaves per-workgroup sums (handles with < 16 streams)
  int knn_queries;          // queries per k_knn workgroup (8, or 4 for handles with >= 16 streams)
  unsigned int* cell_pad;   // [2 S][table_size] early_rebuild: room reserved in the cell for points of the new frame; after the allocation: end of the cell's range
  int used_cap;             // used_cells entries per table (map_cap; early_rebuild: + 8 edge_cap for cells only the padding touches)
  int sorted_cap;           // sorted_pts entries per table (early_rebuild: map_cap + 8 edge_cap of padding + edge_cap of overflow list; else map_cap)
  int ovf_base;             // first entry of the overflow list inside a table's sorted_pts
  float rebuild_delta;      // early_rebuild: a new-frame point may move this far (per axis) between the prediction and the solved pose and still land in a padded cell
  float4* knn_nn;           // [S][edge_cap][5] lock-step batches: the five neighbours of every query (w: found flag, index of NN0, NN1) for k_line_gate
  unsigned int* pipe_flags; // [kEdgePipeBufs + 1] pipelined replay without cross-stream events: [b] = sequence number of the extraction whose edges
                            // are complete in edge buffer b; [kEdgePipeBufs] = number of the last odometry (of this handle) that has completed entirely
  unsigned long long* pose_xch;   // [S][32] early_rebuild: solved pose handed to the workgroups that append the new frame (tagged 8-byte granules)
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
  unsigned long long* pose_xch0;   // [S][kOvReplicas][512] the first solve's result (odom[12], q[4], t[3]) as 38 tagged granules, replicated over memory channels
  unsigned int* knn_done;          // [S][knn_grid] sequence number of the latest overlapped second pass workgroup b has completed
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

// =============================================================================================
// Ring split = stable counting sort of the scan by ring id (the reference appends every valid
// point to its ring's cloud in input order, feature_extractor.cc:115-175).
//
// k_classify      512 threads, one tile of 2048 consecutive points per workgroup: coalesced 16-B
//                 loads, isValidPoint + elevation bin in FP64, one id byte per point, and the
//                 tile's ring histogram (LDS atomics) -> tile_hist[tile][ring].
// k_ring_scatter  same tiling.  Offsets of (tile, ring) = ring start + column prefix of tile_hist
//                 (every workgroup sums the small table itself: no separate scan launch).  The
//                 stable rank inside the tile comes from 64-bit lane masks per (wave chunk, ring)
//                 built with ds_or_b64: rank = popc(mask & lanes_below) + DPP prefix over the 32
//                 chunks.  Points are re-read coalesced and written to their sorted position, so
//                 every ring is contiguous for k_ring_extract (no H-fold id scan, no strided
//                 gathers: 8x less fabric traffic than the first version, profiles/r01_c_*).
// =============================================================================================
constexpr int kTilePts = 2048;
constexpr int kTileThreads = 512;
constexpr int kTileChunks = kTilePts / 64;   // 32

__global__ __launch_bounds__(kTileThreads) void k_classify(DevView v, int s0, const float4* __restrict__ in,
                                                           size_t in_stride, int n, int height, int width) {
  __shared__ int hist[256];
  const int s = s0 + blockIdx.y;
  const int tile = blockIdx.x;
  const int H = v.scan_lines;
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  __syncthreads();
  float4 p[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = tile * kTilePts + j * kTileThreads + threadIdx.x;
    if (i < n) p[j] = in[(size_t)blockIdx.y * in_stride + i];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = tile * kTilePts + j * kTileThreads + threadIdx.x;
    unsigned char id = 0xFF;
    if (i < n) {
      // Fast decision in float for the points that are nowhere near a decision boundary (99.9 %): the FP64 sqrt + atan
      // of the reference's expressions (~350 instructions per point) made this kernel FP64-bound, not bandwidth-bound.
      // The float range / elevation angle are within 1e-4 relative / 1e-5 degrees of the double values, so a point whose
      // float range is further than 1e-4 (relative) from both range limits and whose ring is the same at angle -+ 1e-4
      // degrees gets exactly the reference's verdict; every other point takes the reference's FP64 expressions below.
      bool sure = false;
      int r_fast = -1;
      if (v.lidar_type == 0) {
        const float px = p[j].x, py = p[j].y, pz = p[j].z;
        const bool fin = (px - px) == 0.f && (py - py) == 0.f && (pz - pz) == 0.f;
        if (!fin) {
          sure = true;                                   // isValidPoint: not finite (:89-92)
        } else {
          const float df = sqrtf(px * px + py * py);
          const float lo = (float)v.min_range, hi = (float)v.max_range;
          const bool range_sure = fabsf(df - lo) > 1e-4f * lo + 1e-6f && fabsf(df - hi) > 1e-4f * hi + 1e-6f && df < 1e18f;
          if (range_sure && (df < lo || df > hi)) {
            sure = true;                                 // out of range (:96-97)
          } else if (range_sure) {
            const float a = atanf(pz / df) * 57.29577951308232f;
            const int r0 = velodyne_ring_from_angle((double)(a - 1e-4f), H), r1 = velodyne_ring_from_angle((double)(a + 1e-4f), H);
            sure = r0 == r1;
            r_fast = r0;
          }
        }
      }
      double dist;
      if (sure) {
        if (r_fast >= 0) { id = (unsigned char)r_fast; atomicAdd(&hist[r_fast], 1); }
      } else if (valid_point((double)p[j].x, (double)p[j].y, (double)p[j].z, v.min_range, v.max_range, &dist)) {
        int r;
        if (v.lidar_type == 0) {
          r = velodyne_ring((double)p[j].z, dist, H);
        } else {
          r = (width > 0) ? i / width : -1;     // ring = row (feature_extractor.cc:160-173)
          if (r >= H || r >= height) r = -1;
        }
        if (r >= 0) { id = (unsigned char)r; atomicAdd(&hist[r], 1); }
      }
      v.ring_id[(size_t)s * v.ring_id_stride + i] = id;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < H)
    v.tile_hist[((size_t)s * v.tile_cap + tile) * H + threadIdx.x] = (unsigned short)hist[threadIdx.x];
}

// Row stride of the (chunk, ring) tables in LDS: odd, so that the per-ring prefix pass (32 lanes =
// 32 chunks of one ring) does not land all its 8-byte reads on one bank pair.
__host__ __device__ __forceinline__ int ring_scatter_stride(int H) { return H | 1; }
// LDS: phase A = lane masks [32][Hp] u64 + chunk prefixes [32][Hp] u16; phase B reuses the same
// bytes as the staging tile {float4 point, int dst, int src} x 2048; then rbase / lofs / wtot.
__host__ __device__ __forceinline__ size_t ring_scatter_stage_bytes(int H) {
  const int Hp = ring_scatter_stride(H);
  const size_t a = (size_t)kTileChunks * Hp * 8 + (size_t)((kTileChunks * Hp * 2 + 15) & ~15);
  const size_t b = (size_t)kTilePts * 24;
  return a > b ? a : b;
}
__host__ __device__ __forceinline__ size_t ring_scatter_lds_bytes(int H) {
  return ring_scatter_stage_bytes(H) + (size_t)(2 * H + 2 * 16) * 4;
}

// staging-slot swizzles of k_ring_scatter (bijections on [0, 2048)): 16-byte elements have 16 bank groups (low 4 bits of
// the slot), 4-byte elements 64 banks (low 6 bits); the XOR term is constant over an aligned run of 32 / 64 slots
__device__ __forceinline__ int stage_swz16(int e) { return e ^ ((e >> 5) & 15); }
__device__ __forceinline__ int stage_swz4(int e) { return e ^ ((e >> 6) & 63); }

__global__ __launch_bounds__(kTileThreads) void k_ring_scatter(DevView v, int s0, const float4* __restrict__ in,
                                                               size_t in_stride, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int s = s0 + blockIdx.y;
  const int tile = blockIdx.x, ntiles = gridDim.x;
  const int H = v.scan_lines;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hp = ring_scatter_stride(H);
  unsigned long long* wmask = reinterpret_cast<unsigned long long*>(smem);          // [32][Hp]   (phase A)
  unsigned short* cbase = reinterpret_cast<unsigned short*>(wmask + kTileChunks * Hp);  // [32][Hp]   (phase A)
  float4* spts = reinterpret_cast<float4*>(smem);                                   // [2048]     (phase B, same bytes)
  int* sdst = reinterpret_cast<int*>(smem + (size_t)kTilePts * 16);                 // [2048]
  int* ssrc = sdst + kTilePts;                                                      // [2048]
  int* rbase = reinterpret_cast<int*>(smem + ring_scatter_stage_bytes(H));          // [H] ring start + tile prefix
  int* lofs = rbase + H;                                                            // [H] first staging slot of the ring
  int* wtot = lofs + H;                                                             // [8] ring totals per wave
  int* wloc = wtot + 16;                                                            // [8] this tile's counts per wave
  for (int k = tid; k < kTileChunks * Hp; k += kTileThreads) wmask[k] = 0ull;
  // column prefix / totals of the histogram table for "my" ring (thread r < H); 8 loads in flight
  int pre = 0, tot = 0, mine = 0;
  if (tid < H) {
    const unsigned short* th = v.tile_hist + (size_t)s * v.tile_cap * H + tid;
    for (int t0 = 0; t0 < ntiles; t0 += 8) {
      int c[8];
#pragma unroll
      for (int u = 0; u < 8; u++) c[u] = (t0 + u < ntiles) ? (int)th[(size_t)(t0 + u) * H] : 0;
#pragma unroll
      for (int u = 0; u < 8; u++) { tot += c[u]; if (t0 + u < tile) pre += c[u]; if (t0 + u == tile) mine = c[u]; }
    }
  }
  // exclusive scans over the (<= 254) rings: ring totals -> ring starts; this tile's counts -> staging offsets
  const int incl = wave_incl_scan_i32(tot);
  const int incl_l = wave_incl_scan_i32(mine);
  if (lane == 63) { wtot[wave] = incl; wloc[wave] = incl_l; }
  __syncthreads();
  {
    int base = 0, base_l = 0;
    for (int w = 0; w < wave; w++) { base += wtot[w]; base_l += wloc[w]; }
    const int rstart = base + incl - tot;
    if (tid < H) {
      rbase[tid] = rstart + pre;
      lofs[tid] = base_l + incl_l - mine;
      if (tile == 0) { v.ring_start[(size_t)s * (H + 1) + tid] = rstart; v.ring_len[(size_t)s * H + tid] = tot; }
    }
    if (tile == 0 && tid == H - 1) v.ring_start[(size_t)s * (H + 1) + H] = rstart + tot;
  }
  // lane masks per (chunk, ring)
  const unsigned char* ids = v.ring_id + (size_t)s * v.ring_id_stride;
  int id[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = tile * kTilePts + j * kTileThreads + tid;
    id[j] = (i < n) ? (int)ids[i] : 0xFF;
    if (id[j] != 0xFF) atomicOr(&wmask[(j * (kTileThreads / 64) + wave) * Hp + id[j]], 1ull << lane);
  }
  __syncthreads();
  // prefix over the 32 chunks for every ring: one half-wave per ring
  for (int r = wave * 2 + (lane >> 5); r < H; r += 2 * (kTileThreads / 64)) {
    const int c = lane & 31;
    const int cnt = __popcll(wmask[c * Hp + r]);
    const int ic = half_incl_scan_i32(cnt);
    cbase[c * Hp + r] = (unsigned short)(ic - cnt);
  }
  __syncthreads();
  // rank of every point inside (tile, ring) -> staging slot and final position
  const unsigned long long below = (1ull << lane) - 1ull;
  int slot[4], dst[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    slot[j] = -1; dst[j] = 0;
    if (id[j] != 0xFF) {
      const int chunk = j * (kTileThreads / 64) + wave;
      const int rank = (int)cbase[chunk * Hp + id[j]] + __popcll(wmask[chunk * Hp + id[j]] & below);
      slot[j] = lofs[id[j]] + rank;
      dst[j] = rbase[id[j]] + rank;
    }
  }
  __syncthreads();                  // masks / prefixes are dead: their bytes become the staging tile
  // Bank swizzle of the staging tile.  In firing order the lanes of a wave hold consecutive rings, so their staging
  // slots lie ~32 apart (a tile holds ~32 points of each ring): unswizzled, the 64 16-byte stores of a wave land on
  // one group of four banks (64-way conflict; the SQ counters had 38 % of this kernel's CU cycles in LDS bank
  // conflicts, profiles/r03_n_sq.txt).  XOR-ing the low bits of the slot with the bits above them spreads slots 32
  // apart over all banks and keeps an aligned run of consecutive slots (the read-out below) a permutation of itself.
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (slot[j] >= 0) {
      const int i = tile * kTilePts + j * kTileThreads + tid;
      spts[stage_swz16(slot[j])] = in[(size_t)blockIdx.y * in_stride + i];       // coalesced read
      sdst[stage_swz4(slot[j])] = dst[j];
      ssrc[stage_swz4(slot[j])] = i;
    }
  }
  __syncthreads();
  // Staging slots are ring-major, so consecutive lanes now write consecutive positions of a ring:
  // ~32-point (512-byte) runs instead of 64 different rings per wave store.
  float4* out = v.ring_pts + (size_t)s * v.max_points;
  int* osrc = v.ring_src + (size_t)s * v.max_points;
  const int nvalid = wloc[0] + wloc[1] + wloc[2] + wloc[3] + wloc[4] + wloc[5] + wloc[6] + wloc[7];
  for (int p = tid; p < nvalid; p += kTileThreads) {
    const int d = sdst[stage_swz4(p)];
    out[d] = spts[stage_swz16(p)];
    osrc[d] = ssrc[stage_swz4(p)];
  }
}

// =============================================================================================
// k_row_compact (lidar_type 1: organised clouds, ring = row, feature_extractor.cc:158-175): the ring split needs no
// sort — row r of the input IS ring r once its invalid points are dropped.  One workgroup per (row, stream): coalesced
// 16-B loads of the row, isValidPoint, stable compaction (wave ballots + a prefix over the (round, wave) counts) into
// the row's own segment of the ring-sorted copy (ring_start = row * width: fixed, nothing to count first).  Replaces
// k_classify + k_ring_scatter for these clouds: 36 N bytes of traffic instead of 54 N, no id bytes, no histograms.
// =============================================================================================
constexpr int kRowThreads = 512;
constexpr int kRowRounds = 4;              // columns per thread in flight (rows of up to 2048 points in one sweep)
__global__ __launch_bounds__(kRowThreads) void k_row_compact(DevView v, int s0, const float4* __restrict__ in, size_t in_stride,
                                                             int n, int height, int width) {
  __shared__ int s_cnt[kRowRounds][kRowThreads / 64];  // [round of the sweep][wave] valid points
  __shared__ int s_base;
  const int s = s0 + blockIdx.y, row = blockIdx.x, H = v.scan_lines;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4* src = in + (size_t)blockIdx.y * in_stride;
  float4* out = v.ring_pts + (size_t)s * v.max_points + (size_t)row * width;
  int* osrc = v.ring_src + (size_t)s * v.max_points + (size_t)row * width;
  if (tid == 0) { s_base = 0; v.ring_start[(size_t)s * (H + 1) + row] = row * width; if (row == H - 1) v.ring_start[(size_t)s * (H + 1) + H] = H * width; }
  int total = 0;
  if (row < height && (size_t)(row + 1) * (size_t)width <= (size_t)v.max_points) {
    for (int c0 = 0; c0 < width; c0 += kRowRounds * kRowThreads) {
      float4 p[kRowRounds];
      bool ok[kRowRounds];
      unsigned long long mask[kRowRounds];
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        const int c = c0 + j * kRowThreads + tid;
        const long long i = (long long)row * width + c;
        ok[j] = c < width && i < (long long)n;
        if (ok[j]) p[j] = src[i];
      }
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        double dist;
        ok[j] = ok[j] && valid_point((double)p[j].x, (double)p[j].y, (double)p[j].z, v.min_range, v.max_range, &dist);   // :162-166
        mask[j] = __ballot(ok[j]);
        if (lane == 0) s_cnt[j][wave] = __popcll(mask[j]);
      }
      __syncthreads();
      // exclusive prefix over the (round, wave) counts of this sweep, in column order
      int pre[kRowRounds] = {0, 0, 0, 0};
      int run = s_base;
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
#pragma unroll
        for (int w = 0; w < kRowThreads / 64; w++) { if (w == wave) pre[j] = run; run += s_cnt[j][w]; }
      }
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        if (ok[j]) {
          const int pos = pre[j] + __popcll(mask[j] & ((1ull << lane) - 1ull));
          out[pos] = p[j];
          osrc[pos] = row * width + c0 + j * kRowThreads + tid;
        }
      }
      __syncthreads();
      if (tid == 0) s_base = run;
      total = run;
      __syncthreads();
    }
  }
  if (tid == 0) v.ring_len[(size_t)s * H + row] = total;
}

// =============================================================================================
// k_ring_extract: one workgroup per (ring, stream), ONE 16-LANE DPP ROW PER REGION (four regions per
// wave, ceil(R / 4) waves per workgroup: 128 threads for the default 8 regions).  Nothing per point is
// staged in LDS.
//   keys     lane l of a row owns the 16 consecutive items 16 l .. 16 l + 15 of its region (regions of up
//            to 256 items) and loads their 26 points straight from the ring-sorted copy (contiguous,
//            L1/L2 resident).  Smoothness in registers exactly as the reference evaluates it (float 11-tap
//            sums, squares in double, feature_extractor.cc:196-229), kept as ONE 32-bit key per item: the
//            float image of the double (monotone: float(c1) > float(c2) implies c1 > c2), 0 for items
//            below the 0.1 threshold (they can never be picked: the sorted walk breaks at the first of
//            them, :270).  From the same registers: the "continuity" bit of every owned point (squared gap
//            to its predecessor <= 0.05, :281-291,297-307), OR-ed into an LDS bit array (1 bit per
//            point), so the +-5 suppression extent of any pick is a bit scan.
//   select   per region: repeat { row argmax of the keys (4 DPP steps; lowest ring index on ties by a second
//            row reduction); stop when nothing is left or after epr + 1 picks; zero the keys of the pick's
//            +-5 neighbourhood as far as the continuity bits reach }.  Only when two items of a region
//            share the maximal float image are their doubles recomputed and compared exactly, so the pick
//            is always the reference's: largest double, lowest index on ties.  The four rows of a wave run
//            their regions side by side: a pick costs ~1/4 of the wave instructions of a 64-lane argmax,
//            which is what bounds the kernel on lock-step batches (VALU issue).
//   carry    the reference walks regions in order because suppression carries across region boundaries
//            (SURVEY.md §0 fact 4).  Here all regions run speculatively assuming no carry; the in-order walk
//            is the fixed point of "region r = select(region r | forward spill of region r-1)", and a spill
//            reaches at most the first 5 items of the next region (a 5-bit mask), so every region whose
//            incoming mask changed AND hits one of its picks is re-run with that mask until no mask changes
//            (marking an item a run never picked cannot change that run).  Region 0 is final after the
//            speculative pass, region r after at most r more rounds; typically none or one.
//   emit     all threads write the picks in region order.
// Rings / parameter sets outside this shape (regions longer than 256 items or shorter than a spill, more
// than 64 regions, rings longer than kGapBitsCap) take the generic path: curvature and marks in global
// scratch, regions walked in order by one wave — any ring length, no capacity flag.
// =============================================================================================
constexpr int kExLPR = 16;               // lanes per region (one DPP row)
constexpr int kExIPL = 16;               // items per lane -> regions of up to 256 items ...
constexpr int kExIPLBig = 24;            // ... or 384 (the last region takes the remainder of the split: Ouster 2048 / 8 -> 260); the host
                                         // picks the instance from the expected ring width, longer regions take the generic path
constexpr int kGapBitsCap = 16384;       // points per ring covered by the LDS continuity bits (2 KB)
constexpr int kExMaxRegions = 64;

__host__ __device__ __forceinline__ int ring_extract_threads(int regions) {
  const int waves = (regions + 3) / 4;
  return 64 * (waves < 1 ? 1 : (waves > 16 ? 16 : waves));
}
__host__ __device__ __forceinline__ size_t ring_extract_lds_bytes(int slots, int regions) {
  size_t b = (size_t)(kGapBitsCap / 32 + 4) * 4;          // continuity bits + pad words
  b += (size_t)slots * 4;                                 // pick_idx
  b += (size_t)((slots + 15) / 16 * 16);                  // pick_nfnb
  b += (size_t)regions * 4 + 2 * kExMaxRegions * 4 + 64;  // region_cnt, masks, flags
  return (b + 15) / 16 * 16;
}

__device__ __forceinline__ unsigned int row_max_u32(unsigned int v) {
  unsigned int o;
  o = (unsigned int)dpp_i32<DPP_XOR1>((int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<DPP_XOR2>((int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<DPP_HALF_MIRROR>((int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<DPP_MIRROR>((int)v); v = o > v ? o : v;
  return v;   // uniform over the 16-lane row
}
__device__ __forceinline__ int row_min_i32(int v) {
  int o;
  o = dpp_i32<DPP_XOR1>(v); v = o < v ? o : v;
  o = dpp_i32<DPP_XOR2>(v); v = o < v ? o : v;
  o = dpp_i32<DPP_HALF_MIRROR>(v); v = o < v ? o : v;
  o = dpp_i32<DPP_MIRROR>(v); v = o < v ? o : v;
  return v;
}
// the 16 ballot bits of this lane's row, != 0 iff `p` holds on any lane of the row
__device__ __forceinline__ bool row_any(bool p, int lane) {
  return ((__ballot(p) >> (lane & 48)) & 0xFFFFull) != 0ull;
}

// +-5 suppression extent of pick j from the continuity bits (bit k: gap(k-1, k) <= 0.05):
// forward marks l = 1..5 stop at the first k = j + l whose bit is clear (:280-294), backward marks at the
// first k = j - l + 1 whose bit is clear (:296-310).  Returns nf | nb << 4.
__device__ __forceinline__ int suppression_extent_bits(const unsigned int* gb, int j) {
  const int k0 = j - 4;                                   // >= 1 for any pick (j >= 5)
  const int w = k0 >> 5, sh = k0 & 31;
  const unsigned long long win = ((((unsigned long long)gb[w + 1]) << 32) | gb[w]) >> sh;   // bit i <-> k = k0 + i
  const unsigned int back = (unsigned int)win & 31u;      // k = j-4 .. j   (i = 0..4)
  const unsigned int fwd = (unsigned int)(win >> 5) & 31u;   // k = j+1 .. j+5
  const unsigned int invf = ~fwd & 31u, invb = ~back & 31u;
  const int nf = invf ? (__ffs(invf) - 1) : 5;
  const int nb = invb ? (4 - (31 - __clz(invb))) : 5;     // highest clear bit p: k = j-4+p breaks, nb = 4 - p
  return nf | (nb << 4);
}
// The same test on the points themselves (generic path).
__device__ __forceinline__ int suppression_extent_pts(const float4* rp, int j) {
  int nf = 5, nb = 5;
  for (int l = 1; l <= 5; l++) {
    const float4 a = rp[j + l], b = rp[j + l - 1];
    if (gap_sq3(a.x, a.y, a.z, b.x, b.y, b.z) > 0.05) { nf = l - 1; break; }
  }
  for (int l = 1; l <= 5; l++) {
    const float4 a = rp[j - l], b = rp[j - l + 1];
    if (gap_sq3(a.x, a.y, a.z, b.x, b.y, b.z) > 0.05) { nb = l - 1; break; }
  }
  return nf | (nb << 4);
}

// Smoothness of ring point j from 11 consecutive points q[0..10] = ring points j-5 .. j+5 (:196-229).
__device__ __forceinline__ double curvature_pts(const float4* q) {
  const double dx = stencil_sum(q[0].x, q[1].x, q[2].x, q[3].x, q[4].x, q[5].x, q[6].x, q[7].x, q[8].x, q[9].x, q[10].x);
  const double dy = stencil_sum(q[0].y, q[1].y, q[2].y, q[3].y, q[4].y, q[5].y, q[6].y, q[7].y, q[8].y, q[9].y, q[10].y);
  const double dz = stencil_sum(q[0].z, q[1].z, q[2].z, q[3].z, q[4].z, q[5].z, q[6].z, q[7].z, q[8].z, q[9].z, q[10].z);
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ double curvature_at(const float4* __restrict__ rp, int j) {
  float4 q[11];
#pragma unroll
  for (int i = 0; i < 11; i++) q[i] = rp[j - 5 + i];
  return curvature_pts(q);
}

// Keys of the 16 items owned by this lane (region-array indices k0 .. k0 + 15, ring indices + 5): float image
// of the smoothness, 0 = unavailable (outside the region, below 0.1, picked or suppressed).  Also returns the
// continuity bits of the owned points k = k0 + 5 + i (bit i).  Loads are unconditional (no branch per load, all
// in flight together, one base address + immediate offsets): two batches of 18 points for 8 items each.
typedef float f3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ double curvature_f3(const f3v* q) {
  const double dx = stencil_sum(q[0].x, q[1].x, q[2].x, q[3].x, q[4].x, q[5].x, q[6].x, q[7].x, q[8].x, q[9].x, q[10].x);
  const double dy = stencil_sum(q[0].y, q[1].y, q[2].y, q[3].y, q[4].y, q[5].y, q[6].y, q[7].y, q[8].y, q[9].y, q[10].y);
  const double dz = stencil_sum(q[0].z, q[1].z, q[2].z, q[3].z, q[4].z, q[5].z, q[6].z, q[7].z, q[8].z, q[9].z, q[10].z);
  return dx * dx + dy * dy + dz * dz;
}
template <int IPL>
__device__ __forceinline__ unsigned int region_keys_load(unsigned int (&kf)[IPL], const float4* __restrict__ rp, int nr, int k0,
                                                         int n_own, double* curv_out) {
  unsigned int gbits = 0;
  (void)nr;
#pragma unroll
  for (int c0 = 0; c0 < IPL; c0 += 8) {
    f3v q[18];                                              // x y z only: 12-byte loads, 54 registers per batch
#pragma unroll
    for (int i = 0; i < 18; i++) {
      // ring index k0 + c0 + i; item c0 + t uses q[t .. t + 10].  Lanes at the end of the ring read up to 26
      // points past it (the next ring, or the padding the allocation carries): those items are not owned.
      q[i] = *reinterpret_cast<const f3v*>(rp + k0 + c0 + i);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const bool own = c0 + t < n_own;
      const double c = curvature_f3(q + t);
      kf[c0 + t] = (own && !(c < 0.1)) ? (unsigned int)__float_as_int((float)c) : 0u;      // :270 threshold folded in
      if (curv_out && own) curv_out[k0 + 5 + c0 + t] = c;
      const bool ok = !(gap_sq3(q[t + 5].x, q[t + 5].y, q[t + 5].z, q[t + 4].x, q[t + 4].y, q[t + 4].z) > 0.05);
      gbits |= (own && ok) ? (1u << (c0 + t)) : 0u;
    }
    if (c0 + 8 < IPL) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }   // second batch of loads after the first batch's arithmetic
  }
  return gbits;
}

// Greedy selection of one region per 16-lane row; rows whose `active` is false idle.  Keys are consumed
// (picked / suppressed items zeroed).  premask: bit o set = item rs + o (o < 5) was suppressed by the previous
// region's picks.  Returns the number of picks (uniform over the row).
template <int IPL>
__device__ __forceinline__ int select_region_row(unsigned int (&kf)[IPL], const float4* __restrict__ rp, const unsigned int* gb,
                                                 bool active, int rs, int k0, int epr, int lane, int premask, int* out_idx,
                                                 unsigned char* out_nfnb) {
  const int j0 = k0 + 5;
  const int rl = lane & (kExLPR - 1);
  if (premask) {
#pragma unroll
    for (int i = 0; i < IPL; i++) {
      const int o = k0 + i - rs;
      if (o < 5 && ((premask >> o) & 1)) kf[i] = 0;
    }
  }
  int picks = 0;
  while (__ballot(active) != 0ull) {
    unsigned int bf = 0;
#pragma unroll
    for (int i = 0; i < IPL; i++) bf = kf[i] > bf ? kf[i] : bf;
    const unsigned int m32 = row_max_u32(active ? bf : 0u);
    active = active && m32 != 0u && picks <= epr;                  // nothing left above 0.1, or epr + 1 picks made (:270)
    unsigned int eqm = 0;                                          // bit i: item i carries the maximal float image
#pragma unroll
    for (int i = 0; i < IPL; i++) eqm |= (kf[i] == m32) ? (1u << i) : 0u;
    const bool has = active && eqm != 0u;
    const int first = __ffs(eqm) - 1;
    int j = row_min_i32(has ? j0 + first : 0x7fffffff);            // lowest ring index among the maximal float images
    const bool tie = row_any(has && ((eqm & (eqm - 1u)) != 0u || j0 + first != j), lane);
    if (__ballot(tie) != 0ull) {
      // several items share the maximal float image: their exact doubles decide (recomputed from the points)
      unsigned long long bk = 0;
      int bj = 0x7fffffff;
      unsigned int rem = (tie && has) ? eqm : 0u;
#pragma unroll 1
      while (rem) {
        const int i = __ffs(rem) - 1;                              // ascending i: the lowest index wins among equals
        rem &= rem - 1u;
        const unsigned long long ck = (unsigned long long)__double_as_longlong(curvature_at(rp, j0 + i));
        if (ck > bk) { bk = ck; bj = j0 + i; }
      }
      const unsigned long long m64 = row_max_u64(bk);
      const int jt = row_min_i32((tie && has && bk == m64) ? bj : 0x7fffffff);
      j = tie ? jt : j;
    }
    int ext = 0;
    if (active) ext = suppression_extent_bits(gb, j);
    const int nf = ext & 15, nb = ext >> 4;
    if (active && rl == 0) { out_idx[picks] = j; out_nfnb[picks] = (unsigned char)ext; }
    const unsigned int span = (unsigned int)(nf + nb);
    const int lo = j - nb - j0;
#pragma unroll
    for (int i = 0; i < IPL; i++) {
      if (active && (unsigned int)(i - lo) <= span) kf[i] = 0;     // the pick and its marked neighbours (:277,293,309)
    }
    picks += active ? 1 : 0;
  }
  return picks;
}

// Generic in-order selection of one region on global scratch (any region length).  One wave.
__device__ int select_region_generic(const double* c, const float4* rp, volatile unsigned char* vpicked, int rs, int re,
                                     int epr, int lane, int* out_idx, unsigned char* out_nfnb) {
  int picks = 0;
  while (true) {
    unsigned long long bkey = 0;
    int bidx = 0x7fffffff;
    for (int k = rs + lane; k < re; k += 64) {
      const int j = k + 5;
      if (!vpicked[j]) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(c[j]);
        if (bidx == 0x7fffffff || key > bkey) { bkey = key; bidx = j; }
      }
    }
    const unsigned long long has = __ballot(bidx != 0x7fffffff);
    if (!has) break;                                               // every item already picked
    const unsigned long long m = wave_max_u64(bidx != 0x7fffffff ? bkey : 0ull);
    const double best = __longlong_as_double((long long)m);
    if (best < 0.1 || picks > epr) break;                          // :270
    const int j = wave_min_i32((bidx != 0x7fffffff && bkey == m) ? bidx : 0x7fffffff);   // ties: lowest index
    const int ext = suppression_extent_pts(rp, j);
    const int nf = ext & 15, nb = ext >> 4;
    if (lane == 0) { out_idx[picks] = j; out_nfnb[picks] = (unsigned char)ext; vpicked[j] = 1; }   // :275-277
    if (lane >= 1 && lane <= nf) vpicked[j + lane] = 1;            // :293
    if (lane >= 9 && lane <= 8 + nb) vpicked[j - (lane - 8)] = 1;  // :309
    picks++;                                                       // :276
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
  }
  return picks;
}

// The register path of k_ring_extract for one ring (keys, continuity bits, speculative selection, carry fixed
// point); IPL items per lane.
template <int IPL>
__device__ __forceinline__ void ring_select_rows(const DevView& v, const float4* __restrict__ rpts, double* rc, bool dump, int nr,
                                                 int total, int sector, int R, int epr, int ppr, unsigned int* gb, int* pick_idx,
                                                 unsigned char* pick_nfnb, int* region_cnt, int* used_mask, int* new_mask, int* flags,
                                                 bool dbgb, int& dbg_rounds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
    const int row = lane >> 4, rl = lane & 15;
    const int reg = wave * 4 + row;                     // this row's region
    const bool rvalid = reg < R;
    const int rs = sector * (rvalid ? reg : 0);
    const int re = !rvalid ? rs : ((reg == R - 1) ? total : sector * (reg + 1));   // :242-247
    const int k0 = rs + rl * IPL;                    // region-array index of this lane's first item (ring index + 5)
    const int n_own = re - k0;                          // owned items inside the region (<= 0: none)
    for (int w = tid; w < ((nr + 31) >> 5) + 3; w += nthreads) gb[w] = 0u;
    if (tid < R) { used_mask[tid] = 0; new_mask[tid] = 0; }
    unsigned int kf[IPL], kf0[IPL];                    // kf0: the keys as loaded (a carry re-run starts from them again)
    unsigned int gbits = region_keys_load<IPL>(kf, rpts, nr, k0, n_own, dump ? rc : nullptr);
#pragma unroll
    for (int i = 0; i < IPL; i++) kf0[i] = kf[i];
    // the ring's first / last points are owned by no item: their continuity bits (k = 1..4, nr-5..nr-1) separately
    unsigned int edge_bit = 0;
    int edge_k = 0;
    if (tid < 9) {
      edge_k = tid < 4 ? tid + 1 : nr - 9 + tid;
      const float4 a = rpts[edge_k], b = rpts[edge_k - 1];
      edge_bit = !(gap_sq3(a.x, a.y, a.z, b.x, b.y, b.z) > 0.05) ? 1u : 0u;
    }
    __syncthreads();                                    // bit array zeroed
    if (gbits) {
      const int kb = k0 + 5;                            // ring index of bit 0
      const unsigned long long sh = (unsigned long long)gbits << (kb & 31);
      atomicOr(&gb[kb >> 5], (unsigned int)sh);
      if ((unsigned int)(sh >> 32)) atomicOr(&gb[(kb >> 5) + 1], (unsigned int)(sh >> 32));
    }
    if (edge_bit) atomicOr(&gb[edge_k >> 5], 1u << (edge_k & 31));
    __syncthreads();
    DBG_STAMP(v, dbgb, 0, 2);
    // ---- speculative selection, all regions side by side ----
    {
      const int cntp = select_region_row<IPL>(kf, rpts, gb, rvalid && re > rs, rs, k0, epr, lane, 0, pick_idx + (rvalid ? reg : 0) * ppr,
                                         pick_nfnb + (rvalid ? reg : 0) * ppr);
      if (rvalid && rl == 0) region_cnt[reg] = cntp;
    }
    __syncthreads();
    DBG_STAMP(v, dbgb, 0, 5);
    // ---- carry resolution: fixed point over the 5-bit spill masks ----
    for (int round = 0; round <= R; round++) {
      if (rvalid && reg + 1 < R) {                       // spill of region reg into region reg + 1
        const int end_j = sector * (reg + 1) + 5;        // first ring index of region reg + 1
        const int cntp = region_cnt[reg];
        int m = 0;
        for (int k = rl; k < cntp; k += kExLPR) {
          const int j = pick_idx[reg * ppr + k];
          const int nf = pick_nfnb[reg * ppr + k] & 15;
          for (int l = 1; l <= nf; l++) if (j + l >= end_j) m |= 1 << (j + l - end_j);
        }
        m |= dpp_i32<DPP_XOR1>(m); m |= dpp_i32<DPP_XOR2>(m); m |= dpp_i32<DPP_HALF_MIRROR>(m); m |= dpp_i32<DPP_MIRROR>(m);
        if (rl == 0) new_mask[reg + 1] = m;
      }
      if (tid == 0) flags[0] = 0;
      __syncthreads();
      bool need = false;
      int m = 0;
      if (rvalid) {
        m = new_mask[reg];
        const int um = used_mask[reg];
        if (m != um) {
          need = true;
          if (um == 0) {      // picks of an unmarked run stay valid unless the mask hits one of them
            const int cntp = region_cnt[reg];
            bool hit = false;
            for (int k = rl; k < cntp; k += kExLPR) {
              const int o = pick_idx[reg * ppr + k] - (rs + 5);
              hit = hit || (o < 5 && ((m >> o) & 1));
            }
            need = row_any(hit, lane);
            if (!need && rl == 0) used_mask[reg] = 0;    // still the unmarked run's picks, valid for this mask too
          }
        }
      }
      if (__ballot(need) != 0ull) {                      // (wave-uniform) some row of this wave re-runs its region
        // (the keys are restored from the register copy: reloading the 26 points per lane and recomputing 16 FP64
        //  smoothness values cost 3.7 us per round — more than the re-run itself on most rings)
#pragma unroll
        for (int i = 0; i < IPL; i++) kf[i] = kf0[i];
        const int cntp = select_region_row<IPL>(kf, rpts, gb, need, rs, k0, epr, lane, m, pick_idx + (rvalid ? reg : 0) * ppr,
                                           pick_nfnb + (rvalid ? reg : 0) * ppr);
        if (need && rl == 0) { region_cnt[reg] = cntp; used_mask[reg] = m; flags[0] = 1; }
      }
      __syncthreads();
      if (flags[0] == 0) break;
      dbg_rounds++;
      __syncthreads();
    }
    DBG_STAMP(v, dbgb, 0, 6);
}

template <int kMaxThreads, int IPL>
__global__ __launch_bounds__(kMaxThreads) void k_ring_extract(DevView v, int s0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ring = blockIdx.x;
  const int s = s0 + blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int R = v.scan_regions, epr = v.edges_per_region, slots = v.slots_per_ring;
  unsigned int* gb = reinterpret_cast<unsigned int*>(smem);                          // [kGapBitsCap / 32 + 4]
  int* pick_idx = reinterpret_cast<int*>(gb + kGapBitsCap / 32 + 4);                 // [R][epr+1]
  unsigned char* pick_nfnb = reinterpret_cast<unsigned char*>(pick_idx + slots);
  int* region_cnt = reinterpret_cast<int*>(pick_nfnb + (slots + 15) / 16 * 16);     // [R]
  int* used_mask = region_cnt + R;          // [kExMaxRegions] pre-marks of the run that produced the current picks
  int* new_mask = used_mask + kExMaxRegions;   // [kExMaxRegions] spill of the predecessor's current picks
  int* flags = new_mask + kExMaxRegions;    // [4]

  const int H = v.scan_lines;
  int* nedges_out = v.ring_nedges + (size_t)s * H + ring;
  int* npoints_out = v.ring_npoints + (size_t)s * H + ring;
  const unsigned long long t_begin = (kInstrument && (v.debug & 32)) ? wall_clock64() : 0ull;
  const bool dbgb = (ring == (((v.debug >> 8) & 0xFF) ? ((v.debug >> 8) & 0xFF) : 40) % H) && (s == 0) && (tid == 0);
  DBG_STAMP(v, dbgb, 0, 0);
  int dbg_rounds = 0;
  // ---- the ring's points are contiguous in the ring-sorted copy written by k_ring_scatter ----
  const int rbeg = v.ring_start[(size_t)s * (H + 1) + ring];
  const int nr = v.ring_len[(size_t)s * H + ring];
  const float4* rpts = v.ring_pts + (size_t)s * v.max_points + rbeg;
  const int* rsrc = v.ring_src + (size_t)s * v.max_points + rbeg;
  double* rc = v.ring_c + (size_t)s * v.max_points + rbeg;                // debug dump / generic-path scratch
  const bool dump = (v.debug & 1) != 0;
  if (tid == 0) *npoints_out = nr;
  // rings below min_points_per_scan are skipped (feature_extractor.cc:188)
  if ((long long)nr < v.min_points_per_scan || nr < 11) {
    if (tid == 0) *nedges_out = 0;
    if (dump) for (int j = tid; j < nr; j += nthreads) rc[j] = __longlong_as_double(0x7ff8000000000000ll);
    return;
  }
  const int total = nr - 10;                            // :238
  const int sector = total / R;                         // :239
  const int last_len = total - sector * (R - 1);
  const int max_len = sector > last_len ? sector : last_len;
  const int ppr = epr + 1;                              // picks per region (:270)
  const bool fast = max_len <= kExLPR * IPL && sector >= 5 && R <= kExMaxRegions && R <= 4 * (nthreads >> 6) && nr <= kGapBitsCap;
  if (dump) {
    for (int j = tid; j < 5; j += nthreads) { rc[j] = __longlong_as_double(0x7ff8000000000000ll); rc[nr - 1 - j] = rc[j]; }
  }
  if (fast) {
    ring_select_rows<IPL>(v, rpts, rc, dump, nr, total, sector, R, epr, ppr, gb, pick_idx, pick_nfnb, region_cnt, used_mask, new_mask, flags, dbgb, dbg_rounds);
  } else {
    // ---- generic path: curvature + marks in global scratch, regions in order on one wave ----
    unsigned char* picked = v.ring_picked + (size_t)s * v.max_points + rbeg;
    for (int j = 5 + tid; j < nr - 5; j += nthreads) {
      rc[j] = curvature_at(rpts, j);
      picked[j] = 0;                                                // :230
    }
    __threadfence();
    __syncthreads();
    if (wave == 0) {
      for (int reg = 0; reg < R; reg++) {
        const int rs = sector * reg;
        const int re = (reg == R - 1) ? total : sector * (reg + 1);
        int cntp = 0;
        if (re > rs) cntp = select_region_generic(rc, rpts, picked, rs, re, epr, lane, pick_idx + reg * ppr, pick_nfnb + reg * ppr);
        if (lane == 0) region_cnt[reg] = cntp;
      }
    }
    __syncthreads();
  }
  // ---- emit in region order, pick order (:275) ----
  float4* eout = v.edges_pad + ((size_t)s * H + ring) * slots;
  int2* mout = v.edges_pad_meta + ((size_t)s * H + ring) * slots;
  // one flat pass over all pick slots (region-major): slot (reg, k) goes to position
  // sum of the earlier regions' counts + k — one round of loads instead of one per region
  for (int q = tid; q < R * ppr; q += nthreads) {
    const int reg = q / ppr, k = q - reg * ppr;
    if (k < region_cnt[reg]) {
      int base = 0;
      for (int r2 = 0; r2 < reg; r2++) base += region_cnt[r2];
      const int j = pick_idx[q];
      eout[base + k] = rpts[j];                                    // :275 (XYZ + intensity unchanged)
      mout[base + k] = make_int2(j, rsrc[j]);
    }
  }
  if (tid == 0) {
    int total_picks = 0;
    for (int r2 = 0; r2 < R; r2++) total_picks += region_cnt[r2];
    *nedges_out = total_picks;
  }
  DBG_STAMP(v, dbgb, 0, 7);
  if ((kInstrument && (v.debug & 32)) && s == 0 && tid == 0 && ring < 64) { v.dbg_clk[128 + ring] = wall_clock64() - t_begin; v.dbg_clk[96 + (ring & 31)] = (unsigned long long)dbg_rounds | ((unsigned long long)*nedges_out << 8); }
}

// =============================================================================================
// Pipelined replay: the two HIP streams of a handle (extraction / odometry) depend on each other twice per scan.  As
// hipStreamWaitEvent / hipEventRecord pairs those dependencies cost ~11 us of idle odometry stream per scan (the barrier
// packets are processed when the preceding kernel retires, measured with the host far ahead as well); as flags in
// device memory they cost one early load per workgroup.  A flag is written by a kernel that follows the producer in
// stream order (so the producer's launch has ended and its writes have left the caches) and polled by thread 0 of the
// consumer's workgroups before they touch the data; a consumer that really had to wait also invalidates its caches.
// The wait is bounded (~0.3 s): a producer that cannot run beside the consumer — a profiler that serialises kernels across
// streams, e.g. rocprofv3 --pmc: use LIODOM_PIPE_FLAGS=0 there — raises LIODOM_STATUS_PIPE_TIMEOUT instead of hanging; the waiting
// workgroups then skip their work (nothing reads a half-written buffer or overwrites one still in use), the host reports the scan
// as failed (wait_pose) and the handle falls back to events.
__device__ __forceinline__ bool pipe_wait(const unsigned int* flag, unsigned int want, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  __shared__ int s_pipe_ok;
  if (threadIdx.x == 0) {
    unsigned int spins = 0;
    bool ok = true;
    while ((int)(__hip_atomic_load((gu32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > 1500000u) { atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); ok = false; break; }
    }
    if (spins) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_pipe_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return s_pipe_ok != 0;      // false: the producer never arrived — the caller must not touch the buffer (it returns)
}
// Gate in front of a scan's first k_knn launch for handles whose launch is too large to poll the flag itself (its polling
// workgroups would fill the GPU and starve the extraction they wait for): one wave waits for the extraction's flag and
// publishes that the previous odometry has completed; the launches behind it start when it retires.
__global__ void k_pipe_gate(DevView v, int s0, int eb, unsigned int wait_edges, unsigned int signal_odo) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  if (signal_odo && threadIdx.x == 0) __hip_atomic_store((gu32*)(v.pipe_flags + kEdgePipeBufs), signal_odo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wait_edges && !pipe_wait(v.pipe_flags + eb, wait_edges, &v.state[s0].status)) {
    // the launches behind the gate check the status bit of their own stream (k_knn) and skip the scan
    for (int s = (int)threadIdx.x; s < v.n_streams; s += (int)blockDim.x) atomicOr(&v.state[s].status, LIODOM_STATUS_PIPE_TIMEOUT);
  }
}
// =============================================================================================
// Overlapped second kNN pass (one-stream handles with the streamed rebuild and flags).  The odometry chain of a scan is
// kNN, solve, kNN, solve; as four launches of one HIP stream every link costs a launch boundary (~0.7 us idle), the ramp of
// the next launch (kernel arguments, state words, first loads: ~2 us of dependent round trips) and the tail of the previous
// one.  The second kNN pass depends on the first solve only through the 19 doubles of its result, and everything else it
// reads — the edge, what the first pass saved for the re-ranking, the saved candidates themselves — is known when the
// first pass has completed.  So this pass is launched on a HIP stream of its own (stream_k) right behind the first solve's
// launch and waits INSIDE the kernel, twice:
//   1. for ov_flags[s] == seq, stored by the first solve's launch when it starts (it follows the first kNN pass in stream
//      order, so that pass has completed and its writes are visible) -> the workgroups load their edges and the saved
//      candidates (two dependent round trips) while the solve runs;
//   2. for the solve's result, published as tagged granules (the data is the flag) in kOvReplicas copies 4 KiB apart, so
//      that the polling workgroups do not queue on one memory channel -> transform, re-rank, gate, partial sums.
// Every workgroup of the pass then stores seq into knn_done[s][b] (after a release fence), and the finalising solve's
// launch — which follows the first solve in stream order and therefore starts while this pass still runs — polls those
// flags in its solving workgroups before it reads the pass's results, and in the workgroups that clear the searched table
// before they touch it.  All waits are bounded (LIODOM_STATUS_PIPE_TIMEOUT, as pipe_wait); workgroups that wait never
// hold more than a third of the GPU's wave slots, and a waiting workgroup depends only on launches enqueued before its own.
// Launch order on the host: kNN(0) [stream], solve(0) [stream], kNN(1) [stream_k], solve(1) [stream].
// =============================================================================================
// Stores / loads that are visible across the XCDs without cache maintenance: agent-scope relaxed atomics go through the
// XCD's L2 to the memory side.  (The alternative — plain accesses plus release / acquire fences — costs an L2 write-back or
// invalidate per fence on a part whose eight L2s are not coherent with each other: with one per workgroup of a 352-workgroup
// launch the solve running beside it took 80 us instead of 24.)
__device__ __forceinline__ void wt_store_u32(void* p, unsigned int x) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  __hip_atomic_store((gu32*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wt_store_u64(void* p, unsigned long long x) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  __hip_atomic_store((gu64*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wt_store_f4(float4* p, const float4& x) {
  wt_store_u64(p, ((unsigned long long)__float_as_uint(x.y) << 32) | __float_as_uint(x.x));
  wt_store_u64(reinterpret_cast<char*>(p) + 8, ((unsigned long long)__float_as_uint(x.w) << 32) | __float_as_uint(x.z));
}
__device__ __forceinline__ void wt_store_u8(void* p, unsigned char x) {
  typedef __attribute__((address_space(1))) unsigned char gu8;
  __hip_atomic_store((gu8*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int coh_load_u32(const void* p) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the first solve's result leaves its workgroup: vals = odom[12], q[4], t[3] in LDS; threads 0 .. kOvReplicas * kOvGranules - 1
__device__ __forceinline__ void ov_publish_pose(const DevView& v, int s, const double* vals, unsigned int tag, int tid) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  if (tid >= kOvReplicas * kOvGranules) return;
  const int rep = tid / kOvGranules, gi = tid % kOvGranules;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(vals[gi >> 1]);
  const unsigned int word = (gi & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
  __hip_atomic_store((gu64*)(v.pose_xch0 + ((size_t)s * kOvReplicas + rep) * 512 + gi), ((unsigned long long)tag << 32) | word,
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// whole workgroup; the first wave polls replica rep until every granule carries the tag; out19: LDS.  false: gave up.
__device__ __forceinline__ bool ov_wait_pose(const DevView& v, int s, int rep, unsigned int tag, double* out19, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  __shared__ int s_ov_ok;
  const int tid = (int)threadIdx.x;
  if (tid < 64) {
    const unsigned long long* base = v.pose_xch0 + ((size_t)s * kOvReplicas + rep) * 512;
    unsigned long long g = 0;
    unsigned int spins = 0;
    bool ok;
    // (all 38 lanes poll: one round trip after the publication instead of two; what had congested the memory fabric in the
    //  first version of this pass were release / acquire fences — an L2 write-back / invalidate each —, not these loads)
    while (true) {
      if (tid < kOvGranules) g = __hip_atomic_load((gu64*)(base + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = tid >= kOvGranules || (unsigned int)(g >> 32) == tag;
      if (__all(ok)) break;
      if (++spins > 2000000u) break;
      __builtin_amdgcn_s_sleep(6);
    }
    const bool all_ok = __all(ok);
    const unsigned long long lo = __shfl(g, 2 * (tid % 19)), hi = __shfl(g, 2 * (tid % 19) + 1);
    if (tid < 19) out19[tid] = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
    if (tid == 0) { s_ov_ok = all_ok ? 1 : 0; if (!all_ok) atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); }
  }
  __syncthreads();
  return s_ov_ok != 0;
}
// whole workgroup, every exit path of an overlapped k_knn workgroup: its results are visible before the flag is
__device__ __forceinline__ void ov_signal_knn_done(const DevView& v, int s, int b, unsigned int seq) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the pass's results are write-through stores: acknowledged = visible to every XCD)
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store((gu32*)(v.knn_done + (size_t)s * v.knn_grid + b), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// whole workgroup (finalising solve's launch): every workgroup of the overlapped second pass has completed
__device__ __forceinline__ void ov_wait_knn_done(const DevView& v, int s, unsigned int seq, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  const unsigned int* f = v.knn_done + (size_t)s * v.knn_grid;
  for (int b = (int)threadIdx.x; b < v.knn_grid; b += (int)blockDim.x) {
    unsigned int spins = 0;
    while ((int)(__hip_atomic_load((gu32*)(f + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > 6000000u) { atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); break; }
    }
  }
  // (no acquire fence — an L2 invalidate per waiting workgroup: this launch started, with clean caches, before the pass wrote
  //  any of its results, and reads none of them before this point; the pass's stores are write-through)
  __syncthreads();
}

__global__ void k_set_flag(unsigned int* flag, unsigned int value) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  __hip_atomic_store((gu32*)flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// k_compact_edges: one workgroup per stream; ring-padded edges -> dense edge cloud (edge buffer
// `eb`) in the reference's output order.
// =============================================================================================
// grid (kCompactBlocks, streams): every workgroup scans the <= 256 ring counts itself (cheaper than a
// second launch) and copies its interleaved share of the edges.
constexpr int kCompactBlocks = 8;
__global__ __launch_bounds__(256) void k_compact_edges(DevView v, int s0, int eb, unsigned int wait_odo) {
  __shared__ int pre[257];
  __shared__ int cntr[256];
  // (pipelined replay) the odometry that last read edge buffer eb must have completed before it is rewritten
  if (wait_odo && !pipe_wait(v.pipe_flags + kEdgePipeBufs, wait_odo, &v.state[s0 + blockIdx.y].status)) return;
  const int s = s0 + blockIdx.y;
  const int H = v.scan_lines;
  const int* rn = v.ring_nedges + (size_t)s * H;
  {
    // exclusive prefix over the H <= 256 ring counts: DPP wave scan + 4 wave totals
    const int mine = ((int)threadIdx.x < H) ? rn[threadIdx.x] : 0;
    const int incl = wave_incl_scan_i32(mine);
    if ((threadIdx.x & 63) == 63) cntr[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += cntr[w];
    pre[threadIdx.x] = base + incl - mine;
    if (threadIdx.x == 255) pre[256] = base + incl;
    __syncthreads();
    // threads >= H contribute 0, so pre[H] already equals the total
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int acc = pre[256];
    v.state[s].n_edges_buf[eb] = acc > v.edge_cap ? v.edge_cap : acc;
  }
  const int E = pre[H] > v.edge_cap ? v.edge_cap : pre[H];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += kCompactBlocks * 256) {
    int lo = 0, hi = H;            // largest r with pre[r] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= e) lo = mid; else hi = mid; }
    const int r = lo, k = e - pre[r];
    const size_t pi = ((size_t)s * H + r) * v.slots_per_ring + k;
    const size_t eo = ((size_t)eb * v.n_streams + s) * v.edge_cap + e;
    v.edges[eo] = v.edges_pad[pi];
    const int2 m = v.edges_pad_meta[pi];
    v.edges_meta[eo] = make_int4(r, m.x, m.y, 0);
  }
}

// For liodom_odometry_step (edges supplied by the caller): set counts and reset diagnostics.
__global__ void k_set_edges(DevView v, int s0, int n_edges, int eb) {
  const int s = s0 + blockIdx.x;
  if (threadIdx.x == 0) {
    StreamState& st = v.state[s];
    st.n_edges_buf[eb] = n_edges;
    st.info.matches[0] = 0; st.info.matches[1] = 0;
  }
}

// =============================================================================================
// k_knn: 32 lanes (one half-wave) per edge, 8 edges per 256-thread workgroup (4 per 128 threads on lock-step batches).
//   lane c < 27 probes the voxel hash for neighbour cell c of the query's 1 m cell (occupancy bit, then one 16-B
//   slot load; a 27-cell search is exact for every edge that can pass the sq_dist[4] < 1.0 gate);
//   lane 27 contributes the overflow list of the streamed rebuild.  ALL candidates of the neighbourhood go through
//   one flat pass (no pruning rounds, no bound refreshes: with ~12 VALU instructions per candidate slot the rounds
//   cost more than the ~3x candidates they saved) in which every lane keeps only its two nearest candidates and
//   the distance of its third ("Best2").  The five nearest of the query are then popped from the 64 kept entries
//   with five half-wave minima — exact whenever no lane saw three candidates at or below the fifth popped distance
//   and the six smallest kept distances are pairwise different (FLANN orders equal distances by index, which this
//   path never looks at).  The ~1-2 % of the queries that fail either check repeat the stream with a per-lane sorted
//   list of five (distance, window index) keys and a 64-bit merge ("Top5"): exact in every case.
//   Line gate in FP64, then NN0 / NN1 are written as the line points (laser_odometry.cc:351-357).
//   Round-2 design for the record: per-lane Top5 lists for every query, cells streamed in four
//   rounds of increasing box distance with the bound refreshed in between, second pass re-ranking the first pass's
//   saved lists: ~750 VALU wave instructions per query, 56 % of them fixed cost.
// =============================================================================================
// Cell edge of the kNN hash: 1 m, so that the 27-cell neighbourhood covers the sq_dist[4] < 1.0 gate (:324) exactly.
constexpr float kCellInv = 1.0f;
constexpr double kCellSize = 1.0;

// Per-lane sorted list of the five best candidates.  Key = (float distance bits << 32) | window
// index: distances are non-negative, so the unsigned 64-bit order is exactly "distance, then window
// index" (FLANN result order with the lower index winning ties).
struct Top5 {
  unsigned long long k0, k1, k2, k3, k4;   // ascending
  int p0, p1, p2, p3, p4;                  // position in the cell-sorted array
};
constexpr unsigned long long kTop5Empty = (0x7f800000ull << 32) | 0x7fffffffull;   // (+inf, INT_MAX)
__device__ __forceinline__ float top5_dist(unsigned long long k) { return __int_as_float((int)(k >> 32)); }
__device__ __forceinline__ int top5_index(unsigned long long k) { return (int)(unsigned int)(k & 0xFFFFFFFFull); }
// One compare-exchange stage: the smaller of (slot, carry) stays in the slot, the larger is carried on.
#define TOP5_STAGE(K, P)                                          \
  {                                                               \
    const bool lt = ck < (K);                                     \
    const unsigned long long nk = lt ? ck : (K);                  \
    const int np = lt ? cp : (P);                                 \
    ck = lt ? (K) : ck;                                           \
    cp = lt ? (P) : cp;                                           \
    (K) = nk; (P) = np;                                           \
  }
// Branch-free insertion (the kernel is VALU-issue bound and most waves have some lane inserting in
// every iteration: 5 x (one 64-bit compare + 6 selects) instead of a nest of exec-mask branches).
__device__ __forceinline__ void top5_insert(Top5& t, float d, int wi, int pos) {
  unsigned long long ck = ((unsigned long long)(unsigned int)__float_as_int(d) << 32) | (unsigned int)wi;
  int cp = pos;
  if (ck < t.k4) {
    TOP5_STAGE(t.k0, t.p0)
    TOP5_STAGE(t.k1, t.p1)
    TOP5_STAGE(t.k2, t.p2)
    TOP5_STAGE(t.k3, t.p3)
    TOP5_STAGE(t.k4, t.p4)
  }
}
#undef TOP5_STAGE
__device__ __forceinline__ void top5_clear(Top5& t) {
  t.k0 = t.k1 = t.k2 = t.k3 = t.k4 = kTop5Empty;
  t.p0 = t.p1 = t.p2 = t.p3 = t.p4 = -1;
}
struct Top5Acc {
  Top5 t;
  __device__ __forceinline__ void consider(bool ok, float d, int wi, int pos) { if (ok && d <= top5_dist(t.k4)) top5_insert(t, d, wi, pos); }   // cheap reject first
};

// Fast path: the two nearest candidates a lane has seen (distance + position) and the DISTANCE of its third nearest.
// Straight-line code: ~9 VALU instructions per candidate next to the ~7 of the distance (the sorted list above: ~50).
struct Best2Acc {
  float m1, m2, m3;
  int p1, p2;
  __device__ __forceinline__ void clear() { m1 = m2 = m3 = __int_as_float(0x7f800000); p1 = p2 = -1; }
  __device__ __forceinline__ void consider(bool ok, float d0, int /*wi*/, int pos) {
    const float d = ok ? d0 : __int_as_float(0x7f800000);
    const bool lt1 = d < m1, lt2 = d < m2;
    m3 = __builtin_amdgcn_fmed3f(m2, m3, d);      // third smallest of {m1 <= m2 <= m3, d}
    const int q2 = lt2 ? pos : p2;
    p2 = lt1 ? p1 : q2;
    m2 = __builtin_amdgcn_fmed3f(m1, m2, d);
    p1 = lt1 ? pos : p1;
    m1 = lt1 ? d : m1;
  }
};

// Streams the candidates of the cells selected by (start, cnt) [one cell per lane of the half-wave] through the
// per-lane accumulators: populous cells cell-major (all 32 lanes walk the same cell: no search for "which cell does
// flat index i belong to"), the small ones as one flat list (population prefix by DPP scan, monotone cell cursor per
// lane).  UB / U independent 16-B loads in flight per lane; loads are unconditional (index clamped into the segment,
// the result masked), so that they leave together and the loop body is straight-line code.
// Two tunings of the same code (template parameter kDeep of k_knn / knn_block):
//   lock-step batches (k_knn<128>, VALU-issue bound, 7 waves per SIMD hide the latency): 2 loads in flight per lane, cells
//     of >= 64 points cell-major, phase 1 of the first pass = own cell + neighbours within 6 cm (measured at 256 streams,
//     us per pass: loads 4/4 + cells >= 128: 576; 2/2 + >= 64: 511; 1/1 + >= 32: 534; per-lane cursor instead of the binary
//     search: +6 %; phase-1 radius 0 / 6 / 14 cm: 509 / 511 / 510);
//   few streams (k_knn<256>, one wave per SIMD, bound by the dependent memory round trips of its slowest query): 4
//     loads in flight per lane, cells of >= 128 points (8 loads / >= 256: no difference)
//     cell-major, phase 1 = own cell + neighbours within 20 cm (fewer queries need the second phase's round trip).
#ifndef LIODOM_TUNE_B_BIG            // (lock-step instance; overridable for experiments: tools/variant_build.sh)
#define LIODOM_TUNE_B_BIG 64
#define LIODOM_TUNE_B_LOADS_BIG 2
#define LIODOM_TUNE_B_LOADS_FLAT 2
#define LIODOM_TUNE_B_CURSOR false
#define LIODOM_TUNE_B_NEAR 0.0036f
#endif
#ifndef LIODOM_TUNE_B_WAVES
#define LIODOM_TUNE_B_WAVES 7        // waves per SIMD the lock-step instance is compiled for (72 VGPRs)
#endif
#ifndef LIODOM_TUNE_D_BIG            // (few-stream instance)
#define LIODOM_TUNE_D_BIG 128
#define LIODOM_TUNE_D_LOADS 4
#define LIODOM_TUNE_D_NEAR 0.04f
#endif
template <bool kDeep> struct KnnTune {
  static constexpr int kBigCell = kDeep ? LIODOM_TUNE_D_BIG : LIODOM_TUNE_B_BIG;
  static constexpr int kLoadsBig = kDeep ? LIODOM_TUNE_D_LOADS : LIODOM_TUNE_B_LOADS_BIG;
  static constexpr int kLoadsFlat = kDeep ? LIODOM_TUNE_D_LOADS : LIODOM_TUNE_B_LOADS_FLAT;
  static constexpr float kNearSq = kDeep ? LIODOM_TUNE_D_NEAR : LIODOM_TUNE_B_NEAR;
  static constexpr bool kProbeBoth = kDeep;
  static constexpr bool kHoistLoads = kDeep;
  static constexpr bool kCursor = kDeep ? false : LIODOM_TUNE_B_CURSOR;      // flat list: per-lane cursor instead of the binary search
};
constexpr int kKnnGridDiv = 2;         // k_knn grid = half of the query blocks the edge capacity allows: a workgroup takes block b and, if the scan has that many edges, b + grid
template <class Acc, int UB, int U, int kBigCell, bool kCursor = false>
__device__ __forceinline__ void knn_stream_cells(Acc& t, const float4* sp, int* s_incl, int* s_adj,
                                                 unsigned int start, unsigned int cnt, int hl,
                                                 float qx, float qy, float qz, unsigned int* dbg = nullptr) {
  const unsigned long long dbg_t0 = dbg ? wall_clock64() : 0ull;
  {
    const int half_base = (threadIdx.x & 32);
    unsigned int big = (unsigned int)((__ballot(cnt >= (unsigned int)kBigCell) >> half_base) & 0xFFFFFFFFull);
    while (big) {
      const int l = __ffs(big) - 1;
      big &= big - 1u;
      const int cs = __shfl((int)start, l, kKnnGroup), cc = __shfl((int)cnt, l, kKnnGroup);
      const float4* cp = sp + cs;
      for (int i = hl; i < cc; i += UB * kKnnGroup) {
        float4 m[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) { const int iu = i + u * kKnnGroup; m[u] = cp[iu < cc ? iu : cc - 1]; }
#pragma unroll
        for (int u = 0; u < UB; u++) {
          const int iu = i + u * kKnnGroup;
          t.consider(iu < cc, sqdist_f(qx, qy, qz, m[u].x, m[u].y, m[u].z), __float_as_int(m[u].w), cs + iu);
        }
      }
    }
    if (cnt >= (unsigned int)kBigCell) cnt = 0;       // done; the flat pass below takes the small cells
  }
  const unsigned int dbg_nseg = dbg ? (unsigned int)__popc((unsigned int)(__ballot(cnt > 0) >> (threadIdx.x & 32))) : 0u;
  const int incl = half_incl_scan_i32((int)cnt);
  s_incl[hl] = incl;
  s_adj[hl] = (int)start - (incl - (int)cnt);
  __builtin_amdgcn_wave_barrier();
  const int T = s_incl[kKnnGroup - 1];
  const unsigned long long dbg_t1 = dbg ? wall_clock64() : 0ull;
  int cur = 0;
  for (int i = hl; i < T; i += U * kKnnGroup) {
    // owner segment of flat index iu = number of segments whose inclusive prefix is <= iu: a 5-step binary search over
    // the 32 prefixes in LDS, the U searches of a lane side by side (a per-lane cursor loop — dependent LDS reads behind
    // divergent branches — cost 2-3 us per round on a single stream)
    int a[U];
    int c[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int iu = i + u * kKnnGroup; a[u] = iu < T ? iu : T - 1; c[u] = 0; }
    if (kCursor) {
      // (lock-step batches: fewer instructions) monotone per-lane cursor: the flat index only grows
#pragma unroll
      for (int u = 0; u < U; u++) {
        while (s_incl[cur] <= a[u]) cur++;
        c[u] = cur;
      }
    } else {
#pragma unroll
      for (int step = kKnnGroup / 2; step >= 1; step >>= 1) {
        int pv[U];
#pragma unroll
        for (int u = 0; u < U; u++) pv[u] = s_incl[c[u] + step - 1];
#pragma unroll
        for (int u = 0; u < U; u++) c[u] += pv[u] <= a[u] ? step : 0;
      }
    }
    int adj[U];
#pragma unroll
    for (int u = 0; u < U; u++) adj[u] = s_adj[c[u]];
#pragma unroll
    for (int u = 0; u < U; u++) a[u] += adj[u];
    float4 m[U];
#pragma unroll
    for (int u = 0; u < U; u++) m[u] = sp[a[u]];
#pragma unroll
    for (int u = 0; u < U; u++)
      t.consider(i + u * kKnnGroup < T, sqdist_f(qx, qy, qz, m[u].x, m[u].y, m[u].z), __float_as_int(m[u].w), a[u]);
  }
  __builtin_amdgcn_wave_barrier();
  if (dbg && hl == 0) {
    dbg[0] = (unsigned int)(dbg_t1 - dbg_t0);                      // big-cell part, 10 ns ticks
    dbg[1] = (unsigned int)(wall_clock64() - dbg_t1);              // flat part
    dbg[2] = ((unsigned int)((T + U * kKnnGroup - 1) / (U * kKnnGroup)) << 16) | ((unsigned int)T << 20);   // flat rounds, flat candidates
    dbg[3] = dbg_nseg;
  }
}

// Pops the five nearest of the half-wave's query from the lanes' Best2 entries (position of the r-th nearest ->
// pos[r], fifth distance -> d5).  Returns true when that result is certain:
//   d5 <  1.0: the six smallest kept distances are pairwise different (no index tie-break needed) and every lane's
//              third-nearest distance lies above d5 (so every candidate at or below d5 is among the kept entries);
//   d5 >= 1.0: no lane's third nearest is below 1.0, i.e. fewer than five candidates exist inside the 1.0 gate (:324).
__device__ __forceinline__ bool best2_select(const Best2Acc& t, int hl, int half_shift, float& d5, int (&pos)[5]) {
  unsigned int v = (unsigned int)__float_as_int(t.m1), w = (unsigned int)__float_as_int(t.m2);
  int ph = t.p1, pn = t.p2;
  unsigned int g[6];
#pragma unroll
  for (int r = 0; r < 5; r++) {
    g[r] = half_min_u32(v);                        // non-negative floats order as unsigned ints
    const unsigned int win = (unsigned int)((__ballot(v == g[r]) >> half_shift) & 0xFFFFFFFFull);
    const int l = __ffs(win) - 1;
    pos[r] = __shfl(ph, l, kKnnGroup);
    const bool mine = hl == l;
    v = mine ? w : v;
    w = mine ? 0x7f800000u : w;
    ph = mine ? pn : ph;
  }
  g[5] = half_min_u32(v);
  const unsigned int s3 = half_min_u32((unsigned int)__float_as_int(t.m3));
  const unsigned int one = 0x3f800000u;
  d5 = __int_as_float((int)g[4]);
  if (g[4] < one) return g[0] < g[1] && g[1] < g[2] && g[2] < g[3] && g[3] < g[4] && g[4] < g[5] && g[4] < s3;
  return s3 >= one;
}


// Merges the 32 per-lane lists of a half-wave: afterwards every lane holds the global top-5
// (ascending by distance, ties by window index) in g.
__device__ __forceinline__ void knn_merge(Top5& t, Top5& g, int hl, int half_shift) {
  unsigned long long gk[5]; int gp[5];
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const unsigned long long key = t.k0;
    // reduce on the 32-bit distance (half the DPP traffic of a 64-bit reduction); only when several
    // lanes tie on the distance the full (distance, index) key decides
    const unsigned int dmin = half_min_u32((unsigned int)(key >> 32));
    unsigned int win = (unsigned int)((__ballot((unsigned int)(key >> 32) == dmin) >> half_shift) & 0xFFFFFFFFull);
    unsigned long long mk;
    if (__popc(win) == 1) {
      mk = ((unsigned long long)dmin << 32) | (unsigned int)__shfl((int)(unsigned int)(key & 0xFFFFFFFFull), __ffs(win) - 1, kKnnGroup);
    } else {
      mk = half_min_u64(key);
      win = (unsigned int)((__ballot(key == mk) >> half_shift) & 0xFFFFFFFFull);
    }
    const int wl = __ffs(win) - 1;
    gp[r] = __shfl(t.p0, wl, kKnnGroup);
    gk[r] = mk;
    if (hl == wl) {   // pop
      t.k0 = t.k1; t.k1 = t.k2; t.k2 = t.k3; t.k3 = t.k4; t.k4 = kTop5Empty;
      t.p0 = t.p1; t.p1 = t.p2; t.p2 = t.p3; t.p3 = t.p4; t.p4 = -1;
    }
  }
  g.k0 = gk[0]; g.k1 = gk[1]; g.k2 = gk[2]; g.k3 = gk[3]; g.k4 = gk[4];
  g.p0 = gp[0]; g.p1 = gp[1]; g.p2 = gp[2]; g.p3 = gp[3]; g.p4 = gp[4];
}

// Workgroup size = 32 lanes x queries.  A workgroup lasts as long as its slowest query, so few queries per workgroup
// win: 8 (256 threads) on handles with few streams, 4 (128 threads) on lock-step batches; two instances, chosen by
// the host from the stream count.
__device__ void rebuild_alloc(const DevView& v, int s, StreamState& st, int block, int nblocks);

// LDS of one k_knn workgroup (kQ queries)
template <int kQ>
struct KnnShared {
  int incl[kQ][kKnnGroup];       // inclusive candidate prefix per cell
  int adj[kQ][kKnnGroup];        // cell start - exclusive prefix
  float nn[kQ][16];              // the five neighbours of every query (xyz)
  int res[kQ][4];                // distance gate passed, window index of NN0, NN1, line gate passed
  double part[kQ][32];           // normal-equation terms of every query's residual block at the solve's start pose
  double blk[kQ][24];            // that block's J[18], rho' r [3], rho, rho', validity (0 none, 1 valid, 2 non-finite)
};

// Upper bound of the half-wave query's fifth-nearest distance from the entries kept so far: the smallest of a
// ladder of thresholds at or below which at least five kept entries lie (1.0, the gate of :324, if none does).
__device__ __forceinline__ float best2_bound(const Best2Acc& t, int half_shift) {
  float B = 1.0f;
  const float thr[4] = {0.36f, 0.09f, 0.0225f, 0.0036f};      // (0.6 m, 0.3 m, 0.15 m, 0.06 m) squared, descending
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int c = __popc((unsigned int)(__ballot(t.m1 <= thr[k]) >> half_shift)) + __popc((unsigned int)(__ballot(t.m2 <= thr[k]) >> half_shift));
    B = c >= 5 ? thr[k] : B;
  }
  return B;
}

// One segment of candidates per lane of the half-wave: the cell of (cx, cy, cz) and its 26 neighbours (lanes 0..26; the own
// cell is lane 13) and, on lane 27, the overflow list of the streamed rebuild.  lb = lower bound of the float squared
// distance from q to any point of the segment (see knn_block).
template <class Tune>
__device__ __forceinline__ void knn_probe_cells(const DevView& v, const StreamState& st, int hl, int cx, int cy, int cz, float qx, float qy, float qz,
                                                const CellSlot* cells, const unsigned int* bits, unsigned int tmask,
                                                unsigned int& start, unsigned int& cnt, float& lb) {
  if (hl < 27) {
    const int dx = hl % 3 - 1, dy = (hl / 3) % 3 - 1, dz = hl / 9 - 1;
    const unsigned long long key = pack_cell(cx + dx, cy + dy, cz + dz);
    unsigned int h = hash_cell(key, tmask);
    if (Tune::kProbeBoth) {
      // (few streams: latency counts) occupancy bit and slot of the first probe leave together — one round trip instead of
      // two; the slots of empty cells (most of the 27) are loaded for nothing, 16 B each
      const unsigned int word = bits[h >> 5];
      const uint4 raw = *reinterpret_cast<const uint4*>(cells + h);
      const unsigned long long k = ((unsigned long long)raw.y << 32) | raw.x;
      bool more = ((word >> (h & 31)) & 1u) != 0u;
      if (more && k == key) { start = raw.z; cnt = raw.w; more = false; }
      for (int pr = 1; more && pr < v.table_size; pr++) {       // (collision chain: rare)
        h = (h + 1) & tmask;
        if (!((bits[h >> 5] >> (h & 31)) & 1u)) break;
        const uint4 r2 = *reinterpret_cast<const uint4*>(cells + h);
        if ((((unsigned long long)r2.y << 32) | r2.x) == key) { start = r2.z; cnt = r2.w; break; }
      }
    } else {
      for (int pr = 0; pr < v.table_size; pr++) {
        if (!((bits[h >> 5] >> (h & 31)) & 1u)) break;           // empty slot: cell not in the map
        const uint4 raw = *reinterpret_cast<const uint4*>(cells + h);
        const unsigned long long k = ((unsigned long long)raw.y << 32) | raw.x;
        if (k == key) { start = raw.z; cnt = raw.w; break; }
        h = (h + 1) & tmask;
      }
    }
    const float cs = (float)kCellSize;
    const float lx = (float)(cx + dx) * cs, ly = (float)(cy + dy) * cs, lz = (float)(cz + dz) * cs;
    const float ex = qx < lx ? lx - qx : (qx > lx + cs ? qx - (lx + cs) : 0.0f);
    const float ey = qy < ly ? ly - qy : (qy > ly + cs ? qy - (ly + cs) : 0.0f);
    const float ez = qz < lz ? lz - qz : (qz > lz + cs ? qz - (lz + cs) : 0.0f);
    lb = (ex * ex + ey * ey + ez * ez) * (1.0f - 1e-5f);
  } else if (hl == 27 && v.early_rebuild) {
    start = (unsigned int)v.ovf_base;
    cnt = (unsigned int)st.n_ovf[LD_TAB_PARITY(v, st.frame_count)];
  }
}

// What a query of the OVERLAPPED second pass re-ranks, collected while the first solve still runs (knn_presearch).
struct KnnPre {
  float gsq;        // guard: no map point outside the collected set is closer to the first pass's query than sqrt(gsq) (0: nothing collected)
  float4 sq;        // the first pass's query and its fifth-nearest distance
  int p[5];         // this lane's collected candidates (positions in the cell-sorted array; -1: none)
  float4 c[5];      // ... and the points there (w: window index)
};
// Overlapped second pass, before the first solve's result is there: an exact search around the FIRST pass's query q_old (the
// second query will be millimetres away) that collects every map point within sqrt(d5_old) + kOvMargin of it — sorted
// per-lane lists of five, the sentinel-initialised Top5 lists of the exact path — and loads the collected points.  With the
// result of the solve the block only re-ranks these (knn_block, kPre): d = |q_new - q_old| is far below the margin, so the
// re-ranked five are certified by the same guard argument as the non-overlapped re-ranking, practically always — the
// search a non-certified query falls back to (which a launch lasts as long as) disappears from the critical path.
constexpr float kOvMargin = 0.03f;
template <int kKnnThreads>
__device__ __forceinline__ void knn_presearch(const DevView& v, int s, const StreamState& st, int e, int E,
                                              KnnShared<kKnnThreads / kKnnGroup>& sh, KnnPre& pre) {
  typedef KnnTune<(kKnnThreads >= 256)> Tune;
  const int grp = threadIdx.x / kKnnGroup, hl = threadIdx.x & (kKnnGroup - 1);
  const int ec = e < v.edge_cap ? e : v.edge_cap - 1;
  pre.gsq = 0.f;
  pre.sq = make_float4(0.f, 0.f, 0.f, __int_as_float(0x7f800000));
#pragma unroll
  for (int k = 0; k < 5; k++) { pre.p[k] = -1; pre.c[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
  if (!v.knn_save_q) return;
  pre.sq = v.knn_save_q[(size_t)s * v.edge_cap + ec];
  const float qx = pre.sq.x, qy = pre.sq.y, qz = pre.sq.z;
  const bool act = e < E && !v.knn_exact_only && ld_isfinite((double)qx) && ld_isfinite((double)qy) && ld_isfinite((double)qz) &&
                   fabsf(qx) < 1.0e9f && fabsf(qy) < 1.0e9f && fabsf(qz) < 1.0e9f;
  if (!act) return;                                // (uniform over the half-wave)
  const int cx = (int)floorf(qx * kCellInv), cy = (int)floorf(qy * kCellInv), cz = (int)floorf(qz * kCellInv);
  const unsigned int tmask = st.table_mask;
  const int stab = s + LD_TAB_PARITY(v, st.frame_count) * v.n_streams;
  const CellSlot* cells = v.cells + (size_t)stab * v.table_size;
  const unsigned int* bits = v.cell_bits + (size_t)stab * (v.table_size >> 5);
  const float4* sp = v.sorted_pts + (size_t)stab * v.sorted_cap;
  unsigned int start = 0, cnt = 0;
  float lb = 0.0f;
  knn_probe_cells<Tune>(v, st, hl, cx, cy, cz, qx, qy, qz, cells, bits, tmask, start, cnt, lb);
  // everything within sqrt(min(d5_old, 1)) + margin of q_old (beyond the 1.0 gate nothing can matter: :324)
  const float r = fminf(sqrtf(pre.sq.w), 1.0f) + kOvMargin;
  const float B = r * r * (1.0f + 1e-5f);
  Top5Acc ta;
  {
    const unsigned long long sentinel = ((unsigned long long)(unsigned int)__float_as_int(B) << 32) | 0x7fffffffull;
    ta.t.k0 = ta.t.k1 = ta.t.k2 = ta.t.k3 = ta.t.k4 = sentinel;
    ta.t.p0 = ta.t.p1 = ta.t.p2 = ta.t.p3 = ta.t.p4 = -1;
  }
  const bool all = cnt > 0 && !(lb > B);
  knn_stream_cells<Top5Acc, 2, 2, 64>(ta, sp, sh.incl[grp], sh.adj[grp], start, all ? cnt : 0u, hl, qx, qy, qz);
  // guard: B itself (segments with lb > B and points beyond B were left out), the fifth entry of a lane whose list is full
  // (it may have dropped candidates at or beyond that distance), the border of the 27-cell block
  float gl = B;
  if (ta.t.p4 >= 0) gl = fminf(gl, top5_dist(ta.t.k4));
  float guard = __int_as_float((int)half_min_u32((unsigned int)__float_as_int(gl)));
  {
    const float cs = (float)kCellSize;
    const float fx = qx - (float)cx * cs, fy = qy - (float)cy * cs, fz = qz - (float)cz * cs;
    float edge = fminf(fminf(fminf(fx, cs - fx), fminf(fy, cs - fy)), fminf(fz, cs - fz));
    edge = edge > 0.f ? edge : 0.f;
    const float outer = (cs + edge) * (cs + edge) * (1.0f - 1e-6f);
    guard = guard < outer ? guard : outer;
  }
  pre.gsq = guard;
  pre.p[0] = ta.t.p0; pre.p[1] = ta.t.p1; pre.p[2] = ta.t.p2; pre.p[3] = ta.t.p3; pre.p[4] = ta.t.p4;
#pragma unroll
  for (int k = 0; k < 5; k++) pre.c[k] = sp[pre.p[k] >= 0 ? pre.p[k] : 0];
}

// One block of kKnnThreads / 32 queries (virtual block index bv).  Whole workgroup; returns are workgroup-uniform.
// kPre (overlapped second pass): what the re-ranking loads is in `pre` already, and the solve's start point (q, t) comes
// from qt (LDS) — the stream's state is still being written by the first solve's launch.
template <int kKnnThreads, bool kPre = false, bool kTail = true>
__device__ __forceinline__ void knn_block(const DevView& v, int s, StreamState& st, int outer_it, int eb, int bv, int E,
                                          KnnShared<kKnnThreads / kKnnGroup>& sh, const float4& p_in, const double (&T_in)[12],
                                          const KnnPre& pre, const double* qt) {
  constexpr int kKnnQueries = kKnnThreads / kKnnGroup;
  typedef KnnTune<(kKnnThreads >= 256)> Tune;
  const int grp = threadIdx.x / kKnnGroup;
  const int e = bv * kKnnQueries + grp;
  const int hl = threadIdx.x & (kKnnGroup - 1);
  const int half_shift = (threadIdx.x & 32);     // 0 or 32: which half of the wave
  const bool dbgb = (bv == 5) && (s == 0) && (threadIdx.x == 0) && (outer_it == 0);
  const unsigned long long t_blk = (kInstrument && (v.debug & 32)) ? wall_clock64() : 0ull;
  DBG_STAMP(v, dbgb, 1, 0);
  bool active = e < E;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    // (few streams: edge and pose were loaded by the caller, beside the stream's state words; lock-step batches load them
    //  here — hoisted they would cost 10 VGPRs, i.e. a wave per SIMD)
    float4 p = p_in;
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = T_in[i];
    if (!Tune::kHoistLoads) {
      p = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + e];
#pragma unroll
      for (int i = 0; i < 12; i++) T[i] = st.odom[i];
    }
    transform_point(T, p.x, p.y, p.z, &qx, &qy, &qz);          // :307-308
    if (v.knn_q && hl == 0) v.knn_q[((size_t)s * 2 + outer_it) * v.edge_cap + e] = make_float4(qx, qy, qz, 0.f);
    active = ld_isfinite((double)qx) && ld_isfinite((double)qy) && ld_isfinite((double)qz) &&
             fabsf(qx) < 1.0e9f && fabsf(qy) < 1.0e9f && fabsf(qz) < 1.0e9f;
  }
  DBG_STAMP(v, dbgb, 1, 1); DBG_QSTAMP(1);
  if (hl == 0) { sh.res[grp][0] = 0; sh.res[grp][1] = -1; sh.res[grp][2] = -1; }
  float d5 = __int_as_float(0x7f800000);
  if (active) {                                    // uniform over each 32-lane half
    const int cx = (int)floorf(qx * kCellInv), cy = (int)floorf(qy * kCellInv), cz = (int)floorf(qz * kCellInv);
    const unsigned int tmask = st.table_mask;
    const int stab = s + LD_TAB_PARITY(v, st.frame_count) * v.n_streams;
    const CellSlot* cells = v.cells + (size_t)stab * v.table_size;
    const unsigned int* bits = v.cell_bits + (size_t)stab * (v.table_size >> 5);
    const float4* sp = v.sorted_pts + (size_t)stab * v.sorted_cap;
    float d5_r = __int_as_float(0x7f800000);
    int pos5[5] = {-1, -1, -1, -1, -1};
    // ---- second pass of a scan: re-rank what the first pass kept.  The first solve moves the pose by millimetres, so
    // almost every query has the same neighbours as before.  Pass 0 saved the two candidates every lane kept (64 positions:
    // a superset of the five nearest) and a guard g: no map point outside that set was closer to the old query than
    // sqrt(g) (the lanes' third-nearest distances, the box distances of the pruned cells, the distance to the border of
    // the 27-cell block).  With d = |q_new - q_old| every unsaved point is now at least sqrt(g) - d away; if the fifth of
    // the re-ranked set is strictly closer than that (rounding margins included) — or nothing unsaved can be inside the
    // 1.0 gate — it is the exact answer and the query needs no probe and no stream; otherwise it searches below. ----
    bool reranked = false;
    if (kPre) {
      // overlapped pass: re-rank what knn_presearch collected around the first pass's query (sorted lists, exact merge:
      // ties by window index as in the exact path)
      if (pre.gsq > 0.f) {                                       // (uniform over the half-wave)
        Top5 t, g;
        top5_clear(t);
#pragma unroll
        for (int k = 0; k < 5; k++) {
          if (pre.p[k] >= 0) top5_insert(t, sqdist_f(qx, qy, qz, pre.c[k].x, pre.c[k].y, pre.c[k].z), __float_as_int(pre.c[k].w), pre.p[k]);
        }
        knn_merge(t, g, hl, half_shift);
        const float d5n = g.p4 >= 0 ? top5_dist(g.k4) : __int_as_float(0x7f800000);
        const double ddx = (double)qx - (double)pre.sq.x, ddy = (double)qy - (double)pre.sq.y, ddz = (double)qz - (double)pre.sq.z;
        const double delta = sqrt(ddx * ddx + ddy * ddy + ddz * ddz) * (1.0 + 1e-12);
        const double r = sqrt((double)pre.gsq) * (1.0 - 2e-7) - delta;     // every point outside the collected set is at least this far now
        const double limit = r > 0.0 ? r * r * (1.0 - 1e-6) : 0.0;         // (float rounding of the new distances included)
        reranked = (double)d5n < limit || limit > 1.0;                     // beyond the 1.0 gate nothing uncollected can matter
        if (reranked) {
          d5_r = d5n;
          pos5[0] = g.p0; pos5[1] = g.p1; pos5[2] = g.p2; pos5[3] = g.p3; pos5[4] = g.p4;
          if (d5n < 1.0f) {
            // the five neighbours are among the points the lanes hold: whoever holds the r-th hands it over (no second fetch)
#pragma unroll
            for (int k = 0; k < 5; k++) {
              if (pre.p[k] >= 0) {
                const int r = pre.p[k] == g.p0 ? 0 : pre.p[k] == g.p1 ? 1 : pre.p[k] == g.p2 ? 2 : pre.p[k] == g.p3 ? 3 : pre.p[k] == g.p4 ? 4 : -1;
                if (r >= 0) {
                  sh.nn[grp][r * 3 + 0] = pre.c[k].x; sh.nn[grp][r * 3 + 1] = pre.c[k].y; sh.nn[grp][r * 3 + 2] = pre.c[k].z;
                  if (r < 2) sh.res[grp][1 + r] = __float_as_int(pre.c[k].w);     // window indices of NN0, NN1
                }
              }
            }
            if (hl == 0) sh.res[grp][0] = 1;
          }
        }
      }
    } else
    if (outer_it == 1 && v.knn_save_pos && !v.knn_exact_only) {
      const float gsq = v.knn_save_g[(size_t)s * v.edge_cap + e];
      if (gsq > 0.f) {                                           // (uniform over the half-wave)
        const float4 sq = v.knn_save_q[(size_t)s * v.edge_cap + e];
        const int2 sv = v.knn_save_pos[((size_t)s * v.edge_cap + e) * kKnnGroup + hl];
        const float4 m0 = sp[sv.x >= 0 ? sv.x : 0], m1 = sp[sv.y >= 0 ? sv.y : 0];
        Best2Acc br;
        br.clear();
        br.consider(sv.x >= 0, sqdist_f(qx, qy, qz, m0.x, m0.y, m0.z), 0, sv.x);
        br.consider(sv.y >= 0, sqdist_f(qx, qy, qz, m1.x, m1.y, m1.z), 0, sv.y);
        float d5n;
        int p5[5];
        const bool sel_ok = best2_select(br, hl, half_shift, d5n, p5);
        const double ddx = (double)qx - (double)sq.x, ddy = (double)qy - (double)sq.y, ddz = (double)qz - (double)sq.z;
        const double delta = sqrt(ddx * ddx + ddy * ddy + ddz * ddz) * (1.0 + 1e-12);
        const double r = sqrt((double)gsq) * (1.0 - 2e-7) - delta;         // every unsaved point is at least this far now
        const double limit = r > 0.0 ? r * r * (1.0 - 1e-6) : 0.0;         // (float rounding of the new distances included)
        reranked = sel_ok && ((double)d5n < limit || limit > 1.0);         // beyond the 1.0 gate nothing unsaved can matter
        if (reranked) {
          d5_r = d5n;
#pragma unroll
          for (int k = 0; k < 5; k++) pos5[k] = p5[k];
        }
      }
    }
    if ((kInstrument && (v.debug & 64)) && hl == 0 && outer_it == 1) atomicAdd(&v.dbg_clk[259 + (reranked ? 0 : 1)], 1ull);
    if (!reranked) {
    // One segment of candidates per lane: the query's cell and its 26 neighbours (lanes 0..26; the own cell is lane 13),
    // and on lane 27 the overflow list of the streamed rebuild (points of the newest frame that moved out of their
    // padded cells: empty unless the solve corrected the prediction by more than rebuild_delta).  lb = lower bound of
    // the float squared distance from q to any point of the segment: the box distance of the cell, shrunk by 1e-5 so
    // that rounding of the candidate distances (float, ~3e-7 relative) or of the bound itself can never make a
    // pruned point look closer than the bound.
    unsigned int start = 0, cnt = 0;
    float lb = 0.0f;
    knn_probe_cells<Tune>(v, st, hl, cx, cy, cz, qx, qy, qz, cells, bits, tmask, start, cnt, lb);
    DBG_STAMP(v, dbgb, 1, 2); DBG_QSTAMP(2);
    // Pruning bound B: an upper bound of the query's fifth-nearest distance (never above the 1.0 gate: points at
    // >= 1.0 cannot be part of a match, :324); a segment is skipped only if lb > B, so the result is exact.
    //   second pass of a scan: the map has not changed and the first solve moved the query by delta (millimetres), so
    //   the five neighbours the first pass found are now within sqrt(d5_first) + delta: B is known before anything is
    //   streamed, one phase.
    //   first pass: phase 1 streams the own cell (+ the neighbours within Tune::kNearSq of q, + the overflow list), B
    //   comes from the entries kept so far (best2_bound), phase 2 streams what B leaves of the other cells.
    float B = 1.0f;
    bool have_b = false;
    if (outer_it == 1 && v.knn_save_q) {
      const float4 sq = kPre ? pre.sq : v.knn_save_q[(size_t)s * v.edge_cap + e];
      if (sq.w < 1.0f) {                                         // (uniform over the half-wave; inf / >= 1: nothing to gain)
        const float ddx = qx - sq.x, ddy = qy - sq.y, ddz = qz - sq.z;
        const float delta = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
        const float r = sqrtf(sq.w) * (1.0f + 1e-6f) + delta * (1.0f + 1e-6f) + 1e-7f;
        B = fminf(1.0f, r * r * (1.0f + 1e-5f));
        have_b = true;
      }
    }
    bool pend = cnt > 0;
    Best2Acc b2;
    b2.clear();
    int dbg_n = 0;
    bool dbg_two_phase = false;
    {
      const bool now = pend && (have_b ? !(lb > B) : (hl == 13 || hl == 27 || lb <= Tune::kNearSq));
      if (kInstrument && (v.debug & 32)) dbg_n = __shfl(half_incl_scan_i32(now ? (int)cnt : 0), kKnnGroup - 1, kKnnGroup);
      knn_stream_cells<Best2Acc, Tune::kLoadsBig, Tune::kLoadsFlat, Tune::kBigCell, Tune::kCursor>(b2, sp, sh.incl[grp], sh.adj[grp], start, now ? cnt : 0u, hl, qx, qy, qz,
                                                                                     ((kInstrument && (v.debug & 32)) && s == 0 && e < E) ? v.dbg_q + ((size_t)outer_it * v.edge_cap + e) * 12 + 8 : nullptr);
      pend = pend && !now;
    }
    DBG_STAMP(v, dbgb, 1, 3); DBG_QSTAMP(3);
    if (!have_b && ((__ballot(pend) >> half_shift) & 0xFFFFFFFFull)) {      // (uniform over the half-wave)
      B = best2_bound(b2, half_shift);
      pend = pend && !(lb > B);
      if ((__ballot(pend) >> half_shift) & 0xFFFFFFFFull) {
        if (kInstrument && (v.debug & 32)) { dbg_n += __shfl(half_incl_scan_i32(pend ? (int)cnt : 0), kKnnGroup - 1, kKnnGroup); dbg_two_phase = true; }
        knn_stream_cells<Best2Acc, Tune::kLoadsBig, Tune::kLoadsFlat, Tune::kBigCell, Tune::kCursor>(b2, sp, sh.incl[grp], sh.adj[grp], start, pend ? cnt : 0u, hl, qx, qy, qz);
      }
    } else {
      pend = false;
    }
    DBG_STAMP(v, dbgb, 1, 4); DBG_QSTAMP(4);
    const bool certain = best2_select(b2, hl, half_shift, d5, pos5) && !v.knn_exact_only;
    if ((kInstrument && (v.debug & 32)) && s == 0 && e < E && hl == 0) v.dbg_q[((size_t)outer_it * v.edge_cap + e) * 12] = (unsigned int)dbg_n | (dbg_two_phase ? 0x40000000u : 0u) | (certain ? 0u : 0x80000000u);
    if ((kInstrument && (v.debug & 64)) && hl == 0) {       // (debug) fast-path results / exact-list repeats / queries with a second phase; candidates streamed
      atomicAdd(&v.dbg_clk[256 + (certain ? 0 : 1)], 1ull);
      if (dbg_two_phase) atomicAdd(&v.dbg_clk[258], 1ull);
      atomicAdd(&v.dbg_clk[384 + (dbg_n / 64 < 63 ? dbg_n / 64 : 63)], 1ull);
    }
    if (!certain) {                                  // (uniform over the half-wave) exact path: sorted (distance, index) lists
      // every segment the fast path streamed (its pruning was exact): lb <= B, or the phase-1 set
      const bool all = cnt > 0 && (!(lb > B) || (!have_b && (hl == 13 || hl == 27 || lb <= Tune::kNearSq)));
      // The fast path's fifth popped distance bounds the true fifth-nearest distance from above whenever it is finite (five
      // kept entries lie at or below it), and nothing at or beyond the 1.0 gate can matter: the lists start filled with the
      // sentinel (bound, INT_MAX), so only the handful of candidates at or below the bound are ever inserted — this repeat
      // costs about as much as the fast stream (every launch has a query or two that need it, and a launch lasts as long
      // as its slowest query).
      Top5Acc ta;
      Top5 g;
      {
        const float bnd = d5 < 1.0f ? d5 : 1.0f;
        const unsigned long long sentinel = ((unsigned long long)(unsigned int)__float_as_int(bnd) << 32) | 0x7fffffffull;
        ta.t.k0 = ta.t.k1 = ta.t.k2 = ta.t.k3 = ta.t.k4 = sentinel;
        ta.t.p0 = ta.t.p1 = ta.t.p2 = ta.t.p3 = ta.t.p4 = -1;
      }
      knn_stream_cells<Top5Acc, 2, 2, 64>(ta, sp, sh.incl[grp], sh.adj[grp], start, all ? cnt : 0u, hl, qx, qy, qz);
      knn_merge(ta.t, g, hl, half_shift);
      d5 = g.p4 >= 0 ? top5_dist(g.k4) : __int_as_float(0x7f800000);       // (a sentinel among the five: fewer than five candidates inside the gate)
      pos5[0] = g.p0; pos5[1] = g.p1; pos5[2] = g.p2; pos5[3] = g.p3; pos5[4] = g.p4;
    }
    if (outer_it == 0 && v.knn_save_pos) {
      // what the second pass re-ranks: the lanes' kept candidates and the guard (see above)
      const float sk = (cnt > 0 && !(!(lb > B) || (!have_b && (hl == 13 || hl == 27 || lb <= Tune::kNearSq)))) ? lb : __int_as_float(0x7f800000);   // pruned, non-empty segment
      unsigned int gd = half_min_u32((unsigned int)__float_as_int(sk));
      const unsigned int m3m = half_min_u32((unsigned int)__float_as_int(b2.m3));
      gd = m3m < gd ? m3m : gd;
      float guard = __int_as_float((int)gd);
      {
        // points outside the 27 cells: at least 1 + (distance of q to the nearest face of its own cell) away
        const float cs = (float)kCellSize;
        const float fx = qx - (float)cx * cs, fy = qy - (float)cy * cs, fz = qz - (float)cz * cs;
        float edge = fminf(fminf(fminf(fx, cs - fx), fminf(fy, cs - fy)), fminf(fz, cs - fz));
        edge = edge > 0.f ? edge : 0.f;
        const float outer = (cs + edge) * (cs + edge) * (1.0f - 1e-6f);
        guard = guard < outer ? guard : outer;
      }
      v.knn_save_pos[((size_t)s * v.edge_cap + e) * kKnnGroup + hl] = make_int2(b2.p1, b2.p2);
      if (hl == 0) v.knn_save_g[(size_t)s * v.edge_cap + e] = guard < 3.0e38f ? guard : 3.0e38f;
    }
    if ((kInstrument && (v.debug & 64)) && s == 0 && hl == 0) {
      const int bin = (int)((wall_clock64() - t_blk) / 100ull);
      atomicAdd(&v.dbg_clk[320 + (bin < 63 ? bin : 63)], 1ull);
      const int nb = dbg_n < 64 ? 0 : dbg_n < 128 ? 1 : dbg_n < 256 ? 2 : dbg_n < 512 ? 3 : dbg_n < 1024 ? 4 : 5;
      atomicAdd(&v.dbg_clk[448 + (bin / 4 < 7 ? bin / 4 : 7) * 8 + nb + (dbg_two_phase ? 0 : 0)], 1ull);
      if (!certain) atomicAdd(&v.dbg_clk[448 + (bin / 4 < 7 ? bin / 4 : 7) * 8 + 7], 1ull);
      if (dbg_two_phase) atomicAdd(&v.dbg_clk[448 + (bin / 4 < 7 ? bin / 4 : 7) * 8 + 6], 1ull);
    }
    }   // (!reranked)
    else d5 = d5_r;
    DBG_STAMP(v, dbgb, 1, 5); DBG_QSTAMP(5);
    if (d5 < 1.0f && !(kPre && reranked)) {          // :324 (inf when < 5 candidates)
      const int mypos = hl == 0 ? pos5[0] : hl == 1 ? pos5[1] : hl == 2 ? pos5[2] : hl == 3 ? pos5[3] : pos5[4];
      if (hl < 5) {
        const float4 m = sp[mypos];
        sh.nn[grp][hl * 3 + 0] = m.x; sh.nn[grp][hl * 3 + 1] = m.y; sh.nn[grp][hl * 3 + 2] = m.z;
        if (hl < 2) sh.res[grp][1 + hl] = __float_as_int(m.w);     // window indices of NN0, NN1
      }
      if (hl == 0) sh.res[grp][0] = 1;
    }
  }
  // what the second pass prunes with: the query and its fifth-nearest distance (inf: fewer than five candidates / no query)
  if (outer_it == 0 && v.knn_save_q && e < E && hl == 0) {
    v.knn_save_q[(size_t)s * v.edge_cap + e] = make_float4(qx, qy, qz, d5);
    if (!active && v.knn_save_g) v.knn_save_g[(size_t)s * v.edge_cap + e] = 0.f;       // (no query: nothing to re-rank)
  }
  if (!kTail) return;          // (overlapped pass: the line gates / partial sums of the workgroup's two blocks run side by side, knn_tail_dual)
  __syncthreads();
  DBG_STAMP(v, dbgb, 1, 6); DBG_QSTAMP(6);
  if (kKnnThreads < 256 && v.knn_nn) {
    // Lock-step batches (VALU-issue bound): the line gates of a workgroup's four queries would occupy a whole wave's
    // instruction stream for four lanes.  The neighbours go to memory instead (80 B per query) and k_line_gate runs
    // the gates with one query per lane on full waves.
    if (threadIdx.x < kKnnQueries * 5) {
      const int q = threadIdx.x / 5, j = threadIdx.x % 5;
      const int eq = bv * kKnnQueries + q;
      if (eq < v.edge_cap) {
        const int w = j == 0 ? sh.res[q][0] : (j == 1 ? sh.res[q][1] : (j == 2 ? sh.res[q][2] : 0));
        v.knn_nn[((size_t)s * v.edge_cap + eq) * 5 + j] = make_float4(sh.nn[q][j * 3], sh.nn[q][j * 3 + 1], sh.nn[q][j * 3 + 2], __int_as_float(w));
      }
    }
    return;
  }
  // Line gate (:325-344): one lane per query, so the FP64 eigenvalue iteration runs once per 32
  // queries instead of once per query.
  if (threadIdx.x < kKnnQueries) {
    const int q = threadIdx.x;
    const int eq = bv * kKnnQueries + q;
    bool valid = (eq < E) && (sh.res[q][0] != 0);
    float nx[5], ny[5], nz[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { nx[j] = sh.nn[q][j * 3]; ny[j] = sh.nn[q][j * 3 + 1]; nz[j] = sh.nn[q][j * 3 + 2]; }
    if (valid) valid = line_gate(nx, ny, nz);
    if (eq < E) {
      float4* ca = v.corr_a + (size_t)s * v.edge_cap + eq;
      float4* cb = v.corr_b + (size_t)s * v.edge_cap + eq;
      int2* cidx = v.corr_idx + ((size_t)s * 2 + outer_it) * v.edge_cap + eq;
      const float4 oa = valid ? make_float4(nx[0], ny[0], nz[0], 1.0f) : make_float4(0, 0, 0, 0);              // :351-353
      const float4 ob = valid ? make_float4(nx[1], ny[1], nz[1], 0.0f) : make_float4(0, 0, 0, 0);              // :355-357
      const int2 oi = valid ? make_int2(sh.res[q][1], sh.res[q][2]) : make_int2(-1, -1);
      if (kPre) {
        // (overlapped pass: the finalising solve's launch is already running on other XCDs — write-through stores)
        wt_store_f4(ca, oa); wt_store_f4(cb, ob);
        wt_store_u64(cidx, ((unsigned long long)(unsigned int)oi.y << 32) | (unsigned int)oi.x);
      } else {
        *ca = oa; *cb = ob; *cidx = oi;
      }
    }
    const unsigned long long vb = __ballot(valid);
    const int nvalid = __popcll(vb);
    if (q == 0) {
      // :346 — with knn_partials the count travels as entry 29 of the workgroup's partial sums (no same-address atomic of
      // every workgroup: hot-address atomics delay whatever else maps to that memory channel by microseconds)
      if (nvalid && !v.knn_partials) atomicAdd(&st.info.matches[outer_it], nvalid);
      unsigned char* cm = &v.corr_mask[((size_t)s * 2 + outer_it) * v.knn_blocks + bv];   // bit q = query q accepted
      if (kPre) wt_store_u8(cm, (unsigned char)vb); else *cm = (unsigned char)vb;
    }
    sh.res[q][3] = valid ? 1 : 0;
  } else if (kKnnThreads > 64 && v.knn_partials && threadIdx.x >= 64 && threadIdx.x < 64 + kKnnQueries) {
    // The solve that follows starts at (param_q, param_t) — Ceres evaluates the residuals with the quaternion,
    // not with the matrix the neighbours were searched with (:186-195,205-206) — which is already known here.
    // So the residual block of every query that found five neighbours is evaluated right away and the accepted
    // ones are summed per workgroup: k_lm_solve's first evaluation becomes a reduction of these partial sums
    // instead of a pass over all correspondences.  One lane per query on the SECOND wave, beside the line gates
    // of the first (the block does not depend on the gate's verdict; it is simply dropped if the gate says no).
    const int q = threadIdx.x - 64;
    const int eq = bv * kKnnQueries + q;
    double flag = 0.0;
    if (eq < E && sh.res[q][0] != 0) {
      double Rm[12], pq[4], pt[3];
#pragma unroll
      for (int i = 0; i < 4; i++) pq[i] = kPre ? qt[i] : st.param_q[i];
#pragma unroll
      for (int i = 0; i < 3; i++) pt[i] = kPre ? qt[4 + i] : st.param_t[i];
      iso_from_qt(pq, pt, Rm);
      const float4 pe = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + eq];
      const double p[3] = {(double)pe.x, (double)pe.y, (double)pe.z};       // :347-349 sensor frame
      const double a[3] = {(double)sh.nn[q][0], (double)sh.nn[q][1], (double)sh.nn[q][2]};
      const double b[3] = {(double)sh.nn[q][3], (double)sh.nn[q][4], (double)sh.nn[q][5]};
      double J[18], rs[3], rho0, rho1;
      const bool ok = residual_block(Rm, p, a, b, v.min_range, v.max_range, J, rs, &rho0, &rho1);
#pragma unroll
      for (int i = 0; i < 18; i++) sh.blk[q][i] = J[i];
      sh.blk[q][18] = rs[0]; sh.blk[q][19] = rs[1]; sh.blk[q][20] = rs[2]; sh.blk[q][21] = rho0; sh.blk[q][22] = rho1;
      flag = ok ? 1.0 : 2.0;
    }
    sh.blk[q][23] = flag;
  }
  if (!v.knn_partials) return;                           // (uniform) lock-step batches: the solve evaluates everything itself
  __syncthreads();
  // entry hl of the block's contribution by lane hl of the query's own 32-lane group (J is read from LDS, so
  // the 29-entry accumulator never occupies registers in this kernel)
  if (hl < kAccN) {
    const double flag = sh.res[grp][3] ? sh.blk[grp][23] : 0.0;     // (gate's verdict, block's finiteness)
    double x = 0.0;
    if (flag == 1.0) x = residual_entry(sh.blk[grp], sh.blk[grp] + 18, sh.blk[grp][21], sh.blk[grp][22], hl);
    else if (flag == 2.0 && hl == 28) x = 1.0;            // non-finite block: counted, contributes nothing else
    sh.part[grp][hl] = x;
  } else if (hl == kAccN) {
    sh.part[grp][hl] = sh.res[grp][3] ? 1.0 : 0.0;      // entry 29: accepted correspondences (:346)
  }
  __syncthreads();
  if (threadIdx.x <= kAccN) {
    double x = 0.0;
#pragma unroll
    for (int q = 0; q < kKnnQueries; q++) x += sh.part[q][threadIdx.x];      // fixed order: deterministic
    double* dst = &v.knn_part[(((size_t)s * 2 + outer_it) * v.knn_blocks + bv) * 32 + threadIdx.x];
    if (kPre) wt_store_u64(dst, (unsigned long long)__double_as_longlong(x)); else *dst = x;
  }
  DBG_STAMP(v, dbgb, 1, 7); DBG_QSTAMP(7);
  if ((kInstrument && (v.debug & 64)) && s == 0 && threadIdx.x == 0) {      // histogram of workgroup durations, 1 us bins
    const unsigned long long d = wall_clock64() - t_blk;
    const int bin = (int)(d / 100ull);
    atomicAdd(&v.dbg_clk[192 + (bin < 63 ? bin : 63)], 1ull);
  }
}

// Overlapped second pass: line gates and partial sums of the workgroup's two query blocks side by side (the same steps as
// the tail of knn_block, which runs them for one block: there waves 2 and 3 idle while lanes 0..7 of wave 0 run the
// gates and lanes 0..7 of wave 1 the residual blocks; here block A uses waves 0 / 1 and block B waves 2 / 3 — after the
// first solve's result has arrived this tail IS the launch's critical path).  Results leave as write-through stores.
template <int kKnnThreads>
__device__ __forceinline__ void knn_tail_dual(const DevView& v, int s, int outer_it, int eb, int bvA, int bvB, bool haveB, int E,
                                              KnnShared<kKnnThreads / kKnnGroup>& shA, KnnShared<kKnnThreads / kKnnGroup>& shB, const double* qt) {
  constexpr int kKnnQueries = kKnnThreads / kKnnGroup;
  static_assert(kKnnThreads == 256, "two blocks of 8 queries on four waves");
  const int grp = threadIdx.x / kKnnGroup, hl = threadIdx.x & (kKnnGroup - 1);
  const int half = (int)threadIdx.x >> 7, t = (int)threadIdx.x & 127;
  KnnShared<kKnnQueries>& sh = half ? shB : shA;
  const int bv = half ? bvB : bvA;
  const bool live = half ? haveB : true;
  if (t < kKnnQueries && live) {
    const int q = t;
    const int eq = bv * kKnnQueries + q;
    bool valid = (eq < E) && (sh.res[q][0] != 0);
    float nx[5], ny[5], nz[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { nx[j] = sh.nn[q][j * 3]; ny[j] = sh.nn[q][j * 3 + 1]; nz[j] = sh.nn[q][j * 3 + 2]; }
    if (valid) valid = line_gate(nx, ny, nz);                                                                 // :325-344
    if (eq < E) {
      const float4 oa = valid ? make_float4(nx[0], ny[0], nz[0], 1.0f) : make_float4(0, 0, 0, 0);              // :351-353
      const float4 ob = valid ? make_float4(nx[1], ny[1], nz[1], 0.0f) : make_float4(0, 0, 0, 0);              // :355-357
      const int2 oi = valid ? make_int2(sh.res[q][1], sh.res[q][2]) : make_int2(-1, -1);
      wt_store_f4(v.corr_a + (size_t)s * v.edge_cap + eq, oa);
      wt_store_f4(v.corr_b + (size_t)s * v.edge_cap + eq, ob);
      wt_store_u64(v.corr_idx + ((size_t)s * 2 + outer_it) * v.edge_cap + eq, ((unsigned long long)(unsigned int)oi.y << 32) | (unsigned int)oi.x);
    }
    const unsigned long long vb = __ballot(valid);
    if (q == 0) wt_store_u8(&v.corr_mask[((size_t)s * 2 + outer_it) * v.knn_blocks + bv], (unsigned char)vb);   // bit q = query q accepted
    sh.res[q][3] = valid ? 1 : 0;
  } else if (t >= 64 && t < 64 + kKnnQueries && live) {
    // the residual block of every query that found five neighbours, at the finalising solve's start point (see knn_block)
    const int q = t - 64;
    const int eq = bv * kKnnQueries + q;
    double flag = 0.0;
    if (eq < E && sh.res[q][0] != 0) {
      double Rm[12], pq[4], pt[3];
#pragma unroll
      for (int i = 0; i < 4; i++) pq[i] = qt[i];
#pragma unroll
      for (int i = 0; i < 3; i++) pt[i] = qt[4 + i];
      iso_from_qt(pq, pt, Rm);
      const float4 pe = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + eq];
      const double p[3] = {(double)pe.x, (double)pe.y, (double)pe.z};       // :347-349 sensor frame
      const double a[3] = {(double)sh.nn[q][0], (double)sh.nn[q][1], (double)sh.nn[q][2]};
      const double b[3] = {(double)sh.nn[q][3], (double)sh.nn[q][4], (double)sh.nn[q][5]};
      double J[18], rs[3], rho0, rho1;
      const bool ok = residual_block(Rm, p, a, b, v.min_range, v.max_range, J, rs, &rho0, &rho1);
#pragma unroll
      for (int i = 0; i < 18; i++) sh.blk[q][i] = J[i];
      sh.blk[q][18] = rs[0]; sh.blk[q][19] = rs[1]; sh.blk[q][20] = rs[2]; sh.blk[q][21] = rho0; sh.blk[q][22] = rho1;
      flag = ok ? 1.0 : 2.0;
    }
    sh.blk[q][23] = flag;
  }
  __syncthreads();
  // entry hl of every block's contribution, by lane hl of the 32-lane group with the query's number (both blocks)
#pragma unroll
  for (int b = 0; b < 2; b++) {
    if (b == 1 && !haveB) break;
    KnnShared<kKnnQueries>& shb = b ? shB : shA;
    if (hl < kAccN) {
      const double flag = shb.res[grp][3] ? shb.blk[grp][23] : 0.0;     // (gate's verdict, block's finiteness)
      double x = 0.0;
      if (flag == 1.0) x = residual_entry(shb.blk[grp], shb.blk[grp] + 18, shb.blk[grp][21], shb.blk[grp][22], hl);
      else if (flag == 2.0 && hl == 28) x = 1.0;            // non-finite block: counted, contributes nothing else
      shb.part[grp][hl] = x;
    } else if (hl == kAccN) {
      shb.part[grp][hl] = shb.res[grp][3] ? 1.0 : 0.0;      // entry 29: accepted correspondences (:346)
    }
  }
  __syncthreads();
  if (t <= kAccN && live) {
    double x = 0.0;
#pragma unroll
    for (int q = 0; q < kKnnQueries; q++) x += sh.part[q][t];      // fixed order: deterministic
    wt_store_u64(&v.knn_part[(((size_t)s * 2 + outer_it) * v.knn_blocks + bv) * 32 + t], (unsigned long long)__double_as_longlong(x));
  }
}

// grid.x = v.knn_grid workgroups per stream (+ the streamed rebuild's ALLOC workgroups on the second pass): workgroup b
// takes the query blocks b, b + knn_grid, ... below ceil(E / queries) — the grid is sized for the usual edge count
// (half of the capacity), not for edge_cap: on lock-step batches two thirds of an edge_cap-sized grid were workgroups
// that found nothing to do.
// kOv: the overlapped second pass (see "Overlapped second kNN pass" above; one-stream handles, the 256-thread instance):
// launched on stream_k beside the scan's first solve, seq = the launch sequence number the flags carry.
template <int kKnnThreads, bool kOv>
__device__ __forceinline__ void knn_pass(const DevView& v, int s, int bxi, int byi, int outer_it, int eb, unsigned int wait_edges,
                                         unsigned int signal_odo, unsigned int seq, KnnShared<kKnnThreads / kKnnGroup>& sh, KnnShared<kKnnThreads / kKnnGroup>& sh2, double* sh_ov) {
  constexpr int kKnnQueries = kKnnThreads / kKnnGroup;
  StreamState& st = v.state[s];
  if (kOv) {
    // the scan's first solve launch has started: the first kNN pass (and everything before it) has completed
    // (k_ov_gate in front of this launch has seen the flag already: the launch started, with clean caches, after the first pass ended)
    if (!pipe_wait(v.ov_flags + s, seq, &st.status)) return;
    OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 9); OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 13);
  } else if (v.early_rebuild) {
    if (bxi >= v.knn_grid) { if (outer_it == 1) rebuild_alloc(v, s, st, bxi - v.knn_grid, (int)gridDim.x - v.knn_grid); return; }
    // streamed rebuild, bookkeeping before the first builders start (next launch): the frame count the build refers to
    // (the finalising solve advances it beside them), an empty list of occupied slots for the table being built, and the
    // prediction the scan starts from
    if (outer_it == 0 && bxi == 0 && threadIdx.x == 0) {
      st.reb_frame_count = st.frame_count; st.n_used_tab[(st.frame_count + 1) & 1] = 0; st.reb_initialized = st.initialized;
    }
    if (outer_it == 0 && bxi == 0 && threadIdx.x >= 64 && threadIdx.x < 76) st.pred_odom[threadIdx.x - 64] = st.odom[threadIdx.x - 64];
  }
  if (!kOv && outer_it == 0) {
    // (pipelined replay) this launch follows odometry `signal_odo` in stream order: that odometry has completed entirely;
    // and the extraction that fills edge buffer eb (other stream) must have completed before anything of it is read
    if (signal_odo && bxi == 0 && byi == 0 && threadIdx.x == 0) {
      typedef __attribute__((address_space(1))) unsigned int gu32;
      __hip_atomic_store((gu32*)(v.pipe_flags + kEdgePipeBufs), signal_odo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wait_edges && !pipe_wait(v.pipe_flags + eb, wait_edges, &st.status)) return;
  }
  // The block's first loads — its edge, the pose — leave together with the stream's state words instead of behind the
  // branches on them (one memory round trip less on the launch's critical path; the edge index is clamped, an unused
  // edge costs nothing).
  typedef KnnTune<(kKnnThreads >= 256)> Tune;
  const int e_first = bxi * kKnnQueries + (int)(threadIdx.x / kKnnGroup);
  float4 p_first = make_float4(0.f, 0.f, 0.f, 0.f);
  double T[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (Tune::kHoistLoads) {
    p_first = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (e_first < v.edge_cap ? e_first : v.edge_cap - 1)];
    if (!kOv) {
#pragma unroll
      for (int i = 0; i < 12; i++) T[i] = st.odom[i];
    }
  }
  const unsigned int st_status = st.status;
  const int st_init = st.initialized;
  const int E = st.n_edges_buf[eb];
  if (st_status & LIODOM_STATUS_PIPE_TIMEOUT) return;      // (uniform) a wait of this handle gave up: the edge buffer may be incomplete
  if (!st_init) return;                            // uniform over the workgroup
  // (two explicit calls, not a loop over bv: as a loop body the block needs 160 VGPRs instead of 69)
  static_assert(kKnnGridDiv == 2, "k_knn handles exactly two query blocks per workgroup");
  if (bxi * kKnnQueries >= E) {             // no query here: empty validity bytes for the solve's compaction
    if (threadIdx.x == 0) {
      unsigned char* cm = &v.corr_mask[((size_t)s * 2 + outer_it) * v.knn_blocks + bxi];
      if (kOv) wt_store_u8(cm, 0); else *cm = 0;
      if (bxi + v.knn_grid < v.knn_blocks) { if (kOv) wt_store_u8(cm + v.knn_grid, 0); else cm[v.knn_grid] = 0; }
    }
    return;
  }
  const int bv2 = bxi + v.knn_grid;
  const int e_second = bv2 * kKnnQueries + (int)(threadIdx.x / kKnnGroup);
  const bool second = bv2 < v.knn_blocks && bv2 * kKnnQueries < E;
  KnnPre pre1, pre2;
  float4 p_second = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kOv) {
    // everything the two blocks need apart from the solve's result; then wait for that
    if (second) p_second = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (e_second < v.edge_cap ? e_second : v.edge_cap - 1)];
    knn_presearch<kKnnThreads>(v, s, st, e_first, E, sh, pre1);
    if (second) knn_presearch<kKnnThreads>(v, s, st, e_second, E, sh2, pre2);
    else pre2.gsq = 0.f;
    if (!ov_wait_pose(v, s, bxi % kOvReplicas, seq, sh_ov, &st.status)) return;
    OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 10); OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 14);
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = sh_ov[i];
  }
  if constexpr (kOv) {
    // both blocks' queries, then their gates and partial sums side by side
    knn_block<kKnnThreads, true, false>(v, s, st, outer_it, eb, bxi, E, sh, p_first, T, pre1, sh_ov + 12);
    if (second) knn_block<kKnnThreads, true, false>(v, s, st, outer_it, eb, bv2, E, sh2, p_second, T, pre2, sh_ov + 12);
    else if (bv2 < v.knn_blocks && threadIdx.x == 0) wt_store_u8(&v.corr_mask[((size_t)s * 2 + outer_it) * v.knn_blocks + bv2], 0);
    __syncthreads();
    knn_tail_dual<kKnnThreads>(v, s, outer_it, eb, bxi, bv2, second, E, sh, sh2, sh_ov + 12);
    return;
  }
  knn_block<kKnnThreads, false>(v, s, st, outer_it, eb, bxi, E, sh, p_first, T, pre1, nullptr);
  if (bv2 >= v.knn_blocks) return;
  if (!second) {
    if (threadIdx.x == 0) { unsigned char* cm = &v.corr_mask[((size_t)s * 2 + outer_it) * v.knn_blocks + bv2]; if (kOv) wt_store_u8(cm, 0); else *cm = 0; }
    return;
  }
  __syncthreads();                          // (the second block reuses the LDS)
  double T2[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (reloaded: kept live across the first block the pose would cost 24 VGPRs)
  if (Tune::kHoistLoads) {
    p_second = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (e_second < v.edge_cap ? e_second : v.edge_cap - 1)];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 12; i++) T2[i] = st.odom[i];
  }
  knn_block<kKnnThreads, false>(v, s, st, outer_it, eb, bv2, E, sh, p_second, T2, pre2, nullptr);
}

template <int kKnnThreads, bool kOv = false>
__global__ __launch_bounds__(kKnnThreads, (kKnnThreads >= 256 ? 1 : LIODOM_TUNE_B_WAVES)) void k_knn(DevView v, int s0, int outer_it, int eb, unsigned int wait_edges, unsigned int signal_odo, unsigned int seq) {
  constexpr int kKnnQueries = kKnnThreads / kKnnGroup;
  __shared__ KnnShared<kKnnQueries> shs[kOv ? 2 : 1];      // (overlapped pass: one per query block — their tails run side by side)
  KnnShared<kKnnQueries>& sh = shs[0];
  __shared__ double sh_ov[kOv ? 20 : 1];       // overlapped pass: the first solve's odom[12], q[4], t[3]
  int bxi = (int)blockIdx.x, byi = (int)blockIdx.y;
  xcd_remap(bxi, byi);
  const int s = s0 + byi;
  if (kOv) { OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 8); OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 12); }
  else if (outer_it == 0) OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 16);
  knn_pass<kKnnThreads, kOv>(v, s, bxi, byi, outer_it, eb, wait_edges, signal_odo, seq, sh, shs[kOv ? 1 : 0], sh_ov);
  if (kOv) ov_signal_knn_done(v, s, bxi, seq);       // (every exit of the pass is workgroup-uniform)
  if (kOv) { OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 11); OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 15); }
  else if (outer_it == 0) OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 17);
}

// One wave in front of the overlapped pass on stream_k: the pass's workgroups must not become resident before the first
// solve's launch is (k_lm_solve needs CUs whose registers are all free — 2 waves x 256 VGPRs per SIMD — and 352 polling
// k_knn workgroups leave none: the solve could not start, the pass would wait for it forever).  The launch behind this
// gate starts when it retires, i.e. once the solve's workgroups are on their CUs.
__global__ void k_ov_gate(DevView v, int s, unsigned int seq) {
  OV_STAMP(v, threadIdx.x == 0, 6);
  (void)pipe_wait(v.ov_flags + s, seq, &v.state[s].status);
  OV_STAMP(v, threadIdx.x == 0, 7);
}

// k_line_gate (lock-step batches): the line gate of laser_odometry.cc:325-344 for the queries of one kNN pass, one query
// per lane; writes the correspondences (:351-357), counts the matches (:346) and leaves the validity bytes the solve's
// compaction reads (bit q of byte b = query q of k_knn workgroup b).
__global__ __launch_bounds__(256) void k_line_gate(DevView v, int s0, int outer_it, int eb) {
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  if (!st.initialized) return;
  const int E = st.n_edges_buf[eb];
  const int eq = blockIdx.x * 256 + threadIdx.x;
  const int Q = v.knn_queries;
  if (eq >= v.knn_blocks * Q) return;                   // (whole waves: knn_blocks * Q is a multiple of 16... see below)
  float nx[5], ny[5], nz[5];
  int found = 0, i0 = -1, i1 = -1;
  if (eq < E) {
    const float4* k = v.knn_nn + ((size_t)s * v.edge_cap + eq) * 5;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const float4 m = k[j];
      nx[j] = m.x; ny[j] = m.y; nz[j] = m.z;
      if (j == 0) found = __float_as_int(m.w);
      if (j == 1) i0 = __float_as_int(m.w);
      if (j == 2) i1 = __float_as_int(m.w);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 5; j++) { nx[j] = 0.f; ny[j] = 0.f; nz[j] = 0.f; }
  }
  bool valid = (eq < E) && (found != 0);
  if (valid) valid = line_gate(nx, ny, nz);
  if (eq < E) {
    float4* ca = v.corr_a + (size_t)s * v.edge_cap + eq;
    float4* cb = v.corr_b + (size_t)s * v.edge_cap + eq;
    int2* cidx = v.corr_idx + ((size_t)s * 2 + outer_it) * v.edge_cap + eq;
    if (valid) {
      *ca = make_float4(nx[0], ny[0], nz[0], 1.0f);              // :351-353
      *cb = make_float4(nx[1], ny[1], nz[1], 0.0f);              // :355-357
      *cidx = make_int2(i0, i1);
    } else {
      *ca = make_float4(0, 0, 0, 0); *cb = make_float4(0, 0, 0, 0); *cidx = make_int2(-1, -1);
    }
  }
  const unsigned long long vb = __ballot(valid);
  const int lane = threadIdx.x & 63;
  if (lane == 0) { const int nvalid = __popcll(vb); if (nvalid) atomicAdd(&st.info.matches[outer_it], nvalid); }   // :346
  if ((lane % Q) == 0) v.corr_mask[((size_t)s * 2 + outer_it) * v.knn_blocks + eq / Q] = (unsigned char)((vb >> lane) & ((1ull << Q) - 1ull));
}

// =============================================================================================
// k_lm_solve: one 512-thread workgroup per stream runs the whole Ceres-style solve.
//   eval: every thread accumulates the 29-entry normal-equation accumulator over its edges
//   (fused residual + analytic Jacobian + Huber), wavefront shfl butterfly, then a fixed-order
//   cross-wave sum through LDS (deterministic, no atomics, no MFMA: this is a 6x6 reduction).
//   Thread 0 runs the LM controller (liodom_math.h) between evaluations.
//   finalize (second outer iteration, or the very first frame): pose log, constant-velocity
//   prediction for the next scan, window bookkeeping, hash-generation counters.
// =============================================================================================
// Indices of the edges with an accepted correspondence, in edge order (deterministic), built once
// per solve in LDS so that every evaluation runs over C dense items instead of E sparse ones.
// dynamic LDS of k_lm_solve: index list + reduction scratch (full transposed matrix if it fits the
// 160 KB of a CU next to ~3 KB of static LDS, else one partial per 16-lane row)
__host__ __device__ __forceinline__ bool lm_lds_reduce_fits(int edge_cap) {
  return (size_t)((edge_cap + 3) & ~3) * sizeof(int) + (size_t)kAccN * kLmEvalThreads * sizeof(double) + 8192 <= 160 * 1024;
}
__host__ __device__ __forceinline__ size_t lm_lds_bytes(int edge_cap) {
  return (size_t)((edge_cap + 3) & ~3) * sizeof(int) +
         (lm_lds_reduce_fits(edge_cap) ? (size_t)kAccN * kLmEvalThreads : (size_t)(kLmThreads / 16) * kAccN) * sizeof(double);
}


// Compaction of the accepted correspondences from the validity bytes k_knn left (bit q of byte b = query q of
// k_knn workgroup b): edge indices in ascending order into idx[].  Called by every evaluator wave on its own —
// each writes the same values, so no cross-wave synchronisation is needed before a wave reads its entries.
__device__ int lm_compact_bits(const DevView& v, int s, int outer_it, int E, int* idx /*LDS [edge_cap]*/) {
  const int lane = threadIdx.x & 63;
  const int Q = v.knn_queries;
  const int nb = (E + Q - 1) / Q;                          // k_knn workgroups that had queries
  const int nwords = (nb + 3) >> 2;
  const unsigned int* mw = reinterpret_cast<const unsigned int*>(v.corr_mask + ((size_t)s * 2 + outer_it) * v.knn_blocks);
  int run = 0;
  for (int w0 = 0; w0 < nwords; w0 += 64) {
    const int w = w0 + lane;
    unsigned int word = (w < nwords) ? mw[w] : 0u;
    const int pop = __popc(word);
    const int incl = wave_incl_scan_i32(pop);
    int o = run + incl - pop;
    while (word) {
      const int b = __ffs(word) - 1;
      word &= word - 1u;
      const int bit = w * 32 + b;
      idx[o++] = (bit >> 3) * Q + (bit & 7);
    }
    run += readlane_i32(incl, 63);
  }
  return run;
}

// The correspondences of a solve do not change between its evaluations: every evaluator thread keeps its
// first kLmCached triples (p, a, b) in registers (loaded once by lm_cache_load), so an evaluation
// of up to kLmCached * kLmEvalThreads blocks touches no memory before the reduction.
constexpr int kLmCached = 1;
struct LmCache { float4 P[kLmCached], A[kLmCached], B[kLmCached]; };
__device__ __forceinline__ void lm_cache_load(const DevView& v, int s, int eb, int c_lo, int c_hi, const int* idx, LmCache& k) {
  const float4* ed = v.edges + ((size_t)eb * v.n_streams + s) * v.edge_cap;
  const float4* ca = v.corr_a + (size_t)s * v.edge_cap;
  const float4* cb = v.corr_b + (size_t)s * v.edge_cap;
  const int et = (int)threadIdx.x;
#pragma unroll
  for (int j = 0; j < kLmCached; j++) {
    const int c = c_lo + et + j * kLmEvalThreads;
    // (the controller's wave fetches its blocks inside every evaluation)
    if (et < kLmCtl && c < c_hi) { const int e = idx[c]; k.A[j] = ca[e]; k.B[j] = cb[e]; k.P[j] = ed[e]; }
  }
}

// Evaluation of the blocks c_lo .. c_hi of the compacted list by the evaluator waves, then the reduction by
// everybody.  part: [kAccN][kLmEvalThreads] or [kLmThreads/16][kAccN].
__device__ __forceinline__ void lm_eval(const DevView& v, int s, int eb, int c_lo, int c_hi, const int* idx, const double* Rm_sh,
                                        double* part, double* acc_out /*[kAccN]*/, const LmCache& k) {
  const int et = (int)threadIdx.x;
  const bool cached = et < kLmCtl;
  double acc[kAccN];
#pragma unroll
  for (int i = 0; i < kAccN; i++) acc[i] = 0.0;
  {
    double Rm[12];
#pragma unroll
    for (int i = 0; i < 12; i++) Rm[i] = Rm_sh[i];
    const float4* ed = v.edges + ((size_t)eb * v.n_streams + s) * v.edge_cap;
    const float4* ca = v.corr_a + (size_t)s * v.edge_cap;
    const float4* cb = v.corr_b + (size_t)s * v.edge_cap;
    int c = c_lo + et;
    if (cached) {
#pragma unroll
      for (int j = 0; j < kLmCached; j++, c += kLmEvalThreads) {
        if (c < c_hi) {
          const double p[3] = {(double)k.P[j].x, (double)k.P[j].y, (double)k.P[j].z};     // :347-349 sensor frame
          const double a[3] = {(double)k.A[j].x, (double)k.A[j].y, (double)k.A[j].z};
          const double b[3] = {(double)k.B[j].x, (double)k.B[j].y, (double)k.B[j].z};
          residual_accumulate(Rm, p, a, b, v.min_range, v.max_range, acc);
        }
      }
    }
    for (; c < c_hi; c += kLmEvalThreads) {
      const int e = idx[c];
      const float4 A = ca[e];
      const float4 B = cb[e];
      const float4 P = ed[e];
      const double p[3] = {(double)P.x, (double)P.y, (double)P.z};
      const double a[3] = {(double)A.x, (double)A.y, (double)A.z};
      const double b[3] = {(double)B.x, (double)B.y, (double)B.z};
      residual_accumulate(Rm, p, a, b, v.min_range, v.max_range, acc);
    }
  }
  if (v.lm_lds_reduce) {
    // Reduction through LDS, transposed: every evaluator stores its 29 partial sums as column et of
    // red[29][kLmEvalThreads] (conflict-free 8-byte stores); then thread (v, r) = (t / 16, t % 16) sums
    // the elements r, r + 16, r + 32, ... of row v (conflict-free loads, 28 adds), a 4-step DPP row
    // sum finishes row v.  Fixed order -> deterministic, no atomics.
    // (only the columns of threads that hold a block — with several workgroups per solve about half of them — rounded up
    //  to whole 16-lane rows: the other threads' partial sums are zero and neither written nor read)
    const int nb_here = c_hi - c_lo;
    const int ncol = ((nb_here < kLmEvalThreads ? nb_here : kLmEvalThreads) + 15) & ~15;
    if (et < ncol) {
#pragma unroll
      for (int i = 0; i < kAccN; i++) part[i * kLmEvalThreads + et] = acc[i];
    }
    __syncthreads();
    const int vrow = threadIdx.x >> 4, r = threadIdx.x & 15;
    double x = 0.0;
    if (vrow < kAccN) {
      const double* rowp = part + vrow * kLmEvalThreads + r;
      const int nk = ncol >> 4;
#pragma unroll 8
      for (int kk = 0; kk < nk; kk++) x += rowp[kk * 16];
    }
    x = row_sum_f64(x);
    if (vrow < kAccN && r == 0) acc_out[vrow] = x;
    __syncthreads();
    return;
  }
  // Large edge capacities (the matrix no longer fits beside the index list): DPP butterfly
  // inside each 16-lane row, one partial per row into LDS, then a fixed-order sum of the partials.
#pragma unroll
  for (int i = 0; i < kAccN; i++) acc[i] = row_sum_f64(acc[i]);
  const int row = threadIdx.x >> 4;
  if ((threadIdx.x & 15) == 0) {
#pragma unroll
    for (int i = 0; i < kAccN; i++) part[row * kAccN + i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < kAccN) {
    double x = 0.0;
    for (int w = 0; w < kLmThreads / 16; w++) x += part[w * kAccN + threadIdx.x];
    acc_out[threadIdx.x] = x;
  }
  __syncthreads();
}

// Resets the hash slots occupied by the build that this scan searched (list used_cells[0 .. nup));
// the last kNN pass of the scan has completed before the finalising k_lm_solve launch starts.
__device__ void hash_clear_used(const DevView& v, int s /*table: stream + parity * n_streams*/, int nup, int t, int nt) {
  CellSlot empty; empty.key = kEmptyKey; empty.start = 0; empty.cnt = 0;
  const int* used = v.used_cells + (size_t)s * v.used_cap;
  for (int u0 = t; u0 < nup; u0 += 8 * nt) {   // 8 index loads in flight per thread
    int hh[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int u = u0 + k * nt; hh[k] = (u < nup) ? used[u] : -1; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (hh[k] >= 0) {
        const size_t ti = (size_t)s * v.table_size + hh[k];
        v.cells[ti] = empty;
        v.cell_bits[ti >> 5] = 0u;   // every set bit of that word belongs to a slot of this list
        if (v.cell_pad) v.cell_pad[ti] = 0u;
      }
    }
  }
}

__device__ __forceinline__ void publish_final_pose(const DevView& v, int s, const double* T, int raw, unsigned int tag, int lane);

// Called by the whole workgroup.  sh_cnt: LDS scratch of kMaxFrames + 1 ints.
// Thread 64 publishes the result (pose log, host-mapped record) while thread `ctl` (the one that wrote st.odom)
// computes the prediction and the window bookkeeping; the remaining threads fetch the frame sizes.
__device__ void finalize_scan(const DevView& v, int s, StreamState& st, int* sh_cnt, int eb, bool clear_hash, int ctl) {
  const int P = v.prev_frames;
  const int tid = threadIdx.x;
  // LocalMapManager::addPointCloud (:34-60) on a ring of P frame slots: the new frame goes
  // into slot frame_count % P (overwriting the oldest once the window is full)
  const int fc_new = st.frame_count + 1;
  const int nf = fc_new < P ? fc_new : P;
  const int new_slot = st.frame_count % P;
  int* wn = v.win_n + (size_t)s * P;
  int* wb = v.win_base + (size_t)s * (P + 1);
  int* ws = v.win_slot + (size_t)s * P;
  const int n_edges = st.n_edges_buf[eb];
  const int nup = st.n_used_tab[0];      // cells of the build that this scan searched (cleared below)
  if (tid == ctl) { for (int i = 0; i < 12; i++) st.final_odom[i] = st.odom[i]; }   // (ctl wrote st.odom itself)
  for (int j = tid; j < nf; j += blockDim.x) {           // frame sizes of the new window (nothing here depends on the pose)
    const int sl = (fc_new - nf + j) % P;
    sh_cnt[j] = (sl == new_slot) ? n_edges : wn[sl];     // independent loads, one round trip
    ws[j] = sl;
  }
  __syncthreads();
  // early_rebuild: hand the pose to the workgroups that append the new frame (they have been waiting for it)
  if (v.early_rebuild && tid < 25) publish_final_pose(v, s, st.final_odom, st.append_raw, (unsigned int)st.reb_frame_count + 1u, tid);
  if (tid == 64) {
    // pose as published (laser_odometry.cc:403-412 with identity laser_to_base)
    double q[4];
    quat_from_pose(st.final_odom, v.rotation_mode, q);             // :403 q_current(odom_base_link.rotation())
    const int k = st.scan_counter;
    st.info.scan_index = k;
    st.info.status = st.status;
    if (k < v.pose_log_cap) {
      double* pl = v.pose_log + ((size_t)s * v.pose_log_cap + k) * 7;
      pl[0] = q[0]; pl[1] = q[1]; pl[2] = q[2]; pl[3] = q[3];
      pl[4] = st.final_odom[3]; pl[5] = st.final_odom[7]; pl[6] = st.final_odom[11];
      v.info_log[(size_t)s * v.pose_log_cap + k] = st.info;
    }
    st.scan_counter = k + 1;
    if (v.host_out) {
      // zero-copy publication: payload, system-scope fence, then the sequence word the host polls
      HostOut* ho = v.host_out + (size_t)s * 2 + (k & 1);      // two records per stream: the host may read scan k while scan k + 1 publishes
      ho->pose[0] = q[0]; ho->pose[1] = q[1]; ho->pose[2] = q[2]; ho->pose[3] = q[3];
      ho->pose[4] = st.final_odom[3]; ho->pose[5] = st.final_odom[7]; ho->pose[6] = st.final_odom[11];
      ho->info = st.info;
      // (the system-scope release orders this thread's payload stores before the sequence word: no separate fence)
      __hip_atomic_store(&ho->seq, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    st.info.matches[0] = 0; st.info.matches[1] = 0;   // counters of the next scan's two kNN passes
  }
  if (tid == ctl) {
    // prediction for the next scan: odom * (prev^-1 * odom)   (:148-150)
    double inv[12], rel[12], pred[12];
    iso_inverse(st.prev_odom, inv);
    iso_mul(inv, st.final_odom, rel);
    iso_mul(st.final_odom, rel, pred);
    for (int i = 0; i < 12; i++) { st.prev_odom[i] = st.final_odom[i]; st.odom[i] = pred[i]; }
    quat_from_pose(pred, v.rotation_mode, st.param_q);               // :186-190 q_curr(odom_.rotation())
    st.param_t[0] = pred[3]; st.param_t[1] = pred[7]; st.param_t[2] = pred[11];   // :192-195
    wn[new_slot] = n_edges;
    st.frame_count = fc_new;
    st.n_frames = nf;
    int acc = 0;
    for (int j = 0; j < nf; j++) { const int c = sh_cnt[j]; sh_cnt[j] = acc; acc += c; }
    sh_cnt[nf] = acc;
    st.n_map = acc;
    if (!v.early_rebuild) st.n_used_tab[0] = 0;
    else { st.n_search = acc; st.n_filt = 0; }        // (k_window_insert's job otherwise)
    st.cursor = 0;
  }
  __syncthreads();
  for (int j = tid; j <= nf; j += blockDim.x) wb[j] = sh_cnt[j];
  // (first frame only; in steady state the finalising solve clears the table beside its first
  // controller step instead of extending the kernel by ~4.5 us here)
  if (clear_hash && !v.early_rebuild) hash_clear_used(v, s, nup, tid, (int)blockDim.x);
}

// All-to-all exchange of the 29 partial sums between the G workgroups of a stream, inside the
// launch (MI355X guide, G16 form R2: the data is the flag).  Every double travels as two 8-byte
// granules {epoch tag, 32 data bits}: no fences, no separate flag.  Buffers are
// double-buffered by epoch parity (a workgroup cannot publish epoch e+2 before it has read every
// epoch e+1, which the others publish only after reading epoch e).  Every workgroup adds the G
// partials in the same order and so continues with bit-identical totals.  Spins are bounded.
// Two transports: (memory side, placement independent) relaxed agent-scope stores — sc1, write-through, the line
// leaves the XCD's L2 — and agent-scope loads, ~2-3 us per exchange under load; (local) when the G workgroups sit on
// ONE XCD — they are launched on block indices 0, 8, 16, ... which the dispatcher hands to the same XCD, and every
// exchange carries the workgroups' XCC ids so that this is verified, never assumed — plain stores keep the granules in
// that XCD's L2, where the L1-bypassing loads of the others find them.  The first exchange of a launch always takes the
// memory-side transport and tells whether the later ones may go local.
__device__ __forceinline__ unsigned int xcc_id() { return (unsigned int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xFu; }   // HW_REG_XCC_ID[3:0]
__device__ void lm_exchange(const DevView& v, int s, int g, int G, unsigned int epoch,
                            const double* acc_local, double* acc_total, unsigned int* status, bool local, int* same_xcc /*LDS*/) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  unsigned long long* base = v.lm_xch + ((size_t)s * 2 + (epoch & 1u)) * kLmGroupsMax * 64;
  const int tid = threadIdx.x;
  if (tid <= 2 * kAccN) {
    unsigned long long half;
    if (tid < 2 * kAccN) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(acc_local[tid >> 1]);
      half = (tid & 1) ? (bits >> 32) : (bits & 0xFFFFFFFFull);
    } else {
      half = xcc_id();                                   // granule 58: where this workgroup runs
    }
    const unsigned long long word = ((unsigned long long)epoch << 32) | half;
    if (local) __hip_atomic_store((gu64*)(base + g * 64 + tid), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store: stays in the XCD's L2
    else __hip_atomic_store((gu64*)(base + g * 64 + tid), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < 64) {
    double tot = 0.0;
    unsigned int spins = 0;
    bool same = true;
    while (true) {
      bool ok = true;
      tot = 0.0;
      same = true;
      if (tid <= kAccN) {
        // all 2 G loads in flight at once (a loop over the runtime G waits for every pair: G round trips per poll)
        unsigned long long lo[kLmGroupsMax], hi[kLmGroupsMax];
        const int i0 = tid < kAccN ? 2 * tid : 2 * kAccN, i1 = tid < kAccN ? 2 * tid + 1 : 2 * kAccN;   // lane 29: the XCC ids
#pragma unroll
        for (int gg = 0; gg < kLmGroupsMax; gg++) {
          const int gq = gg < G ? gg : 0;
          lo[gg] = __hip_atomic_load((gu64*)(base + gq * 64 + i0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hi[gg] = __hip_atomic_load((gu64*)(base + gq * 64 + i1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int gg = 0; gg < kLmGroupsMax; gg++) {
          if (gg < G) {
            ok = ok && ((unsigned int)(lo[gg] >> 32) == epoch) && ((unsigned int)(hi[gg] >> 32) == epoch);
            tot += __longlong_as_double((long long)((hi[gg] << 32) | (lo[gg] & 0xFFFFFFFFull)));
            same = same && ((unsigned int)lo[gg] == (unsigned int)lo[0]);
          }
        }
      }
      if (__all(ok)) break;
      if (++spins > 4000000u) { if (tid == 0) atomicOr(status, LIODOM_STATUS_LM_SYNC_TIMEOUT); same = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (tid < kAccN) acc_total[tid] = tot;
    if (tid == kAccN) *same_xcc = same ? 1 : 0;
  }
  __syncthreads();
}

// early_rebuild: the solved pose travels from the solving workgroup to the workgroups that append the new frame inside
// the same launch (MI355X guide, G16 form R2: the data is the flag): 12 doubles as 24 granules {tag, 32 data bits} + one
// granule of flags, relaxed agent-scope stores, one granule per lane (a single thread storing all 25 took 4.7 us);
// tag = frames appended so far + 1 (never 0, the reset value).
__device__ __forceinline__ void publish_final_pose(const DevView& v, int s, const double* T, int raw, unsigned int tag, int lane /*0..24*/) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  unsigned long long* base = v.pose_xch + (size_t)s * 32;
  unsigned int word = (unsigned int)raw;
  if (lane < 24) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(T[lane >> 1]);
    word = (lane & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
  }
  __hip_atomic_store((gu64*)(base + lane), ((unsigned long long)tag << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// use_imu (laser_odometry.cc:152-183): the prediction (made when the previous scan finished) gets
// the roll and pitch of the latest IMU orientation before the first kNN pass; one thread per stream.
__global__ void k_imu_override(DevView v, int s0, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  StreamState& st = v.state[s0 + i];
  if (!st.initialized) return;
  double odom[12], out[12], l2b[12], q[4];
#pragma unroll
  for (int k = 0; k < 12; k++) { odom[k] = st.odom[k]; l2b[k] = v.laser_to_base[k]; }
#pragma unroll
  for (int k = 0; k < 4; k++) q[k] = v.imu_q[(size_t)(s0 + i) * 4 + k];
  imu_override(odom, q, l2b, v.rotation_mode, out);
#pragma unroll
  for (int k = 0; k < 12; k++) st.odom[k] = out[k];
  quat_from_pose(out, v.rotation_mode, st.param_q);                                // :186-190
  st.param_t[0] = out[3]; st.param_t[1] = out[7]; st.param_t[2] = out[11];         // :192-195
}

__device__ void rebuild_beside_solve(const DevView& v, int s, StreamState& st, int eb, int outer_it, int block, int nblocks, unsigned int seq, int* sbase, int* sslot);

__global__ __launch_bounds__(kLmThreads) void k_lm_solve(DevView v, int s0, int outer_it, int eb, unsigned int seq) {
  __shared__ double sh_pose[12];
  __shared__ double sh_acc[kAccN];
  __shared__ LmState lm;
  __shared__ int sh_flag;
  __shared__ int sh_C;
  const int s = s0 + blockIdx.y;
  // G cooperating workgroups per stream, on block indices 0, 8, 16, ... when G > 1 (workgroups are handed to the XCDs
  // round-robin by linear index, so these share an XCD — see lm_exchange); every other block is a rebuild workgroup
  const int G = v.lm_groups, gstride = G > 1 ? 8 : 1, bxl = (int)blockIdx.x;
  const bool is_solver = bxl < G * gstride && (bxl % gstride) == 0;
  const int g = is_solver ? bxl / gstride : G + (bxl < G * gstride ? bxl - (bxl / gstride + 1) : bxl - G);
  StreamState& st = v.state[s];
  __shared__ int sh_cnt[kMaxFrames + 1];
  __shared__ int sh_same_xcc;
  // seq != 0: the scan's second kNN pass runs beside this launch ("Overlapped second kNN pass"): the first solve's launch says
  // that it has started (= the first pass has completed), the finalising one waits for the second pass where it needs it
  OV_STAMP(v, bxl == 0 && threadIdx.x == 0, outer_it == 0 ? 0 : 3);
  if (seq && outer_it == 0 && bxl == 0 && threadIdx.x == 0) {
    typedef __attribute__((address_space(1))) unsigned int gu32;
    __hip_atomic_store((gu32*)(v.ov_flags + s), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (g >= G) {
    if (!v.early_rebuild) return;                  // (filler blocks between the solvers)
    // early_rebuild: the workgroups behind the solve build the next cell hash (see "streamed rebuild" below)
    __shared__ int sh_slot[kMaxFrames];
    rebuild_beside_solve(v, s, st, eb, outer_it, g - G, (int)gridDim.x - G, seq, sh_cnt, sh_slot);
    return;
  }
  __shared__ double sh_loc[kAccN];
  __shared__ double sh_red[16][32];
  extern __shared__ __attribute__((aligned(16))) int sh_idx[];   // [edge_cap] compacted correspondence indices, then the reduction matrix
  double* sh_part = reinterpret_cast<double*>(sh_idx + ((v.edge_cap + 3) & ~3));   // [kAccN][kLmEvalThreads]
  const int tid = threadIdx.x;
  const bool prep = tid < kLmCtl;               // waves 0..6: compaction + register cache while the controller lane works
  if (outer_it == 0 && tid == 0 && g == 0) {     // per-scan diagnostics (matches are counted by k_knn)
    st.info.n_edges = st.n_edges_buf[eb];
    st.info.map_points = st.n_search;
    for (int k = 0; k < 2; k++) {
      st.info.lm[k].iterations = 0; st.info.lm[k].accepted = 0; st.info.lm[k].termination = LM_TERM_NO_RESIDUALS;
      st.info.lm[k].pad = 0; st.info.lm[k].initial_cost = 0.0; st.info.lm[k].final_cost = 0.0;
    }
  }
  if (!st.initialized) {
    // first frame (:108-136): no solve; pose stays identity, edges enter the window raw
    if (outer_it == 1 && g == 0) {
      if (tid == 0) st.append_raw = 1;
      finalize_scan(v, s, st, sh_cnt, eb, true, 0);
      if (tid == 0) st.initialized = 1;
    }
    return;
  }
  // The second kNN pass of this scan has completed when the finalising solve starts, so the cell hash it
  // searched is no longer needed: waves 0..6 reset its occupied slots while the controller lane works on its
  // first update step (they would idle at the barrier otherwise).
  bool clr_pending = outer_it == 1 && g == 0 && prep && !v.early_rebuild;
  auto clear_hash_slots = [&]() {
    hash_clear_used(v, s, st.n_used_tab[0], tid, kLmCtl);
    clr_pending = false;
  };
  const bool dbgb = (s == 0) && (g == 0) && (tid == kLmCtl) && (outer_it == 1);
  const bool dbge = (s == 0) && (g == 0) && (tid == 0) && (outer_it == 1);
  DBG_STAMP(v, dbgb, 2, 0);
  const int E = st.n_edges_buf[eb];
  int nblocks = st.info.matches[outer_it];      // (lock-step batches: counted by k_line_gate; else from k_knn's partial sums below)
  __shared__ int sh_nmatch;
  __shared__ double sh_scale[8];
  const unsigned int epoch0 = ((unsigned int)(st.scan_counter + 1) << 6) | ((unsigned int)outer_it << 5);
  unsigned int n_eval = 0;
  bool xch_local = false;       // the G workgroups were seen on one XCD: exchanges through its L2 (lm_exchange)
  LmCache cache;
  int c_lo = 0, c_hi = 0;
  auto my_share = [&](int C) {                           // this workgroup's contiguous share of the compacted blocks
    const int chunk = (C + G - 1) / G;
    c_lo = g * chunk < C ? g * chunk : C;
    c_hi = (g + 1) * chunk < C ? (g + 1) * chunk : C;
  };
  if (seq && outer_it == 1) ov_wait_knn_done(v, s, seq, &st.status);      // the overlapped second kNN pass has completed
  OV_STAMP(v, g == 0 && tid == 0 && outer_it == 1, 4);
  OV_STAMP(v, g == 0 && tid == 0 && outer_it == 0, 18);
  if (v.knn_partials) {
    // ---- first evaluation = sum of the partial normal equations the k_knn workgroups left, in workgroup order ----
    const int Q = v.knn_queries;
    const int nb = (E + Q - 1) / Q;
    const double* part = v.knn_part + ((size_t)s * 2 + outer_it) * v.knn_blocks * 32;
    const int i = tid & 31, r0 = tid >> 5;               // 16 row classes x 32 columns (29 used)
    double x0 = 0.0, x1 = 0.0;
    if (i <= kAccN) {                                    // (entry 29: the number of accepted correspondences)
      // 16 independent loads in flight per pass (one memory round trip for up to 256 k_knn workgroups)
      for (int rb = r0; rb < nb; rb += 256) {
        double xs[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { const int r = rb + 16 * u; xs[u] = (r < nb) ? part[(size_t)r * 32 + i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; u += 2) { x0 += xs[u]; x1 += xs[u + 1]; }
      }
    }
    sh_red[r0][i] = x0 + x1;
    __syncthreads();
    if (tid <= kAccN) {
      double x = 0.0;
#pragma unroll
      for (int r = 0; r < 16; r++) x += sh_red[r][tid];
      if (tid < kAccN) sh_acc[tid] = x;
      else { sh_nmatch = (int)x; if (g == 0) st.info.matches[outer_it] = (int)x; }      // :346 (sum of small integers: exact)
    }
    __syncthreads();
    nblocks = sh_nmatch;
  } else {
    // ---- lock-step batches: k_knn leaves only the validity bytes; compaction, then an ordinary first evaluation ----
    if (prep) {
      const int C = lm_compact_bits(v, s, outer_it, E, sh_idx);
      if (tid == 0) sh_C = C;
    } else if (tid == kLmCtl) {
      iso_from_qt(st.param_q, st.param_t, sh_pose);
    }
    __syncthreads();
    my_share(sh_C);
    lm_cache_load(v, s, eb, c_lo, c_hi, sh_idx, cache);
    if (G > 1) { lm_eval(v, s, eb, c_lo, c_hi, sh_idx, sh_pose, sh_part, sh_loc, cache); lm_exchange(v, s, g, G, epoch0 | ++n_eval, sh_loc, sh_acc, &st.status, xch_local, &sh_same_xcc); xch_local = sh_same_xcc != 0; }
    else lm_eval(v, s, eb, c_lo, c_hi, sh_idx, sh_pose, sh_part, sh_acc, cache);
  }
  DBG_STAMP(v, dbgb, 2, 2);
  // ---- trust-region loop.  Controller step on lane 0 of the last wave; beside it waves 0..6 prepare the
  // evaluations (step 0: validity bytes -> index list, triples into registers) or reset the hash slots of the
  // build this scan searched (step 1 of the finalising solve) ----
  int dbg_it = 0;
  for (int step = 0;; step++) {
    if (step == 0 && tid > kLmCtl && tid <= kLmCtl + 6) {
      // the six Jacobi scales of lm_begin (an FP64 square root and a division each) on six lanes of the controller's wave
      const int j = tid - kLmCtl - 1;
      sh_scale[j] = 1.0 / (1.0 + sqrt(sh_acc[7 + h_idx(j, j)]));
    }
    if (step == 0) __builtin_amdgcn_wave_barrier();
    if (tid == kLmCtl) {
      // (a wave-parallel controller — lane 8 r + c holding entry (r, c) of the 6 x 6 matrices, Cholesky columns
      // broadcast through LDS, solves on readlane'd entries — was measured slower than this single lane:
      // 3.9-5.9 us per step against 3.1)
      const int f = step == 0 ? lm_begin(lm, st.param_q, st.param_t, sh_acc, nblocks, v.apply_on_ftol, sh_scale) : lm_update(lm, sh_acc);
      sh_flag = f;
      if (f == LM_NEED_EVAL) iso_from_qt(lm.cand_q, lm.cand_t, sh_pose);
      if (step == 0) DBG_STAMP(v, dbgb, 2, 23);
    } else if (!prep) {
      // (the other lanes of the controller's wave wait at the barrier)
    } else if (step == 0 && v.knn_partials) {
      DBG_STAMP(v, dbge, 2, 24);
      const int C = lm_compact_bits(v, s, outer_it, E, sh_idx);
      if (tid == 0) sh_C = C;
      DBG_STAMP(v, dbge, 2, 25);
      my_share(C);
      lm_cache_load(v, s, eb, c_lo, c_hi, sh_idx, cache);
    } else if (clr_pending) {
      clear_hash_slots();
    }
    __syncthreads();
    if (step == 0) { if (v.knn_partials && !prep) my_share(sh_C); DBG_STAMP(v, dbgb, 2, 3); }
    else { DBG_STAMP(v, dbgb && dbg_it < 5, 2, 5 + 2 * dbg_it); dbg_it++; }
    if (sh_flag != LM_NEED_EVAL) break;
    if (G > 1) { lm_eval(v, s, eb, c_lo, c_hi, sh_idx, sh_pose, sh_part, sh_loc, cache); DBG_STAMP(v, dbgb && dbg_it < 4, 2, 12 + dbg_it); lm_exchange(v, s, g, G, epoch0 | ++n_eval, sh_loc, sh_acc, &st.status, xch_local, &sh_same_xcc); xch_local = sh_same_xcc != 0; }
    else lm_eval(v, s, eb, c_lo, c_hi, sh_idx, sh_pose, sh_part, sh_acc, cache);
    DBG_STAMP(v, dbgb && dbg_it < 5, 2, 4 + 2 * dbg_it);
  }
  if (clr_pending) clear_hash_slots();                   // (the solve ended at its first step)
  DBG_STAMP(v, dbgb, 2, 20);
  if ((kInstrument && (v.debug & 32)) && dbgb) v.dbg_clk[2 * 32 + 27] = (xch_local ? 100ull : 0ull) + 10ull * xcc_id() + (unsigned long long)n_eval;   // (debug) exchange transport, XCC, evaluations
  if (g != 0) return;        // every workgroup reached the same result; workgroup 0 records it
  if (seq && outer_it == 0) {
    // overlapped second kNN pass: its workgroups are waiting for exactly these 19 doubles — they leave first
    __shared__ double sh_ov[20];
    if (tid == kLmCtl) {
      double q[4], t[3], T[12];
      for (int k = 0; k < 4; k++) q[k] = lm.q[k];
      for (int k = 0; k < 3; k++) t[k] = lm.t[k];
      iso_from_qt(q, t, T);
      for (int k = 0; k < 12; k++) sh_ov[k] = T[k];
      for (int k = 0; k < 4; k++) sh_ov[12 + k] = q[k];
      for (int k = 0; k < 3; k++) sh_ov[16 + k] = t[k];
    }
    __syncthreads();
    ov_publish_pose(v, s, sh_ov, seq, tid);
    OV_STAMP(v, tid == 0, 1);
  }
  if (tid == kLmCtl) {
    for (int k = 0; k < 4; k++) st.param_q[k] = lm.q[k];
    for (int k = 0; k < 3; k++) st.param_t[k] = lm.t[k];
    iso_from_qt(st.param_q, st.param_t, st.odom);                  // :222-227
    liodom_lm_trace_t& tr = st.info.lm[outer_it];
    tr.iterations = lm.iter; tr.accepted = lm.accepted; tr.termination = lm.termination; tr.pad = 0;
    tr.initial_cost = lm.initial_cost; tr.final_cost = lm.cost;
    if (outer_it == 1) st.append_raw = 0;
  }
  DBG_STAMP(v, dbgb, 2, 21);
  if (outer_it == 1) {
    finalize_scan(v, s, st, sh_cnt, eb, false, kLmCtl);
    DBG_STAMP(v, dbgb, 2, 22);
  }
  DBG_STAMP(v, dbgb, 2, 28);
  OV_STAMP(v, tid == 0, outer_it == 0 ? 2 : 5);
}

// =============================================================================================
// Sliding window + voxel hash rebuild.
// =============================================================================================
// (re)initialise every slot of the voxel hash (handle creation / reset)
__global__ __launch_bounds__(256) void k_init_cells(DevView v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per = (size_t)v.n_streams * v.table_size;
  const size_t total = per * (v.early_rebuild ? 2 : 1);          // early_rebuild: two cell hashes per stream
  if (i >= total) return;
  CellSlot empty; empty.key = kEmptyKey; empty.start = 0; empty.cnt = 0;
  v.cells[i] = empty;
  if ((i & 31) == 0) v.cell_bits[i >> 5] = 0u;
  if (v.cell_pad) v.cell_pad[i] = 0u;
  if (v.vox_cells && i < per) { v.vox_cells[i] = empty; v.vox_fill[i] = 0; }
}

// Counts window point m (position pt) into its 1 m cell of the build in progress: atomicCAS insert of the cell key,
// atomicAdd of the cell's count.  The value the count had before is the point's rank inside the cell, so the scatter
// pass needs no second atomic (and no per-cell fill counter to keep clean).  Called by whole waves (inactive lanes
// pass live = false): the slots a wave creates are appended to the list of occupied slots with one atomic.
__device__ __forceinline__ void hash_count_point(const DevView& v, int s, int par /*table*/, StreamState& st, int m, float4 pt, bool live) {
  int* pc = v.pt_cell + (size_t)s * v.map_cap + m;
  const bool fin = live && ld_isfinite((double)pt.x) && ld_isfinite((double)pt.y) && ld_isfinite((double)pt.z) &&
                   fabsf(pt.x) < 1.0e9f && fabsf(pt.y) < 1.0e9f && fabsf(pt.z) < 1.0e9f;
  const unsigned int tmask = (unsigned int)v.table_size - 1u;
  const int sp = s + par * v.n_streams;
  CellSlot* cells = v.cells + (size_t)sp * v.table_size;
  unsigned int h = 0;
  int found = -1;
  bool created = false;
  if (fin) {
    const unsigned long long key = pack_cell((int)floorf(pt.x * kCellInv), (int)floorf(pt.y * kCellInv), (int)floorf(pt.z * kCellInv));
    h = hash_cell(key, tmask);
    for (int probe = 0; probe < v.table_size; probe++) {
      const unsigned long long prev = atomicCAS(&cells[h].key, kEmptyKey, key);
      if (prev == kEmptyKey) {
        atomicOr(&v.cell_bits[((size_t)sp * v.table_size + h) >> 5], 1u << (h & 31));
        found = (int)h;
        created = true;
        break;
      }
      if (prev == key) { found = (int)h; break; }
      h = (h + 1) & tmask;
    }
  }
  // list of occupied slots: one atomic per wave for all the slots its lanes created
  {
    const unsigned long long cm = __ballot(created);
    if (cm) {
      const int lane = threadIdx.x & 63;
      int base = 0;
      if (lane == (int)__builtin_ctzll(cm)) base = atomicAdd(&st.n_used_tab[par], (int)__popcll(cm));
      base = __shfl(base, (int)__builtin_ctzll(cm));
      if (created) v.used_cells[(size_t)sp * v.used_cap + base + (int)__popcll(cm & ((1ull << lane) - 1ull))] = (int)h;
    }
  }
  if (!live) return;
  if (!fin) { *pc = -1; return; }
  if (found < 0) { atomicOr(&st.status, LIODOM_STATUS_HASH_FULL); *pc = -1; return; }
  v.pt_rank[(size_t)s * v.map_cap + m] = (int)atomicAdd(&cells[found].cnt, 1u);
  *pc = found;
}

// The new frame's edges (dense edge buffer eb, sensor frame) are transformed with the solved pose
// (laser_odometry.cc:231-232), then stored in the window slot (:235).  Every point of the window is counted
// into its 1 m cell (hash_count_point).  (Not launched by handles with early_rebuild: see "streamed rebuild".)
__global__ __launch_bounds__(256) void k_window_insert(DevView v, int s0, int eb) {
  __shared__ int sbase[kMaxFrames + 1];
  __shared__ int sslot[kMaxFrames];
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  const int M = st.n_map;
  const int MT = M + (v.mapping ? st.n_recv : 0);      // window ++ received map (:310-314)
  const int P = v.prev_frames, nf = st.n_frames;
  const int m_first = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && !filter_active(v, st)) { st.n_search = MT; st.n_filt = 0; }
  if (m_first + (int)(blockIdx.x * 256) >= MT) return;
  for (int j = threadIdx.x; j <= nf; j += 256) sbase[j] = v.win_base[(size_t)s * (P + 1) + j];
  for (int j = threadIdx.x; j < nf; j += 256) sslot[j] = v.win_slot[(size_t)s * P + j];
  __syncthreads();
  const int m = m_first + blockIdx.x * 256 + threadIdx.x;
  const bool live = m < MT;
  float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && m < M) {
    int lo = 0, hi = nf;             // largest j with sbase[j] <= m
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sbase[mid] <= m) lo = mid; else hi = mid; }
    const int j = lo, idx = m - sbase[j], slot = sslot[j];
    float4* wp = v.win_pts + ((size_t)s * P + slot) * v.edge_cap + idx;
    if (j == nf - 1 && eb >= 0) {      // eb < 0: rebuild only (the newest frame is already stored)
      const float4 e = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + idx];
      if (st.append_raw) {
        pt = e;
      } else {
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; i++) T[i] = st.final_odom[i];
        transform_point(T, e.x, e.y, e.z, &pt.x, &pt.y, &pt.z);
        pt.w = e.w;
      }
      *wp = pt;
    } else {
      pt = *wp;
    }
  } else if (live) {
    pt = v.recv_pts[(size_t)s * v.recv_cap + (m - M)];
  }
  if (filter_active(v, st)) return;     // the kNN structure is built from the filtered cloud instead
  hash_count_point(v, s, LD_TAB_PARITY(v, st.frame_count), st, m, pt, live);
}

// =============================================================================================
// Streamed rebuild (early_rebuild; handles with <= 4 streams, no mapping / filtered map).
// The cell hash of the NEXT scan is built while the current scan is solved, by extra workgroups riding on the four
// launches of the scan; nothing is left between the finalising solve and the next scan's first kNN pass.  Two tables
// per stream: scan F (frame_count = F when it starts) searches table F & 1 and builds table (F + 1) & 1.
//   k_knn      it 0   bookkeeping (frame_count snapshot, empty slot list for the table being built)
//   k_lm_solve it 0   COUNT the frames that stay in the window (all but the oldest once it is full,
//                     LocalMapManager::addPointCloud :34-60) into their cells, under the window indices they will have
//                     after the append; PAD: every edge of the new scan, transformed with the PREDICTED pose, reserves
//                     one place in each cell it can reach if the solve moves it by less than rebuild_delta per axis
//   k_knn      it 1   ALLOC: start of every occupied cell; room = counted + padded (when that pass is overlapped with the first
//                     solve on its own stream: k_rebuild_alloc, a launch of its own between the two solve launches)
//   k_lm_solve it 1   SCATTER the kept points to start + rank; APPEND: the first workgroups wait for the solved pose
//                     (publish_final_pose), transform the scan's edges (laser_odometry.cc:231-232), store them in the new
//                     frame's window slot (:235) and put every point into its cell at start + count++ — the place its
//                     padding reserved.  A point that moved further than rebuild_delta (or whose cell is missing) goes
//                     to the table's overflow list instead, which every kNN query of the next scan also scans: exact in
//                     every case, and empty unless the solve corrected the prediction by decimetres.
//                     CLEAR the table this scan searched (dead since the second kNN pass) for the scan after the next.
// Builders use only state the solves do not write: the frame_count snapshot, the sizes of the kept slots, the edge count,
// the saved prediction.  (A waiting workgroup depends only on the solving workgroup of its own stream, which has a
// lower block index and so was dispatched before it.)
// =============================================================================================
constexpr int kRebuildAuxBlocks = 8;      // workgroups for ALLOC (inside k_knn) and for CLEAR (k_lm_solve)
constexpr int kRebuildAllocBlocks = 32;   // k_rebuild_alloc: the ~8 000 occupied cells of a headline scan in one sweep (it sits between the two solve launches)

// Prefix table of the kept frames (chronological): sbase[0 .. nk], sslot[0 .. nk).  Returns nk; whole workgroup.
__device__ __forceinline__ int kept_frames_table(const DevView& v, int s, const StreamState& st, int* sbase, int* sslot) {
  const int P = v.prev_frames, fc = st.reb_frame_count;
  const int nf_old = fc < P ? fc : P;
  const int drop = nf_old == P ? 1 : 0;
  const int nk = nf_old - drop;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j = tid; j < nk; j += nt) {
    const int sl = (fc - nf_old + drop + j) % P;
    sslot[j] = sl;
    sbase[j + 1] = v.win_n[(size_t)s * P + sl];
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int j = 0; j < nk; j++) { const int c = sbase[j + 1]; sbase[j] = acc; acc += c; }
    sbase[nk] = acc;
  }
  __syncthreads();
  return nk;
}
__device__ __forceinline__ float4 kept_point(const DevView& v, int s, int m, int nk, const int* sbase, const int* sslot) {
  int lo = 0, hi = nk;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sbase[mid] <= m) lo = mid; else hi = mid; }
  return v.win_pts[((size_t)s * v.prev_frames + sslot[lo]) * v.edge_cap + (m - sbase[lo])];
}
__device__ __forceinline__ bool point_finite(const float4& pt) {
  return ld_isfinite((double)pt.x) && ld_isfinite((double)pt.y) && ld_isfinite((double)pt.z) &&
         fabsf(pt.x) < 1.0e9f && fabsf(pt.y) < 1.0e9f && fabsf(pt.z) < 1.0e9f;
}
// the scan's edge idx at the pose the scan started from (the first frame enters the window untransformed, :123)
__device__ __forceinline__ float4 predicted_point(const StreamState& st, const float4& e) {
  if (!st.reb_initialized) return e;
  double T[12];
#pragma unroll
  for (int i = 0; i < 12; i++) T[i] = st.pred_odom[i];
  float4 q;
  transform_point(T, e.x, e.y, e.z, &q.x, &q.y, &q.z);
  q.w = e.w;
  return q;
}

// COUNT (block < nC) and PAD (the nP blocks behind them)
__device__ void rebuild_count_and_pad(const DevView& v, int s, StreamState& st, int eb, int block, int* sbase, int* sslot) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
  const int par = (st.reb_frame_count + 1) & 1;
  const int nC = (v.edge_cap * (v.prev_frames > 1 ? v.prev_frames - 1 : 1) + nt - 1) / nt;
  if (block < nC) {
    const int nk = kept_frames_table(v, s, st, sbase, sslot);
    const int Mk = nk > 0 ? sbase[nk] : 0;
    if (block * nt >= Mk) return;
    const int m = block * nt + tid;
    const bool live = m < Mk;
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) pt = kept_point(v, s, m, nk, sbase, sslot);
    hash_count_point(v, s, par, st, m, pt, live);
    return;
  }
  const int n_new = st.n_edges_buf[eb];
  const int idx = (block - nC) * nt + tid;
  if ((block - nC) * nt >= n_new) return;
  const bool live = idx < n_new;
  const float4 q = predicted_point(st, v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (live ? idx : 0)]);
  const bool fin = live && point_finite(q);
  const float d = v.rebuild_delta;
  const int lx = (int)floorf((q.x - d) * kCellInv), hx = (int)floorf((q.x + d) * kCellInv);
  const int ly = (int)floorf((q.y - d) * kCellInv), hy = (int)floorf((q.y + d) * kCellInv);
  const int lz = (int)floorf((q.z - d) * kCellInv), hz = (int)floorf((q.z + d) * kCellInv);
  const int sp = s + par * v.n_streams;
  const unsigned int tmask = (unsigned int)v.table_size - 1u;
  CellSlot* cells = v.cells + (size_t)sp * v.table_size;
#pragma unroll 1
  for (int c = 0; c < 8; c++) {        // rebuild_delta < half a cell: at most two cells per axis
    const int cx = (c & 1) ? hx : lx, cy = (c & 2) ? hy : ly, cz = (c & 4) ? hz : lz;
    const bool act = fin && !((c & 1) && hx == lx) && !((c & 2) && hy == ly) && !((c & 4) && hz == lz);
    unsigned int h = 0;
    bool created = false, found = false;
    if (act) {
      const unsigned long long key = pack_cell(cx, cy, cz);
      h = hash_cell(key, tmask);
      for (int probe = 0; probe < v.table_size; probe++) {
        const unsigned long long prev = atomicCAS(&cells[h].key, kEmptyKey, key);
        if (prev == kEmptyKey) { atomicOr(&v.cell_bits[((size_t)sp * v.table_size + h) >> 5], 1u << (h & 31)); created = found = true; break; }
        if (prev == key) { found = true; break; }
        h = (h + 1) & tmask;
      }
      if (found) atomicAdd(&v.cell_pad[(size_t)sp * v.table_size + h], 1u);
      else atomicOr(&st.status, LIODOM_STATUS_HASH_FULL);
    }
    const unsigned long long cm = __ballot(created);   // list of occupied slots: one atomic per wave
    if (cm) {
      int base = 0;
      if (lane == (int)__builtin_ctzll(cm)) base = atomicAdd(&st.n_used_tab[par], (int)__popcll(cm));
      base = __shfl(base, (int)__builtin_ctzll(cm));
      if (created) v.used_cells[(size_t)sp * v.used_cap + base + (int)__popcll(cm & ((1ull << lane) - 1ull))] = (int)h;
    }
  }
}

// ALLOC, by the extra workgroups of the scan's second k_knn launch — or, when that pass is overlapped with the first solve
// on a stream of its own, by k_rebuild_alloc between the two solve launches (a launch boundary must separate ALLOC from
// COUNT / PAD before it and from SCATTER / APPEND behind it) —: start offsets of the occupied cells (any order:
// only contiguity per cell matters), room = points counted + places padded; cell_pad becomes the end of the range.
__device__ void rebuild_alloc(const DevView& v, int s, StreamState& st, int block, int nblocks) {
  const int par = (st.reb_frame_count + 1) & 1, sp = s + par * v.n_streams;
  const int nu = st.n_used_tab[par];
  const int nt = blockDim.x;
  for (int u0 = block * nt; u0 < nu; u0 += nblocks * nt) {
    const int u = u0 + (int)threadIdx.x;
    size_t ti = 0;
    int room = 0;
    if (u < nu) {
      ti = (size_t)sp * v.table_size + v.used_cells[(size_t)sp * v.used_cap + u];
      room = (int)v.cells[ti].cnt + (int)v.cell_pad[ti];
    }
    const int incl = wave_incl_scan_i32(room);
    const int total = readlane_i32(incl, 63);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && total > 0) base = atomicAdd(&st.cursor, total);
    base = __builtin_amdgcn_readfirstlane(base);
    if (u < nu) {
      const int start = base + incl - room;
      v.cells[ti].start = (unsigned int)start;
      v.cell_pad[ti] = (unsigned int)(start + room);
    }
  }
}

__global__ __launch_bounds__(256) void k_rebuild_alloc(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  rebuild_alloc(v, s, st, (int)blockIdx.x, (int)gridDim.x);
}

// SCATTER / APPEND / CLEAR, beside the finalising solve
__device__ void rebuild_finish(const DevView& v, int s, StreamState& st, int eb, int block, unsigned int seq, int* sbase, int* sslot) {
  __shared__ double sh_T[12];
  __shared__ int sh_hand;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int P = v.prev_frames, fc = st.reb_frame_count;
  const int par = (fc + 1) & 1, sp = s + par * v.n_streams;
  const int nP = (v.edge_cap + nt - 1) / nt;
  float4* sorted = v.sorted_pts + (size_t)sp * v.sorted_cap;
  CellSlot* cells = v.cells + (size_t)sp * v.table_size;
  if (block >= nP && block < nP + kRebuildAuxBlocks) {
    // CLEAR: the table this scan searched, its padding and its overflow list (dead since the second kNN pass — which, when
    // it runs beside this launch, has to have completed first)
    if (seq) ov_wait_knn_done(v, s, seq, &st.status);
    const int dead = s + (1 - par) * v.n_streams;
    hash_clear_used(v, dead, st.n_used_tab[1 - par], (block - nP) * nt + tid, kRebuildAuxBlocks * nt);
    if (block == nP && tid == 0) st.n_ovf[1 - par] = 0;
    return;
  }
  const int nk = kept_frames_table(v, s, st, sbase, sslot);
  const int Mk = nk > 0 ? sbase[nk] : 0;
  if (block >= nP) {
    // SCATTER the kept points
    const int m = (block - nP - kRebuildAuxBlocks) * nt + tid;
    if (m >= Mk) return;
    const int h = v.pt_cell[(size_t)s * v.map_cap + m];
    if (h < 0) return;
    const float4 pt = kept_point(v, s, m, nk, sbase, sslot);
    const unsigned int pos = cells[h].start + (unsigned int)v.pt_rank[(size_t)s * v.map_cap + m];
    sorted[pos] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
    return;
  }
  // APPEND the new frame
  const int n_new = st.n_edges_buf[eb];
  if (block * nt >= n_new) return;
  const int idx = block * nt + tid;
  const bool live = idx < n_new;
  const float4 e = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (live ? idx : 0)];     // (loaded before the wait)
  const float4 q = predicted_point(st, e);
  // the cell the prediction puts the point into is where it ends up almost always: look its slot up before the wait
  const unsigned int tmask = (unsigned int)v.table_size - 1u;
  const bool q_fin = point_finite(q);
  const unsigned long long key_pred = q_fin ? pack_cell((int)floorf(q.x * kCellInv), (int)floorf(q.y * kCellInv), (int)floorf(q.z * kCellInv)) : kEmptyKey;
  int h_pred = -1;
  unsigned int start_pred = 0, end_pred = 0;
  if (live && q_fin) {
    unsigned int h = hash_cell(key_pred, tmask);
    for (int probe = 0; probe < v.table_size; probe++) {
      const unsigned long long k = cells[h].key;
      if (k == key_pred) { h_pred = (int)h; start_pred = cells[h].start; end_pred = v.cell_pad[(size_t)sp * v.table_size + h]; break; }
      if (k == kEmptyKey) break;
      h = (h + 1) & tmask;
    }
  }
  if (tid < 64) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    const unsigned long long* base = v.pose_xch + (size_t)s * 32;
    const unsigned int tag = (unsigned int)fc + 1u;
    unsigned long long g = 0;
    unsigned int spins = 0;
    bool ok;
    while (true) {
      if (tid < 25) g = __hip_atomic_load((gu64*)(base + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = tid >= 25 || (unsigned int)(g >> 32) == tag;
      if (__all(ok)) break;
      if (++spins > 4000000u) break;
      __builtin_amdgcn_s_sleep(2);
    }
    const bool all_ok = __all(ok);
    const unsigned long long gflags = __shfl(g, 24);                 // flags granule: low word = append_raw
    if (tid == 0) sh_hand = all_ok ? (int)(unsigned int)gflags + 1 : 0;   // 0: timed out, else raw + 1
    const unsigned long long lo = __shfl(g, 2 * (tid % 12)), hi = __shfl(g, 2 * (tid % 12) + 1);
    if (tid < 12) sh_T[tid] = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
    if (!all_ok && tid == 0) atomicOr(&st.status, LIODOM_STATUS_LM_SYNC_TIMEOUT);
  }
  __syncthreads();
  const bool dbga = (s == 0) && (block == 0) && (tid == 0);
  DBG_STAMP(v, dbga, 2, 29);
  if (sh_hand == 0 || !live) return;
  float4 pt;
  if (sh_hand == 2) {
    pt = e;
  } else {
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = sh_T[i];
    transform_point(T, e.x, e.y, e.z, &pt.x, &pt.y, &pt.z);
    pt.w = e.w;
  }
  v.win_pts[((size_t)s * P + (fc % P)) * v.edge_cap + idx] = pt;
  if (!point_finite(pt)) return;                       // (never part of the map, as in k_window_insert)
  const int m = Mk + idx;
  // inside the padded cells for certain?  (1e-3 covers the float rounding of the two transforms and of q -+ delta)
  const float dc = v.rebuild_delta - 1.0e-3f;
  bool placed = false;
  if (q_fin && fabsf(pt.x - q.x) < dc && fabsf(pt.y - q.y) < dc && fabsf(pt.z - q.z) < dc) {
    const unsigned long long key = pack_cell((int)floorf(pt.x * kCellInv), (int)floorf(pt.y * kCellInv), (int)floorf(pt.z * kCellInv));
    int hf = -1;
    unsigned int start = 0, end = 0;
    if (key == key_pred) {
      hf = h_pred; start = start_pred; end = end_pred;
    } else {                                             // crossed into a neighbour cell (also padded)
      unsigned int h = hash_cell(key, tmask);
      for (int probe = 0; probe < v.table_size; probe++) {
        const unsigned long long k = cells[h].key;
        if (k == key) { hf = (int)h; start = cells[h].start; end = v.cell_pad[(size_t)sp * v.table_size + h]; break; }
        if (k == kEmptyKey) break;
        h = (h + 1) & tmask;
      }
    }
    if (hf >= 0) {
      const unsigned int pos = start + atomicAdd(&cells[hf].cnt, 1u);
      if (pos < end) sorted[pos] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
      else atomicOr(&st.status, LIODOM_STATUS_HASH_FULL);        // (cannot happen: the padding reserved the place)
      placed = true;
    }
  }
  if (!placed) {
    const int i = atomicAdd(&st.n_ovf[par], 1);
    sorted[v.ovf_base + i] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
  }
  DBG_STAMP(v, dbga, 2, 30);
}

__device__ void rebuild_beside_solve(const DevView& v, int s, StreamState& st, int eb, int outer_it, int block, int nblocks, unsigned int seq, int* sbase, int* sslot) {
  (void)nblocks;
  if (outer_it == 0) rebuild_count_and_pad(v, s, st, eb, block, sbase, sslot);
  else rebuild_finish(v, s, st, eb, block, seq, sbase, sslot);
}

// Start offsets of the occupied cells (any order: only contiguity per cell matters).  One atomic
// per wave: the 64 counts are scanned in the wave and lane 0 reserves the wave's total — 3 500
// same-address atomics serialise in L2 (measured 7.5 us for this launch), 55 do not.
__global__ __launch_bounds__(256) void k_hash_alloc(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  const int par = LD_TAB_PARITY(v, st.frame_count), sp = s + par * v.n_streams;
  const int nu = st.n_used_tab[par];
  if ((int)(blockIdx.x * 256) >= nu) return;
  const int u = blockIdx.x * 256 + threadIdx.x;
  CellSlot* slot = nullptr;
  int cnt = 0;
  if (u < nu) {
    slot = v.cells + (size_t)sp * v.table_size + v.used_cells[(size_t)sp * v.used_cap + u];
    cnt = (int)slot->cnt;
  }
  const int incl = wave_incl_scan_i32(cnt);
  const int total = readlane_i32(incl, 63);
  int base = 0;
  if ((threadIdx.x & 63) == 0 && total > 0) base = atomicAdd(&st.cursor, total);
  base = __builtin_amdgcn_readfirstlane(base);
  if (slot) slot->start = (unsigned int)(base + incl - cnt);
}

__global__ __launch_bounds__(256) void k_hash_scatter(DevView v, int s0) {
  __shared__ int sbase[kMaxFrames + 1];
  __shared__ int sslot[kMaxFrames];
  const int s = s0 + blockIdx.y;
  const StreamState& st = v.state[s];
  const int M = st.n_map;
  const int MT = M + (v.mapping ? st.n_recv : 0);
  const int par = LD_TAB_PARITY(v, st.frame_count), sp = s + par * v.n_streams;
  if ((int)(blockIdx.x * 256) >= MT || filter_active(v, st)) return;
  const int P = v.prev_frames, nf = st.n_frames;
  for (int j = threadIdx.x; j <= nf; j += 256) sbase[j] = v.win_base[(size_t)s * (P + 1) + j];
  for (int j = threadIdx.x; j < nf; j += 256) sslot[j] = v.win_slot[(size_t)s * P + j];
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= MT) return;
  const int h = v.pt_cell[(size_t)s * v.map_cap + m];
  if (h < 0) return;
  float4 pt;
  if (m < M) {
    int lo = 0, hi = nf;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sbase[mid] <= m) lo = mid; else hi = mid; }
    pt = v.win_pts[((size_t)s * P + sslot[lo]) * v.edge_cap + (m - sbase[lo])];
  } else {
    pt = v.recv_pts[(size_t)s * v.recv_cap + (m - M)];
  }
  const size_t ti = (size_t)sp * v.table_size + h;
  const unsigned int pos = v.cells[ti].start + (unsigned int)v.pt_rank[(size_t)s * v.map_cap + m];
  v.sorted_pts[(size_t)sp * v.sorted_cap + pos] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
}

// Clears the cell hash of the current build so that it can be rebuilt without a new frame
// (liodom_set_received_map: the kNN cloud changed between two scans).
__global__ __launch_bounds__(256) void k_hash_reset(DevView v, int s) {
  StreamState& st = v.state[s];
  const int nup = st.n_used_tab[0];
  hash_clear_used(v, s, nup, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
  if (st.table_mask != (unsigned int)v.table_size - 1u) {      // LDS-built table: slots [0, kLdsSlotsC)
    CellSlot empty; empty.key = kEmptyKey; empty.start = 0; empty.cnt = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 8192; i += gridDim.x * 256) {
      v.cells[(size_t)s * v.table_size + i] = empty;
      if (i < 8192 / 32) v.cell_bits[(size_t)s * (v.table_size >> 5) + i] = 0u;
    }
  }
}
__global__ void k_hash_reset_done(DevView v, int s) {
  StreamState& st = v.state[s];
  st.n_used_tab[0] = 0; st.cursor = 0; st.table_mask = (unsigned int)v.table_size - 1u;
}

// =============================================================================================
// k_hash_build: window append + complete rebuild of the 1 m cell hash by ONE workgroup per stream,
// with LDS atomics.  The table of one stream is small (headline: ~3 500 occupied cells for 37 000
// points), so an 8192-slot table {key u64, cnt u32, cursor u32} = 128 KiB fits the 160 KiB LDS of
// a CU: slot claim (ds_cmpst_b64) and counting (ds_add) never leave the CU, the exclusive prefix
// over the slots runs in place, points are scattered to cell-contiguous order with LDS cursors,
// and the finished table is written out once (it replaces the previous one wholesale: nothing to
// clear).  One launch instead of three, no L2 atomics: the multi-block version spent ~450 us on
// 64 lock-step streams (L2-atomic bound), this one works on 64 CUs in parallel.
// If more than kLdsCellsMax cells are occupied the workgroup falls back to the global-memory
// table (full v.table_size, global atomics), which any map size fits.
// With filter_local_map active only the new frame is stored here; the k_voxel_* / k_filt_*
// kernels build the table from the filtered cloud.
// =============================================================================================
struct WinIndex {
  int sbase[kMaxFrames + 1];
  int sslot[kMaxFrames];
};
__device__ __forceinline__ void win_index_load(const DevView& v, int s, int nf, WinIndex& w, int tid, int nt) {
  const int P = v.prev_frames;
  for (int j = tid; j <= nf; j += nt) w.sbase[j] = v.win_base[(size_t)s * (P + 1) + j];
  for (int j = tid; j < nf; j += nt) w.sslot[j] = v.win_slot[(size_t)s * P + j];
}
__device__ __forceinline__ float4 win_point(const DevView& v, int s, int nf, const WinIndex& w, int m, int* jc = nullptr) {
  int lo = 0, hi = nf;             // largest j with sbase[j] <= m
  if (jc) { lo = *jc; while (lo + 1 < nf && w.sbase[lo + 1] <= m) lo++; *jc = lo; }
  else { while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (w.sbase[mid] <= m) lo = mid; else hi = mid; } }
  return v.win_pts[((size_t)s * v.prev_frames + w.sslot[lo]) * v.edge_cap + (m - w.sbase[lo])];
}
__device__ __forceinline__ bool point_ok(const float4& p) {
  // finite and below 1e9 in magnitude (NaN and inf fail the comparisons: no separate finiteness test needed)
  return fabsf(p.x) < 1.0e9f && fabsf(p.y) < 1.0e9f && fabsf(p.z) < 1.0e9f;
}

// Runs of equal cell keys among the valid lanes of a wave (consecutive lanes hold consecutive window points, i.e.
// neighbouring edges of a frame: ~8 points per run).  One lane per run — its head — performs the LDS atomic for the whole
// run; the others take the head's result by a lane read.  Without this the 64 lanes of an atomic instruction queue on a
// handful of addresses: 256 lock-step streams spent 70 us of the build's 174 in the counting pass.
struct KeyRun {
  bool head;        // this lane is the first of its run (valid lanes only)
  int head_lane;    // lane of the run's head
  int rank;         // position inside the run
  int len;          // length of the run (meaningful on the head)
};
__device__ __forceinline__ KeyRun wave_key_runs(bool valid, unsigned long long key, int lane) {
  const unsigned long long vm = __ballot(valid);
  const unsigned int klo = (unsigned int)key, khi = (unsigned int)(key >> 32);
  const unsigned int plo = (unsigned int)__shfl_up((int)klo, 1), phi = (unsigned int)__shfl_up((int)khi, 1);
  const bool prev_valid = lane > 0 && ((vm >> (lane - 1)) & 1ull);
  KeyRun r;
  r.head = valid && (!prev_valid || plo != klo || phi != khi);
  const unsigned long long hm = __ballot(r.head);
  const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);        // lanes 0 .. lane
  const unsigned long long below = hm & upto;
  r.head_lane = below ? 63 - __clzll((long long)below) : lane;
  r.rank = lane - r.head_lane;
  const unsigned long long stop = (hm | ~vm) & ~upto;                                   // next head or invalid lane above
  r.len = (stop ? __ffsll((long long)stop) - 1 : 64) - lane;
  return r;
}

constexpr int kLdsSlots = 8192;
constexpr int kLdsCellsMax = 6144;
constexpr int kBuildThreads = 1024;
constexpr int kBuildUnroll = 4;
__host__ __device__ __forceinline__ size_t hash_build_lds_bytes() { return (size_t)kLdsSlots * 16 + 64; }

// (jc: optional frame cursor of a thread whose m only grows: replaces the binary search by a step)
__device__ __forceinline__ float4 window_point_produce(const DevView& v, int s, const StreamState& st, int eb,
                                                       const WinIndex& w, int nf, int m, int* jc = nullptr) {
  int lo = 0, hi = nf;             // largest j with sbase[j] <= m
  if (jc) { lo = *jc; while (lo + 1 < nf && w.sbase[lo + 1] <= m) lo++; *jc = lo; }
  else { while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (w.sbase[mid] <= m) lo = mid; else hi = mid; } }
  const int j = lo, idx = m - w.sbase[j];
  float4* wp = v.win_pts + ((size_t)s * v.prev_frames + w.sslot[j]) * v.edge_cap + idx;
  if (j != nf - 1 || eb < 0) return *wp;
  // newest frame: edges transformed by the solved pose in FP64, rounded to float (:231-232), stored (:235)
  const float4 e = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + idx];
  float4 pt = e;
  if (!st.append_raw) {
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = st.final_odom[i];
    transform_point(T, e.x, e.y, e.z, &pt.x, &pt.y, &pt.z);
    pt.w = e.w;
  }
  *wp = pt;
  return pt;
}

__global__ __launch_bounds__(kBuildThreads) void k_hash_build(DevView v, int s0, int eb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ WinIndex w;
  __shared__ int sh_used, sh_over, sh_wtot[kBuildThreads / 64];
  unsigned long long* lkey = reinterpret_cast<unsigned long long*>(smem);           // [kLdsSlots]
  unsigned int* lcnt = reinterpret_cast<unsigned int*>(lkey + kLdsSlots);           // [kLdsSlots]
  unsigned int* lstart = lcnt + kLdsSlots;                                          // [kLdsSlots]
  const int s = s0 + blockIdx.x;
  StreamState& st = v.state[s];
  const int tid = threadIdx.x;
  const int Mw = st.n_map, nf = st.n_frames;
  const int M = Mw + (v.mapping ? st.n_recv : 0);      // window ++ received map (:310-314)
  const float4* recv = v.recv_pts + (size_t)s * v.recv_cap;
  const bool filt = filter_active(v, st);
  OV_STAMP(v, tid == 0 && s == 0, 19);
  win_index_load(v, s, nf, w, tid, kBuildThreads);
  if (tid == 0) { sh_used = 0; sh_over = 0; }
  if (!filt) for (int i = tid; i < kLdsSlots; i += kBuildThreads) { lkey[i] = kEmptyKey; lcnt[i] = 0; }
  __syncthreads();
  CellSlot* cells = v.cells + (size_t)s * v.table_size;
  unsigned int* bits = v.cell_bits + (size_t)s * (v.table_size >> 5);
  int* pcell = v.pt_cell + (size_t)s * v.map_cap;
  int* prank = v.pt_rank + (size_t)s * v.map_cap;
  if (filt) {
    // store the new frame only; hand a clean global table to the filtered-cloud build
    const int first_new = w.sbase[nf - 1];
    for (int m = first_new + tid; m < Mw; m += kBuildThreads) (void)window_point_produce(v, s, st, eb, w, nf, m);
    if (st.table_mask != (unsigned int)v.table_size - 1u) {
      CellSlot empty; empty.key = kEmptyKey; empty.start = 0; empty.cnt = 0;
      for (int i = tid; i < kLdsSlots; i += kBuildThreads) { cells[i] = empty; }
      for (int i = tid; i < kLdsSlots / 32; i += kBuildThreads) bits[i] = 0u;
      __syncthreads();
      if (tid == 0) { st.table_mask = (unsigned int)v.table_size - 1u; st.n_used_tab[0] = 0; }
    }
    return;
  }
  OV_STAMP(v, tid == 0 && s == 0, 20);
  // ---- insert + count in LDS (kBuildUnroll point loads in flight per thread) ----
  const unsigned int lmask = kLdsSlots - 1;
  int jc = 0;                      // frame cursor: this thread's m only grows
  const int lane = tid & 63;
  for (int m0 = tid; m0 - lane < M; m0 += kBuildUnroll * kBuildThreads) {
    float4 pt[kBuildUnroll];
#pragma unroll
    for (int k = 0; k < kBuildUnroll; k++) {
      const int m = m0 + k * kBuildThreads;
      pt[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) pt[k] = m < Mw ? window_point_produce(v, s, st, eb, w, nf, m, &jc) : recv[m - Mw];
    }
#pragma unroll
    for (int k = 0; k < kBuildUnroll; k++) {
      // (whole waves walk this loop together: m0 - lane is the same for all lanes, the per-lane tests are masks)
      const int m = m0 + k * kBuildThreads;
      const bool ok = m < M && point_ok(pt[k]);
      const unsigned long long key = pack_cell((int)floorf(pt[k].x * kCellInv), (int)floorf(pt[k].y * kCellInv), (int)floorf(pt[k].z * kCellInv));
      const KeyRun run = wave_key_runs(ok, key, lane);
      if (run.head) {
        unsigned int h = hash_cell(key, lmask);
        int found = -1;
        for (int probe = 0; probe < kLdsSlots; probe++) {
          const unsigned long long prev = atomicCAS(&lkey[h], kEmptyKey, key);
          if (prev == kEmptyKey) { if (atomicAdd(&sh_used, 1) >= v.lds_cells_max) sh_over = 1; found = (int)h; break; }
          if (prev == key) { found = (int)h; break; }
          if (*(volatile int*)&sh_over) break;      // the global-table fallback redoes everything
          h = (h + 1) & lmask;
        }
        if (found >= 0) atomicAdd(&lcnt[found], (unsigned int)run.len);      // the whole run's count
      }
    }
  }
  __syncthreads();
  if (sh_over) {
    // ---- fallback: too many occupied cells for the LDS table -> global table, global atomics ----
    CellSlot empty; empty.key = kEmptyKey; empty.start = 0; empty.cnt = 0;
    if (st.table_mask != (unsigned int)v.table_size - 1u) {
      for (int i = tid; i < kLdsSlots; i += kBuildThreads) { cells[i] = empty; }
      for (int i = tid; i < kLdsSlots / 32; i += kBuildThreads) bits[i] = 0u;
    }
    __syncthreads();
    if (tid == 0) { st.table_mask = (unsigned int)v.table_size - 1u; st.n_used_tab[0] = 0; st.cursor = 0; st.n_search = M; st.n_filt = 0; }
    __syncthreads();
    const unsigned int gmask = (unsigned int)v.table_size - 1u;
    for (int m = tid; m < M; m += kBuildThreads) {
      const float4 pt = m < Mw ? win_point(v, s, nf, w, m) : recv[m - Mw];
      int found = -1;
      if (point_ok(pt)) {
        const unsigned long long key = pack_cell((int)floorf(pt.x * kCellInv), (int)floorf(pt.y * kCellInv), (int)floorf(pt.z * kCellInv));
        unsigned int h = hash_cell(key, gmask);
        for (int probe = 0; probe < v.table_size; probe++) {
          const unsigned long long prev = atomicCAS(&cells[h].key, kEmptyKey, key);
          if (prev == kEmptyKey) {
            const int u = atomicAdd(&st.n_used_tab[0], 1);
            v.used_cells[(size_t)s * v.used_cap + u] = (int)h;
            atomicOr(&bits[h >> 5], 1u << (h & 31));
            found = (int)h;
            break;
          }
          if (prev == key) { found = (int)h; break; }
          h = (h + 1) & gmask;
        }
        if (found >= 0) prank[m] = (int)atomicAdd(&cells[found].cnt, 1u);
        else atomicOr(&st.status, LIODOM_STATUS_HASH_FULL);
      }
      pcell[m] = found;
    }
    __threadfence();
    __syncthreads();
    const int nu = *(volatile int*)&st.n_used_tab[0];
    for (int u = tid; u < nu; u += kBuildThreads) {
      CellSlot* slot = cells + v.used_cells[(size_t)s * v.used_cap + u];
      slot->start = (unsigned int)atomicAdd(&st.cursor, (int)*(volatile unsigned int*)&slot->cnt);
    }
    __threadfence();
    __syncthreads();
    for (int m = tid; m < M; m += kBuildThreads) {
      const int h = pcell[m];
      if (h < 0) continue;
      const float4 pt = m < Mw ? win_point(v, s, nf, w, m) : recv[m - Mw];
      const unsigned int pos = *(volatile unsigned int*)&cells[h].start + (unsigned int)prank[m];
      v.sorted_pts[(size_t)s * v.map_cap + pos] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
    }
    return;
  }
  OV_STAMP(v, tid == 0 && s == 0, 21);
  // ---- exclusive prefix of the counts over the slots (8 consecutive slots per thread) ----
  {
    constexpr int PER = kLdsSlots / kBuildThreads;   // 8
    unsigned int c[PER];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { c[k] = lcnt[tid * PER + k]; sum += (int)c[k]; }
    const int incl = wave_incl_scan_i32(sum);
    if ((tid & 63) == 63) sh_wtot[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < (tid >> 6); q++) base += sh_wtot[q];
    int run = base + incl - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) { lstart[tid * PER + k] = (unsigned int)run; run += (int)c[k]; }
  }
  __syncthreads();
  OV_STAMP(v, tid == 0 && s == 0, 22);
  // ---- scatter to cell-contiguous order: position = start of the cell + rank of the point ----
  // The cell of a point is looked up again (a read-only probe by the run's head) and its position taken from the cell's
  // cursor — lstart[h], advanced by the run's length — instead of a (cell, rank) pair written by the counting pass and
  // read back here: 16 B per point less traffic in a pass that is bandwidth-bound on 256 lock-step streams (3.6 TB/s).
  jc = 0;
  for (int m0 = tid; m0 - lane < M; m0 += kBuildUnroll * kBuildThreads) {
    float4 pt[kBuildUnroll];
#pragma unroll
    for (int k = 0; k < kBuildUnroll; k++) {
      const int m = m0 + k * kBuildThreads;
      pt[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) pt[k] = m < Mw ? win_point(v, s, nf, w, m, &jc) : recv[m - Mw];
    }
#pragma unroll
    for (int k = 0; k < kBuildUnroll; k++) {
      const int m = m0 + k * kBuildThreads;
      const bool ok = m < M && point_ok(pt[k]);
      const unsigned long long key = pack_cell((int)floorf(pt[k].x * kCellInv), (int)floorf(pt[k].y * kCellInv), (int)floorf(pt[k].z * kCellInv));
      const KeyRun run = wave_key_runs(ok, key, lane);
      unsigned int pos0 = 0xFFFFFFFFu;
      if (run.head) {
        unsigned int h = hash_cell(key, lmask);
        int probe = 0;
        while (lkey[h] != key && probe < kLdsSlots) { h = (h + 1) & lmask; probe++; }      // (inserted by the counting pass)
        if (probe < kLdsSlots) pos0 = atomicAdd(&lstart[h], (unsigned int)run.len);
      }
      pos0 = (unsigned int)__shfl((int)pos0, run.head_lane);
      if (ok && pos0 != 0xFFFFFFFFu) v.sorted_pts[(size_t)s * v.map_cap + pos0 + (unsigned int)run.rank] = make_float4(pt[k].x, pt[k].y, pt[k].z, __int_as_float(m));
    }
  }
  __syncthreads();
  OV_STAMP(v, tid == 0 && s == 0, 23);
  // ---- publish the table: slots [0, 8192) of the stream's global table + occupancy bits ----
  for (int i = tid; i < kLdsSlots; i += kBuildThreads) {
    CellSlot o; o.key = lkey[i]; o.cnt = lcnt[i]; o.start = lstart[i] - lcnt[i];      // (the scatter pass advanced the cursors to the cells' ends)
    cells[i] = o;
  }
  for (int i = tid; i < kLdsSlots / 32; i += kBuildThreads) {
    unsigned int word = 0;
#pragma unroll
    for (int b = 0; b < 32; b++) word |= (lkey[i * 32 + b] != kEmptyKey) ? (1u << b) : 0u;
    bits[i] = word;
  }
  if (tid == 0) { st.table_mask = lmask; st.n_used_tab[0] = 0; st.cursor = 0; st.n_search = M; st.n_filt = 0; }
  OV_STAMP(v, tid == 0 && s == 0, 24);
}

// =============================================================================================
// filter_local_map (computeLocalMap, laser_odometry.cc:286-292): when the window is full the
// local map searched by the next scan is pcl::VoxelGrid(0.4 m) of the whole window — one float
// centroid (x, y, z, intensity) per occupied leaf.  PCL sorts (leaf index, point) pairs and sums
// each leaf's points in that order in float; here "that order" is ascending window index (the
// oracle uses a stable sort; std::sort's order inside a leaf is unspecified in the reference).
//   k_voxel_bbox      one workgroup per stream: clear the previous voxel table, min/max of the
//                     window -> PCL's min_b_ / div_b_
//   k_voxel_insert    leaf index per window point, atomicCAS/atomicAdd grouping (as the 1 m cells)
//   k_voxel_alloc / k_voxel_scatter   window indices grouped by leaf
//   k_voxel_centroid  half-wave per leaf: rank the leaf's window indices (ascending), then one
//                     lane sums in that order -> deterministic, PCL's float accumulation
//   k_filt_insert / k_hash_alloc / k_filt_scatter   1 m cell hash over the filtered points; the
//                     tie-break index carried by the points is PCL's leaf index (= the rank order
//                     of the filtered cloud)
// Every kernel exits immediately unless filter_active().
// =============================================================================================
__global__ __launch_bounds__(1024) void k_voxel_bbox(DevView v, int s0) {
  __shared__ WinIndex w;
  __shared__ float red[6][16];
  const int s = s0 + blockIdx.x;
  StreamState& st = v.state[s];
  const int tid = threadIdx.x;
  // clear the voxel table of the previous build (also when the filter just became inactive)
  {
    const int nup = st.vox_used;
    CellSlot empty; empty.key = kEmptyKey; empty.start = 0; empty.cnt = 0;
    for (int u = tid; u < nup; u += 1024) {
      const int h = v.vox_used_list[(size_t)s * v.map_cap + u];
      v.vox_cells[(size_t)s * v.table_size + h] = empty;
      v.vox_fill[(size_t)s * v.table_size + h] = 0;
    }
  }
  __syncthreads();
  if (tid == 0) { st.vox_used = 0; st.vox_cursor = 0; }
  if (!filter_active(v, st)) return;
  const int M = st.n_map, nf = st.n_frames;
  win_index_load(v, s, nf, w, tid, 1024);
  __syncthreads();
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int m = tid; m < M; m += 1024) {
    const float4 p = win_point(v, s, nf, w, m);
    if (!point_ok(p)) continue;
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int d = 0; d < 3; d++) {
    for (int off = 32; off >= 1; off >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], off));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off));
    }
    if ((tid & 63) == 0) { red[d][tid >> 6] = mn[d]; red[3 + d][tid >> 6] = mx[d]; }
  }
  __syncthreads();
  if (tid == 0) {
    for (int d = 0; d < 3; d++) {
      float a = red[d][0], b = red[3 + d][0];
      for (int k = 1; k < 16; k++) { a = fminf(a, red[d][k]); b = fmaxf(b, red[3 + d][k]); }
      const int minb = (int)floorf(a * v.vox_inv);            // PCL: floor(min_p * inverse_leaf_size_)
      const int maxb = (int)floorf(b * v.vox_inv);
      st.vox_minb[d] = minb;
      st.vox_divb[d] = maxb - minb + 1;
    }
  }
}

__global__ __launch_bounds__(256) void k_voxel_insert(DevView v, int s0) {
  __shared__ WinIndex w;
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  if (!filter_active(v, st)) return;
  const int M = st.n_map, nf = st.n_frames;
  if ((int)(blockIdx.x * 256) >= M) return;
  win_index_load(v, s, nf, w, threadIdx.x, 256);
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float4 p = win_point(v, s, nf, w, m);
  int* pv = v.pt_vox + (size_t)s * v.map_cap + m;
  if (!point_ok(p)) { *pv = -1; return; }
  const int i0 = (int)floorf(p.x * v.vox_inv) - st.vox_minb[0];
  const int i1 = (int)floorf(p.y * v.vox_inv) - st.vox_minb[1];
  const int i2 = (int)floorf(p.z * v.vox_inv) - st.vox_minb[2];
  const unsigned int idx = (unsigned int)(i0 + i1 * st.vox_divb[0] + i2 * st.vox_divb[0] * st.vox_divb[1]);
  const unsigned long long key = (unsigned long long)idx;
  const unsigned int tmask = (unsigned int)v.table_size - 1u;
  CellSlot* cells = v.vox_cells + (size_t)s * v.table_size;
  unsigned int h = hash_cell(key, tmask);
  int found = -1;
  for (int probe = 0; probe < v.table_size; probe++) {
    const unsigned long long prev = atomicCAS(&cells[h].key, kEmptyKey, key);
    if (prev == kEmptyKey) {
      const int u = atomicAdd(&st.vox_used, 1);
      v.vox_used_list[(size_t)s * v.map_cap + u] = (int)h;
      found = (int)h;
      break;
    }
    if (prev == key) { found = (int)h; break; }
    h = (h + 1) & tmask;
  }
  if (found < 0) { atomicOr(&st.status, LIODOM_STATUS_HASH_FULL); *pv = -1; return; }
  atomicAdd(&cells[found].cnt, 1u);
  *pv = found;
}

__global__ __launch_bounds__(256) void k_voxel_alloc(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  if (!filter_active(v, st)) return;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= st.vox_used) return;
  CellSlot* slot = v.vox_cells + (size_t)s * v.table_size + v.vox_used_list[(size_t)s * v.map_cap + u];
  slot->start = (unsigned int)atomicAdd(&st.vox_cursor, (int)slot->cnt);
}

__global__ __launch_bounds__(256) void k_voxel_scatter(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  const StreamState& st = v.state[s];
  if (!filter_active(v, st)) return;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= st.n_map) return;
  const int h = v.pt_vox[(size_t)s * v.map_cap + m];
  if (h < 0) return;
  const size_t ti = (size_t)s * v.table_size + h;
  const unsigned int pos = v.vox_cells[ti].start + atomicAdd(&v.vox_fill[ti], 1u);
  v.vox_pts[(size_t)s * v.map_cap + pos] = m;
}

// 32 lanes per leaf, 8 leaves per workgroup.
__global__ __launch_bounds__(256) void k_voxel_centroid(DevView v, int s0) {
  __shared__ WinIndex w;
  constexpr int CAP = 512;
  __shared__ int ord[8][CAP];
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  if (!filter_active(v, st)) return;
  const int nvox = st.vox_used;
  if ((int)(blockIdx.x * 8) >= nvox) return;
  const int nf = st.n_frames;
  win_index_load(v, s, nf, w, threadIdx.x, 256);
  __syncthreads();
  const int grp = threadIdx.x >> 5, hl = threadIdx.x & 31;
  const int u = blockIdx.x * 8 + grp;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st.n_filt = nvox; st.n_search = nvox; }
  if (u >= nvox) return;
  const CellSlot slot = v.vox_cells[(size_t)s * v.table_size + v.vox_used_list[(size_t)s * v.map_cap + u]];
  const int cnt = (int)slot.cnt;
  int* list = v.vox_pts + (size_t)s * v.map_cap + slot.start;
  // rank sort of the leaf's window indices (all distinct): rank = number of smaller indices
  if (cnt <= CAP) {
    for (int i = hl; i < cnt; i += 32) ord[grp][i] = list[i];
    __builtin_amdgcn_wave_barrier();
    int mine[CAP / 32], rank[CAP / 32];
#pragma unroll
    for (int k = 0; k < CAP / 32; k++) { const int i = hl + 32 * k; mine[k] = (i < cnt) ? ord[grp][i] : 0x7fffffff; rank[k] = 0; }
    for (int j = 0; j < cnt; j++) {
      const int o = ord[grp][j];
#pragma unroll
      for (int k = 0; k < CAP / 32; k++) rank[k] += (o < mine[k]) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < CAP / 32; k++) { const int i = hl + 32 * k; if (i < cnt) list[rank[k]] = mine[k]; }
  } else {
    // very crowded leaf: rank against the list in global memory, result staged through ord/global
    for (int i = hl; i < cnt; i += 32) {
      const int mi = list[i];
      int r = 0;
      for (int j = 0; j < cnt; j++) r += (list[j] < mi) ? 1 : 0;
      v.pt_vox[(size_t)s * v.map_cap + slot.start + r] = mi;      // pt_vox is free again: scratch
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (int i = hl; i < cnt; i += 32) list[i] = v.pt_vox[(size_t)s * v.map_cap + slot.start + i];
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  if (hl == 0) {
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (int i0 = 0; i0 < cnt; i0 += 8) {       // 8 loads in flight, summed in order
      float4 p[8];
#pragma unroll
      for (int k = 0; k < 8; k++) if (i0 + k < cnt) p[k] = win_point(v, s, nf, w, list[i0 + k]);
#pragma unroll
      for (int k = 0; k < 8; k++) if (i0 + k < cnt) { sx += p[k].x; sy += p[k].y; sz += p[k].z; si += p[k].w; }
    }
    const float c = (float)cnt;
    v.filt_pts[(size_t)s * v.map_cap + u] = make_float4(sx / c, sy / c, sz / c, __int_as_float((int)(unsigned int)slot.key));
    v.filt_int[(size_t)s * v.map_cap + u] = si / c;
  }
}

// 1 m cell hash over the filtered cloud (same slot protocol as k_window_insert).
__global__ __launch_bounds__(256) void k_filt_insert(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  if (!filter_active(v, st)) return;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= st.n_filt) return;
  const float4 pt = v.filt_pts[(size_t)s * v.map_cap + u];
  int* pc = v.pt_cell + (size_t)s * v.map_cap + u;
  if (!point_ok(pt)) { *pc = -1; return; }
  const unsigned long long key = pack_cell((int)floorf(pt.x * kCellInv), (int)floorf(pt.y * kCellInv), (int)floorf(pt.z * kCellInv));
  const unsigned int tmask = (unsigned int)v.table_size - 1u;
  CellSlot* cells = v.cells + (size_t)s * v.table_size;
  unsigned int h = hash_cell(key, tmask);
  int found = -1;
  for (int probe = 0; probe < v.table_size; probe++) {
    const unsigned long long prev = atomicCAS(&cells[h].key, kEmptyKey, key);
    if (prev == kEmptyKey) {
      const int k = atomicAdd(&st.n_used_tab[0], 1);
      v.used_cells[(size_t)s * v.used_cap + k] = (int)h;
      atomicOr(&v.cell_bits[((size_t)s * v.table_size + h) >> 5], 1u << (h & 31));
      found = (int)h;
      break;
    }
    if (prev == key) { found = (int)h; break; }
    h = (h + 1) & tmask;
  }
  if (found < 0) { atomicOr(&st.status, LIODOM_STATUS_HASH_FULL); *pc = -1; return; }
  v.pt_rank[(size_t)s * v.map_cap + u] = (int)atomicAdd(&cells[found].cnt, 1u);
  *pc = found;
}

__global__ __launch_bounds__(256) void k_filt_alloc(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  if (!filter_active(v, st)) return;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= st.n_used_tab[0]) return;
  CellSlot* slot = v.cells + (size_t)s * v.table_size + v.used_cells[(size_t)s * v.used_cap + u];
  slot->start = (unsigned int)atomicAdd(&st.cursor, (int)slot->cnt);
}

__global__ __launch_bounds__(256) void k_filt_scatter(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  const StreamState& st = v.state[s];
  if (!filter_active(v, st)) return;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= st.n_filt) return;
  const int h = v.pt_cell[(size_t)s * v.map_cap + u];
  if (h < 0) return;
  const size_t ti = (size_t)s * v.table_size + h;
  const unsigned int pos = v.cells[ti].start + (unsigned int)v.pt_rank[(size_t)s * v.map_cap + u];
  v.sorted_pts[(size_t)s * v.map_cap + pos] = v.filt_pts[(size_t)s * v.map_cap + u];
}

}  // namespace liodom_dev
