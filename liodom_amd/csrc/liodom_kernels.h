// liodom_kernels.h — hand-written HIP kernels of the LiODOM hot path for gfx950 (CDNA4).
//
// Kernel map (one launch covers all streams: blockIdx.y = stream):
//   k_classify        A1/A2  isValidPoint + ring id per point      (feature_extractor.cc:84-179)
//   k_ring_extract    A2-A5  per-ring gather -> LDS tile -> FP64 curvature stencil ->
//                            greedy per-region selection with +-5 suppression (:181-313)
//   k_compact_edges          ring-padded edges -> dense edge cloud (output order of :186-252)
//   k_knn             A9     edges -> world, 27-cell voxel-hash 5-NN, FP64 line gate
//                            (laser_odometry.cc:300-366)
//   k_lm_solve        A10/A11 fused point-to-line residual/Jacobian + 6x6 normal equations,
//                            whole Ceres-style LM solve in one workgroup per stream; second call
//                            also finalises the scan (pose log, prediction :148-150, window
//                            bookkeeping :34-60)
//   k_hash_clear / k_window_insert / k_hash_alloc / k_hash_scatter
//                     A6     append transformed edges to the sliding window (:231-235) and
//                            rebuild the flat voxel hash the next scan's kNN searches
//
// All FP on the parity-critical paths is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/liodom_hip.h"
#include "liodom_math.h"

namespace liodom_dev {

constexpr int kWave = 64;
constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr int kLmThreads = 512;
constexpr int kKnnGroup = 32;            // lanes cooperating on one query
constexpr int kMaxFrames = 256;          // window frames supported by the LDS prefix tables

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
  int32_t n_edges;        // edges of the current scan
  int32_t n_map;          // window points covered by the voxel hash
  int32_t n_used;         // occupied hash cells (current build)
  int32_t n_used_prev;    // occupied cells of the previous build (to clear)
  int32_t cursor;         // allocation cursor into the cell-sorted point array
  int32_t scan_counter;
  uint32_t status;
  int32_t pad;
  liodom_step_info_t info;
};

// Everything the kernels need (passed by value).
struct DevView {
  // parameters
  double min_range, max_range;
  int lidar_type, scan_lines, scan_regions, edges_per_region;
  long long min_points_per_scan;
  int prev_frames;
  int apply_on_ftol;
  // capacities
  int n_streams, max_points, ring_cap, slots_per_ring, edge_cap, map_cap, table_size;
  int pose_log_cap;
  int debug;
  // per-stream arrays (stride = capacity)
  StreamState* state;
  unsigned char* ring_id;   size_t ring_id_stride;
  float4* edges_pad;        // [S][H][slots_per_ring]
  int2* edges_pad_meta;     // (idx_in_ring, src)
  int* ring_nedges;         // [S][H]
  int* ring_npoints;        // [S][H]
  double* curv_dbg;         // [S][H][ring_cap] or null
  float4* edges;            // [S][edge_cap] dense
  int4* edges_meta;         // (ring, idx_in_ring, src, 0)
  float4* corr_a;           // [S][edge_cap]  xyz of NN0, w = valid
  float4* corr_b;           // [S][edge_cap]  xyz of NN1
  int2* corr_idx;           // [S][2][edge_cap] window indices of (NN0, NN1), debug/parity
  float4* win_pts;          // [S][P][edge_cap]
  int* win_n;               // [S][P]
  int* win_base;            // [S][P+1] logical prefix (oldest first)
  int* win_slot;            // [S][P]  logical frame -> slot
  unsigned long long* cell_key;  // [S][table_size]
  unsigned int* cell_cnt;
  unsigned int* cell_start;
  unsigned int* cell_fill;
  int* used_cells;          // [S][map_cap]
  int* pt_cell;             // [S][map_cap]
  float4* sorted_pts;       // [S][map_cap]  xyz + window index bits
  double* pose_log;         // [S][pose_log_cap][7]
  liodom_step_info_t* info_log;  // [S][pose_log_cap]
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ unsigned long long pack_cell(int cx, int cy, int cz) {
  const unsigned long long m = 0x1FFFFFull;  // 21 bits per axis; aliasing only adds far candidates
  return ((unsigned long long)(cx & m) << 42) | ((unsigned long long)(cy & m) << 21) |
         (unsigned long long)(cz & m);
}
__device__ __forceinline__ unsigned int hash_cell(unsigned long long k, unsigned int mask) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (unsigned int)k & mask;
}

// =============================================================================================
// k_classify: one thread per point.  Reads 16 B, writes 1 B.  ids beyond n are 0xFF so the ring
// kernels can read whole 16-byte id chunks.
// =============================================================================================
__global__ __launch_bounds__(256) void k_classify(DevView v, int s0, const float4* __restrict__ in,
                                                   size_t in_stride, int n, int height, int width) {
  const int s = s0 + blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned char id = 0xFF;
  if (i < n) {
    const float4 p = in[(size_t)blockIdx.y * in_stride + i];
    double dist;
    if (valid_point((double)p.x, (double)p.y, (double)p.z, v.min_range, v.max_range, &dist)) {
      int r;
      if (v.lidar_type == 0) {
        r = velodyne_ring((double)p.z, dist, v.scan_lines);
      } else {
        r = (width > 0) ? i / width : -1;     // ring = row (feature_extractor.cc:160-173)
        if (r >= v.scan_lines || r >= height) r = -1;
      }
      if (r >= 0) id = (unsigned char)r;
    }
  }
  if ((size_t)i < v.ring_id_stride) v.ring_id[(size_t)s * v.ring_id_stride + i] = id;
}

// =============================================================================================
// k_ring_extract: one 256-thread workgroup per (ring, stream).
//   phase 0  scan the ring-id bytes (16 per lane per step), stable-compact the indices of this
//            ring's points into LDS (wave prefix via shfl + cross-wave LDS)
//   phase 1  gather XYZ of those points (16-B loads) into an SoA LDS tile
//   phase 2  FP64 11-tap curvature stencil out of LDS
//   phase 3  wave 0: per region, repeat { wavefront argmax over not-picked items (shfl butterfly,
//            lowest index on ties); stop below 0.1 or after epr+1 picks; emit; suppress +-5
//            neighbours while consecutive gaps <= 0.05 (ballot) }.  Regions run in order because
//            suppression carries across region boundaries.
// LDS: c[cap] f64 | px py pz [cap] f32 | src[cap] i32 | picked[cap] u8 | scratch
// =============================================================================================
__device__ __forceinline__ size_t ring_extract_lds_bytes(int cap) {
  return (size_t)cap * (8 + 12 + 4 + 1) + 64;
}

__global__ __launch_bounds__(256) void k_ring_extract(DevView v, int s0, const float4* __restrict__ in,
                                                       size_t in_stride, int n, int height, int width) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ring = blockIdx.x;
  const int s = s0 + blockIdx.y;
  const int cap = v.ring_cap;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* c = reinterpret_cast<double*>(smem);
  float* px = reinterpret_cast<float*>(c + cap);
  float* py = px + cap;
  float* pz = py + cap;
  int* src = reinterpret_cast<int*>(pz + cap);
  unsigned char* picked = reinterpret_cast<unsigned char*>(src + cap);
  int* wtot = reinterpret_cast<int*>(picked + cap);   // 4 ints (cap is a multiple of 16)

  const float4* scan = in + (size_t)blockIdx.y * in_stride;
  const unsigned char* ids = v.ring_id + (size_t)s * v.ring_id_stride;
  const int H = v.scan_lines;
  int* nedges_out = v.ring_nedges + (size_t)s * H + ring;
  int* npoints_out = v.ring_npoints + (size_t)s * H + ring;

  // ---- phase 0: stable compaction of this ring's point indices ----
  int lo = 0, hi = n;
  if (v.lidar_type != 0) {
    lo = ring * width;
    hi = lo + width;
    if (hi > n) hi = n;
    if (lo > hi || ring >= height) { lo = 0; hi = 0; }
  }
  int count = 0;
  for (int chunk = lo & ~15; chunk < hi; chunk += 256 * 16) {
    const int base = chunk + 16 * tid;
    unsigned int m = 0;
    if (base < hi) {
      const uint4 w = *reinterpret_cast<const uint4*>(ids + base);
      const unsigned int words[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const unsigned int byte = (words[b >> 2] >> ((b & 3) * 8)) & 0xFFu;
        const int pos = base + b;
        if (byte == (unsigned int)ring && pos >= lo && pos < hi) m |= (1u << b);
      }
    }
    const int cnt = __popc(m);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int t = wtot[w]; if (w < wave) pre += t; tot += t; }
    int off = count + pre + incl - cnt;
    while (m) {
      const int b = __ffs(m) - 1;
      m &= m - 1;
      if (off < cap) src[off] = base + b;
      off++;
    }
    count += tot;
    __syncthreads();
  }
  const int nr = count;
  if (tid == 0) *npoints_out = nr;
  if (nr > cap) {   // ring does not fit the LDS tile: flagged, ring skipped
    if (tid == 0) { atomicOr(&v.state[s].status, LIODOM_STATUS_RING_OVERFLOW); *nedges_out = 0; }
    return;
  }
  // rings below min_points_per_scan are skipped (feature_extractor.cc:188)
  if ((long long)nr < v.min_points_per_scan || nr < 11) {
    if (tid == 0) *nedges_out = 0;
    if (v.debug && v.curv_dbg) {
      double* dbg = v.curv_dbg + ((size_t)s * H + ring) * cap;
      for (int j = tid; j < nr; j += 256) dbg[j] = __longlong_as_double(0x7ff8000000000000ll);
    }
    return;
  }
  // ---- phase 1: gather the points ----
  for (int k = tid; k < nr; k += 256) {
    const float4 p = scan[src[k]];
    px[k] = p.x; py[k] = p.y; pz[k] = p.z;
    picked[k] = 0;
  }
  __syncthreads();
  // ---- phase 2: curvature ----
  for (int j = 5 + tid; j < nr - 5; j += 256) c[j] = curvature(px, py, pz, j);
  __syncthreads();
  if (v.debug && v.curv_dbg) {
    double* dbg = v.curv_dbg + ((size_t)s * H + ring) * cap;
    for (int j = tid; j < nr; j += 256)
      dbg[j] = (j >= 5 && j < nr - 5) ? c[j] : __longlong_as_double(0x7ff8000000000000ll);
  }
  if (wave != 0) return;

  // ---- phase 3: selection (wave 0) ----
  const int total = nr - 10;                            // :238
  const int R = v.scan_regions;
  const int sector = total / R;                         // :239
  const int epr = v.edges_per_region;
  float4* eout = v.edges_pad + ((size_t)s * H + ring) * v.slots_per_ring;
  int2* mout = v.edges_pad_meta + ((size_t)s * H + ring) * v.slots_per_ring;
  volatile unsigned char* vpicked = picked;
  int nout = 0;
  for (int reg = 0; reg < R; reg++) {
    const int rs = sector * reg;
    const int re = (reg == R - 1) ? total : sector * (reg + 1);   // :242-247
    int picks = 0;
    while (true) {
      // wavefront argmax of smoothness over not-picked items; ties -> lowest index
      unsigned long long bkey = 0;
      int bidx = -1;
      for (int k = rs + lane; k < re; k += 64) {
        const int j = k + 5;
        if (!vpicked[j]) {
          const unsigned long long key = (unsigned long long)__double_as_longlong(c[j]);
          if (bidx < 0 || key > bkey) { bkey = key; bidx = j; }
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long ok = __shfl_xor(bkey, off);
        const int oi = __shfl_xor(bidx, off);
        const bool better = (oi >= 0) && (bidx < 0 || ok > bkey || (ok == bkey && oi < bidx));
        if (better) { bkey = ok; bidx = oi; }
      }
      if (bidx < 0) break;                                         // every item already picked
      const double best = __longlong_as_double((long long)bkey);
      if (best < 0.1 || picks > epr) break;                        // :270
      const int j = bidx;
      if (lane == 0) {
        if (nout < v.slots_per_ring) {
          const int sidx = src[j];
          eout[nout] = make_float4(px[j], py[j], pz[j], scan[sidx].w);   // :275
          mout[nout] = make_int2(j, sidx);
        }
      }
      nout++;
      picks++;                                                     // :276
      // suppression: lanes 0-4 test forward gaps l=1..5, lanes 8-12 backward gaps (:280-310)
      bool brk = false;
      const int l = (lane & 7) + 1;
      if (lane < 5) brk = gap_sq(px, py, pz, j + l, j + l - 1) > 0.05;
      else if (lane >= 8 && lane < 13) brk = gap_sq(px, py, pz, j - l, j - l + 1) > 0.05;
      const unsigned long long bal = __ballot(brk);
      const unsigned int bf = (unsigned int)(bal & 0x1Fu), bb = (unsigned int)((bal >> 8) & 0x1Fu);
      const int nf = bf ? (__ffs(bf) - 1) : 5;    // forward neighbours marked
      const int nb = bb ? (__ffs(bb) - 1) : 5;
      if (lane == 0) vpicked[j] = 1;                               // :277
      if (lane < 5 && (lane & 7) < nf) vpicked[j + l] = 1;         // :293
      if (lane >= 8 && lane < 13 && (lane & 7) < nb) vpicked[j - l] = 1;   // :309
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0) {
    if (nout > v.slots_per_ring) { atomicOr(&v.state[s].status, LIODOM_STATUS_EDGE_OVERFLOW); nout = v.slots_per_ring; }
    *nedges_out = nout;
  }
}

// =============================================================================================
// k_compact_edges: one workgroup per stream; ring-padded edges -> dense edge cloud in the
// reference's output order; resets the per-scan diagnostics.
// =============================================================================================
__global__ __launch_bounds__(256) void k_compact_edges(DevView v, int s0) {
  __shared__ int pre[257];
  const int s = s0 + blockIdx.x;
  const int H = v.scan_lines;
  const int* rn = v.ring_nedges + (size_t)s * H;
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int r = 0; r < H; r++) { pre[r] = acc; acc += rn[r]; }
    pre[H] = acc;
    StreamState& st = v.state[s];
    st.n_edges = acc > v.edge_cap ? v.edge_cap : acc;
    st.info.n_edges = st.n_edges;
    st.info.matches[0] = 0; st.info.matches[1] = 0;
    st.info.map_points = st.n_map;
    for (int k = 0; k < 2; k++) {
      st.info.lm[k].iterations = 0; st.info.lm[k].accepted = 0; st.info.lm[k].termination = LM_TERM_NO_RESIDUALS;
      st.info.lm[k].pad = 0; st.info.lm[k].initial_cost = 0.0; st.info.lm[k].final_cost = 0.0;
    }
  }
  __syncthreads();
  const int E = pre[H] > v.edge_cap ? v.edge_cap : pre[H];
  for (int e = threadIdx.x; e < E; e += 256) {
    int lo = 0, hi = H;            // largest r with pre[r] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= e) lo = mid; else hi = mid; }
    const int r = lo, k = e - pre[r];
    const size_t pi = ((size_t)s * H + r) * v.slots_per_ring + k;
    v.edges[(size_t)s * v.edge_cap + e] = v.edges_pad[pi];
    const int2 m = v.edges_pad_meta[pi];
    v.edges_meta[(size_t)s * v.edge_cap + e] = make_int4(r, m.x, m.y, 0);
  }
}

// For liodom_odometry_step (edges supplied by the caller): set counts and reset diagnostics.
__global__ void k_set_edges(DevView v, int s0, int n_edges) {
  const int s = s0 + blockIdx.x;
  if (threadIdx.x == 0) {
    StreamState& st = v.state[s];
    st.n_edges = n_edges;
    st.info.n_edges = n_edges;
    st.info.matches[0] = 0; st.info.matches[1] = 0;
    st.info.map_points = st.n_map;
    for (int k = 0; k < 2; k++) {
      st.info.lm[k].iterations = 0; st.info.lm[k].accepted = 0; st.info.lm[k].termination = LM_TERM_NO_RESIDUALS;
      st.info.lm[k].pad = 0; st.info.lm[k].initial_cost = 0.0; st.info.lm[k].final_cost = 0.0;
    }
  }
}

// =============================================================================================
// k_knn: 32 lanes per edge (8 edges per 256-thread workgroup).
//   lane c < 27 probes the voxel hash for neighbour cell c of the query's 1 m cell (a 27-cell
//   search is exact for every edge that can pass the sq_dist[4] < 1.0 gate);
//   the half-wave then strides over the points of each occupied cell (coalesced 16-B loads),
//   every lane keeping its own sorted top-5 of (float distance, window index); five shfl
//   min-reductions merge the 32 lists.  Line gate in FP64, then NN0 / NN1 are written as the
//   line points (laser_odometry.cc:351-357).
// =============================================================================================
__global__ __launch_bounds__(256) void k_knn(DevView v, int s0, int outer_it) {
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  if (!st.initialized) return;
  const int E = st.n_edges;
  const int e = blockIdx.x * (256 / kKnnGroup) + (threadIdx.x / kKnnGroup);
  if (e >= E) return;
  const int hl = threadIdx.x & (kKnnGroup - 1);
  const int half_shift = (threadIdx.x & 32);     // 0 or 32: which half of the wave
  const float4 p = v.edges[(size_t)s * v.edge_cap + e];
  float qx, qy, qz;
  {
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = st.odom[i];
    transform_point(T, p.x, p.y, p.z, &qx, &qy, &qz);          // :307-308
  }
  float4* ca = v.corr_a + (size_t)s * v.edge_cap + e;
  float4* cb = v.corr_b + (size_t)s * v.edge_cap + e;
  int2* cidx = v.corr_idx + ((size_t)s * 2 + outer_it) * v.edge_cap + e;
  const bool qfinite = ld_isfinite((double)qx) && ld_isfinite((double)qy) && ld_isfinite((double)qz) &&
                       fabsf(qx) < 1.0e9f && fabsf(qy) < 1.0e9f && fabsf(qz) < 1.0e9f;
  if (!qfinite) {
    if (hl == 0) { *ca = make_float4(0, 0, 0, 0); *cb = make_float4(0, 0, 0, 0); *cidx = make_int2(-1, -1); }
    return;
  }
  const int cx = (int)floorf(qx), cy = (int)floorf(qy), cz = (int)floorf(qz);
  const unsigned int tmask = (unsigned int)v.table_size - 1u;
  const unsigned long long* keys = v.cell_key + (size_t)s * v.table_size;
  unsigned int start = 0, cnt = 0;
  if (hl < 27) {
    const int dx = hl % 3 - 1, dy = (hl / 3) % 3 - 1, dz = hl / 9 - 1;
    const unsigned long long key = pack_cell(cx + dx, cy + dy, cz + dz);
    unsigned int h = hash_cell(key, tmask);
    for (int probe = 0; probe < v.table_size; probe++) {
      const unsigned long long k = keys[h];
      if (k == key) {
        start = v.cell_start[(size_t)s * v.table_size + h];
        cnt = v.cell_cnt[(size_t)s * v.table_size + h];
        break;
      }
      if (k == kEmptyKey) break;
      h = (h + 1) & tmask;
    }
  }
  unsigned int occ = (unsigned int)((__ballot(cnt > 0) >> half_shift) & 0xFFFFFFFFull);
  // per-lane sorted top-5: distance, window index, position in the cell-sorted array
  float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY, d3 = INFINITY, d4 = INFINITY;
  int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff, i3 = 0x7fffffff, i4 = 0x7fffffff;
  int p0 = -1, p1 = -1, p2 = -1, p3 = -1, p4 = -1;
  const float4* sp = v.sorted_pts + (size_t)s * v.map_cap;
  while (occ) {
    const int cl = __ffs(occ) - 1;
    occ &= occ - 1;
    const unsigned int cst = __shfl(start, cl, kKnnGroup);
    const unsigned int ccn = __shfl(cnt, cl, kKnnGroup);
    for (unsigned int i = hl; i < ccn; i += kKnnGroup) {
      const float4 m = sp[cst + i];
      const float d = sqdist_f(qx, qy, qz, m.x, m.y, m.z);
      const int wi = __float_as_int(m.w);
      if (d < d4 || (d == d4 && wi < i4)) {
        d4 = d; i4 = wi; p4 = (int)(cst + i);
        if (d4 < d3 || (d4 == d3 && i4 < i3)) { float td = d3; d3 = d4; d4 = td; int ti = i3; i3 = i4; i4 = ti; ti = p3; p3 = p4; p4 = ti; }
        if (d3 < d2 || (d3 == d2 && i3 < i2)) { float td = d2; d2 = d3; d3 = td; int ti = i2; i2 = i3; i3 = ti; ti = p2; p2 = p3; p3 = ti; }
        if (d2 < d1 || (d2 == d1 && i2 < i1)) { float td = d1; d1 = d2; d2 = td; int ti = i1; i1 = i2; i2 = ti; ti = p1; p1 = p2; p2 = ti; }
        if (d1 < d0 || (d1 == d0 && i1 < i0)) { float td = d0; d0 = d1; d1 = td; int ti = i0; i0 = i1; i1 = ti; ti = p0; p0 = p1; p1 = ti; }
      }
    }
  }
  // merge the 32 per-lane lists: 5 rounds of min-reduction over (distance bits, window index)
  float nd4 = INFINITY;
  int mypos = -1, mywin = -1;
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const unsigned long long key = ((unsigned long long)(unsigned int)__float_as_int(d0) << 32) | (unsigned int)i0;
    unsigned long long mk = key;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(mk, off, kKnnGroup);
      mk = o < mk ? o : mk;
    }
    const unsigned int win = (unsigned int)((__ballot(key == mk) >> half_shift) & 0xFFFFFFFFull);
    const int wl = __ffs(win) - 1;
    const int wpos = __shfl(p0, wl, kKnnGroup);
    if (hl == r) { mypos = wpos; mywin = (int)(unsigned int)(mk & 0xFFFFFFFFull); }
    if (r == 4) nd4 = __int_as_float((int)(mk >> 32));
    if (hl == wl) {   // pop
      d0 = d1; d1 = d2; d2 = d3; d3 = d4; d4 = INFINITY;
      i0 = i1; i1 = i2; i2 = i3; i3 = i4; i4 = 0x7fffffff;
      p0 = p1; p1 = p2; p2 = p3; p3 = p4; p4 = -1;
    }
  }
  bool valid = (nd4 < 1.0f);                                     // :324 (inf when < 5 candidates)
  float4 mine = make_float4(0, 0, 0, 0);
  if (valid && hl < 5) mine = sp[mypos];
  float nx[5], ny[5], nz[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    nx[j] = __shfl(mine.x, j, kKnnGroup);
    ny[j] = __shfl(mine.y, j, kKnnGroup);
    nz[j] = __shfl(mine.z, j, kKnnGroup);
  }
  const int wa = __shfl(mywin, 0, kKnnGroup), wb = __shfl(mywin, 1, kKnnGroup);
  if (valid) valid = line_gate(nx, ny, nz);                      // :325-344
  if (hl == 0) {
    if (valid) {
      *ca = make_float4(nx[0], ny[0], nz[0], 1.0f);              // :351-353
      *cb = make_float4(nx[1], ny[1], nz[1], 0.0f);              // :355-357
      *cidx = make_int2(wa, wb);
      atomicAdd(&st.info.matches[outer_it], 1);                  // :346
    } else {
      *ca = make_float4(0, 0, 0, 0); *cb = make_float4(0, 0, 0, 0); *cidx = make_int2(-1, -1);
    }
  }
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
__device__ __forceinline__ void lm_eval(const DevView& v, int s, int E, const double* Rm_sh,
                                        double* part /*[8][kAccN]*/, double* acc_out /*[kAccN]*/) {
  double Rm[12];
#pragma unroll
  for (int i = 0; i < 12; i++) Rm[i] = Rm_sh[i];
  double acc[kAccN];
#pragma unroll
  for (int i = 0; i < kAccN; i++) acc[i] = 0.0;
  const float4* ed = v.edges + (size_t)s * v.edge_cap;
  const float4* ca = v.corr_a + (size_t)s * v.edge_cap;
  const float4* cb = v.corr_b + (size_t)s * v.edge_cap;
  for (int e = threadIdx.x; e < E; e += kLmThreads) {
    const float4 A = ca[e];
    if (A.w != 0.0f) {
      const float4 B = cb[e];
      const float4 P = ed[e];
      const double p[3] = {(double)P.x, (double)P.y, (double)P.z};     // :347-349 sensor frame
      const double a[3] = {(double)A.x, (double)A.y, (double)A.z};
      const double b[3] = {(double)B.x, (double)B.y, (double)B.z};
      residual_accumulate(Rm, p, a, b, v.min_range, v.max_range, acc);
    }
  }
#pragma unroll
  for (int i = 0; i < kAccN; i++) {
    double x = acc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    acc[i] = x;
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < kAccN; i++) part[wave * kAccN + i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < kAccN) {
    double x = 0.0;
    for (int w = 0; w < kLmThreads / 64; w++) x += part[w * kAccN + threadIdx.x];
    acc_out[threadIdx.x] = x;
  }
  __syncthreads();
}

__device__ void finalize_scan(const DevView& v, int s, StreamState& st) {
  // pose as published (laser_odometry.cc:403-412 with identity laser_to_base)
  double q[4];
  quat_from_rot(st.odom, q);
  const int k = st.scan_counter;
  st.info.scan_index = k;
  st.info.status = st.status;
  if (k < v.pose_log_cap) {
    double* pl = v.pose_log + ((size_t)s * v.pose_log_cap + k) * 7;
    pl[0] = q[0]; pl[1] = q[1]; pl[2] = q[2]; pl[3] = q[3];
    pl[4] = st.odom[3]; pl[5] = st.odom[7]; pl[6] = st.odom[11];
    v.info_log[(size_t)s * v.pose_log_cap + k] = st.info;
  }
  st.scan_counter = k + 1;
  for (int i = 0; i < 12; i++) st.final_odom[i] = st.odom[i];
  // prediction for the next scan: odom * (prev^-1 * odom)   (:148-150)
  double inv[12], rel[12], pred[12];
  iso_inverse(st.prev_odom, inv);
  iso_mul(inv, st.odom, rel);
  iso_mul(st.odom, rel, pred);
  for (int i = 0; i < 12; i++) { st.prev_odom[i] = st.odom[i]; st.odom[i] = pred[i]; }
  quat_from_rot(st.odom, st.param_q);                              // :186-190
  st.param_t[0] = st.odom[3]; st.param_t[1] = st.odom[7]; st.param_t[2] = st.odom[11];   // :192-195
  // LocalMapManager::addPointCloud (:34-60) on a ring of P frame slots
  const int P = v.prev_frames;
  const int slot = st.frame_count % P;
  int* wn = v.win_n + (size_t)s * P;
  wn[slot] = st.n_edges;
  st.frame_count++;
  st.n_frames = st.frame_count < P ? st.frame_count : P;
  int* wb = v.win_base + (size_t)s * (P + 1);
  int* ws = v.win_slot + (size_t)s * P;
  int acc = 0;
  for (int j = 0; j < st.n_frames; j++) {
    const int sl = (st.frame_count - st.n_frames + j) % P;
    ws[j] = sl; wb[j] = acc; acc += wn[sl];
  }
  wb[st.n_frames] = acc;
  st.n_map = acc;
  st.n_used_prev = st.n_used;
  st.n_used = 0;
  st.cursor = 0;
}

__global__ __launch_bounds__(kLmThreads) void k_lm_solve(DevView v, int s0, int outer_it) {
  __shared__ double sh_pose[12];
  __shared__ double sh_part[(kLmThreads / 64) * kAccN];
  __shared__ double sh_acc[kAccN];
  __shared__ LmState lm;
  __shared__ int sh_flag;
  const int s = s0 + blockIdx.x;
  StreamState& st = v.state[s];
  if (!st.initialized) {
    // first frame (:108-136): no solve; pose stays identity, edges enter the window raw
    if (outer_it == 1 && threadIdx.x == 0) {
      st.append_raw = 1;
      finalize_scan(v, s, st);
      st.initialized = 1;
    }
    return;
  }
  const int E = st.n_edges;
  const int nblocks = st.info.matches[outer_it];
  if (threadIdx.x == 0) iso_from_qt(st.param_q, st.param_t, sh_pose);
  __syncthreads();
  lm_eval(v, s, E, sh_pose, sh_part, sh_acc);
  if (threadIdx.x == 0) {
    sh_flag = lm_begin(lm, st.param_q, st.param_t, sh_acc, nblocks, v.apply_on_ftol);
    if (sh_flag == LM_NEED_EVAL) iso_from_qt(lm.cand_q, lm.cand_t, sh_pose);
  }
  __syncthreads();
  while (sh_flag == LM_NEED_EVAL) {
    lm_eval(v, s, E, sh_pose, sh_part, sh_acc);
    if (threadIdx.x == 0) {
      sh_flag = lm_update(lm, sh_acc);
      if (sh_flag == LM_NEED_EVAL) iso_from_qt(lm.cand_q, lm.cand_t, sh_pose);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int k = 0; k < 4; k++) st.param_q[k] = lm.q[k];
    for (int k = 0; k < 3; k++) st.param_t[k] = lm.t[k];
    iso_from_qt(st.param_q, st.param_t, st.odom);                  // :222-227
    liodom_lm_trace_t& tr = st.info.lm[outer_it];
    tr.iterations = lm.iter; tr.accepted = lm.accepted; tr.termination = lm.termination; tr.pad = 0;
    tr.initial_cost = lm.initial_cost; tr.final_cost = lm.cost;
    if (outer_it == 1) {
      st.append_raw = 0;
      finalize_scan(v, s, st);
    }
  }
}

// =============================================================================================
// Sliding window + voxel hash rebuild.
// =============================================================================================
__global__ __launch_bounds__(256) void k_hash_clear(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  const StreamState& st = v.state[s];
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= st.n_used_prev) return;
  const int h = v.used_cells[(size_t)s * v.map_cap + u];
  const size_t ti = (size_t)s * v.table_size + h;
  v.cell_key[ti] = kEmptyKey;
  v.cell_cnt[ti] = 0;
  v.cell_fill[ti] = 0;
}

// One thread per window point (oldest frame first).  Points of the newest frame are produced
// here: edges transformed by the solved pose in FP64 and rounded to float
// (laser_odometry.cc:231-232), then stored in the window slot (:235).  Every point is counted
// into its 1 m cell (atomicCAS insert + atomicAdd count).
__global__ __launch_bounds__(256) void k_window_insert(DevView v, int s0) {
  __shared__ int sbase[kMaxFrames + 1];
  __shared__ int sslot[kMaxFrames];
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  const int M = st.n_map;
  if ((int)(blockIdx.x * 256) >= M) return;
  const int P = v.prev_frames, nf = st.n_frames;
  for (int j = threadIdx.x; j <= nf; j += 256) sbase[j] = v.win_base[(size_t)s * (P + 1) + j];
  for (int j = threadIdx.x; j < nf; j += 256) sslot[j] = v.win_slot[(size_t)s * P + j];
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  int lo = 0, hi = nf;             // largest j with sbase[j] <= m
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sbase[mid] <= m) lo = mid; else hi = mid; }
  const int j = lo, idx = m - sbase[j], slot = sslot[j];
  float4* wp = v.win_pts + ((size_t)s * P + slot) * v.edge_cap + idx;
  float4 pt;
  if (j == nf - 1) {
    const float4 e = v.edges[(size_t)s * v.edge_cap + idx];
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
  int* pc = v.pt_cell + (size_t)s * v.map_cap + m;
  const bool fin = ld_isfinite((double)pt.x) && ld_isfinite((double)pt.y) && ld_isfinite((double)pt.z) &&
                   fabsf(pt.x) < 1.0e9f && fabsf(pt.y) < 1.0e9f && fabsf(pt.z) < 1.0e9f;
  if (!fin) { *pc = -1; return; }
  const unsigned long long key = pack_cell((int)floorf(pt.x), (int)floorf(pt.y), (int)floorf(pt.z));
  const unsigned int tmask = (unsigned int)v.table_size - 1u;
  unsigned long long* keys = v.cell_key + (size_t)s * v.table_size;
  unsigned int h = hash_cell(key, tmask);
  int found = -1;
  for (int probe = 0; probe < v.table_size; probe++) {
    const unsigned long long prev = atomicCAS(&keys[h], kEmptyKey, key);
    if (prev == kEmptyKey) {
      const int u = atomicAdd(&st.n_used, 1);
      v.used_cells[(size_t)s * v.map_cap + u] = (int)h;
      found = (int)h;
      break;
    }
    if (prev == key) { found = (int)h; break; }
    h = (h + 1) & tmask;
  }
  if (found < 0) { atomicOr(&st.status, LIODOM_STATUS_HASH_FULL); *pc = -1; return; }
  atomicAdd(&v.cell_cnt[(size_t)s * v.table_size + found], 1u);
  *pc = found;
}

__global__ __launch_bounds__(256) void k_hash_alloc(DevView v, int s0) {
  const int s = s0 + blockIdx.y;
  StreamState& st = v.state[s];
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= st.n_used) return;
  const int h = v.used_cells[(size_t)s * v.map_cap + u];
  const size_t ti = (size_t)s * v.table_size + h;
  v.cell_start[ti] = (unsigned int)atomicAdd(&st.cursor, (int)v.cell_cnt[ti]);
}

__global__ __launch_bounds__(256) void k_hash_scatter(DevView v, int s0) {
  __shared__ int sbase[kMaxFrames + 1];
  __shared__ int sslot[kMaxFrames];
  const int s = s0 + blockIdx.y;
  const StreamState& st = v.state[s];
  const int M = st.n_map;
  if ((int)(blockIdx.x * 256) >= M) return;
  const int P = v.prev_frames, nf = st.n_frames;
  for (int j = threadIdx.x; j <= nf; j += 256) sbase[j] = v.win_base[(size_t)s * (P + 1) + j];
  for (int j = threadIdx.x; j < nf; j += 256) sslot[j] = v.win_slot[(size_t)s * P + j];
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int h = v.pt_cell[(size_t)s * v.map_cap + m];
  if (h < 0) return;
  int lo = 0, hi = nf;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sbase[mid] <= m) lo = mid; else hi = mid; }
  const float4 pt = v.win_pts[((size_t)s * P + sslot[lo]) * v.edge_cap + (m - sbase[lo])];
  const size_t ti = (size_t)s * v.table_size + h;
  const unsigned int pos = v.cell_start[ti] + atomicAdd(&v.cell_fill[ti], 1u);
  v.sorted_pts[(size_t)s * v.map_cap + pos] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
}

}  // namespace liodom_dev
