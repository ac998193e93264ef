// kernels_rebuild.h — sliding window + cell hash: k_window_insert / k_hash_alloc / k_hash_scatter, the streamed rebuild, k_hash_build.
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, in this order; not a standalone header).
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
      const int slot_u = base + (int)__popcll(cm & ((1ull << lane) - 1ull));
      if (created) { if (slot_u < v.used_cap) v.used_cells[(size_t)sp * v.used_cap + slot_u] = (int)h; else atomicOr(&st.status, LIODOM_STATUS_HASH_FULL); }
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
__device__ __forceinline__ int kept_frames_table(const DevView& v, int s, const StreamState& st, int* sbase, int* sslot, int fc_of = -1) {
  const int P = v.prev_frames, fc = fc_of >= 0 ? fc_of : st.reb_frame_count;
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
__device__ __forceinline__ float4 predicted_point(const StreamState& st, const float4& e, int fc_of = -1) {
  if (fc_of < 0 && !st.reb_initialized) return e;
  const int fc = fc_of >= 0 ? fc_of : st.reb_frame_count;
  double T[12];
#pragma unroll
  for (int i = 0; i < 12; i++) T[i] = st.pred_odom[fc & 1][i];
  float4 q;
  transform_point(T, e.x, e.y, e.z, &q.x, &q.y, &q.z);
  q.w = e.w;
  return q;
}

// COUNT (block < nC) and PAD (the nP blocks behind them)
// keep (chain mode): PAD also copies the scan's edges into edges_keep, which APPEND reads instead of the edge buffer.  APPEND is a
// launch of its own there and may start after the scan's pose has been published — i.e. after the host may have collected it and
// handed the edge buffer's ticket slot to the extraction of a later scan (found by tools/chain_hammer.py: APPEND transformed the
// edges of scan k + 3, for which nothing had been padded).  On the four-launch chain the appending workgroups are resident, with
// their edge in registers, before the pose exists.
__device__ __forceinline__ void rebuild_count_and_pad(const DevView& v, int s, StreamState& st, int eb, int block, int* sbase, int* sslot, bool keep = false) {
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
  const float4 e_in = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (live ? idx : 0)];
  if (keep && live && v.edges_keep) v.edges_keep[(size_t)s * v.edge_cap + idx] = e_in;
  if (keep && idx == 0) st.reb_pad = n_new;                    // (the count APPEND iterates over, for the same reason)
  const float4 q = predicted_point(st, e_in);
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
      const int slot_u = base + (int)__popcll(cm & ((1ull << lane) - 1ull));
      if (created) { if (slot_u < v.used_cap) v.used_cells[(size_t)sp * v.used_cap + slot_u] = (int)h; else atomicOr(&st.status, LIODOM_STATUS_HASH_FULL); }      // (never past the list's end, whatever state a failed scan left)
    }
  }
}

// ALLOC, by the extra workgroups of the scan's second k_knn launch — or, when that pass is overlapped with the first solve
// on a stream of its own, by k_rebuild_alloc between the two solve launches (a launch boundary must separate ALLOC from
// COUNT / PAD before it and from SCATTER / APPEND behind it) —: start offsets of the occupied cells (any order:
// only contiguity per cell matters), room = points counted + places padded; cell_pad becomes the end of the range.
__device__ __forceinline__ void rebuild_alloc(const DevView& v, int s, StreamState& st, int block, int nblocks) {
  const int par = (st.reb_frame_count + 1) & 1, sp = s + par * v.n_streams;
  const int nu = min(st.n_used_tab[par], v.used_cap);      // (bounded: see hash_clear_used)
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
__device__ __forceinline__ void rebuild_finish(const DevView& v, int s, StreamState& st, int eb, int block, unsigned int seq, int* sbase, int* sslot, bool kept = false) {
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
  const int n_new = (kept && v.edges_keep) ? st.reb_pad : st.n_edges_buf[eb];
  if (block * nt >= n_new) return;
  const int idx = block * nt + tid;
  const bool live = idx < n_new;
  const float4 e = (kept && v.edges_keep) ? v.edges_keep[(size_t)s * v.edge_cap + (live ? idx : 0)]
                                          : v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (live ? idx : 0)];     // (loaded before the wait)
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
    const unsigned long long* base = v.pose_xch + (size_t)s * 64;
    const unsigned int tag = (unsigned int)fc + 1u;
    unsigned long long g = 0, t0w = 0;
    unsigned int spins = 0;
    bool ok;
    while (true) {
      if (tid < 25) g = __hip_atomic_load((gu64*)(base + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = tid >= 25 || (unsigned int)(g >> 32) == tag;
      if (__all(ok)) break;
      if (++spins > 4000000u || wait_expired(spins, t0w)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    const bool all_ok = __all(ok);
    const unsigned long long gflags = __shfl(g, 24);                 // flags granule: low word = append_raw
    if (tid == 0) sh_hand = all_ok ? (int)(unsigned int)gflags + 1 : 0;   // 0: timed out, else raw + 1
    const unsigned long long lo = __shfl(g, 2 * (tid % 12)), hi = __shfl(g, 2 * (tid % 12) + 1);
    if (tid < 12) sh_T[tid] = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
    if (!all_ok && tid == 0) atomicOr(&st.status, LIODOM_STATUS_LM_SYNC_TIMEOUT);
    INJECT_DELAY(17);
  }
  __syncthreads();
  const bool dbga = (s == 0) && (block == 0) && (tid == 0);
  DBG_STAMP(v, dbga, 2, 29);
  OV_STAMP(v, dbga, 28);
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
  OV_STAMP(v, dbga, 26);
}

// Speculative hand-over not confirmed (kernels_sync.h; rare): the new frame was appended at the pose the finalising solve held before
// its last evaluation, and that evaluation's step was accepted after all.  Repair, by the first `nblocks` workgroups of k_chain_redo0
// (behind the next scan's first pass, whose bookkeeping has not been committed: the stream's state is as the rebuild left it):
// every workgroup takes back what APPEND did with its points (the same decisions, from the same inputs: the place in the cell is
// given back by decrementing the cell's count — the order inside a cell carries no meaning), all of them meet, then the points
// are appended again at the confirmed pose.  Results: those of an APPEND that had waited for the confirmed pose.
__device__ __forceinline__ bool append_pose_copy(const DevView& v, int s, unsigned int tag, int copy, double* T /*LDS [12]*/, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  __shared__ int sh_ok;
  const int tid = threadIdx.x;
  if (tid < 64) {
    const unsigned long long* base = v.pose_xch + (size_t)s * 64 + copy * 32;
    unsigned long long g = 0, t0w = 0;
    unsigned int spins = 0;
    bool ok;
    while (true) {
      if (tid < 25) g = __hip_atomic_load((gu64*)(base + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = tid >= 25 || (unsigned int)(g >> 32) == tag;
      if (__all(ok)) break;
      if (++spins > 2000000u || wait_expired(spins, t0w)) break;
      __builtin_amdgcn_s_sleep(4);
    }
    const bool all_ok = __all(ok);
    const unsigned long long lo = __shfl(g, 2 * (tid % 12)), hi = __shfl(g, 2 * (tid % 12) + 1);
    if (tid < 12) T[tid] = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
    if (tid == 0) { sh_ok = all_ok ? 1 : 0; if (!all_ok) atomicOr(status, LIODOM_STATUS_LM_SYNC_TIMEOUT); }
    INJECT_DELAY(18);
  }
  __syncthreads();
  return sh_ok != 0;
}
__device__ __forceinline__ void append_fix(const DevView& v, int s, StreamState& st, int block, int nblocks, unsigned int epoch, int fc, int* sbase, int* sslot) {
  __shared__ double sh_Ts[12], sh_Tf[12];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int P = v.prev_frames;      // (fc: frames appended before the scan under repair — the next scan's first pass has advanced the stream's own count)
  const int par = (fc + 1) & 1, sp = s + par * v.n_streams;
  float4* sorted = v.sorted_pts + (size_t)sp * v.sorted_cap;
  CellSlot* cells = v.cells + (size_t)sp * v.table_size;
  const unsigned int tmask = (unsigned int)v.table_size - 1u;
  const unsigned int tag = (unsigned int)fc + 1u;
  const bool have = append_pose_copy(v, s, tag, 0, sh_Ts, &st.status) && append_pose_copy(v, s, tag, 1, sh_Tf, &st.status);
  const int nk = kept_frames_table(v, s, st, sbase, sslot, fc);
  const int Mk = nk > 0 ? sbase[nk] : 0;
  const int n_new = v.edges_keep ? st.reb_pad : 0;
  const int idx = block * nt + tid;
  const bool live = have && idx < n_new;
  const float4 e = v.edges_keep[(size_t)s * v.edge_cap + (live ? idx : 0)];
  const float4 q = predicted_point(st, e, fc);
  const bool q_fin = point_finite(q);
  const float dc = v.rebuild_delta - 1.0e-3f;
  // the cell a point ends up in at pose T (-1: the overflow list; -2: the point is not part of the map) — as rebuild_finish decides it
  auto place_of = [&](const double* Tl, float4& pt) -> int {
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = Tl[i];
    transform_point(T, e.x, e.y, e.z, &pt.x, &pt.y, &pt.z);
    pt.w = e.w;
    if (!point_finite(pt)) return -2;
    if (!(q_fin && fabsf(pt.x - q.x) < dc && fabsf(pt.y - q.y) < dc && fabsf(pt.z - q.z) < dc)) return -1;
    const unsigned long long key = pack_cell((int)floorf(pt.x * kCellInv), (int)floorf(pt.y * kCellInv), (int)floorf(pt.z * kCellInv));
    unsigned int h = hash_cell(key, tmask);
    for (int probe = 0; probe < v.table_size; probe++) {
      const unsigned long long k = cells[h].key;
      if (k == key) return (int)h;
      if (k == kEmptyKey) break;
      h = (h + 1) & tmask;
    }
    return -1;
  };
  float4 pt;
  if (live) {
    const int hf = place_of(sh_Ts, pt);
    if (hf >= 0) atomicSub(&cells[hf].cnt, 1u);
    else if (hf == -1) atomicSub(&st.n_ovf[par], 1);
  }
  // all workgroups of the repair have taken their points back before any of them appends again
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (tid == 0) { epoch_arrive(v.redo_sync + 0, epoch); (void)epoch_wait(v.redo_sync + 0, epoch, (unsigned int)nblocks, &st.status); }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (live) {
    const int hf = place_of(sh_Tf, pt);
    v.win_pts[((size_t)s * P + (fc % P)) * v.edge_cap + idx] = pt;
    const int m = Mk + idx;
    if (hf >= 0) {
      const unsigned int pos = cells[hf].start + atomicAdd(&cells[hf].cnt, 1u);
      if (pos < v.cell_pad[(size_t)sp * v.table_size + hf]) sorted[pos] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
      else atomicOr(&st.status, LIODOM_STATUS_HASH_FULL);
    } else if (hf == -1) {
      const int i = atomicAdd(&st.n_ovf[par], 1);
      sorted[v.ovf_base + i] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
    }
  }
  // ... and all of them have appended before the pass is repeated (the waiting workgroups of k_chain_redo0)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (tid == 0) epoch_arrive(v.redo_sync + 1, epoch);
}

__device__ __forceinline__ void rebuild_beside_solve(const DevView& v, int s, StreamState& st, int eb, int outer_it, int block, int nblocks, unsigned int seq, int* sbase, int* sslot) {
  (void)nblocks;
  if (outer_it == 0) rebuild_count_and_pad(v, s, st, eb, block, sbase, sslot);
  else rebuild_finish(v, s, st, eb, block, seq, sbase, sslot);
}

// Chain mode: the rebuild's steps as launches of their own on the stream of the kNN passes (light kernels: as extra workgroups of
// k_lm_solve every one of them owned a whole CU — 256 VGPRs x 8 waves — and could only be placed on a CU that ran nothing else).
// COUNT + PAD ride on the second kNN pass's launch (k_knn<256, true>: its extra workgroups), ALLOC is k_rebuild_alloc.
#ifndef LIODOM_REBFIN_THREADS
#define LIODOM_REBFIN_THREADS LIODOM_LM_THREADS
#endif
constexpr int kRebFinThreads = LIODOM_REBFIN_THREADS;
__global__ __launch_bounds__(kRebFinThreads) void k_rebuild_fin(DevView v, int s0, int eb) {         // APPEND (waits for the solved pose), CLEAR, SCATTER
  __shared__ int sh_cnt[kMaxFrames + 1];
  __shared__ int sh_slot[kMaxFrames];
  const int s = s0 + blockIdx.y;
  rebuild_finish(v, s, v.state[s], eb, (int)blockIdx.x, 0u, sh_cnt, sh_slot, true);      // (seq 0: the second kNN pass precedes this launch in stream order)
}

// Chain mode, speculative hand-over of the finalising solve's result (kernels_sync.h): the launch behind the next scan's first kNN
// pass.  Verdict 1 (the rule): nothing to do.  Verdict 2: the first nA workgroups re-append the previous scan's frame at the confirmed
// pose (append_fix); the others wait for them and repeat the first pass from the confirmed prediction, commit its bookkeeping and
// count themselves done in their namesakes' place.  with_pass 0: repair only (the handle leaves chain mode or is synchronised: no
// pass has been started from the prediction); with_fix 0: the frame has been re-appended already, by such a launch.
template <int kKnnThreads>
__global__ __launch_bounds__(kKnnThreads, 1) void k_chain_redo0(DevView v, int s0, int eb, unsigned int wait_edges, unsigned int signal_odo, unsigned int seq, int scan_no, int nA, int with_pass, int with_fix) {
  constexpr int kKnnQueries = kKnnThreads / kKnnGroup;
  const int s = s0 + (int)blockIdx.y;
  StreamState& st = v.state[s];
  // (with_pass: the launch's last workgroup is the gate in front of the overlapped second pass — k_ov_gate: the pass's launch, next on
  //  this stream, must not start before the first solve's workgroups are on their CUs; one launch less per scan for the host)
  if (with_pass && (int)blockIdx.x == (int)gridDim.x - 1) {
    (void)pipe_wait(v.ov_flags + s, seq, &st.status);
    return;
  }
  const int bx = (int)blockIdx.x;
  if (pred_verdict_wait(v, s, bx % kOvReplicas, (unsigned int)scan_no, &st.status) != 2) return;
  // (a wait of the handle has given up — beside a saturating second process the appenders may have stopped waiting for the pose before
  //  it came: there may be nothing to take back; the scan has failed through the status bit and the host resets the handle)
  if (st.status & (LIODOM_STATUS_PIPE_TIMEOUT | LIODOM_STATUS_LM_SYNC_TIMEOUT)) return;
  if ((kInstrument && (v.debug & 64)) && threadIdx.x == 0) atomicAdd(&v.dbg_clk[274], 1ull);      // (debug) workgroups of the repair
  if (bx < nA) {
    __shared__ int sh_cnt[kMaxFrames + 1];
    __shared__ int sh_slot[kMaxFrames];
    if (with_fix) append_fix(v, s, st, bx, nA, (unsigned int)scan_no, scan_no - 1, sh_cnt, sh_slot);      // (else: repaired already, by a launch of its own when the host synchronised)
    return;
  }
  if (!with_pass) return;
  __shared__ KnnShared<kKnnQueries> shs[1];
  __shared__ double sh_ov[20];
  __shared__ int sh_go;
  if (threadIdx.x == 0) sh_go = (!with_fix || epoch_wait(v.redo_sync + 1, (unsigned int)scan_no, (unsigned int)nA, &st.status)) ? 1 : 0;
  __syncthreads();
  if (!sh_go) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  int bxi = bx - nA, byi = (int)blockIdx.y;
  const bool pass_ok = knn_pass<kKnnThreads, false, true>(v, s, bxi, byi, 0, eb, wait_edges, signal_odo, seq, scan_no, shs[0], shs[0], sh_ov, 1);
  if (bxi == 0) chain_release_edges(v, signal_odo);
  if (pass_ok) chain_count_done(v.knn_done0 + s);
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

// The cell hash of one stream in its GLOBAL table (v.table_size slots, global atomics): any map size fits.  k_hash_build's way out when
// the window occupies more cells than its LDS table holds (kLdsCellsMax), and — with hash_incr — what k_hash_append runs on the scans
// in between for such a stream (a table without room to append to: rebuilt every scan, as before round 6).  One workgroup; the
// slots of the previous global build have been emptied through used_cells by the finalising solve.  eb >= 0: the new frame has yet
// to be transformed and stored (window_point_produce).
__device__ __forceinline__ void hash_build_global(const DevView& v, int s, StreamState& st, const WinIndex& w, int nf, int Mw, int M, int eb, int tid) {
  CellSlot* cells = v.cells + (size_t)s * v.table_size;
  unsigned int* bits = v.cell_bits + (size_t)s * (v.table_size >> 5);
  int* pcell = v.pt_cell + (size_t)s * v.map_cap;
  int* prank = v.pt_rank + (size_t)s * v.map_cap;
  const float4* recv = v.recv_pts + (size_t)s * v.recv_cap;
  CellSlot empty; empty.key = kEmptyKey; empty.start = 0; empty.cnt = 0;
  if (st.table_mask != (unsigned int)v.table_size - 1u) {
    for (int i = tid; i < kLdsSlots; i += kBuildThreads) { cells[i] = empty; }
    for (int i = tid; i < kLdsSlots / 32; i += kBuildThreads) bits[i] = 0u;
  }
  __syncthreads();
  if (tid == 0) {
    st.table_mask = (unsigned int)v.table_size - 1u; st.n_used_tab[0] = 0; st.cursor = 0; st.n_search = M; st.n_filt = 0;
    st.hb_main_fc = 0; st.hb_shift = 0; st.hb_spill = 0;      // (hash_incr: nothing is ever appended to a global table)
  }
  __syncthreads();
  const unsigned int gmask = (unsigned int)v.table_size - 1u;
  for (int m = tid; m < M; m += kBuildThreads) {
    const float4 pt = m < Mw ? window_point_produce(v, s, st, eb, w, nf, m) : recv[m - Mw];
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
    v.sorted_pts[(size_t)s * v.sorted_cap + pos] = make_float4(pt.x, pt.y, pt.z, __int_as_float(m));
  }
}

// Room a cell gets beyond its population at a rebuild (hash_incr): as much again, 32 at least; a cell k_hash_append creates: 96.
// (Headline stream, tools/knn_budget_cpu.py: a frame puts 2.2 points into a cell it touches, 10 at the 99th percentile, 26 at most —
//  a pole coming into range —, and three frames arrive between rebuilds: with a quarter of the population / 8 / 16 a cell ran out of
//  room in every period, with this rule in none of 24; the point array then holds ~195 000 places per stream for 36 600 points.)
__host__ __device__ __forceinline__ int hash_cell_slack(unsigned int cnt, int slack_min) { return cnt ? ((int)cnt > slack_min ? (int)cnt : slack_min) : 0; }
#ifndef LIODOM_HB_PERIOD
#define LIODOM_HB_PERIOD 4
#endif
constexpr int kHbPeriod = LIODOM_HB_PERIOD;       // scans between two rebuilds from the whole window (<= 8: hb_base)
constexpr int kHbNewRoom = 96;     // room of a cell that k_hash_append creates (DevView::hb_new_room; kHbSlackMin = 32: hb_slack_min)
constexpr int kHbSlackMin = 32;

// =============================================================================================
// k_hash_append (lock-step batches, round 6): the reference appends one frame to the window and drops the oldest
// (laser_odometry.cc:34-60); k_hash_build re-binned all M ~ 36 000 window points for it, every scan, on one CU per stream (210 us
// of a 1.34 ms step at 256 streams: instruction issue of 2 x 36 points per lane).  Now the table lives for kHbPeriod scans — the
// host launches k_hash_build every kHbPeriod-th scan and this kernel in between:
//   * the NEW frame's ~1 750 points are transformed, stored in the window and appended to their cells — every cell got room for its
//     population again at the rebuild (hash_cell_slack); a cell that does not exist yet is created (CAS on the key, room from the
//     stream's cursor);
//   * a point that finds no room — its cell is full, the cursor or the table exhausted — goes to the stream's SPILL LIST at the end
//     of the point array, which every query of the stream scans like a cell until the next rebuild (empty for 32 of 34 appends on
//     the headline stream; exact in every case, only slower: a rebuild decided on the device would have to be a launch the host
//     makes every scan, and an idle k_hash_build launch still waits for a CU with 128 KB of free LDS);
//   * the EVICTED frames' points stay where they are: a stored window index minus hb_shift (the points evicted since the rebuild)
//     is the point's current window index, and a candidate below hb_shift is dead — one compare per candidate in k_knn8, which
//     subtracts the shift from the indices it hands on (FLANN's tie order, the correspondence indices).
// Candidates are a set to k_knn8 (ties go by window index): results are bit-identical to a rebuild every scan (LIODOM_HASH_INCR=0).
// One workgroup per stream: a barrier separates "every cell exists" from "points take their places".
// =============================================================================================
__global__ __launch_bounds__(kBuildThreads) void k_hash_append(DevView v, int s0, int eb) {
  __shared__ WinIndex w;
  __shared__ int sh_flag;
  const int s = s0 + blockIdx.x;
  StreamState& st = v.state[s];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nf = st.n_frames, fc = st.frame_count, Mw = st.n_map;
  const unsigned int tmask = st.table_mask;
  int d = (fc - nf) - st.hb_main_old;                  // frames evicted since the rebuild (< kHbPeriod <= frames of the window then: hb_base knows them)
  d = d < 0 ? 0 : (d > 7 ? 7 : d);
  if (nf < 1) return;
  win_index_load(v, s, nf, w, tid, kBuildThreads);
  if (tid == 0) sh_flag = 0;
  if (tmask != (unsigned int)kLdsSlots - 1u) {            // (uniform) the window outgrew the LDS table: a global table has no room to append to
    __syncthreads();
    hash_build_global(v, s, st, w, nf, Mw, Mw, eb, tid);
    return;
  }
  __syncthreads();
  const int shift = st.hb_base[d];
  const int first_new = w.sbase[nf - 1];
  const int E = Mw - first_new;
  CellSlot* cells = v.cells + (size_t)s * v.table_size;
  unsigned int* bits = v.cell_bits + (size_t)s * (v.table_size >> 5);
  unsigned int* ccap = v.cell_cap + (size_t)s * v.table_size;
  float4* sp = v.sorted_pts + (size_t)s * v.sorted_cap;
  // One sweep of kHbUnroll points per thread (the whole frame at once for up to 2 048 edges): the points stay in registers between the
  // two phases.  Everything shared between the threads — keys, starts, counts — is written and read by THIS workgroup only, with a
  // barrier between creation and use: plain accesses.
  constexpr int kHbUnroll = 2;
  for (int i0 = 0; i0 < E; i0 += kHbUnroll * kBuildThreads) {      // (uniform trip count)
    float4 pt[kHbUnroll];
    unsigned int slot[kHbUnroll];
    // ---- the new frame enters the window; every point's cell exists afterwards ----
#pragma unroll
    for (int u = 0; u < kHbUnroll; u++) {
      const int i = i0 + u * kBuildThreads + tid;
      slot[u] = 0xFFFFFFFFu;
      pt[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < E) pt[u] = window_point_produce(v, s, st, eb, w, nf, first_new + i);
    }
#pragma unroll
    for (int u = 0; u < kHbUnroll; u++) {
      const int i = i0 + u * kBuildThreads + tid;
      if (i >= E || !point_ok(pt[u])) continue;
      const unsigned long long key = pack_cell((int)floorf(pt[u].x * kCellInv), (int)floorf(pt[u].y * kCellInv), (int)floorf(pt[u].z * kCellInv));
      unsigned int h = hash_cell(key, tmask);
      bool placed = false;
      for (unsigned int probe = 0; probe <= tmask; probe++) {
        unsigned long long k = *(volatile unsigned long long*)&cells[h].key;
        if (k == kEmptyKey) {
          k = atomicCAS(&cells[h].key, kEmptyKey, key);
          if (k == kEmptyKey) {                           // this thread created the cell: room from the stream's cursor
            int start = atomicAdd(&st.hb_cursor, v.hb_new_room);
            const bool room = start >= 0 && start + v.hb_new_room <= v.hb_spill_base;
            start = room ? start : 0;                       // (no room left behind the cells: the cell exists, empty and full — its points spill)
            *(volatile unsigned int*)&cells[h].start = (unsigned int)start;
            *(volatile unsigned int*)&cells[h].cnt = 0u;
            *(volatile unsigned int*)&ccap[h] = (unsigned int)(room ? start + v.hb_new_room : start);
            atomicOr(&bits[h >> 5], 1u << (h & 31));
            k = key;
          }
        }
        if (k == key) { placed = true; break; }
        h = (h + 1) & tmask;
      }
      slot[u] = placed ? h : 0xFFFFFFFEu;                 // (0xFFFFFFFE: table full — the point spills)
    }
    __threadfence_block();
    __syncthreads();
    // ---- the points take their places: one atomic per RUN of equal cells in a wave (consecutive edges lie in the same cell) ----
#pragma unroll
    for (int u = 0; u < kHbUnroll; u++) {
      const int i = i0 + u * kBuildThreads + tid;
      const bool ok = slot[u] < 0xFFFFFFFEu;
      const KeyRun run = wave_key_runs(ok, (unsigned long long)slot[u], lane);
      unsigned int pos0 = 0u, cap0 = 0u;
      if (run.head) {
        // (a full cell's count is not raised any further: k_knn8 walks `cnt` places of the cell)
        cap0 = *(volatile unsigned int*)&ccap[slot[u]];
        const unsigned int st0 = *(volatile unsigned int*)&cells[slot[u]].start;
        const unsigned int old = atomicAdd(&cells[slot[u]].cnt, (unsigned int)run.len);
        pos0 = st0 + old;
        if (pos0 + (unsigned int)run.len > cap0) {          // part of the run (or all of it) does not fit: give the surplus back
          const unsigned int fit = pos0 < cap0 ? cap0 - pos0 : 0u;
          atomicSub(&cells[slot[u]].cnt, (unsigned int)run.len - fit);
        }
      }
      pos0 = (unsigned int)__shfl((int)pos0, run.head_lane);
      cap0 = (unsigned int)__shfl((int)cap0, run.head_lane);
      if (slot[u] != 0xFFFFFFFFu) {                       // a point of the frame
        const unsigned int pos = pos0 + (unsigned int)run.rank;
        const float4 rec = make_float4(pt[u].x, pt[u].y, pt[u].z, __int_as_float(first_new + i + shift));
        if (ok && pos < cap0) sp[pos] = rec;
        else { const int k = atomicAdd(&st.hb_spill, 1); if (k < v.sorted_cap - v.hb_spill_base) sp[v.hb_spill_base + k] = rec; sh_flag = 1; atomicAdd(&st.hb_stats[3], 1); }
      }
    }
    __syncthreads();
  }
  if (tid == 0) { st.hb_shift = shift; st.n_search = Mw; st.n_filt = 0; st.hb_stats[1] += 1; st.hb_stats[2] += sh_flag; }
}

__global__ __launch_bounds__(kBuildThreads) void k_hash_build(DevView v, int s0, int eb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ WinIndex w;
  __shared__ int sh_used, sh_over, sh_alloc, sh_wtot[kBuildThreads / 64];
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
    hash_build_global(v, s, st, w, nf, Mw, M, -1, tid);      // (the counting pass above has stored the new frame)
    return;
  }
  OV_STAMP(v, tid == 0 && s == 0, 21);
  // ---- exclusive prefix of the counts over the slots (8 consecutive slots per thread) ----
  // hash_incr: every cell gets ROOM for the points k_hash_append will add until the next rebuild (hash_cell_slack) — if the point
  // array holds that much
  const bool slack = v.hash_incr && !v.mapping && 2ll * M + (long long)v.hb_slack_min * sh_used <= (long long)v.sorted_cap;
  {
    constexpr int PER = kLdsSlots / kBuildThreads;   // 8
    unsigned int c[PER];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { c[k] = lcnt[tid * PER + k]; sum += (int)c[k] + (slack ? hash_cell_slack(c[k], v.hb_slack_min) : 0); }
    const int incl = wave_incl_scan_i32(sum);
    if ((tid & 63) == 63) sh_wtot[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < (tid >> 6); q++) base += sh_wtot[q];
    int run = base + incl - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) { lstart[tid * PER + k] = (unsigned int)run; run += (int)c[k] + (slack ? hash_cell_slack(c[k], v.hb_slack_min) : 0); }
    if (tid == kBuildThreads - 1) sh_alloc = run;      // everything allocated: k_hash_append's new cells go behind it
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
      if (ok && pos0 != 0xFFFFFFFFu) v.sorted_pts[(size_t)s * v.sorted_cap + pos0 + (unsigned int)run.rank] = make_float4(pt[k].x, pt[k].y, pt[k].z, __int_as_float(m));
    }
  }
  __syncthreads();
  OV_STAMP(v, tid == 0 && s == 0, 23);
  // ---- publish the table: slots [0, 8192) of the stream's global table + occupancy bits ----
  unsigned int* ccap = v.cell_cap ? v.cell_cap + (size_t)s * v.table_size : nullptr;
  for (int i = tid; i < kLdsSlots; i += kBuildThreads) {
    CellSlot o; o.key = lkey[i]; o.cnt = lcnt[i]; o.start = lstart[i] - lcnt[i];      // (the scatter pass advanced the cursors to the cells' ends)
    cells[i] = o;
    if (ccap) ccap[i] = lstart[i] + (slack ? (unsigned int)hash_cell_slack(o.cnt, v.hb_slack_min) : 0u);
  }
  for (int i = tid; i < kLdsSlots / 32; i += kBuildThreads) {
    unsigned int word = 0;
#pragma unroll
    for (int b = 0; b < 32; b++) word |= (lkey[i * 32 + b] != kEmptyKey) ? (1u << b) : 0u;
    bits[i] = word;
  }
  if (tid == 0) {
    st.table_mask = lmask; st.n_used_tab[0] = 0; st.n_search = M; st.n_filt = 0;
    st.cursor = 0; st.hb_cursor = sh_alloc; st.hb_stats[0] += 1;
    // what k_hash_append needs to keep this table current: when it was built, and the window's frame offsets then
    st.hb_main_fc = st.frame_count; st.hb_main_old = st.frame_count - nf; st.hb_main_nf = nf; st.hb_shift = 0; st.hb_spill = 0;
  }
  if (tid < 8) st.hb_base[tid] = tid <= nf ? w.sbase[tid] : w.sbase[nf];
  OV_STAMP(v, tid == 0 && s == 0, 24);
}
