// kernels_filter.h — filter_local_map: VoxelGrid(0.4) of the window (k_voxel_*, k_filt_*).
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, in this order; not a standalone header).
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
  v.sorted_pts[(size_t)s * v.sorted_cap + pos] = v.filt_pts[(size_t)s * v.map_cap + u];
}
