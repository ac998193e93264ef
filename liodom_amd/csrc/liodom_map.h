// Device-side liodom::Map (rows A12-A14 of SURVEY.md §8): the mapping node's coarse-cell map,
// src/map.cc:70-189 and include/liodom/map.h:58-116 of the reference, rebuilt for MI355X.
//
//   Map::updateMap   (map.cc:90-129)  transform the edge cloud by the pose (FP64 -> float), key each
//                    point to a coarse cell int(floor(x / size) * size + size / 2) (:103-105), create
//                    missing cells in first-appearance order (cells_vector_, :110-114), append, then
//                    re-filter every modified cell with PCL VoxelGrid(resolution) (:124-128).
//   Map::getLocalMap (map.cc:141-189) and Map::getMap (:131-139): concatenations of cell clouds.
//
// PCL's VoxelGrid output is one centroid per occupied leaf, ordered by ascending leaf index, each
// centroid a float sum taken in cloud order divided by the count.  A modified cell's cloud is
// [previous centroids (ascending leaf), new points (input order)].  Instead of sorting, every
// modified cell gets a DENSE leaf-occupancy bitmap for the update (a 40 x 40 x 50 m cell at 0.4 m
// is 105 x 105 x 130 leaves = 175 KiB of bits incl. margins; HBM is plentiful), so that
//   output position of a leaf = number of occupied leaves before it = word prefix + popcount,
// which is exactly PCL's order (lexicographic z, y, x of the leaf coordinates).  Leaves holding one
// point (the common case: an old centroid nobody touched) are copied straight to their new
// position; leaves with several points collect the members' cloud-order indices and a half-wave
// sums them in ascending order — the same float additions in the same order as PCL.  Nothing here
// assumes that an old centroid still falls into the leaf it came from.
//
// Cells are slabs of fixed capacity, double-buffered (an update writes the other buffer of every
// modified cell and flips it at commit).  All counts live on the device: an update can be enqueued
// behind the odometry kernels with the edge cloud, its size and the pose read from device memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "liodom_math.h"

namespace liodom_dev {

constexpr int kMapLeafMargin = 2;       // leaves of slack around a cell for float rounding at its faces
constexpr int kMapNewCellsMax = 256;    // cells created by one update (LDS list)
constexpr int kMapLocalEntriesMax = 4096;
constexpr unsigned long long kMapEmptyKey = 0xFFFFFFFFFFFFFFFFull;

enum MapStatusBits {
  MAP_STATUS_UPDATE_OVERFLOW = 1,   // more points in one update than max_update_points
  MAP_STATUS_CELLS_FULL = 2,        // more coarse cells than max_cells / more new cells than the LDS list
  MAP_STATUS_MODIFIED_FULL = 4,     // more cells touched by one update than max_modified_cells
  MAP_STATUS_CELL_OVERFLOW = 8,     // a cell cloud outgrew cell_capacity
  MAP_STATUS_LEAF_RANGE = 16,       // a point's leaf fell outside its cell's dense grid (sizes not multiples?)
  MAP_STATUS_KEY_RANGE = 32,        // cell key beyond +-2^20
  MAP_STATUS_LOCAL_OVERFLOW = 64,   // getLocalMap / getMap result larger than the output buffer
};

struct MapState {
  int n_cells;       // cells_vector_.size()
  int n_mod;         // cells modified by the update in flight
  int n_new;         // points of the update in flight
  int n_multi;       // leaves with more than one point (update in flight)
  int cursor;        // allocation cursor of the member lists
  int status;
  int n_entries;     // getLocalMap / getMap plan
  int n_result;      // points of the last getLocalMap / getMap
};

struct MapEntry { int cell, offset, count, pad; };

struct MapView {
  // Map::Map (map.cc:70-81)
  double xy, inv_xy, half_xy, z, inv_z, half_z;
  float leaf_inv;                // PCL inverse_leaf_size_ = 1.0f / (float)resolution
  int gx, gy, gz, words;         // dense leaf grid of one cell (with margins), 32-bit words of its bitmap
  int max_cells, cell_cap, upd_cap, mod_cap, ctable;
  MapState* st;
  unsigned long long* ckey;      // [ctable] packed cell key or kMapEmptyKey   (HashMap cells_, map.h:91)
  int* cslot_cell;               // [ctable] cell id, -1 while the slot is being created
  int* cfirst;                   // [ctable] first input index that touched a slot under creation
  int* cell_key;                 // [max_cells][3] voxel_x, voxel_y, voxel_z as the reference computes them
  int* cell_org;                 // [max_cells][3] leaf coordinates of the dense grid's origin
  int* cell_n;                   // [max_cells] points in the cell
  int* cell_buf;                 // [max_cells] current slab (0 / 1)
  float4* slab;                  // [2][max_cells][cell_cap]
  // scratch of the update in flight
  float4* new_pts;               // [upd_cap] transformed input points
  int* new_cell;                 // [upd_cap] hash slot, then cell id
  int* new_mi;                   // [upd_cap] index of the point's cell in mod_list
  int* new_pos;                  // [upd_cap] leaf bit, then output position
  int* new_rank;                 // [upd_cap]
  int* mod_list;                 // [mod_cap] cell ids
  int* mod_of_cell;              // [max_cells] index in mod_list or -1
  int* mod_out_n;                // [mod_cap] points after filtering
  unsigned int* bitmap;          // [mod_cap][words]
  unsigned int* wprefix;         // [mod_cap][words]
  int* old_pos;                  // [mod_cap][cell_cap] leaf bit, then output position of the old points
  int* old_rank;                 // [mod_cap][cell_cap]
  unsigned int* leaf_cnt;        // [mod_cap][cell_cap] points per output leaf
  unsigned int* leaf_start;      // [mod_cap][cell_cap] member list start (leaves with > 1 point)
  int* members;                  // [mod_cap * cell_cap] cloud-order indices grouped by leaf
  int2* multi;                   // [mod_cap * cell_cap] (mod index, output position) of the multi-point leaves
  MapEntry* entries;             // [kMapLocalEntriesMax] plan of a concatenation
};

__device__ __forceinline__ unsigned int map_hash(unsigned long long k, unsigned int mask) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (unsigned int)k & mask;
}
// HashKey (map.h:58-72): three ints.  Packed with 21 bits each (+-2^20 m).
__device__ __forceinline__ bool map_pack_key(int kx, int ky, int kz, unsigned long long* out) {
  const int B = 1 << 20;
  if (kx < -B || kx >= B || ky < -B || ky >= B || kz < -B || kz >= B) return false;
  *out = ((unsigned long long)(unsigned int)(kx + B) << 42) | ((unsigned long long)(unsigned int)(ky + B) << 21) |
         (unsigned long long)(unsigned int)(kz + B);
  return true;
}
// map.cc:103-105 (also :145,148,151 with an int argument)
__device__ __forceinline__ int map_cell_key(double x, double inv, double size, double half) {
  return (int)(floor(x * inv) * size + half);
}
// read-only lookup (no update in flight)
__device__ __forceinline__ int map_find_cell(const MapView& m, int kx, int ky, int kz) {
  unsigned long long key;
  if (!map_pack_key(kx, ky, kz, &key)) return -1;
  const unsigned int mask = (unsigned int)m.ctable - 1u;
  unsigned int h = map_hash(key, mask);
  for (int probe = 0; probe < m.ctable; probe++) {
    const unsigned long long k = m.ckey[h];
    if (k == key) return m.cslot_cell[h];
    if (k == kMapEmptyKey) return -1;
    h = (h + 1) & mask;
  }
  return -1;
}
__device__ __forceinline__ const float4* map_cell_cur(const MapView& m, int cell) {
  return m.slab + ((size_t)m.cell_buf[cell] * m.max_cells + cell) * m.cell_cap;
}
__device__ __forceinline__ float4* map_cell_next(const MapView& m, int cell) {
  return m.slab + ((size_t)(m.cell_buf[cell] ^ 1) * m.max_cells + cell) * m.cell_cap;
}
// leaf bit of a point inside its cell's dense grid, -1 if outside
__device__ __forceinline__ int map_leaf_bit(const MapView& m, int cell, const float4& p) {
  const int rx = (int)floorf(p.x * m.leaf_inv) - m.cell_org[cell * 3 + 0];
  const int ry = (int)floorf(p.y * m.leaf_inv) - m.cell_org[cell * 3 + 1];
  const int rz = (int)floorf(p.z * m.leaf_inv) - m.cell_org[cell * 3 + 2];
  if (rx < 0 || rx >= m.gx || ry < 0 || ry >= m.gy || rz < 0 || rz >= m.gz) return -1;
  return rx + ry * m.gx + rz * m.gx * m.gy;
}

// ---------------------------------------------------------------------------------------------
// updateMap, step 1 (one workgroup): transform, cell keys, creation of missing cells in
// first-appearance order, list of modified cells.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_map_assign(MapView m, const float4* in, const int* n_ptr, const double* T_ptr) {
  __shared__ double T[12];
  __shared__ int sh_newslot[kMapNewCellsMax];
  __shared__ int sh_nnew, sh_nmod, sh_n;
  const int tid = threadIdx.x;
  MapState& st = *m.st;
  if (tid < 12) T[tid] = T_ptr[tid];
  if (tid == 0) {
    int n = *n_ptr;
    if (n > m.upd_cap) { n = m.upd_cap; atomicOr(&st.status, MAP_STATUS_UPDATE_OVERFLOW); }
    if (n < 0) n = 0;
    sh_n = n; sh_nnew = 0; sh_nmod = 0;
  }
  __syncthreads();
  const int n = sh_n;
  const unsigned int mask = (unsigned int)m.ctable - 1u;
  // phase A: claim / find the hash slot of every point's cell
  for (int i = tid; i < n; i += 1024) {
    const float4 e = in[i];
    float4 p;
    transform_point(T, e.x, e.y, e.z, &p.x, &p.y, &p.z);        // map.cc:93-94
    p.w = e.w;
    m.new_pts[i] = p;
    const int kx = map_cell_key((double)p.x, m.inv_xy, m.xy, m.half_xy);      // :103
    const int ky = map_cell_key((double)p.y, m.inv_xy, m.xy, m.half_xy);      // :104
    const int kz = map_cell_key((double)p.z, m.inv_z, m.z, m.half_z);         // :105
    unsigned long long key;
    int slot = -1;
    if (map_pack_key(kx, ky, kz, &key)) {
      unsigned int h = map_hash(key, mask);
      for (int probe = 0; probe < m.ctable; probe++) {
        const unsigned long long prev = atomicCAS(&m.ckey[h], kMapEmptyKey, key);
        if (prev == kMapEmptyKey) {
          const int u = atomicAdd(&sh_nnew, 1);
          if (u < kMapNewCellsMax) sh_newslot[u] = (int)h;
          slot = (int)h;
          break;
        }
        if (prev == key) { slot = (int)h; break; }
        h = (h + 1) & mask;
      }
      if (slot >= 0 && *(volatile int*)&m.cslot_cell[slot] < 0) atomicMin(&m.cfirst[slot], i);
    } else {
      atomicOr(&st.status, MAP_STATUS_KEY_RANGE);
    }
    m.new_cell[i] = slot;
  }
  __threadfence();
  __syncthreads();
  // phase B: new cells get their ids in the order of the first point that touched them (:110-114)
  int nn = sh_nnew;
  if (nn > kMapNewCellsMax) { nn = kMapNewCellsMax; if (tid == 0) atomicOr(&st.status, MAP_STATUS_CELLS_FULL); }
  const int base_cells = st.n_cells;
  if (tid < nn) {
    const int hs = sh_newslot[tid];
    const int f = *(volatile int*)&m.cfirst[hs];
    int rank = 0;
    for (int u = 0; u < nn; u++) rank += (*(volatile int*)&m.cfirst[sh_newslot[u]] < f) ? 1 : 0;
    const int id = base_cells + rank;
    if (id < m.max_cells) {
      const float4 p = m.new_pts[f];
      m.cell_key[id * 3 + 0] = map_cell_key((double)p.x, m.inv_xy, m.xy, m.half_xy);
      m.cell_key[id * 3 + 1] = map_cell_key((double)p.y, m.inv_xy, m.xy, m.half_xy);
      m.cell_key[id * 3 + 2] = map_cell_key((double)p.z, m.inv_z, m.z, m.half_z);
      // dense leaf grid: origin = leaf of the cell's lower corner minus the margin
      m.cell_org[id * 3 + 0] = (int)floorf((float)(floor((double)p.x * m.inv_xy) * m.xy) * m.leaf_inv) - kMapLeafMargin;
      m.cell_org[id * 3 + 1] = (int)floorf((float)(floor((double)p.y * m.inv_xy) * m.xy) * m.leaf_inv) - kMapLeafMargin;
      m.cell_org[id * 3 + 2] = (int)floorf((float)(floor((double)p.z * m.inv_z) * m.z) * m.leaf_inv) - kMapLeafMargin;
      m.cell_n[id] = 0;
      m.cell_buf[id] = 0;
      m.cslot_cell[hs] = id;
    } else {
      m.cslot_cell[hs] = m.max_cells;       // marker: no room (points of this cell are dropped)
      atomicOr(&st.status, MAP_STATUS_CELLS_FULL);
    }
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) st.n_cells = min(base_cells + nn, m.max_cells);
  // phase C1: resolve ids, claim a place in the list of modified cells
  for (int i = tid; i < n; i += 1024) {
    const int slot = m.new_cell[i];
    int id = slot >= 0 ? *(volatile int*)&m.cslot_cell[slot] : -1;
    if (id >= m.max_cells) id = -1;
    m.new_cell[i] = id;
    if (id >= 0 && atomicCAS(&m.mod_of_cell[id], -1, -2) == -1) {
      const int mi = atomicAdd(&sh_nmod, 1);
      if (mi < m.mod_cap) { m.mod_list[mi] = id; atomicExch(&m.mod_of_cell[id], mi); }
      else { atomicExch(&m.mod_of_cell[id], -3); atomicOr(&st.status, MAP_STATUS_MODIFIED_FULL); }
    }
  }
  __threadfence();
  __syncthreads();
  // phase C2
  for (int i = tid; i < n; i += 1024) {
    const int id = m.new_cell[i];
    int mi = id >= 0 ? *(volatile int*)&m.mod_of_cell[id] : -1;
    if (mi < 0) mi = -1;
    m.new_mi[i] = mi;
  }
  if (tid == 0) { st.n_new = n; st.n_mod = min(sh_nmod, m.mod_cap); st.n_multi = 0; st.cursor = 0; }
}

// step 2: clear the bitmap and the leaf counters of every modified cell.  grid (x, mod_cap)
__global__ __launch_bounds__(256) void k_map_clear(MapView m) {
  const MapState& st = *m.st;
  const int mi = blockIdx.y;
  if (mi >= st.n_mod) return;
  const int cell = m.mod_list[mi];
  const int lim = min(m.cell_cap, m.cell_n[cell] + st.n_new);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < m.words || i < lim; i += gridDim.x * 256) {
    if (i < m.words) m.bitmap[(size_t)mi * m.words + i] = 0u;
    if (i < lim) m.leaf_cnt[(size_t)mi * m.cell_cap + i] = 0u;
  }
}

// Items of an update: row blockIdx.y < mod_cap = old points of modified cell y; row mod_cap = new points.
struct MapItem { int mi, cell, order; float4 p; bool valid, is_new; int idx; };
__device__ __forceinline__ MapItem map_item(const MapView& m) {
  MapItem it;
  it.valid = false;
  const MapState& st = *m.st;
  const int i = blockIdx.x * 256 + threadIdx.x;
  it.idx = i;
  if ((int)blockIdx.y == m.mod_cap) {
    it.is_new = true;
    if (i >= st.n_new) return it;
    it.mi = m.new_mi[i];
    if (it.mi < 0) return it;
    it.cell = m.new_cell[i];
    it.order = m.cell_n[it.cell] + i;          // after every old point, ascending in input order
    it.p = m.new_pts[i];
  } else {
    it.is_new = false;
    it.mi = blockIdx.y;
    if (it.mi >= st.n_mod) return it;
    it.cell = m.mod_list[it.mi];
    if (i >= m.cell_n[it.cell]) return it;
    it.order = i;
    it.p = map_cell_cur(m, it.cell)[i];
  }
  it.valid = true;
  return it;
}
__device__ __forceinline__ int* map_item_pos(const MapView& m, const MapItem& it) {
  return it.is_new ? &m.new_pos[it.idx] : &m.old_pos[(size_t)it.mi * m.cell_cap + it.idx];
}
__device__ __forceinline__ int* map_item_rank(const MapView& m, const MapItem& it) {
  return it.is_new ? &m.new_rank[it.idx] : &m.old_rank[(size_t)it.mi * m.cell_cap + it.idx];
}

// step 3: leaf occupancy.  grid (x, mod_cap + 1)
__global__ __launch_bounds__(256) void k_map_setbits(MapView m) {
  const MapItem it = map_item(m);
  if (!it.valid) return;
  // PCL skips non-finite points (is_dense == false path); transformed edges are always finite
  int bit = -1;
  if (ld_isfinite((double)it.p.x) && ld_isfinite((double)it.p.y) && ld_isfinite((double)it.p.z)) {
    bit = map_leaf_bit(m, it.cell, it.p);
    if (bit < 0) atomicOr(&m.st->status, MAP_STATUS_LEAF_RANGE);
  }
  *map_item_pos(m, it) = bit;
  if (bit >= 0) atomicOr(&m.bitmap[(size_t)it.mi * m.words + (bit >> 5)], 1u << (bit & 31));
}

// step 4: exclusive prefix of the popcounts of a modified cell's bitmap words.  grid (mod_cap)
__global__ __launch_bounds__(1024) void k_map_prefix(MapView m) {
  __shared__ int sh_w[16];
  __shared__ int sh_carry;
  MapState& st = *m.st;
  const int mi = blockIdx.x;
  if (mi >= st.n_mod) return;
  const int tid = threadIdx.x;
  const unsigned int* bm = m.bitmap + (size_t)mi * m.words;
  unsigned int* wp = m.wprefix + (size_t)mi * m.words;
  if (tid == 0) sh_carry = 0;
  __syncthreads();
  for (int base = 0; base < m.words; base += 1024 * 4) {
    const int i0 = base + tid * 4;
    int c[4];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { c[k] = (i0 + k < m.words) ? __popc(bm[i0 + k]) : 0; sum += c[k]; }
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if ((tid & 63) >= off) incl += t; }
    if ((tid & 63) == 63) sh_w[tid >> 6] = incl;
    __syncthreads();
    int run = sh_carry + incl - sum;
    for (int q = 0; q < (tid >> 6); q++) run += sh_w[q];
#pragma unroll
    for (int k = 0; k < 4; k++) { if (i0 + k < m.words) wp[i0 + k] = (unsigned int)run; run += c[k]; }
    __syncthreads();
    if (tid == 1023) sh_carry = run;
    __syncthreads();
  }
  if (tid == 0) {
    int total = sh_carry;
    if (total > m.cell_cap) { total = m.cell_cap; atomicOr(&st.status, MAP_STATUS_CELL_OVERFLOW); }
    m.mod_out_n[mi] = total;
  }
}

// step 5: output position of every point and points per output leaf.  grid (x, mod_cap + 1)
__global__ __launch_bounds__(256) void k_map_count(MapView m) {
  const MapItem it = map_item(m);
  if (!it.valid) return;
  int* pp = map_item_pos(m, it);
  const int bit = *pp;
  if (bit < 0) return;
  const size_t w = (size_t)it.mi * m.words + (bit >> 5);
  const int pos = (int)m.wprefix[w] + __popc(m.bitmap[w] & ((1u << (bit & 31)) - 1u));
  if (pos >= m.cell_cap) { *pp = -1; return; }
  *pp = pos;
  *map_item_rank(m, it) = (int)atomicAdd(&m.leaf_cnt[(size_t)it.mi * m.cell_cap + pos], 1u);
}

// step 6: member-list allocation for the leaves with more than one point.  grid (x, mod_cap)
__global__ __launch_bounds__(256) void k_map_alloc(MapView m) {
  MapState& st = *m.st;
  const int mi = blockIdx.y;
  if (mi >= st.n_mod) return;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= m.mod_out_n[mi]) return;
  const unsigned int c = m.leaf_cnt[(size_t)mi * m.cell_cap + pos];
  if (c > 1u) {
    m.leaf_start[(size_t)mi * m.cell_cap + pos] = (unsigned int)atomicAdd(&st.cursor, (int)c);
    m.multi[atomicAdd(&st.n_multi, 1)] = make_int2(mi, pos);
  }
}

// step 7: single-point leaves are written to their new position; the others list their members.
__global__ __launch_bounds__(256) void k_map_emit(MapView m) {
  const MapItem it = map_item(m);
  if (!it.valid) return;
  const int pos = *map_item_pos(m, it);
  if (pos < 0) return;
  const size_t li = (size_t)it.mi * m.cell_cap + pos;
  const unsigned int c = m.leaf_cnt[li];
  if (c == 1u) {
    // PCL: centroid = 0 + p, then / 1
    float4 o;
    o.x = (0.0f + it.p.x) / 1.0f; o.y = (0.0f + it.p.y) / 1.0f; o.z = (0.0f + it.p.z) / 1.0f; o.w = (0.0f + it.p.w) / 1.0f;
    map_cell_next(m, it.cell)[pos] = o;
  } else {
    m.members[m.leaf_start[li] + (unsigned int)*map_item_rank(m, it)] = it.order;
  }
}

// step 8: centroid of a multi-point leaf, members summed in cloud order.  32 lanes per leaf.
__global__ __launch_bounds__(256) void k_map_centroid(MapView m) {
  const MapState& st = *m.st;
  const int q = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int hl = threadIdx.x & 31;
  for (int leaf = q; leaf < st.n_multi; leaf += gridDim.x * 8) {
    const int2 mp = m.multi[leaf];
    const int cell = m.mod_list[mp.x];
    const size_t li = (size_t)mp.x * m.cell_cap + mp.y;
    const int cnt = (int)m.leaf_cnt[li];
    const int* list = m.members + m.leaf_start[li];
    const int n_old = m.cell_n[cell];
    const float4* cur = map_cell_cur(m, cell);
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    int last = -1;
    for (int k = 0; k < cnt; k++) {
      int best = 0x7fffffff;                      // smallest cloud-order index above `last`
      for (int j = hl; j < cnt; j += 32) { const int o = list[j]; if (o > last && o < best) best = o; }
      for (int off = 16; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off));
      const float4 p = best < n_old ? cur[best] : m.new_pts[best - n_old];
      sx += p.x; sy += p.y; sz += p.z; si += p.w;
      last = best;
    }
    if (hl == 0) {
      const float c = (float)cnt;
      map_cell_next(m, cell)[mp.y] = make_float4(sx / c, sy / c, sz / c, si / c);
    }
  }
}

// step 9: flip the slabs of the modified cells.
__global__ __launch_bounds__(256) void k_map_commit(MapView m) {
  MapState& st = *m.st;
  const int nm = st.n_mod;
  for (int mi = threadIdx.x; mi < nm; mi += 256) {
    const int cell = m.mod_list[mi];
    m.cell_n[cell] = m.mod_out_n[mi];
    m.cell_buf[cell] ^= 1;
    m.mod_of_cell[cell] = -1;
  }
  __syncthreads();
  if (threadIdx.x == 0) { st.n_mod = 0; st.n_new = 0; }
}

// ---------------------------------------------------------------------------------------------
// getLocalMap (map.cc:141-189): the cells of a (2*cells_xy+1)^2 square at the pose's z layer, x outer
// and y inner, then a z column through the centre cell.  Reproduced as written: the translation is
// truncated to int first (:144,147,150); the z column's extent uses voxel_xysize_ but steps by
// voxel_zsize_ (:175-178); the centre cell is visited by both loops; int loop variables advance by
// `i += double`.  One thread plans (<= a few dozen lookups), k_map_gather copies.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void map_plan_add(const MapView& m, int kx, int ky, int kz, int* ne, int* total) {
  const int cell = map_find_cell(m, kx, ky, kz);
  if (cell < 0 || cell >= m.max_cells) return;
  if (*ne >= kMapLocalEntriesMax) { atomicOr(&m.st->status, MAP_STATUS_LOCAL_OVERFLOW); return; }
  MapEntry e; e.cell = cell; e.offset = *total; e.count = m.cell_n[cell]; e.pad = 0;
  m.entries[*ne] = e;
  *ne += 1;
  *total += e.count;
}
__global__ void k_map_local_plan(MapView m, const double* T_ptr, int cells_xy, int cells_z, int out_cap, int* n_out, int sticky) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  MapState& st = *m.st;
  const int x = (int)T_ptr[3];                                                       // :144
  const int voxel_x = map_cell_key((double)x, m.inv_xy, m.xy, m.half_xy);
  const int y = (int)T_ptr[7];                                                       // :147
  const int voxel_y = map_cell_key((double)y, m.inv_xy, m.xy, m.half_xy);
  const int z = (int)T_ptr[11];                                                      // :150
  const int voxel_z = map_cell_key((double)z, m.inv_z, m.z, m.half_z);
  const int init_x = (int)(voxel_x - cells_xy * m.xy);                               // :157-160
  const int end_x = (int)(voxel_x + cells_xy * m.xy);
  const int init_y = (int)(voxel_y - cells_xy * m.xy);
  const int end_y = (int)(voxel_y + cells_xy * m.xy);
  int ne = 0, total = 0, guard = 0;
  for (int i = init_x; i <= end_x && guard < 65536; i = (int)(i + m.xy), guard++) {          // :162
    for (int j = init_y; j <= end_y && guard < 65536; j = (int)(j + m.xy), guard++) {        // :163
      map_plan_add(m, i, j, voxel_z, &ne, &total);
    }
  }
  const int init_z = (int)(voxel_z - cells_z * m.xy);                                // :175 (xy size)
  const int end_z = (int)(voxel_z + cells_z * m.xy);                                 // :176
  for (int i = init_z; i <= end_z && guard < 65536; i = (int)(i + m.z), guard++) {   // :178
    map_plan_add(m, voxel_x, voxel_y, i, &ne, &total);
  }
  if (total > out_cap && sticky) atomicOr(&st.status, MAP_STATUS_LOCAL_OVERFLOW);   // host callers get LIODOM_ERR_CAPACITY instead
  st.n_entries = ne;
  st.n_result = total;
  if (n_out) *n_out = total > out_cap ? out_cap : total;
}

// getMap (map.cc:131-139): every cell in cells_vector_ order.
__global__ __launch_bounds__(1024) void k_map_all_plan(MapView m, int out_cap, int* n_out) {
  __shared__ int sh_w[16];
  __shared__ int sh_carry;
  MapState& st = *m.st;
  const int tid = threadIdx.x;
  const int nc = min(st.n_cells, kMapLocalEntriesMax);
  if (tid == 0) { sh_carry = 0; if (st.n_cells > kMapLocalEntriesMax) atomicOr(&st.status, MAP_STATUS_LOCAL_OVERFLOW); }
  __syncthreads();
  for (int base = 0; base < nc; base += 1024) {
    const int c = base + tid;
    const int cnt = c < nc ? m.cell_n[c] : 0;
    int incl = cnt;
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if ((tid & 63) >= off) incl += t; }
    if ((tid & 63) == 63) sh_w[tid >> 6] = incl;
    __syncthreads();
    int run = sh_carry + incl - cnt;
    for (int q = 0; q < (tid >> 6); q++) run += sh_w[q];
    if (c < nc) { MapEntry e; e.cell = c; e.offset = run; e.count = cnt; e.pad = 0; m.entries[c] = e; }
    __syncthreads();
    if (tid == 1023) sh_carry = run + cnt;
    __syncthreads();
  }
  if (tid == 0) {
    const int total = sh_carry;
    st.n_entries = nc;
    st.n_result = total;
    if (n_out) *n_out = total > out_cap ? out_cap : total;
  }
}

// grid (x, entries): copies the planned cells to the output
__global__ __launch_bounds__(256) void k_map_gather(MapView m, float4* out, int out_cap) {
  const MapState& st = *m.st;
  for (int e = blockIdx.y; e < st.n_entries; e += gridDim.y) {
    const MapEntry en = m.entries[e];
    const float4* src = map_cell_cur(m, en.cell);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < en.count; i += gridDim.x * 256) {
      const int o = en.offset + i;
      if (o < out_cap) out[o] = src[i];
    }
  }
}

// fills the bookkeeping arrays at creation
__global__ __launch_bounds__(256) void k_map_init(MapView m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m.ctable) { m.ckey[i] = kMapEmptyKey; m.cslot_cell[i] = -1; m.cfirst[i] = 0x7fffffff; }
  if (i < m.max_cells) { m.mod_of_cell[i] = -1; m.cell_n[i] = 0; m.cell_buf[i] = 0; }
  if (i == 0) { MapState z = {}; *m.st = z; }
}

}  // namespace liodom_dev
