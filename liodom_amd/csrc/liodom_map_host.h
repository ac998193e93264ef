// Host side of the device map (C-ABI liodom_map_* of include/liodom_hip.h).  Included by
// liodom_hip.hip (same translation unit: shares HIP_TRY / g_last_error).
#pragma once
#include "liodom_map.h"

struct liodom_map {
  liodom_map_config_t cfg;
  liodom_dev::MapView m{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int device = 0;
  std::vector<void*> allocs;
  // staging of the host entry points
  float4* d_in = nullptr;
  int* d_n = nullptr;
  double* d_T = nullptr;
  float4* d_out = nullptr;
  int out_cap = 0;
  int* d_out_n = nullptr;
};

namespace {

using liodom_dev::MapView;

int map_alloc(liodom_map* mp, void** out, size_t bytes) {
  void* raw = nullptr;
  HIP_TRY(hipMalloc(&raw, bytes ? bytes : 16));
  mp->allocs.push_back(raw);
  *out = raw;
  return LIODOM_OK;
}
#define MAP_ALLOC(ptr, count)                                                                  \
  do {                                                                                         \
    void* _raw = nullptr;                                                                      \
    int _rc = map_alloc(mp, &_raw, sizeof(*(ptr)) * (size_t)(count));                          \
    if (_rc) return _rc;                                                                       \
    (ptr) = reinterpret_cast<decltype(ptr)>(_raw);                                             \
  } while (0)

int map_build(liodom_map* mp, const liodom_map_config_t* c, hipStream_t stream) {
  mp->cfg = *c;
  MapView& m = mp->m;
  m.xy = c->voxel_xysize; m.inv_xy = 1.0 / c->voxel_xysize; m.half_xy = c->voxel_xysize / 2.0;     // map.cc:71-73
  m.z = c->voxel_zsize;   m.inv_z = 1.0 / c->voxel_zsize;   m.half_z = c->voxel_zsize / 2.0;       // :74-76
  const float leaf = (float)c->resolution;               // setLeafSize(float, float, float), :80
  m.leaf_inv = 1.0f / leaf;                              // PCL inverse_leaf_size_
  m.gx = m.gy = (int)std::ceil((float)c->voxel_xysize * m.leaf_inv) + 1 + 2 * liodom_dev::kMapLeafMargin;
  m.gz = (int)std::ceil((float)c->voxel_zsize * m.leaf_inv) + 1 + 2 * liodom_dev::kMapLeafMargin;
  const long long leaves = (long long)m.gx * m.gy * m.gz;
  if (leaves > (1ll << 28)) { g_last_error = "liodom_map_create: voxel size / resolution gives more than 2^28 leaves per cell"; return LIODOM_ERR_INVALID_ARG; }
  m.words = (int)((leaves + 31) / 32);
  m.max_cells = c->max_cells; m.cell_cap = c->cell_capacity; m.upd_cap = c->max_update_points; m.mod_cap = c->max_modified_cells;
  int ct = 64;
  while (ct < 4 * m.max_cells) ct <<= 1;
  m.ctable = ct;
  if (stream) { mp->stream = stream; mp->own_stream = false; }
  else { HIP_TRY(hipStreamCreateWithFlags(&mp->stream, hipStreamNonBlocking)); mp->own_stream = true; }
  MAP_ALLOC(m.st, 1);
  MAP_ALLOC(m.ckey, m.ctable); MAP_ALLOC(m.cslot_cell, m.ctable); MAP_ALLOC(m.cfirst, m.ctable);
  MAP_ALLOC(m.cell_key, 3 * (size_t)m.max_cells); MAP_ALLOC(m.cell_org, 3 * (size_t)m.max_cells);
  MAP_ALLOC(m.cell_n, m.max_cells); MAP_ALLOC(m.cell_buf, m.max_cells);
  MAP_ALLOC(m.slab, 2 * (size_t)m.max_cells * m.cell_cap);
  MAP_ALLOC(m.new_pts, m.upd_cap); MAP_ALLOC(m.new_cell, m.upd_cap); MAP_ALLOC(m.new_mi, m.upd_cap);
  MAP_ALLOC(m.new_pos, m.upd_cap); MAP_ALLOC(m.new_rank, m.upd_cap);
  MAP_ALLOC(m.mod_list, m.mod_cap); MAP_ALLOC(m.mod_of_cell, m.max_cells); MAP_ALLOC(m.mod_out_n, m.mod_cap);
  MAP_ALLOC(m.bitmap, (size_t)m.mod_cap * m.words); MAP_ALLOC(m.wprefix, (size_t)m.mod_cap * m.words);
  MAP_ALLOC(m.old_pos, (size_t)m.mod_cap * m.cell_cap); MAP_ALLOC(m.old_rank, (size_t)m.mod_cap * m.cell_cap);
  MAP_ALLOC(m.leaf_cnt, (size_t)m.mod_cap * m.cell_cap); MAP_ALLOC(m.leaf_start, (size_t)m.mod_cap * m.cell_cap);
  MAP_ALLOC(m.members, (size_t)m.mod_cap * m.cell_cap);
  MAP_ALLOC(m.multi, (size_t)m.mod_cap * m.cell_cap / 2 + 1);
  MAP_ALLOC(m.entries, liodom_dev::kMapLocalEntriesMax);
  const int n_init = std::max(m.ctable, m.max_cells);
  hipLaunchKernelGGL(liodom_dev::k_map_init, dim3((n_init + 255) / 256), dim3(256), 0, mp->stream, m);
  HIP_TRY(hipGetLastError());
  return LIODOM_OK;
}

// Map::updateMap enqueued on `q`: points, their count and the pose are read from device memory.
int map_enqueue_update(liodom_map* mp, const float4* d_pts, const int* d_n, const double* d_T, hipStream_t q) {
  using namespace liodom_dev;
  const MapView& m = mp->m;
  const int xb_old = (m.cell_cap + 255) / 256, xb_new = (m.upd_cap + 255) / 256;
  const int xb = std::max(xb_old, xb_new);
  hipLaunchKernelGGL(k_map_assign, dim3(1), dim3(1024), 0, q, m, d_pts, d_n, d_T);
  hipLaunchKernelGGL(k_map_clear, dim3(std::min(64, std::max((m.words + 255) / 256, xb_old)), m.mod_cap), dim3(256), 0, q, m);
  hipLaunchKernelGGL(k_map_setbits, dim3(xb, m.mod_cap + 1), dim3(256), 0, q, m);
  hipLaunchKernelGGL(k_map_prefix, dim3(m.mod_cap), dim3(1024), 0, q, m);
  hipLaunchKernelGGL(k_map_count, dim3(xb, m.mod_cap + 1), dim3(256), 0, q, m);
  hipLaunchKernelGGL(k_map_alloc, dim3(xb_old, m.mod_cap), dim3(256), 0, q, m);
  hipLaunchKernelGGL(k_map_emit, dim3(xb, m.mod_cap + 1), dim3(256), 0, q, m);
  hipLaunchKernelGGL(k_map_centroid, dim3(512), dim3(256), 0, q, m);
  hipLaunchKernelGGL(k_map_commit, dim3(1), dim3(256), 0, q, m);
  HIP_TRY(hipGetLastError());
  return LIODOM_OK;
}

// Map::getLocalMap enqueued on `q`: result and its size stay on the device.
int map_enqueue_local(liodom_map* mp, const double* d_T, int cells_xy, int cells_z, float4* d_out, int out_cap,
                      int* d_out_n, hipStream_t q, int sticky_overflow) {
  using namespace liodom_dev;
  hipLaunchKernelGGL(k_map_local_plan, dim3(1), dim3(64), 0, q, mp->m, d_T, cells_xy, cells_z, out_cap, d_out_n, sticky_overflow);
  hipLaunchKernelGGL(k_map_gather, dim3((mp->m.cell_cap + 255) / 256 > 64 ? 64 : (mp->m.cell_cap + 255) / 256, 64), dim3(256), 0, q, mp->m, d_out, out_cap);
  HIP_TRY(hipGetLastError());
  return LIODOM_OK;
}

int map_ensure_out(liodom_map* mp, int64_t cap) {
  if (cap > 0x7fffffff) cap = 0x7fffffff;
  if (mp->d_out && mp->out_cap >= cap) return LIODOM_OK;
  if (mp->d_out) { (void)hipFree(mp->d_out); mp->d_out = nullptr; }
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&mp->d_out), sizeof(float4) * (size_t)std::max<int64_t>(cap, 1)));
  mp->out_cap = (int)cap;
  return LIODOM_OK;
}

void map_free(liodom_map* mp) {
  if (!mp) return;
  (void)hipSetDevice(mp->device);
  if (mp->stream) (void)hipStreamSynchronize(mp->stream);
  for (void* p : mp->allocs) (void)hipFree(p);
  if (mp->d_out) (void)hipFree(mp->d_out);
  if (mp->own_stream && mp->stream) (void)hipStreamDestroy(mp->stream);
  delete mp;
}

int map_fetch_result(liodom_map* mp, float* xyzi, int64_t cap, int64_t* n_points) {
  int n = 0;
  HIP_TRY(hipMemcpyAsync(&n, mp->d_out_n, sizeof(int), hipMemcpyDeviceToHost, mp->stream));
  HIP_TRY(hipStreamSynchronize(mp->stream));
  liodom_dev::MapState st;
  HIP_TRY(hipMemcpy(&st, mp->m.st, sizeof(st), hipMemcpyDeviceToHost));
  if (n_points) *n_points = st.n_result;
  if ((int64_t)st.n_result > cap) { g_last_error = "liodom_map: output buffer too small"; return LIODOM_ERR_CAPACITY; }
  if (n > 0 && xyzi) HIP_TRY(hipMemcpy(xyzi, mp->d_out, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost));
  return LIODOM_OK;
}

}  // namespace

extern "C" {

void liodom_map_config_default(liodom_map_config_t* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->device = 0;
  c->max_cells = 1024;
  c->voxel_xysize = 40.0; c->voxel_zsize = 50.0; c->resolution = 0.4;     // liodom_mapping_node.cc:115-125
  c->cell_capacity = 65536;
  c->max_update_points = 16384;
  c->max_modified_cells = 128;
}

int liodom_map_create(const liodom_map_config_t* config, liodom_map_t** out) {
  if (!config || !out) { g_last_error = "liodom_map_create: null argument"; return LIODOM_ERR_INVALID_ARG; }
  *out = nullptr;
  if (!(config->voxel_xysize > 0) || !(config->voxel_zsize > 0) || !(config->resolution > 0) || config->max_cells < 1 ||
      config->cell_capacity < 1 || config->max_update_points < 1 || config->max_modified_cells < 1 ||
      config->max_modified_cells > liodom_dev::kMapNewCellsMax) {
    g_last_error = "liodom_map_create: invalid configuration";
    return LIODOM_ERR_INVALID_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_last_error = "liodom_map_create: no HIP device available (this library has no CPU fallback)";
    return LIODOM_ERR_NO_DEVICE;
  }
  if (config->device < 0 || config->device >= ndev) { g_last_error = "liodom_map_create: device ordinal out of range"; return LIODOM_ERR_INVALID_ARG; }
  HIP_TRY(hipSetDevice(config->device));
  liodom_map* mp = new liodom_map();
  mp->device = config->device;
  int rc = map_build(mp, config, nullptr);
  if (rc == LIODOM_OK) {
    void* raw = nullptr;
    rc = map_alloc(mp, &raw, sizeof(float4) * (size_t)config->max_update_points);
    mp->d_in = reinterpret_cast<float4*>(raw);
    if (rc == LIODOM_OK) { rc = map_alloc(mp, &raw, sizeof(int) * 4); mp->d_n = reinterpret_cast<int*>(raw); mp->d_out_n = mp->d_n + 1; }
    if (rc == LIODOM_OK) { rc = map_alloc(mp, &raw, sizeof(double) * 12); mp->d_T = reinterpret_cast<double*>(raw); }
  }
  if (rc == LIODOM_OK && hipStreamSynchronize(mp->stream) != hipSuccess) { g_last_error = "liodom_map_create: initialisation failed"; rc = LIODOM_ERR_HIP; }
  if (rc != LIODOM_OK) { map_free(mp); return rc; }
  *out = mp;
  return LIODOM_OK;
}

void liodom_map_destroy(liodom_map_t* m) { map_free(m); }

int liodom_map_update(liodom_map_t* mp, const float* xyzi, int64_t n, const double* T) {
  if (!mp || !T || (n > 0 && !xyzi) || n < 0) { g_last_error = "liodom_map_update: invalid argument"; return LIODOM_ERR_INVALID_ARG; }
  if (n > mp->cfg.max_update_points) { g_last_error = "liodom_map_update: more points than max_update_points"; return LIODOM_ERR_CAPACITY; }
  HIP_TRY(hipSetDevice(mp->device));
  const int ni = (int)n;
  if (n) HIP_TRY(hipMemcpyAsync(mp->d_in, xyzi, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, mp->stream));
  HIP_TRY(hipMemcpyAsync(mp->d_n, &ni, sizeof(int), hipMemcpyHostToDevice, mp->stream));
  HIP_TRY(hipMemcpyAsync(mp->d_T, T, sizeof(double) * 12, hipMemcpyHostToDevice, mp->stream));
  int rc = map_enqueue_update(mp, mp->d_in, mp->d_n, mp->d_T, mp->stream);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(mp->stream));      // the staged host values must not be overwritten early
  return LIODOM_OK;
}

int liodom_map_get_local(liodom_map_t* mp, const double* T, int cells_xy, int cells_z, float* xyzi, int64_t cap,
                         int64_t* n_points) {
  if (!mp || !T || cap < 0 || (cap > 0 && !xyzi)) { g_last_error = "liodom_map_get_local: invalid argument"; return LIODOM_ERR_INVALID_ARG; }
  HIP_TRY(hipSetDevice(mp->device));
  int rc = map_ensure_out(mp, cap);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(mp->d_T, T, sizeof(double) * 12, hipMemcpyHostToDevice, mp->stream));
  rc = map_enqueue_local(mp, mp->d_T, cells_xy, cells_z, mp->d_out, mp->out_cap, mp->d_out_n, mp->stream, 0);
  if (rc) return rc;
  return map_fetch_result(mp, xyzi, cap, n_points);
}

int liodom_map_get_all(liodom_map_t* mp, float* xyzi, int64_t cap, int64_t* n_points) {
  if (!mp || cap < 0 || (cap > 0 && !xyzi)) { g_last_error = "liodom_map_get_all: invalid argument"; return LIODOM_ERR_INVALID_ARG; }
  HIP_TRY(hipSetDevice(mp->device));
  int rc = map_ensure_out(mp, cap);
  if (rc) return rc;
  using namespace liodom_dev;
  hipLaunchKernelGGL(k_map_all_plan, dim3(1), dim3(1024), 0, mp->stream, mp->m, mp->out_cap, mp->d_out_n);
  hipLaunchKernelGGL(k_map_gather, dim3(16, 256), dim3(256), 0, mp->stream, mp->m, mp->d_out, mp->out_cap);
  HIP_TRY(hipGetLastError());
  return map_fetch_result(mp, xyzi, cap, n_points);
}

int liodom_map_num_cells(liodom_map_t* mp, int* n_cells) {
  if (!mp || !n_cells) { g_last_error = "liodom_map_num_cells: null argument"; return LIODOM_ERR_INVALID_ARG; }
  HIP_TRY(hipSetDevice(mp->device));
  HIP_TRY(hipStreamSynchronize(mp->stream));
  liodom_dev::MapState st;
  HIP_TRY(hipMemcpy(&st, mp->m.st, sizeof(st), hipMemcpyDeviceToHost));
  *n_cells = st.n_cells;
  return LIODOM_OK;
}

int liodom_map_status(liodom_map_t* mp, uint32_t* status) {
  if (!mp || !status) { g_last_error = "liodom_map_status: null argument"; return LIODOM_ERR_INVALID_ARG; }
  HIP_TRY(hipSetDevice(mp->device));
  HIP_TRY(hipStreamSynchronize(mp->stream));
  liodom_dev::MapState st;
  HIP_TRY(hipMemcpy(&st, mp->m.st, sizeof(st), hipMemcpyDeviceToHost));
  *status = (uint32_t)st.status;
  return LIODOM_OK;
}

}  // extern "C"
