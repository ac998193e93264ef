// kernels_extract.h — ring split (k_classify, k_ring_scatter, k_row_compact) and edge extraction (k_ring_extract).
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, in this order; not a standalone header).
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

// isValidPoint + ring id of one point (feature_extractor.cc:84-102,115-175); 0xFF: dropped.
__device__ __forceinline__ unsigned char classify_point(const DevView& v, const float4& pt, int i, int H, int height, int width) {
  // Fast decision in float for the points that are nowhere near a decision boundary (99.9 %): the FP64 sqrt + atan
  // of the reference's expressions (~350 instructions per point) made this kernel FP64-bound, not bandwidth-bound.
  // The float range / elevation angle are within 1e-4 relative / 1e-5 degrees of the double values, so a point whose
  // float range is further than 1e-4 (relative) from both range limits and whose ring is the same at angle -+ 1e-4
  // degrees gets exactly the reference's verdict; every other point takes the reference's FP64 expressions below.
  bool sure = false;
  int r_fast = -1;
  if (v.lidar_type == 0) {
    const float px = pt.x, py = pt.y, pz = pt.z;
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
        // (round 6: the two trial binnings in float, margin 2e-4 degrees: the one-pass split is VALU-issue bound at 68 % on 256
        //  streams — 240 instructions per point —; as FP64 calls they were 1 % of them: 281 -> 275 us)
        const int r0 = velodyne_ring_from_angle_f(a - 2e-4f, H), r1 = velodyne_ring_from_angle_f(a + 2e-4f, H);
        sure = r0 == r1;
        r_fast = r0;
      }
    }
  }
  double dist;
  if (sure) return r_fast >= 0 ? (unsigned char)r_fast : (unsigned char)0xFF;
  if (!valid_point((double)pt.x, (double)pt.y, (double)pt.z, v.min_range, v.max_range, &dist)) return 0xFF;
  int r;
  if (v.lidar_type == 0) {
    r = velodyne_ring((double)pt.z, dist, H);
  } else {
    r = (width > 0) ? i / width : -1;     // ring = row (feature_extractor.cc:160-173)
    if (r >= H || r >= height) r = -1;
  }
  return r >= 0 ? (unsigned char)r : (unsigned char)0xFF;
}

// copy_out (optional): `in` is HOST memory read over PCIe (page-locked, mapped: the scan is never uploaded by a copy call — this
// kernel's coalesced loads are the upload) and every point is also written to copy_out[stream * copy_stride + i], the device
// copy k_ring_scatter re-reads.
__global__ __launch_bounds__(kTileThreads) void k_classify(DevView v, int s0, const float4* __restrict__ in,
                                                           size_t in_stride, int n, int height, int width,
                                                           float4* __restrict__ copy_out, size_t copy_stride) {
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
  if (copy_out) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = tile * kTilePts + j * kTileThreads + threadIdx.x;
      if (i < n) copy_out[(size_t)blockIdx.y * copy_stride + i] = p[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = tile * kTilePts + j * kTileThreads + threadIdx.x;
    if (i < n) {
      const unsigned char id = classify_point(v, p[j], i, H, height, width);
      if (id != 0xFF) atomicAdd(&hist[id], 1);
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
__host__ __device__ __forceinline__ size_t ring_split_lb_lds_bytes(int H) {      // + rlim [H]
  return ring_scatter_stage_bytes(H) + (size_t)(3 * H + 16) * 4;
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
  float4* out = v.ring_pts + (size_t)s * v.ring_stride;
  int* osrc = v.ring_src + (size_t)s * v.ring_stride;
  const int nvalid = wloc[0] + wloc[1] + wloc[2] + wloc[3] + wloc[4] + wloc[5] + wloc[6] + wloc[7];
  for (int p = tid; p < nvalid; p += kTileThreads) {
    const int d = sdst[stage_swz4(p)];
    out[d] = spts[stage_swz16(p)];
    osrc[d] = ssrc[stage_swz4(p)];
  }
}

// =============================================================================================
// k_ring_split (handles of a few streams): k_classify and k_ring_scatter in ONE pass over the scan — the points stay in
// registers between the two, the scan is read once, the id bytes are never written, one launch less on the extraction stream.
// The scatter needs the ring histograms of ALL tiles of its stream (ring starts = totals over every tile), so a workgroup
// publishes its histogram (write-through stores), counts itself on its stream and waits inside the kernel until the stream's
// other tiles have done the same; then it reads the table at agent scope.  Every workgroup of the launch must be resident at
// once for that: the host uses this kernel only when tiles x streams <= 256 (one workgroup per CU always fits: 50 KB of LDS),
// i.e. up to four HDL-64 streams per launch.  Arrivals are counted in split_ctr[2 s + 1]; the k_ring_extract launch that
// follows every split in stream order zeroes them.
// On lock-step batches the same kernel (workgroups taking their tile from a ticket counter in start order, so that waiting
// workgroups never depend on one that has not started) was measured 3x SLOWER than the two launches — 1 270 us against
// 135 + 305 us per 256-stream step: a waiting workgroup holds its 50 KB of LDS, three per CU, and the tiles of a stream are
// spread over eight XCDs that advance at their own pace — so batches keep k_classify + k_ring_scatter.
// =============================================================================================
__global__ __launch_bounds__(kTileThreads) void k_ring_split(DevView v, int s0, const float4* __restrict__ in,
                                                             size_t in_stride, int n, int height, int width) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  typedef __attribute__((address_space(1))) unsigned short gu16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int hist[256];
  __shared__ int sh_ok;
  const int ntiles = gridDim.x;
  const int H = v.scan_lines;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  const int sy = blockIdx.y, tile = blockIdx.x;
  const int s = s0 + sy;
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
  // ---- classify (k_classify) ----
  float4 p[4];
  int id[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = tile * kTilePts + j * kTileThreads + tid;
    if (i < n) p[j] = in[(size_t)sy * in_stride + i];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = tile * kTilePts + j * kTileThreads + tid;
    id[j] = 0xFF;
    if (i < n) {
      id[j] = (int)classify_point(v, p[j], i, H, height, width);
      if (id[j] != 0xFF) atomicAdd(&hist[id[j]], 1);
    }
  }
  __syncthreads();
  // split_hist[s][ring][tile] (row = v.split_pad tiles, a multiple of 8: a ring's row is read back as 16-byte words)
  unsigned short* th_all = v.split_hist + (size_t)s * H * v.split_pad;
  if (tid < H) __hip_atomic_store((gu16*)(th_all + (size_t)tid * v.split_pad + tile), (unsigned short)hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // lane masks per (chunk, ring) meanwhile (k_ring_scatter, phase A)
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (id[j] != 0xFF) atomicOr(&wmask[(j * (kTileThreads / 64) + wave) * Hp + id[j]], 1ull << lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    // arrive; then wait for the stream's other tiles
    atomicAdd(&v.split_ctr[2 * s + 1], 1u);
    unsigned int spins = 0;
    unsigned long long t0w = 0;
    bool ok = true;
    while (__hip_atomic_load((gu32*)(v.split_ctr + 2 * s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)ntiles) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 4000000u || wait_expired(spins, t0w)) { ok = false; break; }
    }
    if (!ok) atomicOr(&v.state[s].status, LIODOM_STATUS_PIPE_TIMEOUT);
    sh_ok = ok ? 1 : 0;
  }
  // prefix over the 32 chunks for every ring: one half-wave per ring (needs only this tile's masks)
  for (int r = wave * 2 + (lane >> 5); r < H; r += 2 * (kTileThreads / 64)) {
    const int c = lane & 31;
    const int cnt = __popcll(wmask[c * Hp + r]);
    const int ic = half_incl_scan_i32(cnt);
    cbase[c * Hp + r] = (unsigned short)(ic - cnt);
  }
  __syncthreads();
  if (!sh_ok) return;
  // row prefix / totals of the histogram table for "my" ring (thread r < H): 16-byte loads that bypass the non-coherent cache
  // levels (sc1: the other tiles' workgroups ran on other XCDs), up to eight in flight, one wait per batch
  int pre = 0, tot = 0;
  const int mine = tid < H ? hist[tid] : 0;
  if (tid < H) {
    const unsigned short* row = th_all + (size_t)tid * v.split_pad;
    for (int t0 = 0; t0 < ntiles; t0 += 64) {
      // (one asm statement: issued as separate statements the compiler moved the destination registers before the wait.  All
      //  eight words are loaded whatever ntiles is — the table is padded by one batch — and masked below.)
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 w[8];
      asm volatile(
          "global_load_dwordx4 %0, %8, off sc1\n\t"
          "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
          "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
          "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
          "global_load_dwordx4 %4, %8, off offset:64 sc1\n\t"
          "global_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
          "global_load_dwordx4 %6, %8, off offset:96 sc1\n\t"
          "global_load_dwordx4 %7, %8, off offset:112 sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
          : "v"(row + t0)
          : "memory");
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const unsigned int ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int t = t0 + 8 * u + k;
          const int c = (t < ntiles) ? (int)((ww[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu) : 0;
          tot += c;
          if (t < tile) pre += c;
        }
      }
    }
  }
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
  __syncthreads();
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
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (slot[j] >= 0) {
      const int i = tile * kTilePts + j * kTileThreads + tid;
      spts[stage_swz16(slot[j])] = p[j];
      sdst[stage_swz4(slot[j])] = dst[j];
      ssrc[stage_swz4(slot[j])] = i;
    }
  }
  __syncthreads();
  float4* out = v.ring_pts + (size_t)s * v.ring_stride;
  int* osrc = v.ring_src + (size_t)s * v.ring_stride;
  const int nvalid = wloc[0] + wloc[1] + wloc[2] + wloc[3] + wloc[4] + wloc[5] + wloc[6] + wloc[7];
  for (int q = tid; q < nvalid; q += kTileThreads) {
    const int d = sdst[stage_swz4(q)];
    out[d] = spts[stage_swz16(q)];
    osrc[d] = ssrc[stage_swz4(q)];
  }
}

// =============================================================================================
// k_ring_split_lb (lock-step batches, round 6): the ring split in ONE pass over the scan for any number of streams.
// k_ring_split above waits for ALL tiles of its stream (ring starts of the compact layout = totals over every tile), which needs
// every workgroup of the launch resident and lost 3x on batches.  Here the rings have a FIXED PITCH — ring r of a stream starts at
// r * ring_pitch (9/8 of the nominal ring length; k_ring_extract takes starts and lengths from ring_start / ring_len, as for the
// organised clouds of k_row_compact) — so a tile needs only the counts of the tiles BEFORE it:
//   * workgroups take their (stream, tile) from a ticket counter in start order, stream-major: a tile's predecessors have all
//     started before it — no workgroup ever waits for one that has not started, whatever the launch's size;
//   * a tile publishes its per-ring counts as tagged 8-byte words {launch tag, count} (agent-scope stores: the eight L2s are not
//     coherent) right after the classification, and sums its predecessors' words as they appear — all of them in flight at once
//     (thread (w, r) takes the tiles w, w + G, ... for ring r; a stream's 57 tiles start within microseconds of each other, so this
//     is one or two round trips, not a chain of look-backs);
//   * the points stay in registers between classification and scatter: the scan is read once, no id bytes, no second launch.
// A ring that outgrows its pitch (more than 9/8 of the nominal length: not a spinning LiDAR, but the reference takes any cloud)
// raises lb_ovf[stream]; the points beyond the pitch are not written, and k_ring_split_fix — one workgroup per stream, behind this
// launch, idle unless the flag is up — redoes that stream's split into the compact layout, tile by tile.
// =============================================================================================
__device__ __forceinline__ void ring_tile_masks_prefix(unsigned long long* wmask, unsigned short* cbase, int H, int Hp, int wave, int lane) {
  // prefix over the 32 chunks for every ring: one half-wave per ring
  for (int r = wave * 2 + (lane >> 5); r < H; r += 2 * (kTileThreads / 64)) {
    const int c = lane & 31;
    const int cnt = __popcll(wmask[c * Hp + r]);
    const int ic = half_incl_scan_i32(cnt);
    cbase[c * Hp + r] = (unsigned short)(ic - cnt);
  }
}

__global__ __launch_bounds__(kTileThreads) void k_ring_split_lb(DevView v, int s0, const float4* __restrict__ in, size_t in_stride, int n,
                                                                int height, int width, int ntiles, unsigned int tag) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int hist[256];
  __shared__ int sh_pre[256];
  __shared__ int sh_ticket, sh_ok;
  const int H = v.scan_lines;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 256) { hist[tid] = 0; sh_pre[tid] = 0; }
  if (tid == 0) { sh_ticket = (int)atomicAdd(v.lb_ticket, 1u); sh_ok = 1; }
  __syncthreads();
  const int sy = sh_ticket / ntiles, tile = sh_ticket - sy * ntiles;      // stream-major: the tiles before this one have started
  const int s = s0 + sy;
  const int Hp = ring_scatter_stride(H);
  unsigned long long* wmask = reinterpret_cast<unsigned long long*>(smem);          // [32][Hp]   (phase A)
  unsigned short* cbase = reinterpret_cast<unsigned short*>(wmask + kTileChunks * Hp);  // [32][Hp]   (phase A)
  float4* spts = reinterpret_cast<float4*>(smem);                                   // [2048]     (phase B, same bytes)
  int* sdst = reinterpret_cast<int*>(smem + (size_t)kTilePts * 16);                 // [2048]
  int* ssrc = sdst + kTilePts;                                                      // [2048]
  int* rbase = reinterpret_cast<int*>(smem + ring_scatter_stage_bytes(H));          // [H] first position of this tile's points of the ring
  int* lofs = rbase + H;                                                            // [H] first staging slot of the ring
  int* rlim = lofs + H;                                                             // [H] end of the ring's segment
  int* wloc = rlim + H;                                                             // [8] this tile's counts per wave
  for (int k = tid; k < kTileChunks * Hp; k += kTileThreads) wmask[k] = 0ull;
  // ---- classify (k_classify) ----
  float4 p[4];
  int id[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = tile * kTilePts + j * kTileThreads + tid;
    if (i < n) p[j] = in[(size_t)sy * in_stride + i];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = tile * kTilePts + j * kTileThreads + tid;
    id[j] = 0xFF;
    if (i < n) {
      id[j] = (int)classify_point(v, p[j], i, H, height, width);
      if (id[j] != 0xFF) atomicAdd(&hist[id[j]], 1);
    }
  }
  __syncthreads();
  // ---- publish this tile's counts; lane masks meanwhile ----
  unsigned long long* desc = v.lb_desc + (size_t)s * v.tile_cap * v.lb_hpad;
  if (tid < H) __hip_atomic_store((gu64*)(desc + (size_t)tile * v.lb_hpad + tid), ((unsigned long long)tag << 32) | (unsigned int)hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (id[j] != 0xFF) atomicOr(&wmask[(j * (kTileThreads / 64) + wave) * Hp + id[j]], 1ull << lane);
  // ---- the counts of the tiles before this one: thread (w, r) takes tiles w, w + G, ... of ring r, eight loads in flight ----
  {
    const int G = kTileThreads / H > 0 ? kTileThreads / H : 1;      // (H <= 254 < 512)
    const int r = tid % H, w = tid / H;
    if (w < G) {
      int sum = 0;
      unsigned int spins = 0;
      unsigned long long t0w = 0;
      for (int t0 = w; t0 < tile; t0 += 8 * G) {         // batches of eight predecessors
        unsigned int pend = 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) pend |= (t0 + k * G < tile) ? (1u << k) : 0u;
        while (pend) {
          unsigned long long d[8];
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int t = t0 + k * G < tile ? t0 + k * G : w;          // (clamped: the loads leave together, results are masked)
            d[k] = __hip_atomic_load((gu64*)(desc + (size_t)t * v.lb_hpad + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int k = 0; k < 8; k++) {
            if (((pend >> k) & 1u) && (unsigned int)(d[k] >> 32) == tag) { sum += (int)(unsigned int)d[k]; pend &= ~(1u << k); }
          }
          if (!pend) break;
          __builtin_amdgcn_s_sleep(2);
          if (++spins > 4000000u || wait_expired(spins, t0w)) { sh_ok = 0; atomicOr(&v.state[s].status, LIODOM_STATUS_PIPE_TIMEOUT); pend = 0u; t0 = tile; }
        }
      }
      if (sum) atomicAdd(&sh_pre[r], sum);
    }
  }
  __syncthreads();
  ring_tile_masks_prefix(wmask, cbase, H, Hp, wave, lane);
  // staging offsets: exclusive scan of this tile's counts over the rings
  const int mine = tid < H ? hist[tid] : 0;
  const int incl_l = wave_incl_scan_i32(mine);
  if (lane == 63) wloc[wave] = incl_l;
  __syncthreads();
  if (!sh_ok) return;
  {
    int base_l = 0;
    for (int w = 0; w < wave; w++) base_l += wloc[w];
    if (tid < H) {
      const int pitch = v.ring_pitch;
      const int pre = sh_pre[tid];
      rbase[tid] = tid * pitch + pre;
      rlim[tid] = (tid + 1) * pitch;
      lofs[tid] = base_l + incl_l - mine;
      if (pre + mine > pitch) atomicOr(&v.lb_ovf[s], 1u);
      if (tile == 0) v.ring_start[(size_t)s * (H + 1) + tid] = tid * pitch;
      if (tile == ntiles - 1) { const int tot = pre + mine; v.ring_len[(size_t)s * H + tid] = tot < pitch ? tot : pitch; }
    }
    if (tile == 0 && tid == H - 1) v.ring_start[(size_t)s * (H + 1) + H] = H * v.ring_pitch;
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  int slot[4], dst[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    slot[j] = -1; dst[j] = -1;
    if (id[j] != 0xFF) {
      const int chunk = j * (kTileThreads / 64) + wave;
      const int rank = (int)cbase[chunk * Hp + id[j]] + __popcll(wmask[chunk * Hp + id[j]] & below);
      slot[j] = lofs[id[j]] + rank;
      const int d = rbase[id[j]] + rank;
      dst[j] = d < rlim[id[j]] ? d : -1;                 // (beyond the pitch: k_ring_split_fix redoes the stream)
    }
  }
  __syncthreads();                  // masks / prefixes are dead: their bytes become the staging tile
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (slot[j] >= 0) {
      const int i = tile * kTilePts + j * kTileThreads + tid;
      spts[stage_swz16(slot[j])] = p[j];
      sdst[stage_swz4(slot[j])] = dst[j];
      ssrc[stage_swz4(slot[j])] = i;
    }
  }
  __syncthreads();
  float4* out = v.ring_pts + (size_t)s * v.ring_stride;
  int* osrc = v.ring_src + (size_t)s * v.ring_stride;
  const int nvalid = wloc[0] + wloc[1] + wloc[2] + wloc[3] + wloc[4] + wloc[5] + wloc[6] + wloc[7];
  for (int q = tid; q < nvalid; q += kTileThreads) {
    const int d = sdst[stage_swz4(q)];
    if (d >= 0) {
      out[d] = spts[stage_swz16(q)];
      osrc[d] = ssrc[stage_swz4(q)];
    }
  }
}

// One workgroup per stream behind k_ring_split_lb: nothing to do unless a ring of the stream outgrew its pitch; then the stream's
// split once more, into the COMPACT layout (ring starts = running totals, as k_ring_scatter writes it: any ring length up to the
// scan itself fits), tile by tile: a first sweep counts, a second places every point at ring start + points of the ring in earlier
// tiles + rank inside the tile (stable: feature_extractor.cc:115-175 appends in input order).  Slow — one workgroup walks the
// whole scan twice — and only ever run for clouds no spinning LiDAR produces.
__global__ __launch_bounds__(kTileThreads) void k_ring_split_fix(DevView v, int s0, const float4* __restrict__ in, size_t in_stride, int n,
                                                                 int height, int width, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int tot[256];          // points of every ring (first sweep), then its running position (second sweep)
  __shared__ int wtot[16];
  const int sy = blockIdx.y, s = s0 + sy;
  if (!v.lb_ovf[s]) return;         // (uniform)
  const int H = v.scan_lines;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hp = ring_scatter_stride(H);
  unsigned long long* wmask = reinterpret_cast<unsigned long long*>(smem);
  unsigned short* cbase = reinterpret_cast<unsigned short*>(wmask + kTileChunks * Hp);
  if (tid < 256) tot[tid] = 0;
  __syncthreads();
  for (int tile = 0; tile < ntiles; tile++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = tile * kTilePts + j * kTileThreads + tid;
      if (i < n) {
        const int id = (int)classify_point(v, in[(size_t)sy * in_stride + i], i, H, height, width);
        if (id != 0xFF) atomicAdd(&tot[id], 1);
      }
    }
  }
  __syncthreads();
  {
    // ring starts: exclusive scan of the totals over the rings (H <= 254)
    const int mine = tid < H ? tot[tid] : 0;
    const int incl = wave_incl_scan_i32(mine);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wtot[w];
    const int rstart = base + incl - mine;
    __syncthreads();
    if (tid < H) {
      v.ring_start[(size_t)s * (H + 1) + tid] = rstart;
      v.ring_len[(size_t)s * H + tid] = mine;
      tot[tid] = rstart;                                 // running position of the ring
    }
    if (tid == H - 1) v.ring_start[(size_t)s * (H + 1) + H] = rstart + mine;
  }
  __syncthreads();
  float4* out = v.ring_pts + (size_t)s * v.ring_stride;
  int* osrc = v.ring_src + (size_t)s * v.ring_stride;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int tile = 0; tile < ntiles; tile++) {
    for (int k = tid; k < kTileChunks * Hp; k += kTileThreads) wmask[k] = 0ull;
    __syncthreads();
    float4 p[4];
    int id[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = tile * kTilePts + j * kTileThreads + tid;
      id[j] = 0xFF;
      if (i < n) {
        p[j] = in[(size_t)sy * in_stride + i];
        id[j] = (int)classify_point(v, p[j], i, H, height, width);
        if (id[j] != 0xFF) atomicOr(&wmask[(j * (kTileThreads / 64) + wave) * Hp + id[j]], 1ull << lane);
      }
    }
    __syncthreads();
    ring_tile_masks_prefix(wmask, cbase, H, Hp, wave, lane);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (id[j] != 0xFF) {
        const int chunk = j * (kTileThreads / 64) + wave;
        const int rank = (int)cbase[chunk * Hp + id[j]] + __popcll(wmask[chunk * Hp + id[j]] & below);
        const int d = tot[id[j]] + rank;
        out[d] = p[j];
        osrc[d] = tile * kTilePts + j * kTileThreads + tid;
      }
    }
    __syncthreads();
    // advance the rings' running positions by this tile's counts (chunk 31's prefix + its own count)
    if (tid < H) tot[tid] += (int)cbase[(kTileChunks - 1) * Hp + tid] + __popcll(wmask[(kTileChunks - 1) * Hp + tid]);
    __syncthreads();
  }
  if (tid == 0) v.lb_ovf[s] = 0u;
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
  float4* out = v.ring_pts + (size_t)s * v.ring_stride + (size_t)row * width;
  int* osrc = v.ring_src + (size_t)s * v.ring_stride + (size_t)row * width;
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
      // ring index k0 + c0 + i; item c0 + t uses q[t .. t + 10].  A region shorter than the 16 * IPL items of the register
      // tile leaves lanes whose items lie beyond its end: they read up to 16 * IPL + 10 points past the START of the ring's
      // last region, i.e. into the next ring or into the padding behind the last ring of the last stream (liodom_create
      // allocates kExLPR * kExIPLBig + 64 points of it); those items are not owned and their values never used.
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

template <int IPL>
__device__ __forceinline__ void ring_extract_ring(const DevView& v, int s, int ring, unsigned char* smem) {
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
  const float4* rpts = v.ring_pts + (size_t)s * v.ring_stride + rbeg;
  const int* rsrc = v.ring_src + (size_t)s * v.ring_stride + rbeg;
  double* rc = v.ring_c + (size_t)s * v.ring_stride + rbeg;                // debug dump / generic-path scratch
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
    unsigned char* picked = v.ring_picked + (size_t)s * v.ring_stride + rbeg;
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

template <int kMaxThreads, int IPL>
__global__ __launch_bounds__(kMaxThreads) void k_ring_extract(DevView v, int s0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (v.split_ctr && blockIdx.x == 0 && threadIdx.x == 0) {      // (k_ring_split's counters, for the next scan)
    v.split_ctr[2 * (s0 + (int)blockIdx.y) + 1] = 0u;
  }
  if (v.lb_ticket && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *v.lb_ticket = 0u;      // (k_ring_split_lb's ticket counter)
  ring_extract_ring<IPL>(v, s0 + (int)blockIdx.y, (int)blockIdx.x, smem);
}
