// kernels_knn.h — k_knn / k_line_gate / k_ov_gate: edge-to-line correspondences against the window's cell hash.
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, in this order; not a standalone header).
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
//   Round-2 design for the record (DESIGN_HISTORY.md): per-lane Top5 lists for every query, cells streamed in four
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
// Candidate distance with packed FP32 arithmetic (experiment LIODOM_KNN_PK, round 5): dx and dy of one candidate share a
// v_pk_add_f32 and a v_pk_mul_f32 (x and y of a loaded float4 are an aligned register pair); the same operations in the same
// order as sqdist_f, each rounded on its own: bit-identical.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sqdist_cand(float qx, float qy, float qz, const float4& m) {
#if defined(LIODOM_KNN_PK)
  const f32x2_t q = {qx, qy};
  const f32x2_t mm = {m.x, m.y};
  const f32x2_t d = q - mm;
  const f32x2_t sq = d * d;
  float r = sq.x;
  r = r + sq.y;
  const float dz = qz - m.z;
  r = r + dz * dz;
  return r;
#else
  return sqdist_f(qx, qy, qz, m.x, m.y, m.z);
#endif
}
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
          t.consider(iu < cc, sqdist_cand(qx, qy, qz, m[u]), __float_as_int(m[u].w), cs + iu);
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
      t.consider(i + u * kKnnGroup < T, sqdist_cand(qx, qy, qz, m[u]), __float_as_int(m[u].w), a[u]);
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
__device__ void rebuild_count_and_pad(const DevView& v, int s, StreamState& st, int eb, int block, int* sbase, int* sslot, bool keep);

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
__device__ __forceinline__ void knn_probe_cells(const DevView& v, const StreamState& st, int fc, int hl, int cx, int cy, int cz, float qx, float qy, float qz,
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
    cnt = (unsigned int)st.n_ovf[LD_TAB_PARITY(v, fc)];
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
#ifndef LIODOM_OV_MARGIN_PER_M
#define LIODOM_OV_MARGIN_PER_M 0.004f
#endif
constexpr float kOvMarginPerMetre = LIODOM_OV_MARGIN_PER_M;      // + 4 mm per metre of range (sensor frame)
constexpr float kOvMarginMax = 0.30f;
template <int kKnnThreads>
__device__ __forceinline__ void knn_presearch(const DevView& v, int s, const StreamState& st, int fc, int e, int E,
                                              KnnShared<kKnnThreads / kKnnGroup>& sh, KnnPre& pre, float range) {
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
  const int stab = s + LD_TAB_PARITY(v, fc) * v.n_streams;
  const CellSlot* cells = v.cells + (size_t)stab * v.table_size;
  const unsigned int* bits = v.cell_bits + (size_t)stab * (v.table_size >> 5);
  const float4* sp = v.sorted_pts + (size_t)stab * v.sorted_cap;
  unsigned int start = 0, cnt = 0;
  float lb = 0.0f;
  knn_probe_cells<Tune>(v, st, fc, hl, cx, cy, cz, qx, qy, qz, cells, bits, tmask, start, cnt, lb);
  // everything within sqrt(min(d5_old, 1)) + margin of q_old (beyond the 1.0 gate nothing can matter: :324).  The margin grows
  // with the point's range: the first solve corrects the predicted pose by a rotation too, which moves a point 50 m out by
  // centimetres (measured: 7.7 % of the queries moved by more than 1 cm, 4.3 % by more than the flat 3 cm margin of round 3 —
  // every one of them a full search on the launch's critical path); far points are sparse, the wider shell adds few candidates
  const float margin = fminf(kOvMargin + kOvMarginPerMetre * range, kOvMarginMax);
  const float r = fminf(sqrtf(pre.sq.w), 1.0f) + margin;
  const float B = r * r * (1.0f + 1e-5f);
  Top5Acc ta;
  {
    const unsigned long long sentinel = ((unsigned long long)(unsigned int)__float_as_int(B) << 32) | 0x7fffffffull;
    ta.t.k0 = ta.t.k1 = ta.t.k2 = ta.t.k3 = ta.t.k4 = sentinel;
    ta.t.p0 = ta.t.p1 = ta.t.p2 = ta.t.p3 = ta.t.p4 = -1;
  }
  const bool all = cnt > 0 && !(lb > B);
  knn_stream_cells<Top5Acc, 2, 2, 64>(ta, sp, sh.incl[grp], sh.adj[grp], start, all ? cnt : 0u, hl, qx, qy, qz);
  // guard for the points INSIDE the 27-cell block that were not collected: B itself (segments with lb > B and points beyond B
  // were left out) and the fifth entry of a lane whose list is full (it may have dropped candidates at or beyond that distance).
  // Points outside the block are bounded by the re-ranking pass from the new query's position (knn_block, kPre).
  float gl = B;
  if (ta.t.p4 >= 0) gl = fminf(gl, top5_dist(ta.t.k4));
  pre.gsq = __int_as_float((int)half_min_u32((unsigned int)__float_as_int(gl)));
  pre.p[0] = ta.t.p0; pre.p[1] = ta.t.p1; pre.p[2] = ta.t.p2; pre.p[3] = ta.t.p3; pre.p[4] = ta.t.p4;
#pragma unroll
  for (int k = 0; k < 5; k++) pre.c[k] = sp[pre.p[k] >= 0 ? pre.p[k] : 0];
}

// One block of kKnnThreads / 32 queries (virtual block index bv).  Whole workgroup; returns are workgroup-uniform.
// kPre (overlapped second pass): what the re-ranking loads is in `pre` already, and the solve's start point (q, t) comes
// from qt (LDS) — the stream's state is still being written by the first solve's launch.
// kWt: the results leave as write-through stores (another launch of the handle — on another XCD, already running — reads them after
// this workgroup's done flag): the overlapped second pass, and the first pass in chain mode.  fc: frames appended so far (table parity).
template <int kKnnThreads, bool kPre = false, bool kTail = true, bool kWt = kPre>
__device__ __forceinline__ void knn_block(const DevView& v, int s, StreamState& st, int fc, int outer_it, int eb, int bv, int E,
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
    const int stab = s + LD_TAB_PARITY(v, fc) * v.n_streams;
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
      if ((kInstrument && (v.debug & 64)) && hl == 0 && !(pre.gsq > 0.f)) atomicAdd(&v.dbg_clk[262], 1ull);
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
        const double r = sqrt((double)pre.gsq) * (1.0 - 2e-7) - delta;     // every uncollected point of the 27-cell block is at least this far now
        double limit = r > 0.0 ? r * r * (1.0 - 1e-6) : 0.0;               // (float rounding of the new distances included)
        {
          // ... and every point outside that block (the cells around the OLD query's cell) at least one cell size plus the new
          // query's distance to the nearest face of that cell (negative once it has left the cell)
          const double cs = kCellSize;
          // (the cell index exactly as knn_presearch formed it)
          const double fx = (double)qx - (double)(int)floorf(pre.sq.x * kCellInv) * cs, fy = (double)qy - (double)(int)floorf(pre.sq.y * kCellInv) * cs,
                       fz = (double)qz - (double)(int)floorf(pre.sq.z * kCellInv) * cs;
          const double edge = fmin(fmin(fmin(fx, cs - fx), fmin(fy, cs - fy)), fmin(fz, cs - fz));
          const double ro = (cs + edge) * (1.0 - 2e-7);
          const double lo = ro > 0.0 ? ro * ro * (1.0 - 1e-6) : 0.0;
          limit = limit < lo ? limit : lo;
        }
        reranked = (double)d5n < limit || limit > 1.0;                     // beyond the 1.0 gate nothing uncollected can matter
        if ((kInstrument && (v.debug & 64)) && hl == 0) {       // (debug) why a query of the overlapped pass is not certified
          atomicAdd(&v.dbg_clk[261], 1ull);
          if (!reranked) {
            const float rn = fminf(sqrtf(pre.sq.w), 1.0f) + kOvMargin;
            atomicAdd(&v.dbg_clk[g.p4 < 0 ? 264 : 265], 1ull);
            if (delta > 0.01) atomicAdd(&v.dbg_clk[266], 1ull);
            if (pre.gsq < rn * rn) atomicAdd(&v.dbg_clk[267], 1ull);
            if (!(pre.sq.w < 1.0f)) atomicAdd(&v.dbg_clk[268], 1ull);
            if (delta > 0.03) atomicAdd(&v.dbg_clk[269], 1ull);
          }
        }
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
    knn_probe_cells<Tune>(v, st, fc, hl, cx, cy, cz, qx, qy, qz, cells, bits, tmask, start, cnt, lb);
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
    if (outer_it == 0 && v.knn_save_pos && !(kWt && !kPre)) {      // (chain mode: the second pass is the overlapped one, which searches around knn_save_q itself)
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
      float4* ca = v.corr_a + ((size_t)s * 2 + outer_it) * v.edge_cap + eq;
      float4* cb = v.corr_b + ((size_t)s * 2 + outer_it) * v.edge_cap + eq;
      int2* cidx = v.corr_idx + ((size_t)s * 2 + outer_it) * v.edge_cap + eq;
      const float4 oa = valid ? make_float4(nx[0], ny[0], nz[0], 1.0f) : make_float4(0, 0, 0, 0);              // :351-353
      const float4 ob = valid ? make_float4(nx[1], ny[1], nz[1], 0.0f) : make_float4(0, 0, 0, 0);              // :355-357
      const int2 oi = valid ? make_int2(sh.res[q][1], sh.res[q][2]) : make_int2(-1, -1);
      if (kWt) {
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
      unsigned char* cm = &v.corr_mask[((size_t)s * 2 + outer_it) * v.mask_stride + bv];   // bit q = query q accepted
      if (kWt) wt_store_u8(cm, (unsigned char)vb); else *cm = (unsigned char)vb;
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
      for (int i = 0; i < 4; i++) pq[i] = kWt ? qt[i] : st.param_q[i];      // (kWt: the state's copy is being written by a launch that may still run)
#pragma unroll
      for (int i = 0; i < 3; i++) pt[i] = kWt ? qt[4 + i] : st.param_t[i];
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
    if (kWt) wt_store_u64(dst, (unsigned long long)__double_as_longlong(x)); else *dst = x;
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
      wt_store_f4(v.corr_a + ((size_t)s * 2 + outer_it) * v.edge_cap + eq, oa);
      wt_store_f4(v.corr_b + ((size_t)s * 2 + outer_it) * v.edge_cap + eq, ob);
      wt_store_u64(v.corr_idx + ((size_t)s * 2 + outer_it) * v.edge_cap + eq, ((unsigned long long)(unsigned int)oi.y << 32) | (unsigned int)oi.x);
    }
    const unsigned long long vb = __ballot(valid);
    if (q == 0) wt_store_u8(&v.corr_mask[((size_t)s * 2 + outer_it) * v.mask_stride + bv], (unsigned char)vb);   // bit q = query q accepted
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

// Chain mode, first pass, workgroup 0: the release of the previous scan's edge buffer to the extraction (signal_odo: that odometry
// has completed its last evaluation) — when the pass has seen the verdict on the prediction it started from (speculative hand-over,
// kernels_sync.h: the prediction may have left the previous scan's finalising solve BEFORE that evaluation; the verdict is
// published after it).
__device__ __forceinline__ void chain_release_edges(const DevView& v, unsigned int signal_odo) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  if (signal_odo && threadIdx.x == 0) INJECT_DELAY(14);
  if (signal_odo && threadIdx.x == 0) __hip_atomic_store((gu32*)(v.pipe_flags + kEdgePipeBufs), signal_odo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid.x = v.knn_grid workgroups per stream (+ the streamed rebuild's ALLOC workgroups on the second pass): workgroup b
// takes the query blocks b, b + knn_grid, ... below ceil(E / queries) — the grid is sized for the usual edge count
// (half of the capacity), not for edge_cap: on lock-step batches two thirds of an edge_cap-sized grid were workgroups
// that found nothing to do.
// kOv: the overlapped second pass (see "Overlapped second kNN pass" above; one-stream handles, the 256-thread instance):
// launched on stream_k beside the scan's first solve, seq = the launch sequence number the flags carry.
// kChain: the FIRST pass of a scan in chain mode (kernels_sync.h "Chain mode"): it runs on the stream of the kNN passes and the
// rebuild, behind the previous scan's APPEND launch; the scan's first solve — on the other stream — is resident already and waits
// for this pass's done flags, so the results leave write-through.  The prediction the scan starts from comes from pred_xch
// (the previous scan's finalize_scan may still be running: its plain stores to the stream's state are not visible yet) — the
// matrix the queries are transformed with AND the quaternion / translation the residual blocks of the partial sums are evaluated at
// (st.param_q / st.param_t: on the short solves of the 16-ring shapes the pass got there before finalize_scan's stores: wrong
// partial sums, poses off by millimetres, found by the two-thread and soak tests);
// everything else this pass needs of that state follows from scan_no, the number of scans completed before this one.
template <int kKnnThreads, bool kOv, bool kChain>
__device__ __forceinline__ bool knn_pass(const DevView& v, int s, int bxi, int byi, int outer_it, int eb, unsigned int wait_edges,
                                         unsigned int signal_odo, unsigned int seq, int scan_no, KnnShared<kKnnThreads / kKnnGroup>& sh, KnnShared<kKnnThreads / kKnnGroup>& sh2, double* sh_ov, int pred_copy = 0) {
  constexpr int kKnnQueries = kKnnThreads / kKnnGroup;
  static_assert(!(kOv && kChain), "the overlapped pass is the second pass, the chain-mode instance the first");
  StreamState& st = v.state[s];
  if (kOv) {
    // the scan's first solve launch has started: the first kNN pass (and everything before it) has completed
    // (k_ov_gate in front of this launch has seen the flag already: the launch started, with clean caches, after the first pass ended)
    if (!pipe_wait(v.ov_flags + s, seq, &st.status)) return false;
    OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 9); OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 13);
  } else if (kChain) {
    // the prediction (12 doubles, tag = scans completed; normally there long before this launch starts) and, in the same round
    // trip, the flag of the extraction that fills edge buffer eb
    const bool pred_ok = pred_wait(v, s, bxi % kOvReplicas, (unsigned int)scan_no, sh_ov, &st.status, v.pipe_flags + eb, wait_edges, pred_copy);
    OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 27);
    // bookkeeping of the streamed rebuild, as below — from scan_no: every scan appends exactly one frame (finalize_scan).  (With a
    // speculative hand-over the prediction may turn out not to be what the previous scan ended with: the repeated pass of
    // k_chain_redo0 then writes the confirmed one; the repair of the previous scan's APPEND works from its own parity of pred_odom.)
    // (ALSO when the wait above gave up — beside a saturating second process it does —: the scan has failed through the status bit, but
    //  the rebuild's launches behind this pass run whatever happens, and on a list of occupied cells that was not emptied they wrote
    //  past its end: a GPU memory fault in the two-process soak on the 64-ring shape, found with the ROCm debug agent)
    if (bxi == 0 && threadIdx.x == 0) { st.reb_frame_count = scan_no; st.n_used_tab[(scan_no + 1) & 1] = 0; st.reb_initialized = 1; st.cursor = 0; }
    if (pred_ok && bxi == 0 && threadIdx.x >= 64 && threadIdx.x < 76) st.pred_odom[scan_no & 1][threadIdx.x - 64] = sh_ov[threadIdx.x - 64];
    if (!pred_ok) return false;
  } else if (v.early_rebuild) {
    if (bxi >= v.knn_grid) { if (outer_it == 1) rebuild_alloc(v, s, st, bxi - v.knn_grid, (int)gridDim.x - v.knn_grid); return true; }
    // streamed rebuild, bookkeeping before the first builders start (next launch): the frame count the build refers to
    // (the finalising solve advances it beside them), an empty list of occupied slots for the table being built, and the
    // prediction the scan starts from
    if (outer_it == 0 && bxi == 0 && threadIdx.x == 0) {
      st.reb_frame_count = st.frame_count; st.n_used_tab[(st.frame_count + 1) & 1] = 0; st.reb_initialized = st.initialized;
      st.cursor = 0;      // (finalize_scan has reset it already, unless the previous scan ran in chain mode)
    }
    if (outer_it == 0 && bxi == 0 && threadIdx.x >= 64 && threadIdx.x < 76) st.pred_odom[st.frame_count & 1][threadIdx.x - 64] = st.odom[threadIdx.x - 64];
  }
  if (!kOv && outer_it == 0) {
    // (pipelined replay) this launch follows odometry `signal_odo` in stream order: that odometry has completed entirely
    // (chain mode: its last reads of its edge buffer have); and the extraction that fills edge buffer eb (other stream) must have
    // completed before anything of it is read
    if (!kChain && signal_odo && bxi == 0 && byi == 0 && threadIdx.x == 0) {
      typedef __attribute__((address_space(1))) unsigned int gu32;
      __hip_atomic_store((gu32*)(v.pipe_flags + kEdgePipeBufs), signal_odo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!kChain && wait_edges && !pipe_wait(v.pipe_flags + eb, wait_edges, &st.status)) return false;
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
    if (!kOv && !kChain) {
#pragma unroll
      for (int i = 0; i < 12; i++) T[i] = st.odom[i];
    }
  }
  if (kChain) {
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = sh_ov[i];
  }
  const unsigned int st_status = st.status;
  const int st_init = kChain ? 1 : st.initialized;
  const int fc = kChain ? scan_no : st.frame_count;
  // (chain mode: the edge count comes from k_compact_edges' write-through copy in a cache line of its own.  The state word shares
  //  its line with fields that the previous scan's finalize_scan — possibly still running on this XCD when this launch started —
  //  has cached; when the extraction finished after that, a workgroup on that XCD read the count of the scan that used this edge
  //  buffer three scans ago: wrong poses in the two-thread binding on a cold box, where the extraction runs late)
  int E;
  if (kChain) {
    typedef __attribute__((address_space(1))) unsigned int gu32;
    E = (int)__hip_atomic_load((gu32*)(v.edge_cnt + eb * 32 + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    E = st.n_edges_buf[eb];
  }
  if (st_status & LIODOM_STATUS_PIPE_TIMEOUT) return false;      // (uniform) a wait of this handle gave up: the edge buffer may be incomplete
  if (!st_init) return true;                            // uniform over the workgroup
  // (two explicit calls, not a loop over bv: as a loop body the block needs 160 VGPRs instead of 69)
  static_assert(kKnnGridDiv == 2, "k_knn handles exactly two query blocks per workgroup");
  constexpr bool kWt = kOv || kChain;
  if (bxi * kKnnQueries >= E) {             // no query here: empty validity bytes for the solve's compaction
    if (threadIdx.x == 0) {
      unsigned char* cm = &v.corr_mask[((size_t)s * 2 + outer_it) * v.mask_stride + bxi];
      if (kWt) wt_store_u8(cm, 0); else *cm = 0;
      if (bxi + v.knn_grid < v.knn_blocks) { if (kWt) wt_store_u8(cm + v.knn_grid, 0); else cm[v.knn_grid] = 0; }
    }
    return true;
  }
  const int bv2 = bxi + v.knn_grid;
  const int e_second = bv2 * kKnnQueries + (int)(threadIdx.x / kKnnGroup);
  const bool second = bv2 < v.knn_blocks && bv2 * kKnnQueries < E;
  KnnPre pre1, pre2;
  float4 p_second = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kOv) {
    // everything the two blocks need apart from the solve's result; then wait for that
    if (second) p_second = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (e_second < v.edge_cap ? e_second : v.edge_cap - 1)];
    knn_presearch<kKnnThreads>(v, s, st, fc, e_first, E, sh, pre1, sqrtf(p_first.x * p_first.x + p_first.y * p_first.y + p_first.z * p_first.z));
    if (second) knn_presearch<kKnnThreads>(v, s, st, fc, e_second, E, sh2, pre2, sqrtf(p_second.x * p_second.x + p_second.y * p_second.y + p_second.z * p_second.z));
    else pre2.gsq = 0.f;
    if (!ov_wait_pose(v, s, bxi % kOvReplicas, seq, sh_ov, &st.status)) return false;
    OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 10); OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 14);
    if (kInstrument && (v.debug & 128) && threadIdx.x == 0) {      // (debug) when this workgroup saw the pose, relative to its publication: 0.25 us bins
      sh_ov[19] = __longlong_as_double((long long)wall_clock64());
      const unsigned long long pub = *(volatile unsigned long long*)&v.dbg_clk[448 + 1];
      const long long d = (long long)wall_clock64() - (long long)pub;
      if (pub && d >= 0) atomicAdd(&v.dbg_clk[320 + (d / 25 < 63 ? d / 25 : 63)], 1ull);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = sh_ov[i];
  }
  if constexpr (kOv) {
    // both blocks' queries, then their gates and partial sums side by side
    knn_block<kKnnThreads, true, false>(v, s, st, fc, outer_it, eb, bxi, E, sh, p_first, T, pre1, sh_ov + 12);
    if (second) knn_block<kKnnThreads, true, false>(v, s, st, fc, outer_it, eb, bv2, E, sh2, p_second, T, pre2, sh_ov + 12);
    else if (bv2 < v.knn_blocks && threadIdx.x == 0) wt_store_u8(&v.corr_mask[((size_t)s * 2 + outer_it) * v.mask_stride + bv2], 0);
    __syncthreads();
    knn_tail_dual<kKnnThreads>(v, s, outer_it, eb, bxi, bv2, second, E, sh, sh2, sh_ov + 12);
    // speculative hand-over (kernels_sync.h): the pose this workgroup worked from may have been the solve's iterate before its last
    // evaluation; what the solve ended with has arrived by now, as a rule — equal bits: done.  Else the workgroup does not count
    // itself done: its namesake in the k_knn_redo launch behind this one repeats the two blocks from the confirmed pose.
    if (!v.speculate) return true;
    const int cf = ov_confirm_pose(v, s, bxi % kOvReplicas, seq, sh_ov, &st.status);
    if ((kInstrument && (v.debug & 64)) && threadIdx.x == 0) atomicAdd(&v.dbg_clk[271 + (cf == 1 ? 0 : 1)], 1ull);      // (debug) confirmed / not
    return cf == 1;
  }
  knn_block<kKnnThreads, false, true, kWt>(v, s, st, fc, outer_it, eb, bxi, E, sh, p_first, T, pre1, kChain ? sh_ov + 12 : nullptr);
  if (bv2 >= v.knn_blocks) return true;
  if (!second) {
    if (threadIdx.x == 0) { unsigned char* cm = &v.corr_mask[((size_t)s * 2 + outer_it) * v.mask_stride + bv2]; if (kWt) wt_store_u8(cm, 0); else *cm = 0; }
    return true;
  }
  __syncthreads();                          // (the second block reuses the LDS)
  double T2[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (reloaded: kept live across the first block the pose would cost 24 VGPRs)
  if (Tune::kHoistLoads) {
    p_second = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + (e_second < v.edge_cap ? e_second : v.edge_cap - 1)];
    asm volatile("" ::: "memory");
    if (kChain) {
#pragma unroll
      for (int i = 0; i < 12; i++) T2[i] = sh_ov[i];
    } else {
#pragma unroll
      for (int i = 0; i < 12; i++) T2[i] = st.odom[i];
    }
  }
  knn_block<kKnnThreads, false, true, kWt>(v, s, st, fc, outer_it, eb, bv2, E, sh, p_second, T2, pre2, kChain ? sh_ov + 12 : nullptr);
  return true;
}

template <int kKnnThreads, bool kOv = false, bool kChain = false>
__global__ __launch_bounds__(kKnnThreads, (kKnnThreads >= 256 ? 1 : LIODOM_TUNE_B_WAVES)) void k_knn(DevView v, int s0, int outer_it, int eb, unsigned int wait_edges, unsigned int signal_odo, unsigned int seq, int scan_no) {
  constexpr int kKnnQueries = kKnnThreads / kKnnGroup;
#if defined(LIODOM_CHAIN_PRIO)
  // Few-stream handles: this kernel is a link of the scan's dependent chain, the next scan's extraction kernels (other HIP stream)
  // are not; where both have waves on one SIMD the issue arbiter goes by wave priority first (MI355X_MICROARCH.md, "two waves per
  // SIMD").  The stream priority only orders dispatch.
  if (kKnnThreads >= 256) __builtin_amdgcn_s_setprio(LIODOM_CHAIN_PRIO);
#endif
  __shared__ KnnShared<kKnnQueries> shs[kOv ? 2 : 1];      // (overlapped pass: one per query block — their tails run side by side)
  KnnShared<kKnnQueries>& sh = shs[0];
  __shared__ double sh_ov[(kOv || kChain) ? 20 : 1];       // overlapped pass: the first solve's odom[12], q[4], t[3]; chain mode: the prediction
  int bxi = (int)blockIdx.x, byi = (int)blockIdx.y;
  xcd_remap(bxi, byi);
  const int s = s0 + byi;
  if constexpr (kOv) {
    // chain mode: COUNT + PAD of the streamed rebuild as extra workgroups of the second pass's launch (light ones: 256 threads at
    // this kernel's register budget), beside the pass's search and its wait for the first solve.  (As a launch of their own in
    // front of the pass they took 20 us, and the pass behind them no longer ran beside the solve.)
    if (bxi >= v.knn_grid) {
      __shared__ int sh_rb_cnt[kMaxFrames + 1];
      __shared__ int sh_rb_slot[kMaxFrames];
      rebuild_count_and_pad(v, s, v.state[s], eb, bxi - v.knn_grid, sh_rb_cnt, sh_rb_slot, true);
      return;
    }
  }
  if (kInstrument && kOv && threadIdx.x == 0) sh_ov[19] = 0.0;
  if (kOv) { OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 8); OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 12); }
  else if (outer_it == 0) {
    if (kInstrument && (v.debug & 128) && threadIdx.x == 0 && bxi == 0) {
      // (debug) the previous scan's stamps are final when the next scan's first pass starts: fold its phases into running sums
      // (dbg_clk[480..]: first pass, first solve, second pass's tail, finalising solve, period, scans, early hand-overs, APPEND)
      volatile unsigned long long* c = v.dbg_clk + 448;
      const unsigned long long t16 = c[16], now = wall_clock64();
      if (t16 && c[18] > t16 && c[1] > c[18] && c[4] > c[1] && c[28] > c[4] && c[26] > c[28]) {
        unsigned long long* a = v.dbg_clk + 480;
        a[0] += c[18] - t16; a[1] += c[1] - c[18]; a[2] += c[4] - c[1]; a[3] += c[28] - c[4]; a[4] += now - t16; a[5] += 1ull;
        a[6] += c[19] > t16 ? 1ull : 0ull; a[7] += c[26] - c[28];
        if (c[20] > c[1]) { a[8] += c[20] - c[1]; a[9] += c[4] - c[20]; }      // second pass's last workgroup after the pose's (confirmed) publication; -> the finalising solve has seen the count
        if (c[19] > t16 && c[21] < t16) { a[10] += c[1] - c[19]; a[11] += c[20] - c[19]; a[12] += 1ull; a[13] += c[4] - c[1]; }      // handed over early and confirmed
        if (c[19] > t16 && c[21] > t16) { a[14] += 1ull; a[15] += c[4] - c[1]; }                                                       // ... and not confirmed
      }
    }
    OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 16);
  }
  const bool pass_ok = knn_pass<kKnnThreads, kOv, kChain>(v, s, bxi, byi, outer_it, eb, wait_edges, signal_odo, seq, scan_no, sh, shs[kOv ? 1 : 0], sh_ov);
  // (every exit of the pass is workgroup-uniform)  chain mode (scan_no >= 0 on this instance): the pass's workgroups count themselves
  // on one word, as the first pass's do; else one flag per workgroup
  // (a workgroup whose own wait gave up leaves the count short: the solve that waits for it gives up in turn and consumes nothing)
  // (speculative hand-over not confirmed: pass_ok is false — the workgroup's namesake in k_knn_redo signals in its place)
  if (kOv) { if (scan_no >= 0) { if (pass_ok) chain_count_done(v.knn_done0 + 32 + s); } else if (pass_ok || !v.speculate) ov_signal_knn_done(v, s, bxi, seq); }
  if (kChain) {
    // speculative hand-over: the prediction this pass started from may have left the previous scan's finalising solve before its last
    // evaluation; the verdict has arrived by now, as a rule.  Not confirmed: the workgroup does not count itself — its namesake in
    // k_chain_redo0, behind this launch, repeats the pass from the confirmed prediction (after the new frame has been re-appended).
    int vd = 1;
    if (v.speculate) vd = pred_verdict_wait(v, s, bxi % kOvReplicas, (unsigned int)scan_no, &v.state[s].status);
    if (vd != 2 && bxi == 0) chain_release_edges(v, signal_odo);      // (also when the wait gave up: the extraction must not starve)
    if (pass_ok && vd == 1) chain_count_done(v.knn_done0 + s);
  }
  if (kInstrument && (v.debug & 128) && threadIdx.x == 0) {
    if (kOv) {                                         // (debug) pose seen -> flag raised, per workgroup with queries: 0.5 us bins
      const long long t_seen = __double_as_longlong(sh_ov[19]);
      if (bxi * kKnnQueries < v.state[s].n_edges_buf[eb] && t_seen) { const long long d = (long long)wall_clock64() - t_seen; atomicAdd(&v.dbg_clk[256 + (d / 50 < 63 ? d / 50 : 63)], 1ull); }
    } else if (outer_it == 0) {                        // (debug) first pass: end of every workgroup relative to the start of workgroup 0: 0.5 us bins
      const unsigned long long t0 = *(volatile unsigned long long*)&v.dbg_clk[448 + 16];
      const long long d = (long long)wall_clock64() - (long long)t0;
      if (t0 && d >= 0 && bxi * kKnnQueries < v.state[s].n_edges_buf[eb]) atomicAdd(&v.dbg_clk[384 + (d / 50 < 63 ? d / 50 : 63)], 1ull);
    }
  }
  if (kOv) {
    OV_STAMP(v, threadIdx.x == 0 && bxi == 0, 11); OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 15);
    if (kInstrument && (v.debug & 128) && threadIdx.x == 0 && bxi * kKnnQueries < v.state[s].n_edges_buf[eb]) atomicMax(&v.dbg_clk[448 + 20], wall_clock64());      // (debug) the pass's last workgroup with queries
  }
  else if (outer_it == 0) OV_STAMP(v, threadIdx.x == 0 && bxi == v.knn_grid - 1, 17);
}

// Speculative hand-over not confirmed (kernels_sync.h; rare): the launch behind the overlapped second pass, same grid.  A workgroup
// whose namesake worked from the pose the solve really ended with (copy 0 of pose_xch0 == the confirmation copy) has nothing to do;
// else it searches the two blocks like the non-overlapped second pass (from what the first pass saved) at the confirmed pose, with
// write-through results, and counts itself done in the namesake's place.  A kernel of its own so that the pass keeps its register
// budget (as a retry loop in k_knn the pass went from 134 to 324 VGPRs).
template <int kKnnThreads>
// alloc_blocks (chain mode): the launch's first workgroups are the streamed rebuild's ALLOC step (k_rebuild_alloc), which follows
// the pass on its stream anyway — one launch less per scan for the host to enqueue.
__global__ __launch_bounds__(kKnnThreads, 1) void k_knn_redo(DevView v, int s0, int eb, unsigned int seq, int scan_no, int alloc_blocks) {
  constexpr int kKnnQueries = kKnnThreads / kKnnGroup;
  __shared__ KnnShared<kKnnQueries> shs[2];
  __shared__ double sh_ov[20];
  __shared__ double sh_first[20];
  if ((int)blockIdx.x < alloc_blocks) { rebuild_alloc(v, s0 + (int)blockIdx.y, v.state[s0 + (int)blockIdx.y], (int)blockIdx.x, alloc_blocks); return; }
  int bxi = (int)blockIdx.x - alloc_blocks, byi = (int)blockIdx.y;
  const int s = s0 + byi;
  StreamState& st = v.state[s];
  const int rep = bxi % kOvReplicas;
  // (both copies are there: this launch follows the pass, whose workgroups have seen them — or given up)
  if (!granules_wait<LIODOM_POLL_POSE>(v.pose_xch0 + ((size_t)s * kOvReplicas + rep) * 512, kOvGranules, seq, sh_first, &st.status)) return;
  if (!granules_wait<LIODOM_POLL_POSE>(v.pose_xch0 + ((size_t)s * kOvReplicas + rep) * 512 + kOvFinalOffset, kOvGranules, seq, sh_ov, &st.status)) return;
  bool same = true;
  for (int i = 0; i < 19; i++) same = same && __double_as_longlong(sh_first[i]) == __double_as_longlong(sh_ov[i]);
  if (same) return;                                              // (uniform)
  if (st.status & (LIODOM_STATUS_PIPE_TIMEOUT | LIODOM_STATUS_LM_SYNC_TIMEOUT)) return;      // (a wait of the handle has given up: the scan has failed, nothing is repeated)
  if ((kInstrument && (v.debug & 64)) && threadIdx.x == 0) atomicAdd(&v.dbg_clk[273], 1ull);      // (debug) workgroups repeated
  OV_STAMP(v, threadIdx.x == 0, 21);
  // (chain mode: the first pass saved its queries and fifth distances but not its candidates — the overlapped pass collects its own —:
  //  search like a first pass, pruned by the saved fifth distance)
  DevView vr = v;
  vr.knn_save_pos = nullptr;
  const int fc = st.frame_count;
  const int E = st.n_edges_buf[eb];
  const int outer_it = 1;
  if (!st.initialized || (st.status & LIODOM_STATUS_PIPE_TIMEOUT)) return;
  const int bv2 = bxi + v.knn_grid;
  if (bxi * kKnnQueries < E) {
    const bool second = bv2 < v.knn_blocks && bv2 * kKnnQueries < E;
    const float4* ed = v.edges + ((size_t)eb * v.n_streams + s) * v.edge_cap;
    const int e_first = bxi * kKnnQueries + (int)(threadIdx.x / kKnnGroup), e_second = bv2 * kKnnQueries + (int)(threadIdx.x / kKnnGroup);
    const float4 p_first = ed[e_first < v.edge_cap ? e_first : v.edge_cap - 1];
    const float4 p_second = ed[e_second < v.edge_cap ? e_second : v.edge_cap - 1];
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = sh_ov[i];
    KnnPre none;
    none.gsq = 0.f;
    knn_block<kKnnThreads, false, false, true>(vr, s, st, fc, outer_it, eb, bxi, E, shs[0], p_first, T, none, sh_ov + 12);
    if (second) knn_block<kKnnThreads, false, false, true>(vr, s, st, fc, outer_it, eb, bv2, E, shs[1], p_second, T, none, sh_ov + 12);
    else if (bv2 < v.knn_blocks && threadIdx.x == 0) wt_store_u8(&v.corr_mask[((size_t)s * 2 + outer_it) * v.mask_stride + bv2], 0);
    __syncthreads();
    knn_tail_dual<kKnnThreads>(v, s, outer_it, eb, bxi, bv2, second, E, shs[0], shs[1], sh_ov + 12);
  }
  // (a workgroup without queries: its namesake wrote the empty validity bytes and returned before it compared anything — it counted itself)
  else return;
  if (scan_no >= 0) chain_count_done(v.knn_done0 + 32 + s); else ov_signal_knn_done(v, s, bxi, seq);
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
  if (v.knn8_cnt && blockIdx.x == 0 && threadIdx.x == 0) v.knn8_cnt[s] = 0;      // (k_knn8's list for k_knn8_exact, the launch before this one: consumed)
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
    float4* ca = v.corr_a + ((size_t)s * 2 + outer_it) * v.edge_cap + eq;
    float4* cb = v.corr_b + ((size_t)s * 2 + outer_it) * v.edge_cap + eq;
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
  if ((lane % Q) == 0) v.corr_mask[((size_t)s * 2 + outer_it) * v.mask_stride + eq / Q] = (unsigned char)((vb >> lane) & ((1ull << Q) - 1ull));
}
