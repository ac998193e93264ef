// kernels_knn8.h — k_knn8: the kNN passes of lock-step batches (handles with >= 16 streams), EIGHT lanes per query.
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, behind kernels_knn.h; not a standalone header).
// =============================================================================================
// Why (round 6, profiles/r06_knn_budget.txt, 256 lock-step HDL-64 streams): k_knn<128> — a half-wave per query — issues 425 VALU
// wave-instructions per query in a scan's first pass and 195 in its second, and is VALU-issue bound (4 instead of 7 waves per SIMD:
// +8 % time).  A query streams ~100 candidates, i.e. three or four rounds of its 32 lanes at ~17 instructions per round: ~12 % of
// those 425.  The rest is per-QUERY work that a wave-instruction does for the wave's TWO queries only: the FP64 transform, 27
// probes, prefix + binary search of the flat candidate list, the ladder bound, five half-wave minima with ballots and LDS-crossbar
// shuffles, the save for the second pass.  Fewer candidates (the round-5 verdict's sub-cell ordering) would shave the 12 %;
// MORE QUERIES PER WAVE-INSTRUCTION shaves the 88 %.
//
// k_knn8: a query is a group of 8 lanes, a wave holds 8 queries, a 256-thread workgroup 32; everything between the lanes of a
// query is DPP within 8 lanes (quad permutes + half-row mirror: three steps per reduction, no LDS crossbar, no barrier — the waves
// of a workgroup never talk to each other).
//   own cell    every lane of the group probes the query's own 1 m cell (same address: one request) and the 8 lanes walk its
//               points, lane j taking points j, j + 8, ...; a lane keeps its THREE nearest candidates (distance + position) and
//               the DISTANCE of its fourth ("Best3": 13 straight-line VALU instructions per candidate).
//   bound       first pass: B = the smallest of a ladder of thresholds at or below which five of the group's 24 kept entries lie
//               (1.0, the gate of laser_odometry.cc:324, if none); second pass: B = (sqrt(d5 of the first pass) + |q_new - q_old|)^2.
//   neighbours  PROBED LAZILY: lane j owns the cells j, j + 8, j + 16, j + 24 of the 26 around the own one; only those whose box
//               distance is <= B are looked up at all (1.4 per query on the headline stream: most queries never touch the hash
//               again), and streamed cell by cell by the whole group.
//   selection   five (+ one, for the tie test) 8-lane minima of 64-bit (distance, position) keys; exact whenever the six smallest
//               kept distances are pairwise different and every lane's fourth distance lies above the fifth popped (else, 0.8 % of
//               the queries, the stream is repeated with sorted (distance, window index) lists that start at the known bound —
//               FLANN's order in every case, as in k_knn).
//   second pass re-ranks the 24 candidates the first pass kept (3 gathers per lane) and certifies the result with the guard argument
//               of k_knn (no map point outside the kept set was closer to the old query than sqrt(guard): kept lanes' fourth
//               distances, the box distances of every cell that was not streamed — probed or not —, the border of the 27-cell
//               block); an uncertified query searches, pruned with the first pass's fifth distance.
// Results: the five neighbours of every query for k_line_gate (knn_nn), exactly what k_knn<128> leaves there.
// LIODOM_KNN8=0 keeps k_knn<128> (tests compare the two bit for bit).
// Measured on top of this form and NOT kept (256 streams, same box, bench.py's slot: search + exact + line gate per pass; this form
// 320-323 us): queries prepared by one thread each and handed to the groups through LDS, box bounds from the six face distances,
// the four probes of a lane in two round trips instead of eight, the cells found walked as one list per group — -5 % VALU
// instructions (1.29e8 -> 1.23e8 in a first pass), 325-335 us in every combination; the workgroup's queries dealt to the groups a
// second time by the length of their neighbour lists (first pass) / by whether they still search (second pass: 9 % do, spread over
// 38 % of the waves) — -14 % instructions, +9 % time (two more barriers per workgroup).  A workgroup here is ONE chain of ~20
// dependent memory round trips; seven of them per SIMD do not hide it, so instructions saved are not time saved.
// =============================================================================================
// (instrumented builds, LIODOM_DEBUG_CLOCKS bit 8: shader cycles per phase of k_knn8, summed over the first lanes of the working waves of
//  all streams — dbg_clk[64 + 8 * pass + phase]; tools/gpu_debug.py knn8phases)
#define KNN8_PHASE(i) do { if (kInstrument && (v.debug & 256) && (threadIdx.x & 63) == 0) { const unsigned long long t_now = __builtin_readcyclecounter(); atomicAdd(&v.dbg_clk[64 + 8 * outer_it + (i)], t_now - t_ph); t_ph = t_now; } } while (0)
constexpr int kG8 = 8;                    // lanes per query
#ifndef LIODOM_KNN8_THREADS
#define LIODOM_KNN8_THREADS 256
#endif
constexpr int kKnn8Threads = LIODOM_KNN8_THREADS;      // 4 waves = 32 queries per workgroup (512 threads: the sort below balances better, but the workgroups pack worse — 373 us against 334 for a first pass at 256 streams)
constexpr int kKnn8Queries = kKnn8Threads / kG8;
constexpr int kKnn8Save = 3 * kG8;        // candidates kept per query for the second pass

// ---- 8-lane group primitives (every lane of the group active) ----
__device__ __forceinline__ unsigned int g8_min_u32(unsigned int v) {
  unsigned int o;
  o = (unsigned int)dpp_i32<DPP_XOR1>((int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<DPP_XOR2>((int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<DPP_HALF_MIRROR>((int)v); v = o < v ? o : v;
  return v;   // uniform over each group of 8 lanes
}
__device__ __forceinline__ unsigned long long g8_min_u64(unsigned long long v) {
  unsigned long long o;
  o = dpp_u64<DPP_XOR1>(v); v = o < v ? o : v;
  o = dpp_u64<DPP_XOR2>(v); v = o < v ? o : v;
  o = dpp_u64<DPP_HALF_MIRROR>(v); v = o < v ? o : v;
  return v;
}
__device__ __forceinline__ int g8_sum_i32(int v) {
  v += dpp_i32<DPP_XOR1>(v);
  v += dpp_i32<DPP_XOR2>(v);
  v += dpp_i32<DPP_HALF_MIRROR>(v);
  return v;
}
// bit j = predicate of lane j of this lane's group
__device__ __forceinline__ unsigned int g8_ballot(bool p, int gshift) { return (unsigned int)((__ballot(p) >> gshift) & 0xFFull); }

// The three nearest candidates a lane has seen (distance + position) and the DISTANCE of its fourth nearest.
struct Best3Acc {
  float m1, m2, m3, m4;
  int p1, p2, p3;
  __device__ __forceinline__ void clear() { m1 = m2 = m3 = m4 = __int_as_float(0x7f800000); p1 = p2 = p3 = -1; }
  __device__ __forceinline__ void consider(bool ok, float d0, int /*wi*/, int pos) {
    const float d = ok ? d0 : __int_as_float(0x7f800000);
    const bool lt1 = d < m1, lt2 = d < m2, lt3 = d < m3;
    m4 = __builtin_amdgcn_fmed3f(m3, m4, d);      // fourth smallest of {m1 <= m2 <= m3 <= m4, d}
    const int q3 = lt3 ? pos : p3;
    p3 = lt2 ? p2 : q3;
    m3 = __builtin_amdgcn_fmed3f(m2, m3, d);
    const int q2 = lt2 ? pos : p2;
    p2 = lt1 ? p1 : q2;
    m2 = __builtin_amdgcn_fmed3f(m1, m2, d);
    p1 = lt1 ? pos : p1;
    m1 = lt1 ? d : m1;
  }
};

// The group's 8 lanes walk one cell (start, cnt: uniform over the group, 0 = nothing; groups of a wave differ).  The loads of round
// r + 1 leave before round r is evaluated (a wave holds eight queries and few waves fit a SIMD: the rounds of a cell would otherwise
// be a chain of dependent memory round trips); loads are unconditional with clamped indices, results masked.
// min_wi (hash_incr, k_hash_append): candidates whose stored window index lies below it belong to frames the window has dropped
// since the table was rebuilt — skipped; the others' indices are handed on minus min_wi (their current window index).
template <class Acc, int U, int kLanes = kG8>      // (kLanes: the lanes that share the cell — a group of 8, or the whole wave in k_knn8_exact)
__device__ __forceinline__ void g8_stream_cell(Acc& t, const float4* sp, int start, int cnt, int j, float qx, float qy, float qz, int min_wi) {
  if (cnt <= 0) return;
  const float4* cp = sp + start;
  float4 cur[U];
#pragma unroll
  for (int u = 0; u < U; u++) { const int iu = j + u * kLanes; cur[u] = cp[iu < cnt ? iu : cnt - 1]; }
  for (int i = j; i < cnt; i += U * kLanes) {
    float4 nxt[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int iu = i + (U + u) * kLanes; nxt[u] = cp[iu < cnt ? iu : cnt - 1]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int iu = i + u * kLanes;
      const int wi = __float_as_int(cur[u].w);
      t.consider(iu < cnt && wi >= min_wi, sqdist_cand(qx, qy, qz, cur[u]), wi - min_wi, start + iu);
    }
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = nxt[u];
  }
}

// One probe of the cell hash (occupancy bit first: the slots of empty cells are never loaded).
__device__ __forceinline__ void g8_probe(const DevView& v, const CellSlot* cells, const unsigned int* bits, unsigned int tmask,
                                         int cx, int cy, int cz, unsigned int& start, unsigned int& cnt) {
  const unsigned long long key = pack_cell(cx, cy, cz);
  unsigned int h = hash_cell(key, tmask);
  for (int pr = 0; pr < v.table_size; pr++) {
    if (!((bits[h >> 5] >> (h & 31)) & 1u)) break;           // empty slot: cell not in the map
    const uint4 raw = *reinterpret_cast<const uint4*>(cells + h);
    const unsigned long long k = ((unsigned long long)raw.y << 32) | raw.x;
    if (k == key) { start = raw.z; cnt = raw.w; break; }
    h = (h + 1) & tmask;
  }
}

// Box distance (squared, float) from q to neighbour cell c27 = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1) of the cell (cx, cy, cz),
// shrunk by 1e-5 as in knn_probe_cells: rounding of the candidate distances can never make a pruned point look closer than the bound.
__device__ __forceinline__ float g8_cell_lb(int c27, int cx, int cy, int cz, float qx, float qy, float qz, int& ox, int& oy, int& oz) {
  const int dx = c27 % 3 - 1, dy = (c27 / 3) % 3 - 1, dz = c27 / 9 - 1;
  ox = cx + dx; oy = cy + dy; oz = cz + dz;
  const float cs = (float)kCellSize;
  const float lx = (float)ox * cs, ly = (float)oy * cs, lz = (float)oz * cs;
  const float ex = qx < lx ? lx - qx : (qx > lx + cs ? qx - (lx + cs) : 0.0f);
  const float ey = qy < ly ? ly - qy : (qy > ly + cs ? qy - (ly + cs) : 0.0f);
  const float ez = qz < lz ? lz - qz : (qz > lz + cs ? qz - (lz + cs) : 0.0f);
  return (ex * ex + ey * ey + ez * ez) * (1.0f - 1e-5f);
}

// Pops the five nearest of the group's query from the lanes' Best3 entries: keys (distance bits << 32 | position) in ascending
// order -> key5[0..4]; returns true when that result is certain (see best2_select): d5 < 1.0: the six smallest kept distances are
// pairwise different and every lane's fourth distance lies above the fifth popped; d5 >= 1.0: no lane dropped anything inside the gate.
__device__ __forceinline__ bool best3_select(const Best3Acc& t, float& d5, int (&pos)[5]) {
  const unsigned long long inf = (0x7f800000ull << 32) | 0xFFFFFFFFull;
  unsigned long long a = ((unsigned long long)(unsigned int)__float_as_int(t.m1) << 32) | (unsigned int)t.p1;
  unsigned long long b = ((unsigned long long)(unsigned int)__float_as_int(t.m2) << 32) | (unsigned int)t.p2;
  unsigned long long c = ((unsigned long long)(unsigned int)__float_as_int(t.m3) << 32) | (unsigned int)t.p3;
  unsigned int g[6];
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const unsigned long long k = g8_min_u64(a);       // positions are unique: exactly one lane holds the minimum (or all hold "empty")
    g[r] = (unsigned int)(k >> 32);
    pos[r] = (int)(unsigned int)(k & 0xFFFFFFFFull);
    const bool mine = a == k;
    a = mine ? b : a;
    b = mine ? c : b;
    c = mine ? inf : c;
  }
  g[5] = g8_min_u32((unsigned int)(a >> 32));
  const unsigned int s4 = g8_min_u32((unsigned int)__float_as_int(t.m4));
  const unsigned int one = 0x3f800000u;
  d5 = __int_as_float((int)g[4]);
  if (g[4] < one) return g[0] < g[1] && g[1] < g[2] && g[2] < g[3] && g[3] < g[4] && g[4] < g[5] && g[4] < s4;
  return s4 >= one;
}

// The neighbour cells of a group: lane j owns cells j, j + 8, j + 16, j + 24 (< 27, not the own one) of the 27-cell block.  Round k:
// every lane looks its cell up if its box distance is <= B (most are not: no probe at all), then the group streams the cells its
// lanes found, one after the other, with all 8 lanes.  Returns the smallest box distance among this lane's cells that were NOT
// streamed (inf: none) — the second pass's guard needs it.
template <class Acc, int U>
__device__ __forceinline__ float g8_neighbours(Acc& t, const DevView& v, const CellSlot* cells, const unsigned int* bits, unsigned int tmask,
                                               const float4* sp, int j, int gshift, int cx, int cy, int cz, float qx, float qy, float qz, float B, int min_wi,
                                               int spill_base, int n_spill) {
  float lb_skipped = __int_as_float(0x7f800000);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int c27 = k * kG8 + j;
    unsigned int start = 0, cnt = 0;
    if (c27 == 31) { start = (unsigned int)spill_base; cnt = (unsigned int)n_spill; }      // (k_hash_append's spill list rides as a 28th "cell" of every query: lane 7, last round)
    if (c27 < 27 && c27 != 13) {
      int ox, oy, oz;
      const float lb = g8_cell_lb(c27, cx, cy, cz, qx, qy, qz, ox, oy, oz);
      if (!(lb > B)) g8_probe(v, cells, bits, tmask, ox, oy, oz, start, cnt);
      else lb_skipped = fminf(lb_skipped, lb);
    }
    unsigned int pend = g8_ballot(cnt > 0, gshift);
    while (pend) {                                   // (divergent between the groups of a wave: exec-masked)
      const int l = __ffs(pend) - 1;
      pend &= pend - 1u;
      const int cs = __shfl((int)start, l, kG8), cc = __shfl((int)cnt, l, kG8);
      g8_stream_cell<Acc, U>(t, sp, cs, cc, j, qx, qy, qz, min_wi);
    }
  }
  return lb_skipped;
}

// Ladder bound (see best2_bound) over the group's 24 kept entries.
__device__ __forceinline__ float best3_bound(const Best3Acc& t) {
  float B = 1.0f;
  const float thr[4] = {0.36f, 0.09f, 0.0225f, 0.0036f};      // (0.6 m, 0.3 m, 0.15 m, 0.06 m) squared, descending
  int w = 0;                                                  // the four counts (<= 24 each) side by side: one group sum
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int c = (t.m1 <= thr[k] ? 1 : 0) + (t.m2 <= thr[k] ? 1 : 0) + (t.m3 <= thr[k] ? 1 : 0);
    w |= c << (8 * k);
  }
  w = g8_sum_i32(w);
#pragma unroll
  for (int k = 0; k < 4; k++) B = ((w >> (8 * k)) & 0xFF) >= 5 ? thr[k] : B;
  return B;
}

// grid.x workgroups per stream walk the query blocks b, b + grid.x, ... below ceil(E / 32); grid.y = streams (XCD-aware mapping).
#ifndef LIODOM_KNN8_SORT
#define LIODOM_KNN8_SORT 1
#endif
constexpr bool kKnn8Sort = LIODOM_KNN8_SORT != 0;      // first pass: queries dealt to the groups in the order of their own cells' populations
#ifndef LIODOM_KNN8_U
#define LIODOM_KNN8_U 2          // candidate loads in flight per lane and round (x 2: the next round's leave before this round is evaluated)
#endif
#ifndef LIODOM_KNN8_WAVES
#define LIODOM_KNN8_WAVES 7
#endif
template <int outer_it>      // the scan's first (0) or second (1) pass: two instances, each without the other's code and registers
__global__ __launch_bounds__(kKnn8Threads, LIODOM_KNN8_WAVES) void k_knn8(DevView v, int s0, int eb) {
  int bxi = (int)blockIdx.x, byi = (int)blockIdx.y;
  xcd_remap(bxi, byi);
  const int s = s0 + byi;
  StreamState& st = v.state[s];
  if (!st.initialized) return;                           // (uniform) first scan: no map yet
  if (st.status & LIODOM_STATUS_PIPE_TIMEOUT) return;
  const int E = st.n_edges_buf[eb];
  const int j = (int)(threadIdx.x & 7);                  // lane of the group
  const int gshift = (int)(threadIdx.x & 56);            // first lane of the group inside its wave
  const int grp = (int)(threadIdx.x >> 3);               // query of the workgroup
  const unsigned int tmask = st.table_mask;
  const int fc = st.frame_count;
  const int stab = s + LD_TAB_PARITY(v, fc) * v.n_streams;
  const CellSlot* cells = v.cells + (size_t)stab * v.table_size;
  const unsigned int* bits = v.cell_bits + (size_t)stab * (v.table_size >> 5);
  const float4* sp = v.sorted_pts + (size_t)stab * v.sorted_cap;
  const float inf = __int_as_float(0x7f800000);
  const int min_wi = v.hash_incr ? st.hb_shift : 0;      // (k_hash_append: stored window indices below it are evicted points)
  const int n_spill = v.hash_incr ? min(st.hb_spill, v.sorted_cap - v.hb_spill_base) : 0;      // ... appended points without a place in a cell: every query scans them
  // first pass: the workgroup's queries are dealt to its groups in the order of their own cells' populations (see below)
  __shared__ float4 sh_q[2][kKnn8Queries];                // query (xyz) and edge number, per parity of the loop
  __shared__ int2 sh_own[2][kKnn8Queries];                // own cell: start, count (-1: no query)
  __shared__ int sh_perm[2][kKnn8Queries];
  int par = 0;
  for (int blk = bxi; blk * kKnn8Queries < E; blk += (int)gridDim.x, par ^= 1) {
    int e = blk * kKnn8Queries + grp;
    unsigned long long t_ph = (kInstrument && (v.debug & 256)) ? __builtin_readcyclecounter() : 0ull;
    // ---- the query: edge -> world with the pose the pass searches at (laser_odometry.cc:307-308) ----
    bool active = e < E;                                  // (uniform over the group)
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (active) {
      const float4 p = v.edges[((size_t)eb * v.n_streams + s) * v.edge_cap + e];
      double T[12];
#pragma unroll
      for (int i = 0; i < 12; i++) T[i] = st.odom[i];
      transform_point(T, p.x, p.y, p.z, &qx, &qy, &qz);
      if (v.knn_q && j == 0) v.knn_q[((size_t)s * 2 + outer_it) * v.edge_cap + e] = make_float4(qx, qy, qz, 0.f);
      active = ld_isfinite((double)qx) && ld_isfinite((double)qy) && ld_isfinite((double)qz) &&
               fabsf(qx) < 1.0e9f && fabsf(qy) < 1.0e9f && fabsf(qz) < 1.0e9f;
    }
    int cx = (int)floorf(qx * kCellInv), cy = (int)floorf(qy * kCellInv), cz = (int)floorf(qz * kCellInv);
    unsigned int own_start = 0, own_cnt = 0;
    if (outer_it == 0 && kKnn8Sort) {
      // A wave walks its eight queries' cells in lock-step: it lasts as long as the query with the most populous cell (a pole or a
      // corner seen in 20 frames holds hundreds of points, the cell next to it a dozen).  In edge order the eight differ widely —
      // 21.5 rounds of the own cell per wave on the headline stream where 9.0 would do if all were equal — so the workgroup sorts
      // its queries by the population of their own cell first and deals them to its groups in that order: 13.9 rounds with 32
      // queries per workgroup (tools/knn_budget_cpu.py).  The queries swap groups through LDS: 24 bytes each.
      if (active) g8_probe(v, cells, bits, tmask, cx, cy, cz, own_start, own_cnt);
      const int key = active ? (int)own_cnt : -1;
      if (j == 0) { sh_q[par][grp] = make_float4(qx, qy, qz, __int_as_float(e)); sh_own[par][grp] = make_int2((int)own_start, key); }
      __syncthreads();
      int below = 0;
#pragma unroll
      for (int i = 0; i < kKnn8Queries / kG8; i++) {
        const int o = i * kG8 + j;
        const int ko = sh_own[par][o].y;
        below += (ko < key || (ko == key && o < grp)) ? 1 : 0;
      }
      below = g8_sum_i32(below);                          // this query's place in the order
      if (j == 0) sh_perm[par][below] = grp;
      __syncthreads();
      const int src = sh_perm[par][grp];                  // the query this group continues with
      const float4 q4 = sh_q[par][src];
      const int2 o2 = sh_own[par][src];
      qx = q4.x; qy = q4.y; qz = q4.z; e = __float_as_int(q4.w);
      own_start = (unsigned int)o2.x; own_cnt = o2.y > 0 ? (unsigned int)o2.y : 0u;
      active = o2.y >= 0;
      cx = (int)floorf(qx * kCellInv); cy = (int)floorf(qy * kCellInv); cz = (int)floorf(qz * kCellInv);
    }
    KNN8_PHASE(0);      // query, own-cell probe, sort
    float d5 = inf;
    int pos5[5] = {-1, -1, -1, -1, -1};
    bool done = !active;                                  // (uniform over the group) the five nearest are known
    bool exact = false;                                   // (uniform over the group) ... are left to k_knn8_exact
    // ---- second pass: re-rank what the first pass kept ----
    float4 sq = make_float4(0.f, 0.f, 0.f, inf);          // the first pass's query and fifth distance
    if (outer_it == 1 && active && v.knn_save_q) sq = v.knn_save_q[(size_t)s * v.edge_cap + e];
    if (outer_it == 1 && active && v.knn_save_pos && !v.knn_exact_only) {
      const float gsq = v.knn_save_g[(size_t)s * v.edge_cap + e];
      if (gsq > 0.f) {                                    // (uniform over the group)
        const int* sv = reinterpret_cast<const int*>(v.knn_save_pos) + ((size_t)s * v.edge_cap + e) * (2 * kKnnGroup) + j * 3;
        const int s0p = sv[0], s1p = sv[1], s2p = sv[2];
        const float4 m0 = sp[s0p >= 0 ? s0p : 0], m1 = sp[s1p >= 0 ? s1p : 0], m2 = sp[s2p >= 0 ? s2p : 0];
        Best3Acc br;
        br.clear();
        br.consider(s0p >= 0, sqdist_f(qx, qy, qz, m0.x, m0.y, m0.z), 0, s0p);
        br.consider(s1p >= 0, sqdist_f(qx, qy, qz, m1.x, m1.y, m1.z), 0, s1p);
        br.consider(s2p >= 0, sqdist_f(qx, qy, qz, m2.x, m2.y, m2.z), 0, s2p);
        float d5n;
        int p5[5];
        const bool sel_ok = best3_select(br, d5n, p5);
        // every unsaved point is at least sqrt(gsq) - |q_new - q_old| away now.  In float, every rounding taken against the
        // certificate (relative margins of 1e-6 on each factor against errors of 6e-8 per operation): never certifies what the
        // FP64 form of k_knn would not; the FP64 square roots cost this pass 60 instructions per wave
        const float ddx = qx - sq.x, ddy = qy - sq.y, ddz = qz - sq.z;
        const float delta = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) * (1.0f + 2e-6f) + 1e-30f;
        const float r = sqrtf(gsq) * (1.0f - 2e-6f) - delta;
        const float limit = r > 0.0f ? r * r * (1.0f - 4e-6f) : 0.0f;      // (float rounding of the new distances included)
        if (sel_ok && (d5n < limit || limit > 1.0f)) {              // beyond the 1.0 gate nothing unsaved can matter
          d5 = d5n;
#pragma unroll
          for (int k = 0; k < 5; k++) pos5[k] = p5[k];
          done = true;
        }
      }
    }
    // ---- search ----
    KNN8_PHASE(1);      // second pass: re-ranking
    if (kInstrument && (v.debug & 256) && outer_it == 1) {          // how many queries of the second pass search, and how many waves that keeps busy
      const unsigned long long srch = __ballot(!done && j == 0);
      if ((threadIdx.x & 63) == 0) { atomicAdd(&v.dbg_clk[64 + 15], (unsigned long long)__popcll(srch)); atomicAdd(&v.dbg_clk[64 + 7], srch ? 1ull : 0ull); atomicAdd(&v.dbg_clk[64 + 23], 1ull); }
    }
    if (!done) {                                          // (uniform over the group)
      float B = 1.0f;
      bool have_b = false;
      if (outer_it == 1 && sq.w < 1.0f) {
        const float ddx = qx - sq.x, ddy = qy - sq.y, ddz = qz - sq.z;
        const float delta = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
        const float r = sqrtf(sq.w) * (1.0f + 1e-6f) + delta * (1.0f + 1e-6f) + 1e-7f;
        B = fminf(1.0f, r * r * (1.0f + 1e-5f));
        have_b = true;
      }
      if (!(outer_it == 0 && kKnn8Sort)) g8_probe(v, cells, bits, tmask, cx, cy, cz, own_start, own_cnt);
      Best3Acc b3;
      b3.clear();
      g8_stream_cell<Best3Acc, LIODOM_KNN8_U>(b3, sp, (int)own_start, (int)own_cnt, j, qx, qy, qz, min_wi);
      if (!have_b) B = best3_bound(b3);
      KNN8_PHASE(2);    // own cell
      const float lb_skipped = g8_neighbours<Best3Acc, LIODOM_KNN8_U>(b3, v, cells, bits, tmask, sp, j, gshift, cx, cy, cz, qx, qy, qz, B, min_wi, v.hb_spill_base, n_spill);
      KNN8_PHASE(3);    // bound + neighbour cells
      // not certain (0.8 % of the queries: three of the nearest in one lane with a fourth at or below the fifth distance, or equal
      // distances; all queries with LIODOM_KNN_EXACT_ONLY): the query goes to k_knn8_exact, the launch behind this one, through
      // its knn_nn record.  (d5, the fifth popped distance, bounds the true fifth-nearest distance from above whenever it is finite.)
      // Measured alternatives: the sorted-list path inside this kernel cost 40 VGPRs — half of the waves a SIMD holds — for every
      // query; a second try with the lanes' shares of every cell rotated, as further rounds of this loop, cost +10 % / +18 % of the
      // passes' instructions and 80 us per pass at 256 streams (a third of the workgroups run a round for one query).
      exact = !(best3_select(b3, d5, pos5) && !v.knn_exact_only);
      KNN8_PHASE(4);    // selection
      if (outer_it == 0 && v.knn_save_pos) {
        // what the second pass re-ranks: the lanes' kept candidates, and the guard (see the header)
        unsigned int gd = g8_min_u32((unsigned int)__float_as_int(b3.m4));
        const unsigned int sk = g8_min_u32((unsigned int)__float_as_int(lb_skipped));
        gd = sk < gd ? sk : gd;
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
        int* sv = reinterpret_cast<int*>(v.knn_save_pos) + ((size_t)s * v.edge_cap + e) * (2 * kKnnGroup) + j * 3;
        sv[0] = b3.p1; sv[1] = b3.p2; sv[2] = b3.p3;
        if (j == 0) v.knn_save_g[(size_t)s * v.edge_cap + e] = guard < 3.0e38f ? guard : 3.0e38f;
      }
    }
    // ---- what the second pass prunes with: the query and its fifth-nearest distance (inf: fewer than five candidates / no query) ----
    KNN8_PHASE(5);      // save for the second pass (and, for a lane whose group is done, the wait for the other groups of its wave)
    if (outer_it == 0 && v.knn_save_q && e < E && j == 0) {
      v.knn_save_q[(size_t)s * v.edge_cap + e] = make_float4(qx, qy, qz, d5);
      if (!active && v.knn_save_g) v.knn_save_g[(size_t)s * v.edge_cap + e] = 0.f;       // (no query: nothing to re-rank)
    }
    // ---- the five neighbours for k_line_gate (w: found flag, window index of NN0, NN1) ----
    if (e < E && j < 5) {
      const bool found = d5 < 1.0f && !exact;             // :324 (inf when < 5 candidates)
      const int mypos = j == 0 ? pos5[0] : j == 1 ? pos5[1] : j == 2 ? pos5[2] : j == 3 ? pos5[3] : pos5[4];
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
      if (found) m = sp[mypos];
      const int wprev = dpp_i32<DPP_ROW_SHR1>(__float_as_int(m.w)) - min_wi;      // lane j: the (current) window index of neighbour j - 1
      int w = j == 0 ? (found ? 1 : 0) : (j <= 2 ? (found ? wprev : -1) : 0);
      if (exact) {
        // record for k_knn8_exact (+ the query's number in the stream's list: that launch is a handful of workgroups): the query,
        // the flag, and an upper bound of its fifth-nearest distance (nothing at or beyond the 1.0 gate can matter)
        if (j == 0) {
          m = make_float4(qx, qy, qz, 0.f); w = 2;
          const int slot = atomicAdd(&v.knn8_cnt[s], 1);
          v.knn8_list[(size_t)s * v.edge_cap + slot] = e;
        }
        if (j == 1) m.x = d5 < 1.0f ? d5 : 1.0f;
      }
      v.knn_nn[((size_t)s * v.edge_cap + e) * 5 + j] = make_float4(m.x, m.y, m.z, __int_as_float(w));
    }
    KNN8_PHASE(6);      // neighbours fetched, records written
  }
}

// The queries k_knn8 could not certify (three of a query's nearest in one lane with a fourth at or below the fifth distance, equal
// distances among the six nearest — FLANN orders those by index — or LIODOM_KNN_EXACT_ONLY): sorted (distance, window index) lists
// per lane, started at the bound k_knn8 left, merged: FLANN's answer in every case.  ONE WAVE PER QUERY here: a dozen queries per
// stream and pass, and what the launch costs is the length of ONE search — 27 lanes probe the 27 cells, the wave's 64 lanes walk
// every cell found, the 64 lists are merged over the wave.  (With a group
// per query — 27 cells one after the other — the launch took 48 us; as a launch of k_knn8's shape that looked for flagged records,
// 57 us: 90 000 waves of two dependent loads each.)  The workgroups of a stream walk its list; k_line_gate, the next launch, empties it.
constexpr int kKnn8ExactBlocks = 4;
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  v = half_min_u64(v);
  const unsigned long long o = xor32_u64(v);
  return o < v ? o : v;
}
__global__ __launch_bounds__(kKnn8Threads) void k_knn8_exact(DevView v, int s0, int outer_it, int eb) {
  const int s = s0 + (int)blockIdx.y;
  StreamState& st = v.state[s];
  const int n = v.knn8_cnt[s];
  if (n <= 0 || !st.initialized) return;
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  constexpr int kWaves = kKnn8Threads / 64;
  const unsigned int tmask = st.table_mask;
  const int stab = s + LD_TAB_PARITY(v, st.frame_count) * v.n_streams;
  const CellSlot* cells = v.cells + (size_t)stab * v.table_size;
  const unsigned int* bits = v.cell_bits + (size_t)stab * (v.table_size >> 5);
  const float4* sp = v.sorted_pts + (size_t)stab * v.sorted_cap;
  const int min_wi = v.hash_incr ? st.hb_shift : 0;
  const int n_spill = v.hash_incr ? min(st.hb_spill, v.sorted_cap - v.hb_spill_base) : 0;
  for (int i = (int)blockIdx.x * kWaves + wave; i < n; i += (int)gridDim.x * kWaves) {      // (uniform over the wave)
    const int e = v.knn8_list[(size_t)s * v.edge_cap + i];
    float4* rec = v.knn_nn + ((size_t)s * v.edge_cap + e) * 5;
    const float4 r0 = rec[0];
    const float qx = r0.x, qy = r0.y, qz = r0.z;
    const float B = rec[1].x;
    const int cx = (int)floorf(qx * kCellInv), cy = (int)floorf(qy * kCellInv), cz = (int)floorf(qz * kCellInv);
    Top5Acc ta;
    {
      const unsigned long long sentinel = ((unsigned long long)(unsigned int)__float_as_int(B) << 32) | 0x7fffffffull;
      ta.t.k0 = ta.t.k1 = ta.t.k2 = ta.t.k3 = ta.t.k4 = sentinel;
      ta.t.p0 = ta.t.p1 = ta.t.p2 = ta.t.p3 = ta.t.p4 = -1;
    }
    // Lane l < 27 looks cell l of the block up (27 probes side by side: two round trips), lane 27 stands for k_hash_append's spill
    // list; then the WHOLE WAVE walks the cells found, one after the other, 64 points per round.  (Until late in round 6 every group of 8
    // lanes took four cells: a query whose own cell holds 300 points — where most uncertain results come from — kept 8 lanes busy
    // for 38 rounds while 56 idled, and the launch lasts as long as its slowest query.)
    unsigned int my_start = 0, my_cnt = 0;
    if (lane < 27) {
      int ox, oy, oz;
      const float lb = g8_cell_lb(lane, cx, cy, cz, qx, qy, qz, ox, oy, oz);
      if (!(lb > B)) g8_probe(v, cells, bits, tmask, ox, oy, oz, my_start, my_cnt);      // (a cell beyond the bound cannot hold one of the five)
    } else if (lane == 27) {
      my_start = (unsigned int)v.hb_spill_base; my_cnt = (unsigned int)n_spill;
    }
    unsigned long long pend = __ballot(my_cnt > 0);
    while (pend) {                                       // (uniform over the wave)
      const int l = (int)__builtin_ctzll(pend);
      pend &= pend - 1ull;
      const int cs = __builtin_amdgcn_readlane((int)my_start, l), cc = __builtin_amdgcn_readlane((int)my_cnt, l);
      g8_stream_cell<Top5Acc, 1, 64>(ta, sp, cs, cc, lane, qx, qy, qz, min_wi);
    }
    // merge of the 64 sorted lists: (distance, window index) keys are unique among real candidates; the sentinel (bound, INT_MAX)
    // may sit in several lanes: the lowest lane pops
    int gp[5];
    unsigned long long gk4 = 0ull;
#pragma unroll
    for (int r = 0; r < 5; r++) {
      const unsigned long long mk = wave_min_u64(ta.t.k0);
      const unsigned long long win = __ballot(ta.t.k0 == mk);
      const int wl = (int)__builtin_ctzll(win);
      gp[r] = __builtin_amdgcn_readlane(ta.t.p0, wl);
      if (r == 4) gk4 = mk;
      if (lane == wl) {   // pop
        ta.t.k0 = ta.t.k1; ta.t.k1 = ta.t.k2; ta.t.k2 = ta.t.k3; ta.t.k3 = ta.t.k4; ta.t.k4 = kTop5Empty;
        ta.t.p0 = ta.t.p1; ta.t.p1 = ta.t.p2; ta.t.p2 = ta.t.p3; ta.t.p3 = ta.t.p4; ta.t.p4 = -1;
      }
    }
    const bool found = gp[4] >= 0 && top5_dist(gk4) < 1.0f;       // (a sentinel among the five: fewer than five candidates inside the gate)
    if (lane < 5) {
      const int mypos = lane == 0 ? gp[0] : lane == 1 ? gp[1] : lane == 2 ? gp[2] : lane == 3 ? gp[3] : gp[4];
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
      if (found) m = sp[mypos];
      const int wprev = dpp_i32<DPP_ROW_SHR1>(__float_as_int(m.w)) - min_wi;
      const int w = lane == 0 ? (found ? 1 : 0) : (lane <= 2 ? (found ? wprev : -1) : 0);
      rec[lane] = make_float4(m.x, m.y, m.z, __int_as_float(w));
    }
  }
}
