// kernels_lm.h — k_lm_solve: the Ceres-style solve, scan finalisation, in-launch exchanges.
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, in this order; not a standalone header).
// =============================================================================================
// k_lm_solve: the whole Ceres-style solve of a stream in one launch, by G = lm_groups cooperating 512-thread workgroups
//   (G = 8 on the headline shape, 4 for >= 512 possible edges, 1 on lock-step batches and in safe mode; block indices 0, 8, 16, ...
//   so that the G workgroups share an XCD; every other block of the launch is a workgroup of the streamed rebuild, kernels_rebuild.h).
//   first evaluation: the reduction of the per-workgroup partial normal equations k_knn left (handles with knn_partials);
//   eval: every workgroup takes a contiguous share of the accepted correspondences; every thread accumulates the 29-entry
//   normal-equation accumulator over its blocks (fused residual + analytic Jacobian + Huber), transposed LDS reduction in a fixed
//   order (deterministic, no atomics, no MFMA: this is a 6x6 reduction); with G > 1 the 29 partial sums are exchanged inside the
//   launch as tagged 8-byte granules (lm_exchange) and every workgroup continues with bit-identical totals.
//   controller: lane 0 of the last wave (tid kLmCtl) runs lm_begin / lm_update (liodom_math.h) between evaluations, redundantly in
//   every workgroup; beside its first step the other waves compact the accepted correspondences and cache their triples in registers.
//   finalize (second outer iteration, or the very first frame; workgroup 0): pose log + host-mapped record, constant-velocity
//   prediction for the next scan, window bookkeeping, and the solved pose handed to the rebuild workgroups that append the frame.
//   Two instances, k_lm_solve<0> (first solve of a scan) and <1> (finalising solve): one kernel with a run-time outer iteration
//   spilled more with every addition to either side.  What a launch would otherwise fetch between the kNN pass it waits for and its
//   first step — start point, previous pose, frame count — is prefetched into LDS while it waits; the solved pose leaves straight
//   from LDS when the loop ends, before any bookkeeping.
//   Speculative hand-over (kernels_sync.h): the candidate's and the iterate's matrices live in two LDS buffers that swap roles when a
//   step is accepted; before an evaluation whose predicted cost change is below spec_theta x the function tolerance workgroup 0 hands
//   the ITERATE to whoever waits for the solve's result (first solve: the overlapped second pass; finalising solve in chain mode:
//   the appenders and — as the prediction formed from it by the controller's wave — the next scan's first pass), and publishes the
//   confirmation (copy / verdict) when the solve has ended.
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
  const unsigned int* mw = reinterpret_cast<const unsigned int*>(v.corr_mask + ((size_t)s * 2 + outer_it) * v.mask_stride);
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
#ifndef LIODOM_LM_CACHED
#define LIODOM_LM_CACHED 1
#endif
constexpr int kLmCached = LIODOM_LM_CACHED;
struct LmCache { float4 P[kLmCached], A[kLmCached], B[kLmCached]; };
__device__ __forceinline__ void lm_cache_load(const DevView& v, int s, int outer_it, int eb, int c_lo, int c_hi, const int* idx, LmCache& k) {
  const float4* ed = v.edges + ((size_t)eb * v.n_streams + s) * v.edge_cap;
  const float4* ca = v.corr_a + ((size_t)s * 2 + outer_it) * v.edge_cap;
  const float4* cb = v.corr_b + ((size_t)s * 2 + outer_it) * v.edge_cap;
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
__device__ __forceinline__ void lm_eval(const DevView& v, int s, int outer_it, int eb, int c_lo, int c_hi, const int* idx, const double* Rm_sh,
                                        double* part, double* acc_out /*[kAccN]*/, const LmCache& k) {
  const int et = (int)threadIdx.x;
#ifdef LIODOM_LM_NOCACHE      // (debugging: every evaluation reads its triples from memory)
  const bool cached = false;
#else
  const bool cached = et < kLmCtl;
#endif
  double acc[kAccN];
#pragma unroll
  for (int i = 0; i < kAccN; i++) acc[i] = 0.0;
  {
    double Rm[12];
#pragma unroll
    for (int i = 0; i < 12; i++) Rm[i] = Rm_sh[i];
    const float4* ed = v.edges + ((size_t)eb * v.n_streams + s) * v.edge_cap;
    const float4* ca = v.corr_a + ((size_t)s * 2 + outer_it) * v.edge_cap;
    const float4* cb = v.corr_b + ((size_t)s * 2 + outer_it) * v.edge_cap;
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
    const int r = threadIdx.x & 15;
#pragma unroll
    for (int vrow = threadIdx.x >> 4; vrow < ((kAccN + kLmThreads / 16 - 1) / (kLmThreads / 16)) * (kLmThreads / 16); vrow += kLmThreads / 16) {
      double x = 0.0;
      if (vrow < kAccN) {
        const double* rowp = part + vrow * kLmEvalThreads + r;
        const int nk = ncol >> 4;
#pragma unroll 8
        for (int kk = 0; kk < nk; kk++) x += rowp[kk * 16];
      }
      x = row_sum_f64(x);
      if (vrow < kAccN && r == 0) acc_out[vrow] = x;
    }
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
  if (nup > v.used_cap) nup = v.used_cap;      // (a scan that failed half-way may have left more cells counted than listed)
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

__device__ __forceinline__ void publish_final_pose(const DevView& v, int s, const double* T, int raw, unsigned int tag, int lane, int copy = 0);

// The prediction for the next scan: odom * (prev^-1 * odom) (:148-150), its quaternion (:186-190 q_curr(odom_.rotation())) and
// translation (:192-195): out19 = matrix [12], quaternion [4], translation [3].  One thread.
__device__ __forceinline__ void predict_next(const double* T, const double* prev, int rotation_mode, double* out19) {
  double fin[12], po[12], inv[12], rel[12], pred[12], q[4];
  for (int i = 0; i < 12; i++) { fin[i] = T[i]; po[i] = prev[i]; }
  iso_inverse(po, inv);
  iso_mul(inv, fin, rel);
  iso_mul(fin, rel, pred);
  quat_from_pose(pred, rotation_mode, q);
  for (int i = 0; i < 12; i++) out19[i] = pred[i];
  for (int i = 0; i < 4; i++) out19[12 + i] = q[i];
  out19[16] = pred[3]; out19[17] = pred[7]; out19[18] = pred[11];
}

// Called by the whole workgroup.  sh_cnt: LDS scratch of kMaxFrames + 1 ints.
// T (LDS): the scan's final pose; raw: the frame enters the window untransformed (first frame); publish_pose: the appenders have
// not been handed T yet (the solve's path publishes it itself, straight after its last step: they have been waiting for it, and the
// next scan's first kNN pass follows them in stream order); prev (LDS or nullptr): copy of st.prev_odom taken when the launch
// started (the finalising solve prefetches it while it waits for the second pass).
// Order: what others wait for leaves first — the pose, then the prediction for the next scan (thread `ctl`, beside the other threads'
// frame-size loads) — and only then the pose log, the host-mapped record and the window bookkeeping.
// chain: the rebuild of this scan runs on the other HIP stream, beside this launch (ALLOC may still be allocating cell ranges from
// st.cursor): the cursor is then reset by the next scan's first kNN pass, which follows the rebuild in stream order.
__device__ __forceinline__ void finalize_scan(const DevView& v, int s, StreamState& st, int* sh_cnt, int eb, bool clear_hash, int ctl, int chain,
                              const double* T, int raw, bool publish_pose, const double* prev, int fc_old, int pred_copies = 3, unsigned int verdict = 1u,
                              const liodom_lm_trace_t* trace1 = nullptr) {
  __shared__ double sh_pred[19];      // the prediction: matrix [12], quaternion [4], translation [3]
  const int P = v.prev_frames;
  const int tid = threadIdx.x;
  // LocalMapManager::addPointCloud (:34-60) on a ring of P frame slots: the new frame goes
  // into slot frame_count % P (overwriting the oldest once the window is full)
  const int fc_new = fc_old + 1;
  const int nf = fc_new < P ? fc_new : P;
  const int new_slot = fc_old % P;
  int* wn = v.win_n + (size_t)s * P;
  int* wb = v.win_base + (size_t)s * (P + 1);
  int* ws = v.win_slot + (size_t)s * P;
  // early_rebuild: hand the pose to the workgroups that append the new frame (they have been waiting for it)
  if (publish_pose && v.early_rebuild && tid < 50) publish_final_pose(v, s, T, raw, (unsigned int)fc_old + 1u, tid % 25, tid / 25);
  const int n_edges = st.n_edges_buf[eb];
  const int nup = st.n_used_tab[0];      // cells of the build that this scan searched (cleared below)
  for (int j = tid; j < nf; j += blockDim.x) {           // frame sizes of the new window (nothing here depends on the pose)
    const int sl = (fc_new - nf + j) % P;
    sh_cnt[j] = (sl == new_slot) ? n_edges : wn[sl];     // independent loads, one round trip
    ws[j] = sl;
  }
  // (the pose log / host-mapped record by thread 64, beside thread `ctl`'s prediction)
  if (tid == 64) {
    // pose as published (laser_odometry.cc:403-412 with identity laser_to_base)
    double fin[12], q[4];
    for (int i = 0; i < 12; i++) fin[i] = T[i];
    quat_from_pose(fin, v.rotation_mode, q);             // :403 q_current(odom_base_link.rotation())
    const int k = st.scan_counter;
    st.info.scan_index = k;
    st.info.status = st.status;
    // trace1 (LDS): the finalising solve's trace.  The controller's thread wrote it before the barrier this call follows — as stores
    // to st.info by that thread, issued just before the call, they raced with the copies below
    if (trace1) st.info.lm[1] = *trace1;
    if (k < v.pose_log_cap) {
      double* pl = v.pose_log + ((size_t)s * v.pose_log_cap + k) * 7;
      pl[0] = q[0]; pl[1] = q[1]; pl[2] = q[2]; pl[3] = q[3];
      pl[4] = fin[3]; pl[5] = fin[7]; pl[6] = fin[11];
      v.info_log[(size_t)s * v.pose_log_cap + k] = st.info;
    }
    st.scan_counter = k + 1;
    if (v.host_out) {
      // zero-copy publication: payload, system-scope fence, then the sequence word the host polls
      HostOut* ho = v.host_out + (size_t)s * 2 + (k & 1);      // two records per stream: the host may read scan k while scan k + 1 publishes
      ho->pose[0] = q[0]; ho->pose[1] = q[1]; ho->pose[2] = q[2]; ho->pose[3] = q[3];
      ho->pose[4] = fin[3]; ho->pose[5] = fin[7]; ho->pose[6] = fin[11];
      ho->info = st.info;
      ho->pad = (int)verdict;      // (speculative hand-over: 1 = confirmed, the host then skips the repair launch when it synchronises)
      // (the system-scope release orders this thread's payload stores before the sequence word: no separate fence)
      __hip_atomic_store(&ho->seq, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    st.info.matches[0] = 0; st.info.matches[1] = 0;   // counters of the next scan's two kNN passes
  }
  if (tid == ctl) {
    double po[12];
    for (int i = 0; i < 12; i++) po[i] = prev ? prev[i] : st.prev_odom[i];
    predict_next(T, po, v.rotation_mode, sh_pred);
  }
  __syncthreads();
  // chain mode: the next scan's first kNN pass runs on the other HIP stream and may start before this launch has ended (it follows
  // the APPEND launch): the prediction it starts from travels as tagged granules (tag = scans completed); pred_copies / verdict:
  // speculative hand-over (kernels_sync.h) — the copy the pass starts from may have left before the solve's last evaluation
  if (v.pred_xch) { pred_publish(v, s, sh_pred, (unsigned int)fc_new, tid, (int)blockDim.x, pred_copies); pred_verdict_publish(v, s, (unsigned int)fc_new, verdict, tid); }
  if (tid == ctl) {
    for (int i = 0; i < 12; i++) { const double f = T[i]; st.final_odom[i] = f; st.prev_odom[i] = f; st.odom[i] = sh_pred[i]; }
    for (int i = 0; i < 4; i++) st.param_q[i] = sh_pred[12 + i];
    st.param_t[0] = sh_pred[16]; st.param_t[1] = sh_pred[17]; st.param_t[2] = sh_pred[18];   // :192-195
    wn[new_slot] = n_edges;
    st.frame_count = fc_new;
    st.n_frames = nf;
    int acc = 0;
    for (int j = 0; j < nf; j++) { const int c = sh_cnt[j]; sh_cnt[j] = acc; acc += c; }
    sh_cnt[nf] = acc;
    st.n_map = acc;
    if (!v.early_rebuild) st.n_used_tab[0] = 0;
    else { st.n_search = acc; st.n_filt = 0; }        // (k_window_insert's job otherwise)
    if (!chain) st.cursor = 0;
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
  INJECT_DELAY(16);
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
    unsigned long long t0w = 0;
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
      if (++spins > 4000000u || __any(wait_expired(spins, t0w))) { if (tid == 0) atomicOr(status, LIODOM_STATUS_LM_SYNC_TIMEOUT); same = false; break; }
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
// copy 0: what the appending workgroups start from; copy 1: what the solve ended with (speculative hand-over: the repair reads both)
__device__ __forceinline__ void publish_final_pose(const DevView& v, int s, const double* T, int raw, unsigned int tag, int lane /*0..24*/, int copy) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  unsigned long long* base = v.pose_xch + (size_t)s * 64 + copy * 32;
  INJECT_DELAY(15);
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

__device__ __forceinline__ void rebuild_beside_solve(const DevView& v, int s, StreamState& st, int eb, int outer_it, int block, int nblocks, unsigned int seq, int* sbase, int* sslot);

// chain != 0 (chain mode, kernels_sync.h; done_target: the first pass's done count to wait for): the launch holds the solving workgroups only (the rebuild rides on the other stream as
// launches of its own), and the FIRST solve's launch is resident while the first kNN pass still runs: it waits for that pass's
// done flags before it touches anything the pass or the extraction wrote.
// (one instance per solve of the scan: the first solve's code holds neither the scan's finalisation nor the appending workgroups,
//  the finalising solve's neither COUNT / PAD nor the hand-over to the second pass — as one kernel with a run-time outer_it the
//  controller's steps spilled more with every addition to either side)
// (round 6: and one instance per MODE — the four-launch chain of the host-fed replay, the strict / serial legs, the lock-step batches
//  and per-kernel profiling never waits for a pass's done count and never hands a result over early: as a run-time `chain` that code
//  cost its finalising solve 200 B of scratch per lane)
template <int kOuterIt, bool kChainMode>
#ifndef LIODOM_LM_WAVES_PER_SIMD
#define LIODOM_LM_WAVES_PER_SIMD 1      // (experiments: 256-thread workgroups at 2 -> 256 registers per lane, half of the CU's register file)
#endif
__global__ __launch_bounds__(kLmThreads, LIODOM_LM_WAVES_PER_SIMD) void k_lm_solve(DevView v, int s0, int eb, unsigned int seq, unsigned int done_target) {
  constexpr int outer_it = kOuterIt;
  constexpr int chain = kChainMode ? 1 : 0;
  // the candidate's matrix and the current iterate's (the candidate of the last accepted step): two buffers that swap roles when a
  // step is accepted, so that the iterate's matrix is at hand — bit for bit the one the solve ends with unless another step is
  // accepted — without being formed again (speculative hand-over, kernels_sync.h; and the launch's last microsecond)
  __shared__ double sh_pose2[2][12];
  __shared__ int sh_pi, sh_ci;           // buffer of the candidate / of the iterate (-1: the start point, no matrix)
  __shared__ int sh_nmoved;              // how often the iterate has moved (a hand-over is confirmed iff it has not moved since)
  __shared__ double sh_pred_s[20];       // finalising solve, speculative hand-over: the prediction formed from the iterate
  __shared__ int sh_unapplied;           // the controller's last step left the iterate where it was
  __shared__ double sh_acc[kAccN];
  __shared__ LmState lm;
  __shared__ int sh_flag;
  __shared__ int sh_C;
  const int s = s0 + blockIdx.y;
#if defined(LIODOM_CHAIN_PRIO)
  if (v.n_streams <= 4) __builtin_amdgcn_s_setprio(LIODOM_CHAIN_PRIO);      // (see k_knn)
#endif
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
    if (!v.early_rebuild || chain) return;         // (filler blocks between the solvers)
    // early_rebuild: the workgroups behind the solve build the next cell hash (see "streamed rebuild" below)
    __shared__ int sh_slot[kMaxFrames];
    rebuild_beside_solve(v, s, st, eb, outer_it, g - G, (int)gridDim.x - G, seq, sh_cnt, sh_slot);
    return;
  }
  __shared__ double sh_loc[kAccN];
  __shared__ double sh_red[16][32];
  // State that earlier launches on THIS stream wrote (the start point: the previous scan's finalising solve, or this scan's first
  // solve; the previous pose and the frame count: the previous scan's finalising solve) is fetched now — in chain mode and beside an
  // overlapped second pass the launch is about to wait for a kNN pass, and each of these loads would otherwise be a memory round
  // trip between that pass's last workgroup and the first trust-region step, or between the last step and the pose's publication.
  __shared__ double sh_x0[8];          // start point: quaternion [4], translation [3]
  __shared__ double sh_prev[12];       // st.prev_odom (finalising solve)
  __shared__ int sh_fc;                // st.frame_count (finalising solve)
  if (threadIdx.x < 4) sh_x0[threadIdx.x] = st.param_q[threadIdx.x];
  else if (threadIdx.x < 7) sh_x0[threadIdx.x] = st.param_t[threadIdx.x - 4];
  else if (threadIdx.x >= 64 && threadIdx.x < 76) sh_prev[threadIdx.x - 64] = st.prev_odom[threadIdx.x - 64];
  else if (threadIdx.x == 76) sh_fc = st.frame_count;
  __shared__ int sh_spec_at;           // speculative hand-over: solves of this kind still to sit out after one that was not confirmed
  if (threadIdx.x == 77) sh_spec_at = st.spec_eval[outer_it];
  // The kNN pass this solve consumes was REPEATED (speculative hand-over not confirmed: rare) — by k_knn_redo / k_chain_redo0, whose
  // workgroups do not sit where their namesakes of the pass's first edition sat.  This launch has been resident since before either
  // wrote: what a first-edition workgroup on THIS XCD stored (write-through, but the line stays in this L2) is what a plain load here
  // still finds after the repeat has overwritten it in memory from another XCD — "nothing of theirs cached here" (ov_wait_knn_done)
  // no longer holds.  Then, and only then, the solve invalidates its caches behind the wait.  (Found with 256-thread solving
  // workgroups on the 16 x 900 shape, predictor forced wrong: one repair in four left a solve with the first edition's partial sums
  // or correspondences — poses off by 1e-8; with 384 / 512 threads the first edition had always finished before this launch began.)
  __shared__ int sh_redo;
  if (threadIdx.x == 78) sh_redo = st.spec_redo[outer_it == 0 ? 1 : 0];
  extern __shared__ __attribute__((aligned(16))) int sh_idx[];   // [edge_cap] compacted correspondence indices, then the reduction matrix
  double* sh_part = reinterpret_cast<double*>(sh_idx + ((v.edge_cap + 3) & ~3));   // [kAccN][kLmEvalThreads]
  const int tid = threadIdx.x;
  const bool prep = tid < kLmCtl;               // the other waves: compaction + register cache while the controller lane works
  // chain mode, first solve: nothing of the stream's state that the extraction writes (n_edges_buf) and nothing of the first pass's
  // results has been read so far; the pass's workgroups store write-through and raise one flag each
  // (dead: a wait of this scan gave up — here or in the pass: nothing the pass was to leave may be consumed; the solve then runs as
  //  one without residual blocks, the pose stays the prediction, and the status bit fails the scan on the host)
  bool dead = false;
  if (chain && outer_it == 0) dead = !chain_wait_count(v.knn_done0 + s, done_target, &st.status);      // (the count the pass's workgroups reach: wrap-safe comparison)
  if (chain && outer_it == 0 && sh_redo) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (uniform; sh_redo: visible behind the wait's barrier)
  // (the edge count: the state word shares a cache line with fields that earlier launches on this XCD have read — since this launch
  //  started, possibly before the extraction wrote it; k_compact_edges leaves a write-through copy in a line of its own)
  int E_in;
  if (chain && outer_it == 0) {
    typedef __attribute__((address_space(1))) unsigned int gu32;
    E_in = (int)__hip_atomic_load((gu32*)(v.edge_cnt + eb * 32 + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    E_in = st.n_edges_buf[eb];
  }
  if (outer_it == 0 && tid == 0 && g == 0) {     // per-scan diagnostics (matches are counted by k_knn)
    st.info.n_edges = E_in;
    st.info.map_points = st.n_search;
    for (int k = 0; k < 2; k++) {
      st.info.lm[k].iterations = 0; st.info.lm[k].accepted = 0; st.info.lm[k].termination = LM_TERM_NO_RESIDUALS;
      st.info.lm[k].pad = 0; st.info.lm[k].initial_cost = 0.0; st.info.lm[k].final_cost = 0.0;
    }
  }
  if (!st.initialized) {
    // first frame (:108-136): no solve; pose stays identity, edges enter the window raw
    if (outer_it == 1 && g == 0) {
      // (the host never overlaps the second kNN pass of an uninitialised stream — reset_state() — but if it did, that pass must
      //  have read `initialized` before it changes: a workgroup that saw 1 would wait for a pose this branch never publishes)
      if (seq) ov_wait_knn_done(v, s, seq, &st.status);
      if (tid == 0) st.append_raw = 1;
      __shared__ double sh_T0[12];
      if (tid < 12) sh_T0[tid] = st.odom[tid];
      __syncthreads();                       // (also: the prefetched sh_prev / sh_fc)
      finalize_scan(v, s, st, sh_cnt, eb, true, 0, 0, sh_T0, 1, true, sh_prev, sh_fc);
      if (tid == 0) st.initialized = 1;
    }
    return;
  }
  // The second kNN pass of this scan has completed when the finalising solve starts, so the cell hash it
  // searched is no longer needed: the other waves reset its occupied slots while the controller lane works on its
  // first update step (they would idle at the barrier otherwise).
  bool clr_pending = outer_it == 1 && g == 0 && prep && !v.early_rebuild;
  auto clear_hash_slots = [&]() {
    hash_clear_used(v, s, st.n_used_tab[0], tid, kLmCtl);
    clr_pending = false;
  };
  const bool dbgb = (s == 0) && (g == 0) && (tid == kLmCtl) && (outer_it == 1);
  const bool dbge = (s == 0) && (g == 0) && (tid == 0) && (outer_it == 1);
  DBG_STAMP(v, dbgb, 2, 0);
  const int E = E_in;
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
  // the overlapped second kNN pass has completed (chain mode: its workgroups count themselves on one word; else a flag each)
  if (seq && outer_it == 1) { if (chain) dead = !chain_wait_count(v.knn_done0 + 32 + s, done_target, &st.status) || dead; else ov_wait_knn_done(v, s, seq, &st.status); }
  if (seq && outer_it == 1) { __syncthreads(); if (sh_redo) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
  // (a pass whose own wait gave up does not count its workgroups: this solve's wait then gives up too)
  OV_STAMP(v, g == 0 && tid == 0 && outer_it == 1, 4);
  OV_STAMP(v, g == 0 && tid == 0 && outer_it == 0, 18);
  if (v.knn_partials) {
    // ---- first evaluation = sum of the partial normal equations the k_knn workgroups left, in workgroup order ----
    const int Q = v.knn_queries;
    const int nb = (E + Q - 1) / Q;
    const double* part = v.knn_part + ((size_t)s * 2 + outer_it) * v.knn_blocks * 32;
    constexpr int kRC = kLmThreads / 32;                 // row classes (8 at 256 threads) x 32 columns (29 used)
    const int i = tid & 31, r0 = tid >> 5;
    double x0 = 0.0, x1 = 0.0;
    if (i <= kAccN) {                                    // (entry 29: the number of accepted correspondences)
      // 16 independent loads in flight per pass (one memory round trip for up to 16 kRC k_knn workgroups).  The first pass's loads
      // do not wait for the edge count (itself a load that has just been issued): rows up to the table's size are fetched and the
      // ones at or beyond ceil(E / Q) — left by earlier scans — are dropped when the count has arrived.
      const int nb_cap = v.knn_blocks;
      for (int rb = r0; rb < (dead ? 0 : (rb == r0 ? nb_cap : nb)); rb += 16 * kRC) {
        double xs[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { const int r = rb + kRC * u; xs[u] = (r < nb_cap) ? part[(size_t)r * 32 + i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; u++) { const int r = rb + kRC * u; if (r >= nb) xs[u] = 0.0; }
#pragma unroll
        for (int u = 0; u < 16; u += 2) { x0 += xs[u]; x1 += xs[u + 1]; }
      }
    }
    sh_red[r0][i] = x0 + x1;
    __syncthreads();
    if (tid <= kAccN) {
      double x = 0.0;
#pragma unroll
      for (int r = 0; r < kRC; r++) x += sh_red[r][tid];
      if (tid < kAccN) sh_acc[tid] = x;
      else { sh_nmatch = (int)x; if (g == 0) st.info.matches[outer_it] = (int)x; }      // :346 (sum of small integers: exact)
    }
    __syncthreads();
    nblocks = sh_nmatch;
  } else {
    // ---- lock-step batches: k_knn leaves only the validity bytes; compaction, then an ordinary first evaluation ----
    __syncthreads();                                     // (the prefetched start point)
    if (prep) {
      const int C = lm_compact_bits(v, s, outer_it, E, sh_idx);
      if (tid == 0) sh_C = C;
    } else if (tid == kLmCtl) {
      iso_from_qt(sh_x0, sh_x0 + 4, sh_pose2[0]);
    }
    __syncthreads();
    my_share(sh_C);
    lm_cache_load(v, s, outer_it, eb, c_lo, c_hi, sh_idx, cache);
    if (G > 1) { lm_eval(v, s, outer_it, eb, c_lo, c_hi, sh_idx, sh_pose2[0], sh_part, sh_loc, cache); lm_exchange(v, s, g, G, epoch0 | ++n_eval, sh_loc, sh_acc, &st.status, xch_local, &sh_same_xcc); xch_local = sh_same_xcc != 0; }
    else { lm_eval(v, s, outer_it, eb, c_lo, c_hi, sh_idx, sh_pose2[0], sh_part, sh_acc, cache); ++n_eval; }
  }
  DBG_STAMP(v, dbgb, 2, 2);
  // ---- trust-region loop.  Controller step on lane 0 of the last wave; beside it the other waves prepare the
  // evaluations (step 0: validity bytes -> index list, triples into registers) or reset the hash slots of the
  // build this scan searched (step 1 of the finalising solve) ----
  int dbg_it = 0;
  int pi = 0, ci = -1, nmoved = 0;     // (controller lane) candidate / iterate buffer, moves of the iterate
  const unsigned int n_eval0 = n_eval;       // evaluations of this launch before the loop (lock-step batches: the one at the start point)
  const bool spec_ok = v.speculate != 0 && v.speculate != 5 && v.speculate != 7 && g == 0 && seq != 0u && outer_it == 0;      // (debugging: 4 / 5 only the first / only the finalising solve; 6 / 7 the same with the predictor forced wrong, as 2 forces both)
  // ... and of the finalising solve's, in chain mode: to the workgroups that append the new frame (the pose) and to the next scan's
  // first kNN pass, which follows them on their stream (the prediction formed from it)
  const bool spec_ok1 = v.speculate != 0 && v.speculate != 4 && v.speculate != 6 && g == 0 && chain != 0 && outer_it == 1 && v.early_rebuild && v.pred_xch != nullptr;
  const bool spec_forced = v.speculate == 2 || v.speculate == 6 || v.speculate == 7;
  bool spec_done = false;
  int spec_moves = -1;
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
      // 3.9-5.9 us per step against 3.1; DESIGN_HISTORY.md)
      const int acc_before = step == 0 ? 0 : lm.accepted;
      const int f = step == 0 ? lm_begin(lm, sh_x0, sh_x0 + 4, sh_acc, nblocks, v.apply_on_ftol, sh_scale) : lm_update(lm, sh_acc);
      sh_flag = f;
      bool moved = step > 0 && lm.accepted != acc_before;
      if (step > 0 && !moved && f != LM_NEED_EVAL && lm.termination == LM_TERM_FUNC_TOL) {      // (apply_on_ftol: the step was applied without counting as accepted)
        moved = true;
        for (int k = 0; k < 4; k++) moved = moved && lm.q[k] == lm.cand_q[k];
        for (int k = 0; k < 3; k++) moved = moved && lm.t[k] == lm.cand_t[k];
      }
      if (moved) { ci = pi; pi ^= 1; nmoved++; }             // the candidate became the iterate: its matrix stays where it is
      if (f == LM_NEED_EVAL) iso_from_qt(lm.cand_q, lm.cand_t, sh_pose2[pi]);
      sh_pi = pi; sh_ci = ci; sh_nmoved = nmoved; sh_unapplied = (step > 0 && !moved) ? 1 : 0;
      if (step == 0) DBG_STAMP(v, dbgb, 2, 23);
    } else if (!prep) {
      // (the other lanes of the controller's wave wait at the barrier)
    } else if (step == 0 && v.knn_partials) {
      DBG_STAMP(v, dbge, 2, 24);
      const int C = dead ? 0 : lm_compact_bits(v, s, outer_it, E, sh_idx);
      if (tid == 0) sh_C = C;
      DBG_STAMP(v, dbge, 2, 25);
      my_share(C);
      lm_cache_load(v, s, outer_it, eb, c_lo, c_hi, sh_idx, cache);
    } else if (clr_pending) {
      clear_hash_slots();
    }
    __syncthreads();
    if (step == 0) { if (v.knn_partials && !prep) my_share(sh_C); DBG_STAMP(v, dbgb, 2, 3); }
    else { DBG_STAMP(v, dbgb && dbg_it < 5, 2, 5 + 2 * dbg_it); dbg_it++; }
    if (sh_flag != LM_NEED_EVAL) break;
    // speculative hand-over (kernels_sync.h): the iterate leaves for the waiting second pass before the evaluation that — going by
    // the previous scan — will end this solve without moving it
    if ((spec_ok || spec_ok1) && !spec_done && sh_ci >= 0 &&
        (spec_forced || (!spec_forced && sh_spec_at == 0 && lm.model_cost_change <= v.spec_theta * 1e-6 * lm.cost))) {
      if (spec_ok) {
        ov_publish_pose(v, s, sh_pose2[sh_ci], &lm.q[0], seq, tid, 1);
        OV_STAMP(v, tid == 0, 19);
      } else if (tid >= kLmCtl) {
        // the controller's wave has no residual blocks (they go to the lowest threads): it hands the pose to the appenders, forms
        // the prediction (one lane, ~0.6 us: less than the evaluation the other waves start meanwhile) and publishes it
        const int l = tid - kLmCtl;
        if (l < 25) publish_final_pose(v, s, sh_pose2[sh_ci], 0, (unsigned int)sh_fc + 1u, l, 0);
        if (l == 0) predict_next(sh_pose2[sh_ci], sh_prev, v.rotation_mode, sh_pred_s);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        pred_publish(v, s, sh_pred_s, (unsigned int)sh_fc + 1u, l, 64, 1);
        OV_STAMP(v, l == 0, 22);
      }
      if ((kInstrument && (v.debug & 64)) && tid == 0) atomicAdd(&v.dbg_clk[270], 1ull);      // (debug) iterates handed over early
      spec_done = true;
      spec_moves = sh_nmoved;
    }
    const double* pose_c = sh_pose2[sh_pi];
    if (G > 1) { lm_eval(v, s, outer_it, eb, c_lo, c_hi, sh_idx, pose_c, sh_part, sh_loc, cache); DBG_STAMP(v, dbgb && dbg_it < 4, 2, 12 + dbg_it); lm_exchange(v, s, g, G, epoch0 | ++n_eval, sh_loc, sh_acc, &st.status, xch_local, &sh_same_xcc); xch_local = sh_same_xcc != 0; }
    else { lm_eval(v, s, outer_it, eb, c_lo, c_hi, sh_idx, pose_c, sh_part, sh_acc, cache); ++n_eval; }
    DBG_STAMP(v, dbgb && dbg_it < 5, 2, 4 + 2 * dbg_it);
  }
  if (clr_pending) clear_hash_slots();                   // (the solve ended at its first step)
  DBG_STAMP(v, dbgb, 2, 20);
  if ((kInstrument && (v.debug & 32)) && dbgb) v.dbg_clk[2 * 32 + 27] = (xch_local ? 100ull : 0ull) + 10ull * xcc_id() + (unsigned long long)n_eval;   // (debug) exchange transport, XCC, evaluations
  if (g != 0) return;        // every workgroup reached the same result; workgroup 0 records it
  // What others are waiting for leaves first: the solved pose — to the overlapped second kNN pass (first solve: matrix, quaternion,
  // translation) or to the workgroups that append the new frame (finalising solve: the matrix) — straight from LDS; the stream's
  // state, the trace and the scan's bookkeeping follow.
  __shared__ double sh_ov[20];
  __shared__ liodom_lm_trace_t sh_trace;      // the solve's trace, from the controller's thread to whoever records it (finalize_scan: thread 64)
  {
    if (tid == kLmCtl) {
      sh_trace.iterations = lm.iter; sh_trace.accepted = lm.accepted; sh_trace.termination = lm.termination; sh_trace.pad = 0;
      sh_trace.initial_cost = lm.initial_cost; sh_trace.final_cost = lm.cost;
    }
    const int cif = sh_ci;
    if (cif >= 0) {                                                  // the iterate's matrix is at hand (:222-227)
      if (tid < 12) sh_ov[tid] = sh_pose2[cif][tid];
      else if (tid >= 64 && tid < 71) sh_ov[12 + tid - 64] = (&lm.q[0])[tid - 64];
    } else if (tid == kLmCtl) {                                      // no step was applied: the start point's
      double q[4], t[3], T[12];
      for (int k = 0; k < 4; k++) q[k] = lm.q[k];
      for (int k = 0; k < 3; k++) t[k] = lm.t[k];
      iso_from_qt(q, t, T);
      for (int k = 0; k < 12; k++) sh_ov[k] = T[k];
      for (int k = 0; k < 4; k++) sh_ov[12 + k] = q[k];
      for (int k = 0; k < 3; k++) sh_ov[16 + k] = t[k];
    }
  }
  __syncthreads();
  if (outer_it == 0) {
    // (the confirmation copy always; the copy the pass starts from unless the iterate left early)
    if (seq) { ov_publish_pose(v, s, sh_ov, sh_ov + 12, seq, tid, spec_done ? 2 : 3); OV_STAMP(v, tid == 0, 1); }
  } else if (v.early_rebuild && tid < 50) {
    // (the confirmation copy always; the copy the appenders start from unless the iterate left early)
    if (tid >= 25 || !spec_done) publish_final_pose(v, s, sh_ov, 0, (unsigned int)sh_fc + 1u, tid % 25, tid / 25);
  }
  if (tid == kLmCtl) {
    if (outer_it == 0) {       // (the finalising solve's finalize_scan leaves the prediction for the next scan there instead)
      for (int k = 0; k < 4; k++) st.param_q[k] = sh_ov[12 + k];
      for (int k = 0; k < 3; k++) st.param_t[k] = sh_ov[16 + k];
      for (int k = 0; k < 12; k++) st.odom[k] = sh_ov[k];
    }
    // speculative hand-over, back-off: a hand-over that was not confirmed costs the receivers a repeated pass (10-25 us); where
    // the model's prediction fails once it tends to fail again (the first scans of a young window converge more slowly): the next
    // spec_backoff (16) solves of this kind hand nothing over early
    st.spec_eval[outer_it] = (spec_done && sh_nmoved != spec_moves) ? v.spec_backoff : (sh_spec_at > 0 ? sh_spec_at - 1 : 0);
    st.spec_redo[outer_it] = (spec_done && sh_nmoved != spec_moves) ? 1 : 0;
    if (spec_done) { st.spec_stats[2 * outer_it] += 1; if (sh_nmoved != spec_moves) st.spec_stats[2 * outer_it + 1] += 1; }
    if (outer_it == 0) st.info.lm[0] = sh_trace;      // (read by the finalising solve's launch; the finalising solve's own trace travels through LDS)
    if (outer_it == 1) st.append_raw = 0;
  }
  DBG_STAMP(v, dbgb, 2, 21);
  if (outer_it == 1) {
    finalize_scan(v, s, st, sh_cnt, eb, false, kLmCtl, chain, sh_ov, 0, false, sh_prev, sh_fc, spec_done ? 2 : 3, (!spec_done || sh_nmoved == spec_moves) ? 1u : 2u, &sh_trace);
    DBG_STAMP(v, dbgb, 2, 22);
  }
  DBG_STAMP(v, dbgb, 2, 28);
  OV_STAMP(v, tid == 0, outer_it == 0 ? 2 : 5);
}
