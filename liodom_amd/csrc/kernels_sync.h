// kernels_sync.h — dependencies between the HIP streams of a handle as flags in device memory: pipelined replay (pipe_wait, k_pipe_gate, k_set_flag) and the overlapped second kNN pass (tagged pose granules, done flags, write-through stores).
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, in this order; not a standalone header).
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
// poll intervals (s_sleep units of 64 cycles) of the overlapped pass's two waits: the pass's workgroups for the first solve's
// pose, the finalising solve's threads for the pass's done flags
#ifndef LIODOM_POLL_POSE
#define LIODOM_POLL_POSE 6
#endif
#ifndef LIODOM_POLL_DONE
#define LIODOM_POLL_DONE 4
#endif
__device__ __forceinline__ bool pipe_wait(const unsigned int* flag, unsigned int want, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  __shared__ int s_pipe_ok;
  if (threadIdx.x == 0) {
    unsigned int spins = 0;
    unsigned long long t0 = 0;
    bool ok = true;
    while ((int)(__hip_atomic_load((gu32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > 1500000u || wait_expired(spins, t0)) { atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); ok = false; break; }
    }
    if (spins) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_pipe_ok = ok ? 1 : 0;
    INJECT_DELAY(1);
  }
  __syncthreads();
  return s_pipe_ok != 0;      // false: the producer never arrived — the caller must not touch the buffer (it returns)
}
// Gate in front of a scan's first k_knn launch for handles whose launch is too large to poll the flag itself (its polling
// workgroups would fill the GPU and starve the extraction they wait for): one wave waits for the extraction's flag and
// publishes that the previous odometry has completed; the launches behind it start when it retires.
__global__ void k_pipe_gate(DevView v, int s0, int eb, unsigned int wait_edges, unsigned int signal_odo) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  INJECT_DELAY(2);
  if (signal_odo && threadIdx.x == 0) __hip_atomic_store((gu32*)(v.pipe_flags + kEdgePipeBufs), signal_odo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wait_edges && !pipe_wait(v.pipe_flags + eb, wait_edges, &v.state[s0].status)) {
    // the launches behind the gate check the status bit of their own stream (k_knn) and skip the scan
    for (int s = (int)threadIdx.x; s < v.n_streams; s += (int)blockDim.x) atomicOr(&v.state[s].status, LIODOM_STATUS_PIPE_TIMEOUT);
  }
}
// =============================================================================================
// Chain mode (round 5; one-stream handles with the streamed rebuild and flags, scans whose edges arrive by flag from the
// extraction stream — the pipelined replay and the ticket API).  The scan's launches are split over two HIP streams by ROLE:
//   stream_k:  kNN(0) | gate | kNN(1) + COUNT + PAD | ALLOC | APPEND + CLEAR + SCATTER            (light kernels; with the speculative
//              hand-overs below the gate is the last workgroup of k_chain_redo0 and ALLOC the first workgroups of k_knn_redo)
//   stream:    solve(0) | solve(1)                                                                 (the solving workgroups alone)
// * solve(0) of scan k follows solve(1) of scan k-1 in stream order, i.e. its launch is RESIDENT while kNN(0) of scan k still runs:
//   the pass's workgroups store their results write-through and count themselves on one word (chain_count_done), one thread per
//   solving workgroup polls it (chain_wait_count) — the hand-off the finalising solve has had from the overlapped second pass
//   since round 3, now also between the first pass and the first solve.
// * kNN(0) of scan k+1 follows APPEND of scan k in stream order: a true kernel boundary, which is what makes the rebuilt cell
//   hash — megabytes written by plain stores and atomics from every XCD — visible to it (a pass pre-launched across the rebuild
//   would have to invalidate its L2: the eight L2s are not coherent).  The one thing it needs from solve(1), which may still be in
//   finalize_scan, is the prediction: pred_xch, tagged granules.  Everything else it reads of the stream's state follows from
//   scan_no (frames appended = scans completed) or was written by launches of its own stream.
// * the rebuild's steps are launches of their own (k_rebuild_alloc, k_rebuild_fin) or extra workgroups of the LIGHT second pass
//   (COUNT + PAD): as extra workgroups of k_lm_solve each of them owned a whole CU (256 VGPRs x 8 waves) and could only be placed
//   on a CU that ran nothing else — the reason why shapes whose second pass has many waiting workgroups (Ouster-128: 704) lost
//   9 % with the overlapped pass; in chain mode the same pass gains 19 % there.
// * finalize_scan must not touch what the rebuild of its own scan still uses on the other stream: st.cursor is reset by the next
//   scan's kNN(0) instead (ALLOC may still be allocating from it).
// What it does NOT buy (measured, in-kernel stamps, profiles/r05_*): the kNN(0) -> solve(0) boundary costs ~1.5 us on this stack,
// and the write-through tail + count + granule read of the chain cost about the same: HDL-64 is unchanged within +-2.5 %.
// Bit-identical to the four-launch path: same workgroups, same partial sums, same order (tools/overlap_equal.py).
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
// Every workgroup of the pass stores its results write-through, waits for their acknowledgement (s_waitcnt — no fence:
// see below) and then stores seq into knn_done[s][b]; the finalising solve's launch — which follows the first solve in
// stream order and therefore starts while this pass still runs — polls those flags in its solving workgroups before it
// reads the pass's results, and in the workgroups that clear the searched table before they touch it.  All waits are bounded (LIODOM_STATUS_PIPE_TIMEOUT, as pipe_wait); workgroups that wait never
// hold more than a third of the GPU's wave slots, and a waiting workgroup depends only on launches enqueued before its own.
// Launch order on the host: kNN(0) [stream], solve(0) [stream], kNN(1) [stream_k], solve(1) [stream].
// =============================================================================================
// Stores / loads that are visible across the XCDs without cache maintenance: agent-scope relaxed atomics go through the
// XCD's L2 to the memory side.  (The alternative — plain accesses plus release / acquire fences — costs an L2 write-back or
// invalidate per fence on a part whose eight L2s are not coherent with each other: with one per workgroup of a 352-workgroup
// launch the solve running beside it took 80 us instead of 24.)
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
// the first solve's result leaves its workgroup: T = odom[12], qt = q[4], t[3] (LDS); one granule per thread and copy.
// copies: bit 0 = the copy the overlapped second pass starts from (granules 0 .. 37 of every replica), bit 1 = the confirmation copy
// (granules kOvFinalOffset ..: what the solve really ended with, see "Speculative hand-over" below).
constexpr int kOvFinalOffset = 64;
__device__ __forceinline__ void ov_publish_pose(const DevView& v, int s, const double* T, const double* qt, unsigned int tag, int tid, int copies) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  const int per = kOvReplicas * kOvGranules;
  INJECT_DELAY(3);
  for (int t = tid; t < 2 * per; t += (int)blockDim.x) {      // (one or two granules per thread at 512 threads)
    const int c = t / per, u = t % per;
    if (!((copies >> c) & 1)) continue;
    const int rep = u / kOvGranules, gi = u % kOvGranules;
    const double val = (gi >> 1) < 12 ? T[gi >> 1] : qt[(gi >> 1) - 12];
    const unsigned long long bits = (unsigned long long)__double_as_longlong(val);
    const unsigned int word = (gi & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
    __hip_atomic_store((gu64*)(v.pose_xch0 + ((size_t)s * kOvReplicas + rep) * 512 + c * kOvFinalOffset + gi), ((unsigned long long)tag << 32) | word,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// whole workgroup; the first wave polls `n_gran` (<= 64, even) tagged granules at `base` until every one carries the tag; out: LDS,
// n_gran / 2 doubles.  false: gave up.
template <int kPollSleep>
__device__ __forceinline__ bool granules_wait(const unsigned long long* base, int n_gran, unsigned int tag, double* out, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  __shared__ int s_ov_ok;
  const int tid = (int)threadIdx.x;
  if (tid < 64) {
    unsigned long long g = 0, t0 = 0;
    unsigned int spins = 0;
    bool ok;
    // (all lanes poll: one round trip after the publication instead of two; what had congested the memory fabric in the
    //  first version of this pass were release / acquire fences — an L2 write-back / invalidate each —, not these loads)
    while (true) {
      if (tid < n_gran) g = __hip_atomic_load((gu64*)(base + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = tid >= n_gran || (unsigned int)(g >> 32) == tag;
      if (__all(ok)) break;
      if (++spins > 2000000u || wait_expired(spins, t0)) break;
      __builtin_amdgcn_s_sleep(kPollSleep);
    }
    const bool all_ok = __all(ok);
    const int nd = n_gran >> 1;
    const unsigned long long lo = __shfl(g, 2 * (tid % nd)), hi = __shfl(g, 2 * (tid % nd) + 1);
    if (tid < nd) out[tid] = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
    if (tid == 0) { s_ov_ok = all_ok ? 1 : 0; if (!all_ok) atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); }
    INJECT_DELAY(4);
  }
  __syncthreads();
  return s_ov_ok != 0;
}
// the first solve's result for the overlapped second pass: replica rep of pose_xch0; out19: LDS
__device__ __forceinline__ bool ov_wait_pose(const DevView& v, int s, int rep, unsigned int tag, double* out19, unsigned int* status) {
  return granules_wait<LIODOM_POLL_POSE>(v.pose_xch0 + ((size_t)s * kOvReplicas + rep) * 512, kOvGranules, tag, out19, status);
}
// =============================================================================================
// Speculative hand-over of a solve's result (round 5).  A solve ends, practically always, with an evaluation whose step is not applied
// (function tolerance: the cost no longer changes); the pose it ends with is then the iterate it held BEFORE that evaluation, and
// the evaluation, its exchange, the controller's last step and the publication (4.4-4.6 us) sit between the result and whatever
// waits for it.  The controller can tell beforehand: the cost change its quadratic model predicts for the step it has just proposed
// (LmState::model_cost_change) is what the evaluation will measure, to a few percent near convergence.  When that prediction is
// below spec_theta (0.8) x the function tolerance, the solving workgroup hands its ITERATE over before the evaluation starts:
//   first solve -> overlapped second pass: copy 0 of pose_xch0 (granules 0..37); what the solve really ended with is always
//     published as the confirmation copy (granules 64..101).  Every workgroup of the pass compares the two when its work is done
//     (ov_confirm_pose) and counts itself done only on equal bits; the others are repeated, from the confirmed pose, by their
//     namesakes in k_knn_redo, the launch behind the pass.
//   finalising solve (chain mode) -> the appending workgroups (pose_xch copy 0) and the next scan's first pass (pred_xch copy 0: the
//     prediction formed from the iterate, by the controller's wave beside the other waves' evaluation).  finalize_scan publishes
//     the confirmed copies and a verdict granule {tag, 1 confirmed / 2 not}.  The first pass waits for the verdict when its work
//     is done: 1 -> it counts itself done; 2 -> it does not, and k_chain_redo0 — the launch behind it, which is also the gate in front
//     of the second pass — takes the frame's points back out of their cells, re-appends them at the confirmed pose (append_fix)
//     and repeats the pass from the confirmed prediction.  When no pass follows, the host enqueues the repair on its own
//     (chain_flush in liodom_hip.hip; not when the scan's host record already says "confirmed").
// Results are those of the non-speculative hand-over in every case (LIODOM_SPECULATE=0 / 1 / 2 = off / model / always as early as
// possible, i.e. practically always wrong: bit-identical pose logs, tools/overlap_equal.py, tools/spec_switches.py); the predictor
// only decides how often the early start pays.  A hand-over that was not confirmed costs the receivers a repeated pass (10-25 us):
// it suspends the hand-overs of that solve for the next 16 scans (StreamState::spec_eval; the first scans behind a freshly filled
// window mispredict in a row).  What the early start required elsewhere: correspondences per kNN pass (corr_a / corr_b [S][2]: a
// pass that starts early writes while the sender's last evaluation still reads), the release of the previous scan's edge buffer
// after the verdict instead of at the first pass's start (chain_release_edges), the scan's prediction kept per parity (pred_odom[2]:
// the repair of scan k runs behind scan k+1's pass).  Once a wait of the handle has given up, nothing is repaired (the scan has
// failed through its status bits; the repairs would work from state that may not have been written).
// =============================================================================================
// returns 1: the pose in io19 is confirmed; 2: it was not — io19 now holds the confirmed one; 0: gave up.
__device__ __forceinline__ int ov_confirm_pose(const DevView& v, int s, int rep, unsigned int tag, double* io19, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  __shared__ int s_cf;
  const unsigned long long* base = v.pose_xch0 + ((size_t)s * kOvReplicas + rep) * 512 + kOvFinalOffset;
  const int tid = (int)threadIdx.x;
  if (tid < 64) {
    unsigned long long g = 0, t0 = 0;
    unsigned int spins = 0;
    bool ok;
    while (true) {
      if (tid < kOvGranules) g = __hip_atomic_load((gu64*)(base + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = tid >= kOvGranules || (unsigned int)(g >> 32) == tag;
      if (__all(ok)) break;
      if (++spins > 2000000u || wait_expired(spins, t0)) break;
      __builtin_amdgcn_s_sleep(LIODOM_POLL_POSE);
    }
    const bool all_ok = __all(ok);
    const unsigned long long lo = __shfl(g, 2 * (tid % 19)), hi = __shfl(g, 2 * (tid % 19) + 1);
    const unsigned long long fin = (hi << 32) | (lo & 0xFFFFFFFFull);
    const bool differs = tid < 19 && fin != (unsigned long long)__double_as_longlong(io19[tid]);
    const bool any = __any(differs);
    if (all_ok && any && tid < 19) io19[tid] = __longlong_as_double((long long)fin);
    if (tid == 0) { s_cf = all_ok ? (any ? 2 : 1) : 0; if (!all_ok) atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); }
    INJECT_DELAY(5);
  }
  __syncthreads();
  return s_cf;
}
// Chain mode: the prediction a scan starts from, published by the previous scan's finalize_scan (threads 0 .. kOvReplicas *
// kPredGranules - 1 of the solving workgroup; vals: 19 doubles in LDS) and read by the scan's first kNN pass (other HIP stream).
// copies: bit 0 = the copy the next scan's first pass starts from (granules 0 .. 37 of every replica), bit 1 = the confirmation copy
// (granules kOvFinalOffset ..); lane / nlanes: the publishing threads (the whole workgroup, or one wave beside the evaluators).
constexpr int kPredVerdict = 127;      // granule of every replica: {tag, 1 = the copy the pass started from is what the solve ended with, 2 = it is not}
__device__ __forceinline__ void pred_publish(const DevView& v, int s, const double* vals, unsigned int tag, int lane, int nlanes, int copies) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  const int per = kOvReplicas * kPredGranules;
  INJECT_DELAY(6);
  for (int t = lane; t < 2 * per; t += nlanes) {
    const int c = t / per, u = t % per;
    if (!((copies >> c) & 1)) continue;
    const int rep = u / kPredGranules, gi = u % kPredGranules;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(vals[gi >> 1]);
    const unsigned int word = (gi & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
    __hip_atomic_store((gu64*)(v.pred_xch + ((size_t)s * kOvReplicas + rep) * 512 + c * kOvFinalOffset + gi), ((unsigned long long)tag << 32) | word,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void pred_verdict_publish(const DevView& v, int s, unsigned int tag, unsigned int verdict, int lane) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  INJECT_DELAY(7);
  if (lane < kOvReplicas) __hip_atomic_store((gu64*)(v.pred_xch + ((size_t)s * kOvReplicas + lane) * 512 + kPredVerdict), ((unsigned long long)tag << 32) | verdict,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// whole workgroup; 0: gave up, else the verdict
__device__ __forceinline__ int pred_verdict_wait(const DevView& v, int s, int rep, unsigned int tag, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  __shared__ int s_vd;
  if (threadIdx.x == 0) {
    const unsigned long long* p = v.pred_xch + ((size_t)s * kOvReplicas + rep) * 512 + kPredVerdict;
    unsigned int spins = 0;
    unsigned long long t0 = 0, g;
    int r = 0;
    while (true) {
      g = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned int)(g >> 32) == tag) { r = (int)(unsigned int)g; break; }
      if (++spins > 2000000u || wait_expired(spins, t0)) { atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); break; }
      __builtin_amdgcn_s_sleep(4);
    }
    s_vd = r;
    INJECT_DELAY(8);
  }
  __syncthreads();
  return s_vd;
}
// Self-resetting arrival counter for the rare repair paths: the word is {epoch, arrivals}; the first arrival of a new epoch resets it.
__device__ __forceinline__ void epoch_arrive(unsigned int* word, unsigned int epoch) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  unsigned int old = __hip_atomic_load((gu32*)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    const unsigned int want = ((old >> 12) == (epoch & 0xFFFFFu)) ? old + 1u : (((epoch & 0xFFFFFu) << 12) | 1u);
    const unsigned int seen = atomicCAS(word, old, want);
    if (seen == old) break;
    old = seen;
  }
}
__device__ __forceinline__ bool epoch_wait(const unsigned int* word, unsigned int epoch, unsigned int n, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  unsigned int spins = 0;
  unsigned long long t0 = 0;
  while (__hip_atomic_load((gu32*)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (((epoch & 0xFFFFFu) << 12) | n)) {
    __builtin_amdgcn_s_sleep(8);
    if (++spins > 2000000u || wait_expired(spins, t0)) { atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); return false; }
  }
  return true;
}
// whole workgroup: the prediction (lanes 0 .. 37 of the first wave: matrix, quaternion, translation) and, in the same round trip,
// the extraction's flag (lane 63; edge_flag may be null / want 0: nothing to wait for).  out12: 19 doubles.  false: one of them never arrived.
__device__ __forceinline__ bool pred_wait(const DevView& v, int s, int rep, unsigned int tag, double* out12, unsigned int* status,
                                          const unsigned int* edge_flag, unsigned int edge_want, int copy = 0) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  typedef __attribute__((address_space(1))) unsigned int gu32;
  __shared__ int s_pw_ok;
  const unsigned long long* base = v.pred_xch + ((size_t)s * kOvReplicas + rep) * 512 + copy * kOvFinalOffset;
  const int tid = (int)threadIdx.x;
  if (tid < 64) {
    unsigned long long g = 0, t0 = 0;
    unsigned int spins = 0;
    bool ok;
    while (true) {
      ok = true;
      if (tid < kPredGranules) { g = __hip_atomic_load((gu64*)(base + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = (unsigned int)(g >> 32) == tag; }
      else if (tid == 63 && edge_flag && edge_want) ok = (int)(__hip_atomic_load((gu32*)edge_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - edge_want) >= 0;
      if (__all(ok)) break;
      if (++spins > 1500000u || wait_expired(spins, t0)) break;
      __builtin_amdgcn_s_sleep(4);
    }
    const bool all_ok = __all(ok);
    if (spins) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (it really waited: as pipe_wait)
    const unsigned long long lo = __shfl(g, 2 * (tid % 19)), hi = __shfl(g, 2 * (tid % 19) + 1);
    if (tid < 19) out12[tid] = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
    if (tid == 0) { s_pw_ok = all_ok ? 1 : 0; if (!all_ok) atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); }
    INJECT_DELAY(9);
  }
  __syncthreads();
  return s_pw_ok != 0;
}
// Chain mode, first pass -> first solve: every workgroup of the pass counts itself on ONE word once its write-through results are
// acknowledged; one thread per solving workgroup polls that word.  (With a flag per pass workgroup, as the second pass has them,
// the eight solving workgroups polled 352 words with 352 threads each for the whole length of the pass — which is bound by the
// latency of its own loads: it got 3-5 us slower.)
__device__ __forceinline__ void chain_count_done(unsigned int* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) INJECT_DELAY(10);
  if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (returns false when the wait gave up: the caller must not consume what it waited for)
__device__ __forceinline__ bool chain_wait_count(const unsigned int* counter, unsigned int target, unsigned int* status) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  __shared__ int s_cw_ok;
  if (threadIdx.x == 0) {
    unsigned int spins = 0;
    unsigned long long t0 = 0;
    bool ok = true;
    while ((int)(__hip_atomic_load((gu32*)counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > 6000000u || wait_expired(spins, t0)) { atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); ok = false; break; }
    }
    s_cw_ok = ok ? 1 : 0;
    INJECT_DELAY(11);
  }
  // (no acquire fence: the argument of ov_wait_knn_done — write-through producers, nothing of theirs cached here before this point)
  __syncthreads();
  return s_cw_ok != 0;
}
// whole workgroup, every exit path of an overlapped k_knn workgroup: its results are visible before the flag is
__device__ __forceinline__ void ov_signal_knn_done(const DevView& v, int s, int b, unsigned int seq, unsigned int* done = nullptr) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the pass's results are write-through stores: acknowledged = visible to every XCD)
  __syncthreads();
  if (threadIdx.x == 0) INJECT_DELAY(12);
  if (threadIdx.x == 0) __hip_atomic_store((gu32*)((done ? done : v.knn_done) + (size_t)s * v.knn_grid + b), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// whole workgroup (finalising solve's launch): every workgroup of the overlapped second pass has completed
__device__ __forceinline__ void ov_wait_knn_done(const DevView& v, int s, unsigned int seq, unsigned int* status, const unsigned int* done = nullptr) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  const unsigned int* f = (done ? done : v.knn_done) + (size_t)s * v.knn_grid;
  for (int b = (int)threadIdx.x; b < v.knn_grid; b += (int)blockDim.x) {
    unsigned int spins = 0;
    unsigned long long t0 = 0;
    while ((int)(__hip_atomic_load((gu32*)(f + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
      __builtin_amdgcn_s_sleep(LIODOM_POLL_DONE);
      if (++spins > 6000000u || wait_expired(spins, t0)) { atomicOr(status, LIODOM_STATUS_PIPE_TIMEOUT); break; }
    }
  }
  INJECT_DELAY(13);
  // No acquire fence (an L2 invalidate per waiting workgroup; with one per workgroup of the pass the solve beside it took 80 us
  // instead of 24).  What this relies on instead — ASSUMPTIONS OUTSIDE THE HIP MEMORY MODEL, written down in DESIGN.md §3 and
  // guarded by tests/test_gpu_parity.py::test_overlapped_pass_long_replay_is_bit_identical and tools/soak_two_process.py:
  //   (1) every result of the pass that this launch reads (corr_a / corr_b / corr_idx / corr_mask / knn_part of THIS pass) is
  //       written with agent-scope relaxed atomic stores = sc1, write-through: once acknowledged (s_waitcnt vmcnt(0) in
  //       ov_signal_knn_done) the bytes are at the memory side, in no XCD's L2 as a dirty line;
  //   (2) this launch holds none of those lines in its L1 / L2 when it reads them: it has not touched them before this point
  //       (the code above reads only state words), no other kernel of this handle reads them between the pass's stores and
  //       here, and the lines it did cache earlier belong to other buffers — the two passes' halves of corr_mask are padded to
  //       different 128-byte lines (mask_stride), corr_idx / knn_part halves are multiples of 128 bytes apart;
  //       (round 6: EXCEPT when the pass was repeated after a speculative hand-over that was not confirmed — its first edition's
  //       workgroups on this XCD have left their lines here: k_lm_solve then invalidates, StreamState::spec_redo);
  //   (3) the launch may well have STARTED after some of the pass's stores (it follows k_rebuild_alloc in stream order): a kernel
  //       start invalidates the caches, so that order is harmless; the order the argument needs is only (1) before the flag.
  // A later edit that makes this launch read one of those buffers before this wait breaks (2) silently: don't.
  __syncthreads();
}

// liodom_create's probe: do kernels of two HIP streams of this process run side by side?  One wave waits (bounded, ~2 ms) for a
// flag that a launch on the other stream sets.  Under a profiler or debugger that serialises kernels across streams (rocprofv3
// --pmc, AMD_SERIALIZE_KERNEL) the setter cannot start before the waiter has given up: result[0] = 1 seen / 2 gave up.
__global__ void k_probe_wait(const unsigned int* flag, unsigned int* result) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  if (threadIdx.x != 0) return;
  unsigned int spins = 0, seen = 2u;
  while (spins++ < 12000u) {        // ~ 2 ms at s_sleep 8 (~0.17 us per round)
    if (__hip_atomic_load((gu32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { seen = 1u; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  result[0] = seen;
}

__global__ void k_set_flag(unsigned int* flag, unsigned int value) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  INJECT_DELAY(20);
  __hip_atomic_store((gu32*)flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
