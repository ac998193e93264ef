// kernels_compact.h — k_compact_edges: ring-padded edges -> dense edge cloud.
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, in this order; not a standalone header).
// k_compact_edges: one workgroup per stream; ring-padded edges -> dense edge cloud (edge buffer
// `eb`) in the reference's output order.
// =============================================================================================
// grid (kCompactBlocks, streams): every workgroup scans the <= 256 ring counts itself (cheaper than a
// second launch) and copies its interleaved share of the edges.
constexpr int kCompactBlocks = 8;
__device__ __forceinline__ void compact_edges_body(const DevView& v, int s0, int eb, int mirror, int* pre, int* cntr) {
  const int s = s0 + blockIdx.y;
  const int H = v.scan_lines;
  const int* rn = v.ring_nedges + (size_t)s * H;
  {
    // exclusive prefix over the H <= 256 ring counts: DPP wave scan + 4 wave totals
    const int mine = ((int)threadIdx.x < H) ? rn[threadIdx.x] : 0;
    const int incl = wave_incl_scan_i32(mine);
    if ((threadIdx.x & 63) == 63) cntr[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += cntr[w];
    pre[threadIdx.x] = base + incl - mine;
    if (threadIdx.x == 255) pre[256] = base + incl;
    __syncthreads();
    // threads >= H contribute 0, so pre[H] already equals the total
  }
  // mirror: the device-resident hand-off also leaves the edges in host-mapped memory (slot eb) for the thread that publishes
  // ~edges; the stores cross PCIe while the kernel runs and are complete when it ends (k_publish_edges follows in stream order)
  const bool to_host = mirror != 0 && v.host_edges != nullptr && eb < kEdgePipeBufs;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int acc = pre[256];
    v.state[s].n_edges_buf[eb] = acc > v.edge_cap ? v.edge_cap : acc;
    if (v.edge_cnt && s < 32) {      // (chain mode: the first solve's launch is resident before this extraction may have run)
      typedef __attribute__((address_space(1))) unsigned int gu32;
      __hip_atomic_store((gu32*)(v.edge_cnt + eb * 32 + s), (unsigned int)(acc > v.edge_cap ? v.edge_cap : acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (to_host) v.host_edges_hdr[kEdgePipeBufs + eb] = (unsigned int)(acc > v.edge_cap ? v.edge_cap : acc);
  }
  const int E = pre[H] > v.edge_cap ? v.edge_cap : pre[H];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += kCompactBlocks * 256) {
    int lo = 0, hi = H;            // largest r with pre[r] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= e) lo = mid; else hi = mid; }
    const int r = lo, k = e - pre[r];
    const size_t pi = ((size_t)s * H + r) * v.slots_per_ring + k;
    const size_t eo = ((size_t)eb * v.n_streams + s) * v.edge_cap + e;
    const float4 pt = v.edges_pad[pi];
    v.edges[eo] = pt;
    const int2 m = v.edges_pad_meta[pi];
    v.edges_meta[eo] = make_int4(r, m.x, m.y, 0);
    if (to_host) {
      v.host_edges[(size_t)eb * v.edge_cap + e] = pt;
      v.host_edges_meta[(size_t)eb * v.edge_cap + e] = make_int4(r, m.x, m.y, 0);
    }
  }
}

// pub_value != 0 (round 6): the launch also PUBLISHES the extraction — what k_publish_edges / k_set_flag did in a launch of their
// own behind this one (~5 us of the extraction stream per scan, the stream that bounds the two-thread binding): every thread
// completes its stores (system scope: the mirror in host memory included), the workgroups count themselves on pub_counter[eb],
// and the last one to arrive writes the sequence number for the odometry side's kernels (pub_flag, may be null) and for the host
// thread that waits for the edges (pub_host, may be null).  A launch whose wait gave up still counts and publishes, as the
// separate launch did.
__global__ __launch_bounds__(256) void k_compact_edges(DevView v, int s0, int eb, unsigned int wait_odo, int mirror,
                                                       unsigned int* pub_flag, unsigned int* pub_host, unsigned int pub_value) {
  __shared__ int pre[257];
  __shared__ int cntr[256];
  // (pipelined replay) the odometry that last read edge buffer eb must have completed before it is rewritten
  const bool go = !(wait_odo && !pipe_wait(v.pipe_flags + kEdgePipeBufs, wait_odo, &v.state[s0 + blockIdx.y].status));      // (uniform over the workgroup)
  if (go) compact_edges_body(v, s0, eb, mirror, pre, cntr);
  if (pub_value) {
    typedef __attribute__((address_space(1))) unsigned int gu32;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int total = gridDim.x * gridDim.y;
      const unsigned int prev = __hip_atomic_fetch_add(v.pub_counter + eb, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == total - 1u) {
        __hip_atomic_store(v.pub_counter + eb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the next launch on this buffer follows in stream order)
        INJECT_DELAY(19);
        if (pub_flag) __hip_atomic_store((gu32*)pub_flag, pub_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pub_host) __hip_atomic_store(pub_host, pub_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
// Behind k_compact_edges in stream order (that launch has ended: its stores, to HBM and to host memory, are complete):
// the extraction's sequence number for the odometry side's kernels (dev_flag, as k_set_flag; may be null) and for the host
// thread that waits for the edges (host_seq, system scope; may be null).
__global__ void k_publish_edges(unsigned int* dev_flag, unsigned int* host_seq, unsigned int value) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  INJECT_DELAY(19);
  if (dev_flag) __hip_atomic_store((gu32*)dev_flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (host_seq) __hip_atomic_store(host_seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// For liodom_odometry_step (edges supplied by the caller): set counts and reset diagnostics.
__global__ void k_set_edges(DevView v, int s0, int n_edges, int eb) {
  const int s = s0 + blockIdx.x;
  if (threadIdx.x == 0) {
    StreamState& st = v.state[s];
    st.n_edges_buf[eb] = n_edges;
    st.info.matches[0] = 0; st.info.matches[1] = 0;
  }
}
