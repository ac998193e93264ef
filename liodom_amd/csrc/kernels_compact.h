// kernels_compact.h — k_compact_edges: ring-padded edges -> dense edge cloud.
// Part of liodom_kernels.h (included there, inside namespace liodom_dev, in this order; not a standalone header).
// k_compact_edges: one workgroup per stream; ring-padded edges -> dense edge cloud (edge buffer
// `eb`) in the reference's output order.
// =============================================================================================
// grid (kCompactBlocks, streams): every workgroup scans the <= 256 ring counts itself (cheaper than a
// second launch) and copies its interleaved share of the edges.
constexpr int kCompactBlocks = 8;
__global__ __launch_bounds__(256) void k_compact_edges(DevView v, int s0, int eb, unsigned int wait_odo) {
  __shared__ int pre[257];
  __shared__ int cntr[256];
  // (pipelined replay) the odometry that last read edge buffer eb must have completed before it is rewritten
  if (wait_odo && !pipe_wait(v.pipe_flags + kEdgePipeBufs, wait_odo, &v.state[s0 + blockIdx.y].status)) return;
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
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int acc = pre[256];
    v.state[s].n_edges_buf[eb] = acc > v.edge_cap ? v.edge_cap : acc;
  }
  const int E = pre[H] > v.edge_cap ? v.edge_cap : pre[H];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += kCompactBlocks * 256) {
    int lo = 0, hi = H;            // largest r with pre[r] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= e) lo = mid; else hi = mid; }
    const int r = lo, k = e - pre[r];
    const size_t pi = ((size_t)s * H + r) * v.slots_per_ring + k;
    const size_t eo = ((size_t)eb * v.n_streams + s) * v.edge_cap + e;
    v.edges[eo] = v.edges_pad[pi];
    const int2 m = v.edges_pad_meta[pi];
    v.edges_meta[eo] = make_int4(r, m.x, m.y, 0);
  }
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
