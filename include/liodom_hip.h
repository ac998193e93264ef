/* liodom_hip.h — C-ABI of the MI355X-native LiODOM hot path (libliodom_hip.so).
 *
 * The reference (/root/reference, emiliofidalgo/liodom) has no plugin / FFI surface: its hot
 * path lives in private member functions of two worker classes fed by ROS callbacks.  This
 * header is the boundary a maintainer binds instead of those members; each entry point names
 * the reference interface it replaces (paths relative to /root/reference).  INTEGRATION.md
 * shows the reference-side glue.
 *
 * Conventions: plain C, caller-owned buffers with explicit capacities, int status return
 * (0 = LIODOM_OK, negative = error), no exceptions cross the boundary.  Points are packed
 * float[4] XYZI (pcl::PointXYZI without padding).  Poses are world<-laser, 7 doubles
 * [qx qy qz qw tx ty tz] — the storage order of param_q / param_t
 * (include/liodom/laser_odometry.h:98-99, src/laser_odometry.cc:187-195).
 *
 * One handle drives one GPU and `n_streams` independent LiDAR streams that advance in
 * lock-step (every kernel is launched once over all streams).  n_streams = 1 is the drop-in
 * case for liodom_node; n_streams > 1 serves replay / multi-sensor batches.
 *
 * Threading.  The reference runs a FeatureExtractor thread and a LaserOdometer thread side by side
 * (src/liodom_node.cc:89-91) and hands edge clouds over through a queue (src/shared_data.cc:64-89).
 * The handle mirrors that with two sides: liodom_extract_edges works on the extraction side (its own
 * HIP stream, scratch and edge buffer), liodom_odometry_step on the odometry side (window, local-map
 * hash, pose state, result records).  One thread may call liodom_extract_edges while another calls
 * liodom_odometry_step on the same handle: extraction of scan k+1 then overlaps the odometry of scan
 * k on the GPU, and the poses are bit-identical to the serial order (tests/test_gpu_threads.py).
 * Each side serialises its own callers with a mutex; every other entry point takes both mutexes,
 * i.e. is safe to call from any thread but does not overlap with anything.  While per-kernel
 * profiling is enabled (liodom_set_profiling) all entry points are serialised.
 */
#ifndef LIODOM_HIP_H
#define LIODOM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIODOM_OK 0
#define LIODOM_ERR_INVALID_ARG (-1)
#define LIODOM_ERR_UNSUPPORTED (-2)   /* call not valid for the way the handle was created (e.g. mapping = 0) */
#define LIODOM_ERR_CAPACITY (-3)      /* caller buffer or configured capacity too small */
#define LIODOM_ERR_HIP (-4)           /* HIP runtime failure; see liodom_last_error() */
#define LIODOM_ERR_NO_DEVICE (-5)
#define LIODOM_ERR_BUSY (-6)          /* device-resident hand-off: every slot holds an edge cloud that has not been consumed yet (retry
                                         after liodom_odometry_step_device), or tickets are outstanding where none may be */
#define LIODOM_ERR_NEEDS_SYNC (-7)    /* liodom_extract_edges_device while scans of a resident / host replay (liodom_process_resident*,
                                         liodom_replay_*) still occupy the pipeline edge buffers: retrying cannot help — call liodom_sync()
                                         (or liodom_reset()) first, then hand clouds over by ticket */

/* Sticky per-stream status bits reported in liodom_step_info_t.status */
#define LIODOM_STATUS_RING_OVERFLOW 1u  /* deprecated, never raised (rings of any length are processed); kept so that callers
                                           that test the bit keep compiling */
#define LIODOM_STATUS_EDGE_OVERFLOW 2u
#define LIODOM_STATUS_HASH_FULL 4u
#define LIODOM_STATUS_LM_SYNC_TIMEOUT 8u  /* cooperating LM workgroups did not all arrive (result invalid) */
#define LIODOM_STATUS_PIPE_TIMEOUT 16u    /* pipelined replay: a kernel waited ~0.3 s in vain for the other HIP stream of the handle
                                             (kernels serialised across streams by a profiler / debugger, or a GPU saturated by
                                             another process).  The waiting workgroups skipped their work: the scan's result is
                                             invalid, the entry point that collects it returns LIODOM_ERR_HIP, and the handle
                                             switches to event-based stream dependencies; liodom_reset() before reusing the stream */

/* Field-for-field mirror of liodom::Params (include/liodom/params.h:33-49); defaults are
 * those of Params::readParams (src/params.cc:40-109). */
typedef struct liodom_params_t {
  double min_range;              /* params.h:33  default 3.0  */
  double max_range;              /* params.h:34  default 75.0 */
  int32_t lidar_type;            /* params.h:35  0 Velodyne (elevation binning), 1 Ouster (row = ring) */
  int32_t scan_lines;            /* params.h:36  default 64 */
  int32_t scan_regions;          /* params.h:37  default 8  */
  int32_t edges_per_region;      /* params.h:38  default 10 */
  uint64_t min_points_per_scan;  /* params.h:39  scan_regions*edges_per_region + 10 (params.cc:63) */
  uint64_t local_map_size;       /* params.h:40  "prev_frames", default 5 */
  int32_t save_results;          /* params.h:41 */
  char results_dir[256];         /* params.h:42  default "~/" */
  char fixed_frame[64];          /* params.h:43  default "odom" */
  char base_frame[64];           /* params.h:44  default "base_link" */
  char laser_frame[64];          /* params.h:45  default "" (taken from the header) */
  int32_t use_imu;               /* params.h:46 */
  int32_t filter_local_map;      /* params.h:47 */
  int32_t mapping;               /* params.h:48 */
  int32_t publish_tf;            /* params.h:49  default true */
} liodom_params_t;

/* Engine configuration that has no counterpart in the reference. */
typedef struct liodom_config_t {
  int32_t device;            /* HIP device ordinal */
  int32_t n_streams;         /* independent streams advanced in lock-step (>= 1) */
  int32_t max_points;        /* capacity: points per scan (H*W) */
  int32_t max_width;         /* expected points per ring (0 = max_points / scan_lines): picks the extraction kernel instance whose
                                register tile covers the longest region, (max_width - 10) / scan_regions + remainder items;
                                not a capacity — longer rings are processed by the generic path of the same kernel */
  union {                    /* (anonymous union: C11 / C++) */
    int32_t reserved1;       /* ignored */
    int32_t max_ring_points; /* deprecated name of the same field (earlier headers: capacity of a ring): ignored, rings have no
                                capacity any more; kept so that callers that set it keep compiling */
  };
  int32_t lm_apply_step_on_ftol; /* 0 = Ceres >= 1.12 behaviour (LM section) */
  int32_t pose_log_capacity; /* scans kept in the device-side pose log (resident replay) */
  int32_t debug_buffers;     /* 1 = keep per-ring smoothness dumps for liodom_get_curvature */
  int32_t lm_workgroups;     /* workgroups (CUs) per stream for the pose solve: 0 = auto (<= 4 streams: 8 for >= 8192 possible edges, 4 for >= 4096; else 1), or 1 .. 8 */
  int32_t recv_capacity;     /* mapping = 1: points of the received ~map cloud the kNN structure can take (0 = 262144) */
  int32_t pose_rotation_mode; /* what Eigen::Transform::rotation() returns at src/laser_odometry.cc:164,186,403,420:
                                 1 (default) = Eigen 3.3.x, the README's platform: orthonormal polar factor of linear()
                                 (computeRotationScaling, JacobiSVD) for every Mode; 0 = Eigen >= 3.4: alias of linear()
                                 for an Isometry.  See DESIGN.md §4. */
  int32_t reserved0;
} liodom_config_t;

typedef struct liodom_lm_trace_t {
  int32_t iterations;        /* trust-region iterations used (<= 4, src/laser_odometry.cc:214) */
  int32_t accepted;
  int32_t termination;       /* 0 max-iter 1 param-tol 2 func-tol 3 grad-tol 4 no-residuals 5 eval-failure 6 radius 7 invalid-steps */
  int32_t pad;
  double initial_cost;
  double final_cost;
} liodom_lm_trace_t;

/* Per-scan diagnostics (what the reference logs with ROS_DEBUG at
 * feature_extractor.cc:62, laser_odometry.cc:279,297,365). */
typedef struct liodom_step_info_t {
  int32_t n_edges;
  int32_t map_points;        /* local-map points searched by this scan */
  int32_t matches[2];        /* "Correct matchings" of the two outer iterations */
  liodom_lm_trace_t lm[2];
  uint32_t status;
  int32_t scan_index;
} liodom_step_info_t;

typedef struct liodom_handle liodom_handle_t;
typedef struct liodom_map liodom_map_t;

/* Params::readParams defaults (src/params.cc:40-109). */
void liodom_params_default(liodom_params_t* p);
void liodom_config_default(liodom_config_t* c);

/* Replaces the construction of FeatureExtractor + LaserOdometer
 * (src/liodom_node.cc:85-86; feature_extractor.cc:24-37; laser_odometry.cc:71-95). */
int liodom_create(const liodom_params_t* params, const liodom_config_t* config, liodom_handle_t** out);
void liodom_destroy(liodom_handle_t* h);
const char* liodom_last_error(void);

/* FeatureExtractor::splitPointCloud + extractFeatures (src/feature_extractor.cc:104-254) for one
 * cloud of stream `stream`.  Extraction side: may run concurrently with liodom_odometry_step.  xyzi: n points (host).  For lidar_type 1, height*width == n and
 * the cloud is row-major organised.  Outputs (host, capacity `cap` edges): edges in the
 * reference's output order (ring-major, region-major, pick order); edge_ring / edge_idx /
 * edge_src (each optional) = ring id, index inside the compacted ring, index into xyzi. */
int liodom_extract_edges(liodom_handle_t* h, int stream, const float* xyzi, int64_t n,
                         int height, int width, float* edges_xyzi, int32_t* edge_ring,
                         int32_t* edge_idx, int32_t* edge_src, int cap, int* n_edges);

/* One iteration of LaserOdometer::operator() (src/laser_odometry.cc:107-267) on an edge cloud
 * (sensor frame, host memory; odometry side: may run concurrently with liodom_extract_edges): first call initialises the window, later calls predict, run
 * 2 x [addEdgeConstraints + solve], and append the transformed edges to the sliding window. */
int liodom_odometry_step(liodom_handle_t* h, int stream, const float* edges_xyzi, int n_edges,
                         double stamp, double* pose_out, liodom_step_info_t* info);

/* ---- the same two entry points with the edge cloud staying on the device ----
 * The reference hands every edge cloud from the FeatureExtractor thread to the LaserOdometer thread through a queue
 * (src/shared_data.cc:64-89: pushFeatures / popFeatures).  With the two calls above that queue element is a host cloud: a
 * device-to-host copy on the extraction side, a host-to-device copy on the odometry side, both synchronous, per scan.  Here the
 * queue element is a TICKET: the edges stay in one of three device buffers, the scan's upload is asynchronous, nothing on the
 * extraction side blocks, and the odometry's first kernel waits on the device for the extraction it needs.  One-stream handles.
 *
 *   liodom_extract_edges_device   FeatureExtractor::operator() body up to the hand-over (feature_extractor.cc:49-77): enqueues
 *       upload + extraction of one cloud and returns at once.  LIODOM_ERR_BUSY when all three slots hold clouds that
 *       liodom_odometry_step_device has not taken yet (the reference's queue is unbounded: keep the cloud and retry).
 *       xyzi: any host memory; a buffer from liodom_scan_buffer or one registered with liodom_pin_host_buffer is read
 *       asynchronously (no staging copy) and must then stay untouched until liodom_wait_edges or liodom_odometry_step_device of
 *       the ticket has returned.
 *   liodom_wait_edges             the cloud for the ~edges topic (feature_extractor.cc:70-75): waits until the extraction has
 *       completed and copies the edges out of host-mapped memory the extraction kernel wrote (no device-to-host copy call).
 *       Optional; any thread; before the ticket's liodom_odometry_step_device call returns or — from the extractor thread — before
 *       the ticket is pushed into the queue.
 *   liodom_odometry_step_device   LaserOdometer::operator() body (laser_odometry.cc:107-267) on the ticket's cloud; tickets must
 *       be consumed in the order they were issued.  = liodom_odometry_submit_device + liodom_odometry_collect: a thread that finds
 *       the next ticket already in its queue may submit it before it collects (and publishes) the previous pose — the device then
 *       needs nothing from the host between two scans; at most two scans in flight (LIODOM_ERR_BUSY beyond).  A slot becomes free
 *       for the extraction side when the pose of its scan has been collected.
 *   liodom_scan_buffer            a page-locked buffer (capacity max_points) to assemble the next cloud in — the place of
 *       pcl::fromROSMsg's target in lidarClb (src/liodom_node.cc:43-44); valid until the extraction it is passed to.
 * Poses are bit-identical to liodom_process_scan on the same clouds (tests/test_gpu_threads.py). */
typedef struct liodom_edge_ticket_t {
  uint32_t seq;              /* sequence number of the extraction (never 0) */
  int32_t slot;              /* device edge buffer that holds the cloud */
  int32_t stream;
  int32_t reserved;
} liodom_edge_ticket_t;
int liodom_scan_buffer(liodom_handle_t* h, int stream, float** xyzi, int64_t* capacity_points);
int liodom_extract_edges_device(liodom_handle_t* h, int stream, const float* xyzi, int64_t n, int height, int width,
                                liodom_edge_ticket_t* ticket);
int liodom_wait_edges(liodom_handle_t* h, const liodom_edge_ticket_t* ticket, float* edges_xyzi, int32_t* edge_ring,
                      int32_t* edge_idx, int32_t* edge_src, int cap, int* n_edges);
int liodom_odometry_step_device(liodom_handle_t* h, const liodom_edge_ticket_t* ticket, double stamp, double* pose_out,
                                liodom_step_info_t* info);
int liodom_odometry_submit_device(liodom_handle_t* h, const liodom_edge_ticket_t* ticket, double stamp);
int liodom_odometry_collect(liodom_handle_t* h, int stream, double* pose_out, liodom_step_info_t* info);

/* lidarClb -> FeatureExtractor -> LaserOdometer for one scan without leaving the device
 * (src/liodom_node.cc:40-55 + the two worker loops).  pose_out / info / edges outputs optional. */
int liodom_process_scan(liodom_handle_t* h, int stream, const float* xyzi, int64_t n, int height,
                        int width, double stamp, double* pose_out, liodom_step_info_t* info);

/* mapClb -> SharedData::setLocalMap (src/liodom_node.cc:57-64).  Only with mapping = 1: the next
 * scan's kNN cloud is window ++ this cloud (src/laser_odometry.cc:276-278,310-314). */
int liodom_set_received_map(liodom_handle_t* h, int stream, const float* xyzi, int64_t n);
/* imuClb -> SharedData::setLastIMUOri (src/liodom_node.cc:66-70): latest IMU orientation [x y z w].
 * With use_imu = 1 the roll and pitch of every predicted pose are replaced by the IMU's before the
 * solve (src/laser_odometry.cc:152-183).  Identity until first set. */
int liodom_set_imu_orientation(liodom_handle_t* h, int stream, const double* q_xyzw);
/* laser_to_base_ (TF lookup at src/laser_odometry.cc:110-119): 3 x 4 row-major, identity by default.
 * It enters the IMU override only; poses are returned in the laser frame (odom_). */
int liodom_set_laser_to_base(liodom_handle_t* h, const double* T);
/* The last received ~map cloud of a stream (inspection). */
int liodom_get_received_map(liodom_handle_t* h, int stream, float* xyzi, int64_t cap, int64_t* n_points);

/* ---- resident replay (bench / batched streams): scans live in HBM before timing starts ---- */
int liodom_alloc_resident(liodom_handle_t* h, int n_slots);
int liodom_upload_scan(liodom_handle_t* h, int stream, int slot, const float* xyzi, int64_t n);
/* Advance every stream by one scan read from resident slot `slot` (n, height, width as above,
 * identical for all streams).  If poses_out != NULL (n_streams*7 doubles) the call waits for
 * the poses (per-scan synchronous, as the node publishes ~odom per scan); otherwise it only
 * enqueues and poses are read later from the device-side log. */
int liodom_process_resident(liodom_handle_t* h, int slot, int64_t n, int height, int width,
                            double* poses_out, liodom_step_info_t* infos_out);
/* Same, and additionally issues the extraction of resident slot `next_slot` (if >= 0) on a second
 * HIP stream so that it overlaps this scan's odometry — the reference's own two-thread pipeline
 * (FeatureExtractor / LaserOdometer threads, src/liodom_node.cc:89-91).  The following call must
 * then name that slot. */
int liodom_process_resident_pipelined(liodom_handle_t* h, int slot, int next_slot, int64_t n, int height,
                                      int width, double* poses_out, liodom_step_info_t* infos_out);
/* The consumer loop of the pipelined replay, in C: resident slots first_slot .. first_slot + count - 1 in order, every
 * scan exactly as liodom_process_resident_pipelined (the extraction of scan k+1 is issued beside the odometry of scan k)
 * and every pose read back, in order — like the LaserOdometer thread that publishes ~odom per scan
 * (src/liodom_node.cc:89-91 / laser_odometry.cc:100-107,403-430).
 * depth = 0: strictly synchronous — pose k is read back before the odometry of scan k+1 is submitted (the GPU idles
 * for the host's turn-around, ~6 us per scan).  depth = 1: the odometry of scan k+1 is submitted before pose k is
 * waited for (the device needs nothing from the host between two scans; poses arrive exactly when they would anyway).
 * `ahead` != 0 also issues the extraction of slot first_slot + count at the end (the next call must start there).
 * poses_out: count * n_streams * 7 doubles; infos_out: count * n_streams records (either may be NULL: then the poses
 * are only waited for). */
int liodom_replay_resident(liodom_handle_t* h, int first_slot, int count, int ahead, int depth, int64_t n, int height, int width,
                           double* poses_out, liodom_step_info_t* infos_out);

/* Host-fed replay: the shape the patched liodom_node sees (scans arrive in HOST memory, one per PointCloud2 message,
 * src/liodom_node.cc:40-55 -> shared_data.cc:37-42).  Scan i of stream s is read from
 * xyzi_base + ((size_t)i * n_streams + s) * scan_stride_floats (n points of packed float4).  The upload of scan k+1
 * (hipMemcpyAsync on the extraction stream into a ring of device staging slots) and its extraction overlap the
 * odometry of scan k; every pose is read back, in order (depth as in liodom_replay_resident).  For the copies to be
 * asynchronous the host buffer must be page-locked: liodom_pin_host_buffer / liodom_unpin_host_buffer register a
 * caller-owned buffer (pageable memory works, the copies then stage through the runtime).  The handle's resident
 * scan buffer is (re)allocated as the staging ring (liodom_alloc_resident(h, 3)) if it has fewer than 3 slots; resident slots
 * 0 .. 2 are overwritten by the replayed scans either way. */
int liodom_replay_host(liodom_handle_t* h, const float* xyzi_base, int64_t scan_stride_floats, int count, int depth,
                       int64_t n, int height, int width, double* poses_out, liodom_step_info_t* infos_out);
int liodom_pin_host_buffer(void* p, int64_t bytes);
int liodom_unpin_host_buffer(void* p);
int liodom_sync(liodom_handle_t* h);
int liodom_get_pose_log(liodom_handle_t* h, int stream, int first, int count, double* poses_out,
                        liodom_step_info_t* infos_out);
/* Resets the odometry state (pose, window, map) of every stream; capacities are kept. */
int liodom_reset(liodom_handle_t* h);

/* ---- inspection (tests, parity checks) ---- */
/* Edges of the last scan of a stream as left on the device by process_scan / process_resident. */
int liodom_get_edges(liodom_handle_t* h, int stream, float* edges_xyzi, int32_t* edge_ring,
                     int32_t* edge_idx, int32_t* edge_src, int cap, int* n_edges);
/* Sliding-window points in LocalMapManager order (oldest frame first). */
int liodom_get_window(liodom_handle_t* h, int stream, float* xyzi, int64_t cap, int64_t* n_points,
                      int* n_frames);
/* The cloud the next scan's kNN will search (computeLocalMap, src/laser_odometry.cc:274-298): the
 * window, or — with filter_local_map and a full window — its VoxelGrid(0.4) down-sampling in
 * PCL's output order (ascending leaf index).  *filtered tells which. */
int liodom_get_local_map(liodom_handle_t* h, int stream, float* xyzi, int64_t cap, int64_t* n_points, int* filtered);
/* Correspondences of outer iteration `it` (0/1) of the last step: valid flag and window
 * indices (as in liodom_get_window; PCL leaf indices when the local map is filtered) of the two
 * line points per edge. */
int liodom_get_correspondences(liodom_handle_t* h, int stream, int it, int32_t* valid,
                               int32_t* idx_a, int32_t* idx_b, int cap, int* n);
/* World-frame float queries (edges transformed by the pose entering outer iteration `it`,
 * src/laser_odometry.cc:307-308) of the last step: n x float[4] (x y z 0).  debug_buffers = 1 only.
 * Together with liodom_get_local_map (taken before the step) they are the exact inputs of the 5-NN +
 * line-gate kernel, so that its output can be compared bit for bit with the oracle on identical inputs. */
int liodom_get_knn_queries(liodom_handle_t* h, int stream, int it, float* xyz0, int cap, int* n);
/* Smoothness values of the last extracted scan, ring-major over the compacted rings; also the
 * ring offsets (scan_lines + 1 entries).  NaN where the stencil is undefined. */
int liodom_get_curvature(liodom_handle_t* h, int stream, double* curv, int64_t cap,
                         int32_t* ring_offsets);

/* ---- measurement ---- */
/* When enabled every kernel launch is bracketed by HIP events on the handle's stream. */
int liodom_set_profiling(liodom_handle_t* h, int enable);
#define LIODOM_NUM_KERNELS 12
typedef struct liodom_kernel_stat_t {
  char name[32];
  int64_t launches;
  double total_ms;
} liodom_kernel_stat_t;
/* Drains recorded events (synchronises) and accumulates into the per-kernel table. */
int liodom_get_kernel_stats(liodom_handle_t* h, liodom_kernel_stat_t* stats /*LIODOM_NUM_KERNELS*/);
int liodom_reset_kernel_stats(liodom_handle_t* h);
/* Number of HIP devices visible to this process (0 without a GPU). */
int liodom_device_count(int* count);
/* PCI bus id ("0000:c1:00.0") of HIP device `device`: lets a launcher place the host thread that
 * polls the result records on the GPU's NUMA node (/sys/bus/pci/devices/<id>/local_cpulist). */
int liodom_device_pci_bus_id(int device, char* bus_id, int cap);
/* Device name and compute-unit count of the handle's GPU. */
int liodom_device_info(liodom_handle_t* h, char* name, int name_cap, int* compute_units);
/* The code paths this handle runs, as "key=value key=value ..." (stream-dependency mechanism, hash rebuild variant,
 * workgroups per solve, kNN tuning, test switches picked up from the environment at liodom_create).  Every variant
 * produces the same results (each has an equality test); measurements quote this string next to their numbers. */
int liodom_get_modes(liodom_handle_t* h, char* buf, int cap);


/* ---- liodom::Map on the device (mapping node, src/map.cc, src/liodom_mapping_node.cc) ---- */
typedef struct liodom_map_config_t {
  int32_t device;              /* HIP device ordinal */
  int32_t max_cells;           /* coarse cells the map can hold (cells_vector_) */
  double voxel_xysize;         /* ~voxel_xysize, liodom_mapping_node.cc:115-117  default 40 */
  double voxel_zsize;          /* ~voxel_zsize,  :119-121  default 50 */
  double resolution;           /* ~resolution,   :123-125  default 0.4 */
  int32_t cell_capacity;       /* points per cell (after filtering, and filtered + appended during an update) */
  int32_t max_update_points;   /* points per updateMap call */
  int32_t max_modified_cells;  /* cells touched by one updateMap call (<= 256) */
  int32_t reserved;
} liodom_map_config_t;
/* status bits of liodom_map_status */
#define LIODOM_MAP_UPDATE_OVERFLOW 1u
#define LIODOM_MAP_CELLS_FULL 2u
#define LIODOM_MAP_MODIFIED_FULL 4u
#define LIODOM_MAP_CELL_OVERFLOW 8u
#define LIODOM_MAP_LEAF_RANGE 16u
#define LIODOM_MAP_KEY_RANGE 32u
#define LIODOM_MAP_RESULT_OVERFLOW 64u

void liodom_map_config_default(liodom_map_config_t* c);
/* Map::Map (src/map.cc:70-81) */
int liodom_map_create(const liodom_map_config_t* config, liodom_map_t** out);
void liodom_map_destroy(liodom_map_t* m);
/* Map::updateMap (src/map.cc:90-129): xyzi = n sensor-frame points (host), T = world<-sensor
 * isometry as 12 doubles, row-major 3 x 4 (the Eigen::Isometry3d of liodom_mapping_node.cc:63-64). */
int liodom_map_update(liodom_map_t* m, const float* xyzi, int64_t n, const double* T);
/* Map::getLocalMap (src/map.cc:141-189) with the node's ~cells_xy / ~cells_z
 * (liodom_mapping_node.cc:130-134, defaults 2 and 1).  Output order = the reference's loops. */
int liodom_map_get_local(liodom_map_t* m, const double* T, int cells_xy, int cells_z, float* xyzi,
                         int64_t cap, int64_t* n_points);
/* Map::getMap (src/map.cc:131-139): every cell in creation order. */
int liodom_map_get_all(liodom_map_t* m, float* xyzi, int64_t cap, int64_t* n_points);
int liodom_map_num_cells(liodom_map_t* m, int* n_cells);
/* Wires a map to stream `stream` of an odometry handle created with mapping = 1, replaying the
 * two-node loop of launch/liodom.launch:41-56 synchronously on the device: after every scan k the
 * handle enqueues updateMap(edges_k, pose_k) (lidarClb, liodom_mapping_node.cc:45-69) and
 * getLocalMap(pose_k, cells_xy, cells_z) (:78-86) whose result becomes the received map of scan
 * k+1 (mapClb, liodom_node.cc:57-64) without leaving HBM.  (The reference's two nodes run
 * asynchronously, so which map a scan sees is timing dependent there; this is the zero-latency
 * case.)  The map must live on the handle's device; from here on it uses the handle's HIP stream.
 * Pass map = NULL to detach. */
int liodom_attach_mapper(liodom_handle_t* h, int stream, liodom_map_t* m, int cells_xy, int cells_z);
/* Sticky LIODOM_MAP_* bits raised by the device since creation. */
int liodom_map_status(liodom_map_t* m, uint32_t* status);

#ifdef __cplusplus
}
#endif
#endif /* LIODOM_HIP_H */
