/*
 * dsr.h — C ABI of the MI355X-native dense scene reconstruction (DSR) engine.
 *
 * This is the drop-in boundary for DynSLAM's voxel-hashed TSDF hot path
 * (allocation -> integration -> [swap/decay] -> raycast).  In the reference the
 * boundary is C++ inheritance: `class InfiniTamDriver : public ITMMainEngine`
 * (src/DynSLAM/InfiniTamDriver.h:79) poking at ITMLib's protected members.  There
 * is no FFI layer in the reference, so every entry point below cites the ITMLib
 * call (as seen from DynSLAM's call sites) that it replaces.  The header-only C++
 * shim in shim/ re-creates the ITMLib class/method names on top of these calls.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types.
 *   - 4x4 matrices are float[16] COLUMN-MAJOR, exactly ORUtils::Matrix4f::m
 *     (InfiniTamDriver.cpp:146-163).
 *   - RGBA images are uint8[4*W*H] in R,G,B,A order (InfiniTamDriver.cpp:89-94).
 *   - raw depth is int16 millimetres (InfiniTamDriver.cpp:52,77), float depth is
 *     metres with <=0 meaning "invalid" (InstanceReconstructor.cpp:97,165).
 *   - every call returns a dsr_status (0 == DSR_OK) unless stated otherwise.
 *   - calls on one engine handle must come from one thread at a time; distinct
 *     handles are independent (each owns its HIP stream) and may be driven from
 *     distinct threads — with two things shared per GPU and process: the I/O stream
 *     (frame uploads, previews, view read-backs of ALL engines of that GPU queue on
 *     it in call order) and, for engines a host waits on (sync_status), the view
 *     stream, on which the instance-sized volumes of that GPU also fuse
 *     (dsr_settings.view_pipeline, the SHARED form: their default).  DynSLAM drives all its drivers from one thread
 *     (SURVEY 8b); a host that drives a map and its instance drivers from several
 *     threads gets correct results and head-of-line waits on those streams.
 *     Engines that share a stream explicitly (dsr_engine_share_stream, a
 *     dsr_batch) must be driven from ONE thread.
 *   - "_dev" variants take pointers to device (HBM) memory on the engine's GPU.
 *
 * The same signatures, with the prefix `orc_` instead of `dsr_`, are exported by
 * the CPU oracle (oracle/dsr_oracle.cpp), which is TEST INFRASTRUCTURE only.
 */
#ifndef DSR_H_
#define DSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever a public struct layout or the meaning of a field changes, so that a library and a caller built on
 * different sides of the change refuse each other at load (2: dsr_kernel_time grew bytes_layout/units, dsr_stats was
 * extended, DSR_E_IO; 3: the multi-GPU exchange (dsr_exchange_*), dsr_update_view_bgr, host-buffer calls no longer
 * synchronise with the engine's stream; 4: dsr_view_split_silhouette, dsr_engine_share_stream, dsr_pin_host_thread, the volume
 * batch dsr_batch_*, dsr_exchange_set_collective / _timing; the view pipeline is on by default for engines with sync_status;
 * 5: dsr_settings.view_pipeline, dsr_measure_copy_bandwidth_spread; dsr_prepare / dsr_batch_fuse defer the tracking render of an
 * instance-sized volume to the next call (paired render); dsr_exchange_clear_target is carried out by the next composite) */
#define DSR_ABI_VERSION 5

/* SDF_BLOCK_SIZE / SDF_BLOCK_SIZE3 (InfiniTamDriver.h:243,247). */
#define DSR_BLOCK_SIZE 8
#define DSR_BLOCK_SIZE3 512
/* Upstream ITMLibDefines.h compile-time sizes; runtime-configurable here because
 * the fork already made sdfLocalBlockNum a runtime setting
 * (InstanceReconstructor.cpp:379). */
#define DSR_DEFAULT_BUCKET_NUM 0x100000
#define DSR_DEFAULT_EXCESS_LIST_SIZE 0x20000
#define DSR_DEFAULT_LOCAL_BLOCK_NUM 0x40000
#define DSR_TRANSFER_BLOCK_NUM 0x1000

typedef enum dsr_status {
  DSR_OK = 0,
  DSR_E_ARG = 1,           /* bad argument / unsupported combination          */
  DSR_E_DEVICE = 2,        /* HIP runtime error (ITMSafeCall in the reference) */
  DSR_E_OUT_OF_BLOCKS = 3, /* voxel block array or excess list exhausted; the
                              fork throws std::runtime_error, caught at
                              InstanceReconstructor.cpp:662-671                */
  DSR_E_NO_VIEW = 4,       /* no frame given yet (InfiniTamDriver.cpp:168,185) */
  DSR_E_NOMEM = 5,
  DSR_E_IO = 6             /* file missing or malformed (std::runtime_error at PrecomputedDepthProvider.cpp:40-52) */
} dsr_status;

/* ITMHashEntry (ITMLib/Objects/ITMVoxelBlockHash.h): 16 bytes.
 * pos: block coordinates; offset-1 = index into the excess list of the next
 * entry of the chain (<1: end); ptr >= 0: block index in the voxel block array,
 * -1: swapped out, < -1: unallocated. */
typedef struct dsr_hash_entry {
  int16_t pos[3];
  int16_t _pad;
  int32_t offset;
  int32_t ptr;
} dsr_hash_entry;

/* ITMVoxel_s_rgb (ITMLib/Utils/ITMLibDefines.h), 8 bytes; the array-of-structs
 * exchange format used by the dump/load/swap entry points.  (The HIP engine
 * stores blocks plane-wise in HBM; this is the interchange
 * layout.) */
typedef struct dsr_voxel {
  int16_t sdf;     /* float value = sdf / 32767                */
  uint8_t w_depth; /* integration weight                       */
  uint8_t clr[3];  /* running mean colour                      */
  uint8_t w_color;
  uint8_t _pad;
} dsr_voxel;

/* ITMSceneParams + the ITMLibSettings fields DynSLAM touches
 * (DynSLAMGUI.cpp:1214-1219; InstanceReconstructor.cpp:365-380). */
typedef struct dsr_settings {
  float voxel_size;        /* sceneParams.voxelSize [m]                       */
  float mu;                /* sceneParams.mu, truncation band [m]             */
  int32_t max_w;           /* sceneParams.maxW                                */
  float view_frustum_min;  /* sceneParams.viewFrustum_min [m]                 */
  float view_frustum_max;  /* sceneParams.viewFrustum_max [m]                 */
  int32_t stop_integrating_at_max_w;
  int32_t sdf_local_block_num; /* settings->sdfLocalBlockNum (fork)           */
  int32_t hash_bucket_num;     /* SDF_BUCKET_NUM, power of two                */
  int32_t excess_list_size;    /* SDF_EXCESS_LIST_SIZE                        */
  int32_t use_swapping;        /* settings->useSwapping                       */
  int32_t use_bilateral_filter;/* settings->useBilateralFilter                */
  int32_t device;              /* HIP device ordinal; -1 = current device     */
  int32_t sync_status;         /* 1: dsr_process_frame synchronises and returns
                                  DSR_E_OUT_OF_BLOCKS itself (shim behaviour);
                                  0: fully asynchronous, poll dsr_get_stats   */
  int32_t view_pipeline;       /* (ABI 5) how the view's writers are queued (see "view pipeline" in DESIGN.md 6.5):
                                  DSR_VIEW_PIPELINE_AUTO (0: engines with sync_status get the shared form, others none —
                                  round 5's default; env DSR_PIPELINED_VIEW=0/1/2 overrides AUTO only), _OFF: one stream for
                                  everything — what dsr_engine_share_stream and dsr_batch_create need, also for an engine
                                  whose host waits for its status —, _PER_ENGINE, _SHARED (one view stream per GPU, on which the
                                  instance-sized volumes of that GPU also fuse: AUTO's choice for sync_status engines)   */
  int32_t reserved[6];
} dsr_settings;
#define DSR_VIEW_PIPELINE_AUTO 0
#define DSR_VIEW_PIPELINE_OFF 1
#define DSR_VIEW_PIPELINE_PER_ENGINE 2
#define DSR_VIEW_PIPELINE_SHARED 3

/* ITMIntrinsics::projectionParamsSimple + image size
 * (InfiniTamDriver.cpp:55-68). */
typedef struct dsr_intrinsics {
  float fx, fy, cx, cy;
  int32_t width, height;
} dsr_intrinsics;

/* ITMRGBDCalib (InfiniTamDriver.cpp:49-79): rgb and depth intrinsics, the
 * rgb->depth extrinsic and the affine disparity calibration (0.001, 0). */
typedef struct dsr_calib {
  dsr_intrinsics rgb;
  dsr_intrinsics depth;
  float trafo_rgb_to_depth[16]; /* column-major; identity in DynSLAM          */
  float disparity_calib[2];     /* depth_m = raw * [0] + [1]                  */
} dsr_calib;

/* ITMMainEngine::GetImageType values reached through
 * GetItmVisualization (InfiniTamDriver.cpp:16-34). */
typedef enum dsr_image_type {
  DSR_IMAGE_ORIGINAL_RGB = 0,
  DSR_IMAGE_ORIGINAL_DEPTH = 1,
  DSR_IMAGE_SCENERAYCAST = 2,                       /* kLatestRaycast */
  DSR_IMAGE_FREECAMERA_SHADED = 3,                  /* kGray          */
  DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME = 4,      /* kColor         */
  DSR_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL = 5,      /* kNormal        */
  DSR_IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT = 6,/* kWeight (fork) */
  DSR_IMAGE_FREECAMERA_DEPTH = 7                    /* kDepth (fork, float) */
} dsr_image_type;

/* Counters the host reads for its memory statistics
 * (InfiniTamDriver.h:237-250) plus launch-free status. */
typedef struct dsr_stats {
  int32_t num_allocated_voxel_blocks; /* scene->index.getNumAllocatedVoxelBlocks() == sdf_local_block_num */
  int32_t last_free_block_id;         /* scene->localVBA.lastFreeBlockId            */
  int32_t last_free_excess_list_id;
  int32_t no_visible_blocks;          /* renderState_live->noVisibleBlocks          */
  int32_t no_total_entries;           /* hash_bucket_num + excess_list_size         */
  int32_t voxel_bytes;                /* sizeof(ITMVoxel) == 8                      */
  int32_t block_voxels;               /* SDF_BLOCK_SIZE3 == 512                     */
  int32_t sticky_status;              /* DSR_OK or DSR_E_OUT_OF_BLOCKS seen so far  */
  int64_t decayed_block_count;        /* denseMapper->GetDecayedBlockCount()        */
  int64_t frames_processed;
  int32_t no_visible_blocks_freeview;
  int32_t host_store_slots;           /* ITMGlobalCache: 4 KiB block slots handed out so far (one per
                                         entry ever swapped out; 0 without use_swapping)           */
  int32_t host_store_capacity_slots;  /* slots the allocated (pinned) host store can hold          */
  int32_t reserved[3];
} dsr_stats;

typedef struct dsr_engine dsr_engine; /* opaque: one ITMMainEngine (scene + render
                                         states + view + tracking state)        */

/* ---- lifetime --------------------------------------------------------------- */

int dsr_abi_version(void);
/* Fills upstream ITMLibSettings/ITMSceneParams defaults (voxel 0.005, mu 0.02,
 * maxW 100, frustum [0.2,3.0] are upstream's indoor values; DynSLAM overrides
 * them: DynSLAMGUI.cpp:1214-1219). */
void dsr_default_settings(dsr_settings *s);
/* Thread-local text of the last failure. */
const char *dsr_last_error(void);

/* ITMMainEngine::ITMMainEngine(settings, calib, imgSize_rgb, imgSize_d)
 * (InfiniTamDriver.h:84-91; InstanceReconstructor.cpp:382-389).  Allocates the
 * scene (hash table, excess list, voxel block array), both render states, the
 * view and the tracking state, then resets the scene. */
int dsr_engine_create(const dsr_settings *settings, const dsr_calib *calib, dsr_engine **out);
void dsr_engine_destroy(dsr_engine *e);
/* denseMapper->ResetScene(scene) (InfiniTamDriver.h:282-284). */
int dsr_reset_scene(dsr_engine *e);
/* Blocks until all work queued on the engine's stream is done. */
int dsr_sync(dsr_engine *e);
/* Blocks until all work of EVERY engine of this process is done, on every device an engine was
 * created on (the host's per-frame "final sanity check": ITMSafeCall(cudaDeviceSynchronize()) +
 * cudaGetLastError(), DynSlam.cpp:163-172) — including tracking renders that dsr_prepare / dsr_batch_fuse deferred (ABI 5):
 * they are queued first, so call it from the thread that drives the engines.  Returns DSR_E_DEVICE if a device reports an error. */
int dsr_device_synchronize(void);
/* Free / total memory of a device in bytes (device < 0: the calling thread's current device): the GUI's memory read-out,
 * cudaMemGetInfo at DynSLAMGUI.cpp:912. */
int dsr_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* Stream ordering for the "_dev" entry points, WITHOUT host synchronisation.  Every engine
 * enqueues on its own private HIP stream and "_dev" calls return before the work has run, so a
 * caller that produces the inputs or consumes the outputs on another stream (torch's current
 * stream, the RCCL stream of the preview all-gather) must order the two:
 *   dsr_wait_for_stream(e, s)        — work queued on the engine AFTER this call starts only when
 *                                      everything queued on `s` so far has finished (call it before
 *                                      dsr_update_view_dev / dsr_set_view_float_dev when `s` wrote
 *                                      the buffers);
 *   dsr_stream_wait_for_engine(e, s) — work queued on `s` after this call starts only when
 *                                      everything queued on the engine so far has finished (call it
 *                                      after dsr_get_image_dev, before `s` reads the buffers).
 * `hip_stream` is a hipStream_t (NULL = the legacy default stream).  dsr_sync() remains the
 * blocking alternative. */
int dsr_wait_for_stream(dsr_engine *e, void *hip_stream);
int dsr_stream_wait_for_engine(dsr_engine *e, void *hip_stream);
/* (ABI 4) `e` queues its work on `owner`'s stream from now on (its own stream is released).  For ONE instance volume next to
 * the engine that holds the full frame on a GPU of their own — the one-volume-per-GPU layout: the view split, the fusion and
 * the renders of the pair are then ordered by one queue and the frame contains no cross-stream event.  Both engines on one
 * GPU, driven from one thread, `e` idle, neither with a pipelined view.  LIFETIME: the stream stays `owner`'s — `owner` must
 * outlive every CALL on `e` (a sync, a dump, a frame), not only its destruction; dsr_engine_destroy(e) itself is safe in either
 * order.  The same holds for the volumes of a batch (dsr_batch_create puts them on its source's stream and dsr_batch_destroy
 * does not give them streams of their own back) and for the batch handle: destroy it before the engines it names. */
int dsr_engine_share_stream(dsr_engine *e, dsr_engine *owner);

/* ---- volume batch (ABI 4): the instance volumes of ONE GPU driven together.  An instance frame is eight small launches; N
 * volumes one after the other — the reference's loop, InstanceReconstructor.cpp:315-361 — are N chains of them on a chip each
 * of them leaves almost idle.  A batch takes the volume as a grid dimension: dsr_batch_fuse does, for every listed instance in
 * the host's order, ProcessSilhouette + RemoveSilhouette (:238-263), SetPose, Integrate and PrepareNextStep (:569-700) in
 * 2 + 6 launches for ALL of them; dsr_batch_render their GetImage + GetFloatImage (:956-986) in two.  Results are those of the
 * per-volume calls, bit for bit.  The volumes (1..8, instance-sized: sdf_local_block_num <= 16384 behind a table no larger than
 * upstream's) live on `source`'s GPU and from then on share its stream; `source` holds the full frame the silhouettes are cut
 * from.  Per-volume calls on the same engines (dumps, statistics, decay, a single GetImage) remain valid in between. */
typedef struct dsr_batch dsr_batch;
typedef struct dsr_batch_item {
  int32_t volume;                   /* index into the batch's volumes; -1: the instance's volume lives elsewhere — its silhouette is only blanked here */
  int32_t x0, y0, box_w, box_h;     /* copy mask: bbox-local uint8 in HBM, placed at (x0, y0) */
  int32_t dx0, dy0, dbox_w, dbox_h; /* delete mask */
  int32_t reserved;
  const void *copy_mask_dev;        /* NULL with volume -1 */
  const void *delete_mask_dev;      /* NULL: nothing is blanked for this instance */
  float inv_m[16];                  /* camera -> object pose of the volume for this frame (dsr_set_pose_inv_m) */
} dsr_batch_item;
typedef struct dsr_batch_render_item {
  int32_t volume, reserved;
  void *rgba_out_dev, *depth_out_dev; /* HBM buffers of the caller (an exchange slot); either may be NULL */
  float pose_m[16];                   /* object -> camera pose of the free camera (dsr_get_image's pose_m) */
} dsr_batch_render_item;
int dsr_batch_create(dsr_engine *source, dsr_engine *const *volumes, int n_volumes, dsr_batch **out);
void dsr_batch_destroy(dsr_batch *b);
/* status_out (n_items, or NULL): per item the allocation status of its volume's frame — DSR_E_OUT_OF_BLOCKS where the fork
 * throws — for volumes created with sync_status; DSR_OK elsewhere.  The call itself fails only on bad arguments / device errors. */
int dsr_batch_fuse(dsr_batch *b, const dsr_batch_item *items, int n_items, int32_t *status_out);
int dsr_batch_render(dsr_batch *b, int type, const dsr_batch_render_item *items, int n_items);
/* (ABI 4) Restrict the CALLING thread to the CPUs next to `device` (< 0: the current one) — the PCI device's local_cpulist.
 * DynSLAM's host thread moves ~7.5 MB of frames / previews per frame through pinned memory and polls words the GPU writes; on
 * a multi-socket host that is cheaper from the GPU's own NUMA node.  DSR_OK and no effect where the list is not published. */
int dsr_pin_host_thread(int device);

/* ---- view ------------------------------------------------------------------- */

/* viewBuilder->UpdateView(&view, rgb, rawDepth, useBilateralFilter, ...)
 * (InfiniTamDriver.cpp:222-223): copies RGBA, converts int16 mm -> float m
 * (<=0 or >32000 -> -1), optional bilateral filter.  Host pointers. */
int dsr_update_view(dsr_engine *e, const uint8_t *rgba, const int16_t *depth_mm);
/* Same with inputs already resident in HBM (no PCIe copy). */
int dsr_update_view_dev(dsr_engine *e, const void *rgba_dev, const void *depth_mm_dev);
/* InfiniTamDriver::UpdateView as a whole (InfiniTamDriver.cpp:211-224): CvToItm(rgb_image) — packed BGR u8[3*W*H], the
 * cv::Mat3b the host holds — + CvToItm(raw_depth) + viewBuilder->UpdateView in ONE call: the frame goes up as the 3 + 2
 * bytes per pixel the host has (the RGBA form is 4 + 2 and costs the host a 465 750-pixel loop), the BGR -> RGBA
 * conversion (a = 255) and the depth conversion run in the ingest kernel.  Like dsr_update_view it returns without
 * waiting for the GPU: the frame is copied into a pinned staging slot (the caller's buffers are free on return). */
int dsr_update_view_bgr(dsr_engine *e, const uint8_t *bgr, const int16_t *depth_mm);
/* SetView() with an already converted view: RGBA + float depth in metres.  This
 * is what InstanceReconstructor builds per instance
 * (InstanceReconstructor.cpp:238-263,580) and what it writes back into the main
 * view after masking (:196-197). */
/* Depth values above 1e30 (+inf included) are stored as 1e30: every fusion result is the one the reference's
 * arithmetic gives for them (sdf update with +1, no colour), and dsr_get_view returns 1e30 for such pixels. */
int dsr_set_view_float(dsr_engine *e, const uint8_t *rgba, const float *depth_m);
int dsr_set_view_float_dev(dsr_engine *e, const void *rgba_dev, const void *depth_m_dev);
/* view->rgb / view->depth ->UpdateHostFromDevice()
 * (InstanceReconstructor.cpp:180-181; InfiniTamDriver.h:155-156).  Either
 * pointer may be NULL. */
int dsr_get_view(dsr_engine *e, uint8_t *rgba_out, float *depth_m_out);
/* InfiniTamDriver::PrepareNextStep's "Keep the OpenCV previews up to date" (InfiniTamDriver.h:154-156): the current view as
 * packed BGR bytes (ItmToCv, InfiniTamDriver.cpp:108-120) and int16 millimetres (ItmDepthToCv / FloatDepthmapToShort,
 * :128-144), converted on the GPU from the engine's device-resident view: two D2H copies, one synchronisation.  Either
 * pointer may be NULL. */
int dsr_get_view_previews(dsr_engine *e, uint8_t *bgr_out, int16_t *depth_mm_out);

/* Page-locks a HOST buffer the caller hands to the library again and again — the cv::Mat behind a driver's previews, a frame
 * buffer it refills every frame — (what cudaHostRegister is to a CUDA host): transfers to and from a pinned buffer go straight
 * between it and HBM on the I/O stream, without the staging copy every pageable buffer costs (2.3 MB per preview pair).  The
 * buffer must stay allocated until dsr_unpin_host_buffer.  Purely an optimisation: every entry point takes either kind. */
int dsr_pin_host_buffer(void *ptr, size_t bytes);
int dsr_unpin_host_buffer(void *ptr);

/* ---- pose (trackingState->pose_d) ------------------------------------------- */

/* pose_d->SetInvM(invM) (InfiniTamDriver.h:131-134): invM = camera->world. */
int dsr_set_pose_inv_m(dsr_engine *e, const float inv_m[16]);
/* pose_d->SetM(M): M = world->camera. */
int dsr_set_pose_m(dsr_engine *e, const float m[16]);
int dsr_get_pose(dsr_engine *e, float m_out[16], float inv_m_out[16]);

/* ---- fusion ------------------------------------------------------------------ */

/* denseMapper->SetFusionWeightParams(params) (InfiniTamDriver.h:138). */
int dsr_set_fusion_weight_params(dsr_engine *e, int depth_weighting);
/* denseMapper->ProcessFrame(view, trackingState, scene, renderState_live)
 * (InfiniTamDriver.h:140-145): AllocateSceneFromDepth + IntegrateIntoScene
 * (+ swapping if enabled). */
int dsr_process_frame(dsr_engine *e);
/* The two halves of ProcessFrame, for tests and per-stage timing. */
int dsr_allocate_scene_from_depth(dsr_engine *e);
int dsr_integrate_into_scene(dsr_engine *e);
/* trackingController->Prepare(trackingState, view, renderState_live)
 * (InfiniTamDriver.h:152): CreateExpectedDepths + CreateICPMaps.  A no-op when
 * noVisibleBlocks == 0 (InfiniTamDriver.h:150).
 * (ABI 5) For an instance-sized volume the raycast + ICP maps of this call are DEFERRED: they are queued by the next call that
 * names this engine — together with the preview raycast, as one launch, when that call is dsr_get_image_dev /
 * dsr_exchange_render_slot from a free camera (the reference's order per instance: Integrate, PrepareNextStep, later GetImage);
 * every other call queues them first.  No caller can observe the difference through this API: results, stream order as seen by
 * dsr_stream_wait_for_engine / dsr_sync, dumps.  dsr_batch_fuse defers the tracking render of its volumes the same way (the
 * next dsr_batch_render pairs it, any other call on the source or a volume queues it).  env DSR_PAIR_RENDER=0: queued at once. */
int dsr_prepare(dsr_engine *e);

/* denseMapper->Decay(scene, renderState, maxWeight, minAge, forceAllVoxels)
 * (InfiniTamDriver.h:201-235): voxel garbage collection. */
int dsr_decay(dsr_engine *e, int max_weight, int min_age, int force_all_voxels);

/* ---- rendering --------------------------------------------------------------- */

/* ITMMainEngine::GetImage(out, outFloat, type, pose, intrinsics)
 * (InfiniTamDriver.cpp:178-183,202-207).  pose_m: world->camera of the free
 * camera (NULL: current pose_d); intrinsics: fx,fy,cx,cy (NULL: depth calib).
 * rgba_out (4*W*H bytes) and/or depth_out (W*H floats; metres, 0 = miss) are HOST
 * buffers; either may be NULL. */
int dsr_get_image(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4],
                  uint8_t *rgba_out, float *depth_out);
/* Same, results left in caller-provided HBM buffers (for the multi-GPU
 * composite: the RCCL all-gather reads them in place). */
int dsr_get_image_dev(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4],
                      void *rgba_out_dev, void *depth_out_dev);

/* ---- edges of the path: depth ingest and instance view split ("next" rows) -- */

/* DepthProvider::DepthFromDisparityMap (src/DynSLAM/DepthProvider.h:94-137):
 *   depth_mm = (int32)(1000.0f * scale * ((baseline_m * focal_px) / disparity));
 *   |disparity| < 1e-5 -> 0;  depth_mm outside [ (int32)(min_depth_m*1000), (int32)(max_depth_m*1000) ] -> 0
 * disparity: float[n] (DispNet .pfm / ELAS), out: int16 mm — the format dsr_update_view takes.
 * The _dev variant works on HBM buffers of `device` and enqueues on hip_stream (NULL = default). */
int dsr_depth_from_disparity(const float *disparity, int16_t *depth_mm_out, int n, float baseline_m, float focal_px,
                             float scale, float min_depth_m, float max_depth_m);
int dsr_depth_from_disparity_dev(int device, void *hip_stream, const void *disparity_dev, void *depth_mm_out_dev,
                                 int n, float baseline_m, float focal_px, float scale, float min_depth_m,
                                 float max_depth_m);

/* PrecomputedDepthProvider::ReadPrecomputed (src/DynSLAM/PrecomputedDepthProvider.cpp:22-75): the two on-disk
 * formats of precomputed maps, read into caller-provided host buffers of `capacity` elements
 * (*width / *height are set even when the map does not fit: DSR_E_ARG then — call with capacity 0 to
 * query the size; DSR_E_IO for a missing or malformed file — the reference throws std::runtime_error at
 * :40,:43,:47-52 — leaves them 0 x 0.  Nothing is ever written past `capacity` elements.)
 *   dsr_read_depth_xml  OpenCV FileStorage XML dump of a CV_16SC1 matrix under the node "depth-frame"
 *                       (:35-45; the ELAS depth maps, Input.h:71-78 "%04d.xml"): <rows>, <cols>, <dt>s</dt>,
 *                       <data> as whitespace-separated decimals, row-major.  Any other <dt> is the
 *                       reference's "wrong format" error.
 *   dsr_read_pfm        single-channel PFM ("Pf" header, width height, scale; scale < 0 = little endian) as
 *                       written by DispNet (:27-31, Input.h:112-118 "%06d.pfm"); PFM stores the BOTTOM row
 *                       first, the map is returned top row first (what pfmLib's ReadFilePFM hands to OpenCV).
 *                       pfmLib is an empty submodule of the reference: the public format is followed, the
 *                       values are returned as stored (no multiplication by |scale|).
 *   dsr_clip_depth_mm   the `input_is_depth_` clamp (:55-74) for int16 maps: depth > (int16)round(max_depth_m
 *                       * 1000) -> 0, in place; the _dev variant works on an HBM buffer and enqueues on
 *                       hip_stream. */
int dsr_read_depth_xml(const char *path, int16_t *depth_mm_out, int capacity, int *width, int *height);
int dsr_read_pfm(const char *path, float *out, int capacity, int *width, int *height);
int dsr_clip_depth_mm(int16_t *depth_mm, int n, float max_depth_m);
int dsr_clip_depth_mm_dev(int device, void *hip_stream, void *depth_mm_dev, int n, float max_depth_m);

/* The host's layout shims at the boundary (InfiniTamDriver.cpp:81-144),
 * as kernels, so that frames and previews can stay in HBM in the host's own formats:
 *   dsr_bgr_to_rgba   CvToItm(cv::Mat3b): packed BGR u8[3n] -> RGBA u8[4n], a = 255   (:81-100)
 *   dsr_rgba_to_bgr   ItmToCv(ITMUChar4Image): RGBA u8[4n] -> packed BGR u8[3n]      (:108-120)
 *   dsr_depth_m_to_mm FloatDepthmapToShort: (int16)(metres * 1000)                   (:128-139);
 *                     C leaves out-of-range float->int16 undefined: defined here, as everywhere
 *                     in this library, as the saturating float->int32 conversion wrapped to 16 bits
 * The plain variants take host buffers; the _dev variants work on HBM buffers of `device` and
 * enqueue on hip_stream (NULL = default stream). */
int dsr_bgr_to_rgba(const uint8_t *bgr, uint8_t *rgba_out, int n);
int dsr_bgr_to_rgba_dev(int device, void *hip_stream, const void *bgr_dev, void *rgba_out_dev, int n);
int dsr_rgba_to_bgr(const uint8_t *rgba, uint8_t *bgr_out, int n);
int dsr_rgba_to_bgr_dev(int device, void *hip_stream, const void *rgba_dev, void *bgr_out_dev, int n);
int dsr_depth_m_to_mm(const float *depth_m, int16_t *depth_mm_out, int n);
int dsr_depth_m_to_mm_dev(int device, void *hip_stream, const void *depth_m_dev, void *depth_mm_out_dev, int n);

/* ProcessSilhouette_CPU (InstanceReconstructor.cpp:59-133) on the GPU: the view of `instance`
 * becomes the pixels of `main`'s current view that lie under the copy mask, everything else
 * rgba (255,255,255,255) / depth 0.  mask: HOST uint8[box_h][box_w] (1 = copy), placed at
 * (x0,y0) in the frame (Mask::GetBoundingBox; may stick out of the frame).  Both engines must
 * have the same image size.  They may live on DIFFERENT GPUs (one volume per GPU): the cut-out is
 * produced on main's GPU and sent to the instance's with one peer copy over xGMI, stream-ordered on
 * both sides.  Replaces the D2H -> CPU loop -> H2D round trip of InstanceReconstructor.cpp:180-197,238-263. */
int dsr_view_extract_silhouette(dsr_engine *main_engine, dsr_engine *instance, const uint8_t *mask, int x0, int y0,
                                int box_w, int box_h);
/* RemoveSilhouette_CPU (InstanceReconstructor.cpp:135-170): pixels of the engine's view under
 * the mask become rgba 0 / depth 0.0f. */
int dsr_view_remove_silhouette(dsr_engine *e, const uint8_t *mask, int x0, int y0, int box_w, int box_h);
/* The same two steps with the mask ALREADY IN HBM (e.g. the segmentation network's output, or masks uploaded ahead
 * of the frame): nothing is copied and nothing synchronises — the host variants above must wait for the stream before
 * returning because the caller may reuse its pageable mask buffer.  The mask must stay unmodified until the engine's
 * stream has passed the call (dsr_sync / dsr_stream_wait_for_engine). */
int dsr_view_extract_silhouette_dev(dsr_engine *main_engine, dsr_engine *instance, const void *mask_dev, int x0, int y0,
                                    int box_w, int box_h);
int dsr_view_remove_silhouette_dev(dsr_engine *e, const void *mask_dev, int x0, int y0, int box_w, int box_h);
/* (ABI 4) Both steps of ONE instance in one launch: ProcessSilhouette_CPU with the copy mask, then RemoveSilhouette_CPU with
 * the delete mask (InstanceReconstructor.cpp:238-263 calls them back to back; the reference scales the two masks differently,
 * Utils/Mask.cpp:21-46).  The cut-out sees the pixel as it is before THIS instance's blanking, exactly as the two calls in
 * that order; results are identical to them.  `_dev`: masks already in HBM (may be one and the same buffer). */
int dsr_view_split_silhouette(dsr_engine *main_engine, dsr_engine *instance, const uint8_t *copy_mask, int x0, int y0, int box_w,
                              int box_h, const uint8_t *delete_mask, int dx0, int dy0, int dbox_w, int dbox_h);
int dsr_view_split_silhouette_dev(dsr_engine *main_engine, dsr_engine *instance, const void *copy_mask_dev, int x0, int y0,
                                  int box_w, int box_h, const void *delete_mask_dev, int dx0, int dy0, int dbox_w, int dbox_h);

/* ---- instance compositing (the fused preview) ------------------------------------ */

/* InstanceReconstructor::CompositeInstances (InstanceReconstructor.cpp:933-990) and
 * CompositeInstanceDepthMaps (:911-931) on n_layers instance renders of n_pixels pixels:
 *   - if dim_background: target colour *= (1.0 - 0.10f) in double, truncated (:945-954);
 *   - for each layer IN THE GIVEN ORDER (the host iterates tracks by ascending id): the
 *     instance wins a pixel iff s != 0 && (t == 0 || t > s) (CompositeColor :875-908),
 *     then t = s and colour = min(255, c*(1.0+0.5-tint_strength) + tint*tint_strength) with
 *     tint = kMatplotlib2Palette[track_id % 10] (:44-55), double arithmetic, truncated.
 *   - target_rgba == NULL: depth only (CompositeDepth :851-871; same depth result).
 * layers_rgba: [n_layers][n_pixels][4] bytes, layers_depth: [n_layers][n_pixels] floats
 * (metres, 0 = miss); target_*: [n_pixels], updated in place; track_ids: host int32[n_layers].
 * The _dev variant takes HBM pointers on `device` and enqueues on `hip_stream` (a
 * hipStream_t, NULL = the default stream) without synchronising — it is what runs after
 * the RCCL all-gather of the per-GPU raycast buffers.  The host variant synchronises. */
int dsr_composite_instances_dev(int device, void *hip_stream, void *target_rgba_dev, void *target_depth_dev,
                                const void *layers_rgba_dev, const void *layers_depth_dev,
                                const int32_t *track_ids, int n_layers, int n_pixels, float tint_strength,
                                int dim_background);
/* ... with one pointer PER LAYER (host arrays of HBM pointers; layer_rgba_ptrs may be NULL for depth only): the layers
 * are composited where the collective left them — any slot of the all-gathered exchange buffer, in track-id order —
 * without first being copied into one contiguous array. */
int dsr_composite_layer_ptrs_dev(int device, void *hip_stream, void *target_rgba_dev, void *target_depth_dev,
                                 const void *const *layer_rgba_ptrs, const void *const *layer_depth_ptrs,
                                 const int32_t *track_ids, int n_layers, int n_pixels, float tint_strength,
                                 int dim_background);
int dsr_composite_instances(uint8_t *target_rgba, float *target_depth, const uint8_t *layers_rgba,
                            const float *layers_depth, const int32_t *track_ids, int n_layers, int n_pixels,
                            float tint_strength, int dim_background);

/* ---- multi-GPU: one volume per GPU, the fused preview exchanged over RCCL / xGMI ---------------
 *
 * The reference keeps one InfiniTamDriver per tracked instance (InstanceReconstructor::InitializeReconstruction,
 * InstanceReconstructor.cpp:363-392) next to the static map's, all on one GPU, and fuses / previews them one after
 * the other (:315-361, CompositeInstances :933-990).  Volumes share no data, so each engine may live on its own GPU
 * (dsr_settings.device); the ONE exchange of the path is the fused preview: every volume is raycast from the shared
 * camera on its own GPU, the per-volume layers (float depth plane + RGBA plane = 8 bytes per pixel) are ALL-GATHERED
 * and z-composited where the preview is consumed.  A dsr_exchange owns the layer buffers, the RCCL communicator(s) and
 * one stream per GPU; it calls RCCL itself (librccl is loaded on first use), so a C / C++ host needs nothing else.
 *
 * Ranks: a rank is one slot owner — one GPU's worth of volumes.  Two ways to create the exchange:
 *   dsr_exchange_create       ONE process drives all GPUs (DynSLAM's C++ host): rank r lives on devices[r]; ranks may
 *                             share a device (their layers are then already in place there; with a single device no
 *                             communicator is created at all).  Distinct devices: ncclCommInitAll, grouped all-gather.
 *   dsr_exchange_create_rank  one process per GPU (`torch.distributed`-style launch): rank 0 calls
 *                             dsr_exchange_unique_id, hands the 128 bytes to every rank by its own means, all ranks call
 *                             dsr_exchange_create_rank (ncclCommInitRank) collectively.
 * Every rank contributes `slots_per_rank` layers of n_pixels pixels (unused slots stay empty: depth 0 never wins a
 * pixel).  Calls on one exchange come from one thread at a time. */
typedef struct dsr_exchange dsr_exchange;
int dsr_exchange_create(const int32_t *devices, int n_ranks, int slots_per_rank, int n_pixels, dsr_exchange **out);
int dsr_exchange_unique_id(uint8_t id_out[128]);
int dsr_exchange_create_rank(const uint8_t unique_id[128], int world_size, int rank, int device, int slots_per_rank,
                             int n_pixels, dsr_exchange **out);
void dsr_exchange_destroy(dsr_exchange *x);
/* The exchange's hipStream_t on the GPU of local rank `rank` (gather and composite run on it), for callers that order
 * other streams against it with dsr_wait_for_stream / dsr_stream_wait_for_engine.  NULL for a rank of another process. */
void *dsr_exchange_stream(dsr_exchange *x, int rank);
/* HBM addresses of slot `slot` of local rank `rank`: where that rank's renders go (dsr_get_image_dev's two outputs). */
int dsr_exchange_slot_ptrs(dsr_exchange *x, int rank, int slot, void **rgba_dev, void **depth_dev);
/* ... and of the gathered copy of (rank, slot) on the GPU of local rank `on_rank` (valid after dsr_exchange_gather). */
int dsr_exchange_layer_ptrs(dsr_exchange *x, int on_rank, int rank, int slot, void **rgba_dev, void **depth_dev);
/* GetImage(colour) + GetFloatImage(depth) of one volume from the preview camera (InstanceReconstructor.cpp:923,968:
 * the model view composed with the instance's pose) straight into slot `slot` of local rank `rank`, ordered after the
 * previous gather's reads of that slot and before the next gather — no host synchronisation.  `e` must live on the
 * rank's GPU.  e == NULL: the slot becomes an empty layer (an instance that is not visible in this frame). */
int dsr_exchange_render_slot(dsr_exchange *x, int rank, int slot, dsr_engine *e, int type, const float pose_m[16],
                             const float intrinsics[4]);
/* The all-gather of every rank's layers (one collective; nothing for a single device).  Collective in rank mode. */
int dsr_exchange_gather(dsr_exchange *x);
/* (ABI 4) Which collective dsr_exchange_gather runs between the GPUs: 0 = the in-place all-gather (every GPU ends up with every
 * layer), 1 = a gather to the GPU of `root_rank` only (ncclSend / ncclRecv in one group: the composite has ONE consumer, so every
 * link but the root's carries 1/N of the bytes).  The composite must then run on that rank.  No effect on a one-GPU exchange. */
int dsr_exchange_set_collective(dsr_exchange *x, int gather_to_root, int root_rank);
/* (ABI 4) Measurement: returns (and resets) the HIP-event time the collectives and the composites of this process took on their
 * exchange streams since the last call, and switches the timing on / off for what follows (`enable`).  Waits for that work. */
int dsr_exchange_timing(dsr_exchange *x, int enable, double *gather_ms, double *composite_ms, int32_t *n_gathers, int32_t *n_composites);
/* CompositeInstances on the GPU of local rank `root_rank`: layers (ranks[i], slots[i]) with track_ids[i], in the
 * given order (the host's ascending track ids), over the target — arithmetic of dsr_composite_layer_ptrs_dev.
 * target_*_dev == NULL: the exchange's own target pair on that GPU (dsr_exchange_target_ptrs: render the static map
 * into it first, or clear it).  target_engine (may be NULL): the engine that rendered the target — the composite is
 * ordered after its queued work and its later work after the composite. */
int dsr_exchange_composite(dsr_exchange *x, int root_rank, dsr_engine *target_engine, void *target_rgba_dev,
                           void *target_depth_dev, const int32_t *ranks, const int32_t *slots, const int32_t *track_ids,
                           int n_layers, float tint_strength, int dim_background);
/* dsr_exchange_gather + dsr_exchange_composite (ranks without the root: root_rank < 0 or not local — gather only). */
int dsr_exchange_gather_and_composite(dsr_exchange *x, int root_rank, dsr_engine *target_engine, void *target_rgba_dev,
                                      void *target_depth_dev, const int32_t *ranks, const int32_t *slots,
                                      const int32_t *track_ids, int n_layers, float tint_strength, int dim_background);
/* The exchange's own composite target on the GPU of local rank `rank` (RGBA + float depth, n_pixels each), its
 * clearing (no static map: instances over an empty frame) and its read-back (synchronises).  (ABI 5) The clear is carried
 * out by the next composite over the target (which then does not read it), or by _target_ptrs / _read_target: buffers looked at
 * through pointers obtained EARLIER show it only after one of these. */
int dsr_exchange_target_ptrs(dsr_exchange *x, int rank, void **rgba_dev, void **depth_dev);
int dsr_exchange_clear_target(dsr_exchange *x, int rank);
int dsr_exchange_read_target(dsr_exchange *x, int rank, uint8_t *rgba_out, float *depth_out);
/* Blocks until the exchange's streams are idle. */
int dsr_exchange_sync(dsr_exchange *x);

/* ---- statistics / parity dumps ------------------------------------------------ */

/* renderState_live->noVisibleBlocks, which the host reads right after fusion (InfiniTamDriver.h:150).  The value travels
 * with the status word: after a dsr_process_frame with `sync_status` (or a dsr_get_stats) it is known on the host and this
 * call does not touch the device; otherwise it costs one 12-byte read-back. */
int dsr_get_no_visible_blocks(dsr_engine *e, int32_t *out);
/* Synchronises. */
int dsr_get_stats(dsr_engine *e, dsr_stats *out);

/* Copies of the engine state for parity tests against the oracle.  All host
 * buffers, caller-sized:
 *   hash entries      : no_total_entries * sizeof(dsr_hash_entry)
 *   visible ids       : up to sdf_local_block_num int32 (returns count via *n)
 *   visible types     : no_total_entries uint8
 *   voxel blocks      : sdf_local_block_num * 512 * sizeof(dsr_voxel)  (AoS)
 *   allocation lists  : sdf_local_block_num int32, excess_list_size int32
 */
int dsr_dump_hash_table(dsr_engine *e, dsr_hash_entry *out);
int dsr_dump_visible_list(dsr_engine *e, int freeview, int32_t *ids_out, int32_t *n);
int dsr_dump_visible_types(dsr_engine *e, uint8_t *out);
int dsr_dump_voxel_blocks(dsr_engine *e, int first_block, int n_blocks, dsr_voxel *out);
int dsr_dump_allocation_lists(dsr_engine *e, int32_t *voxel_alloc_list, int32_t *excess_alloc_list);
/* Render-state buffers: which = 0 live, 1 freeview.  Any pointer may be NULL.
 *   minmax        : 2 * ceil(W/8) * ceil(H/8) floats (renderingRangeImage)
 *   raycast_result: 4*W*H floats (voxel units, w = found)
 *   points/normals: 4*W*H floats (trackingState->pointCloud; live only)
 *   raycast_image : 4*W*H bytes   (The range image of the LIVE view is computed right after the visible list,
 * under the integration — so between dsr_process_frame and dsr_prepare a dump already shows the new frame's image; the
 * reference's structure of the same name is internal to its visualisation engine and never read by the host.  For the same
 * reason, after dsr_process_frame calls that were not followed by dsr_prepare, a frame WITHOUT visible blocks — dsr_prepare is
 * skipped, the image "keeps its previous contents" — leaves the image of the last dsr_process_frame here, where the serial engine
 * has the image of its last Prepare(); nothing reads it before the next frame with visible blocks rebuilds it: DESIGN.md 5.) */
int dsr_dump_render_state(dsr_engine *e, int which, float *minmax, float *raycast_result,
                          float *points, float *normals, uint8_t *raycast_image);

/* Host swapping (settings.use_swapping; ITMSwappingEngine + ITMGlobalCache, run inside
 * dsr_process_frame after integration: IntegrateGlobalIntoLocal then SaveToGlobalMemory, at
 * most DSR_TRANSFER_BLOCK_NUM blocks each way per frame).  Parity dumps:
 *   states     : no_total_entries uint8 — ITMHashSwapState::state (0 host only / most recent on
 *                host, 1 both: needs merge, 2 device most recent)
 *   has_stored : no_total_entries uint8 — ITMGlobalCache::HasStoredData(entry)
 * dsr_dump_stored_block copies the host-side copy of one entry's block (AoS); *present = 0 and
 * `out` untouched when the global cache holds nothing for it. */
int dsr_dump_swap_state(dsr_engine *e, uint8_t *states, uint8_t *has_stored);
int dsr_dump_stored_block(dsr_engine *e, int entry, dsr_voxel *out, int *present);

/* ---- self-test ------------------------------------------------------------------- */

/* Compares the engine's shared-reciprocal division (dsr_device.h: the IEEE fp32 divide
 * sequence without its scale/fix-up instructions, used on the integrate path) with the
 * device's IEEE `/` on n pseudo-random "tame" operand pairs, plus the exhaustive small
 * domains short/32767 and uchar/255.  *mismatches receives the number of differing bit
 * patterns (must be 0).  Synchronises. */
int dsr_selftest_division(int device, uint64_t n, uint64_t seed, uint64_t *mismatches);

/* Device-to-device bandwidth of `device` as THIS library's float4 grid-stride copy kernel
 * measures it (the kernel MI355X_MICROARCH.md quotes 6.29 TB/s for): `bytes` read + `bytes`
 * written per pass, `iters` timed passes after one warm-up; *gbps_out = 2*bytes*iters / time.
 * The roofline harness reports it next to the nominal 8 TB/s.  Allocates 2*bytes of HBM
 * for the duration of the call; synchronises. */
int dsr_measure_copy_bandwidth(int device, uint64_t bytes, int iters, double *gbps_out);
/* (ABI 5) The same kernel as the roofline's DENOMINATOR: `bytes` per direction (bench.py: 4 GiB), clocks warmed by ~50 ms of
 * copies first, six launch shapes (4 / 8 / 16 workgroups per CU x plain / non-temporal accesses) timed launch by launch,
 * `repeats` (1..64) rounds; a round's figure is its best launch.  out = {max, median, min} over the rounds, GB/s. */
int dsr_measure_copy_bandwidth_spread(int device, uint64_t bytes, int repeats, double out[3]);

/* ---- per-kernel timing (roofline harness) -------------------------------------- */

typedef struct dsr_kernel_time {
  char name[32];
  double total_ms;   /* sum of HIP-event durations on the engine's stream */
  int64_t launches;
  double bytes;      /* algorithmic bytes accumulated (SURVEY.md 8d model: the reference's AoS voxels) */
  double bytes_layout; /* integrate: compulsory bytes of the plane-wise layout actually used (DESIGN.md
                          "byte model"), from the kernel's own tallies; 0 for the other kernels       */
  double units;      /* integrate: visible blocks processed over all launches                     */
  double store_lanes;   /* integrate: the kernel's own tallies behind bytes_layout — lanes (= x rows of a block) that wrote
                           their 24 B of sdf + w_depth back, ...                                       */
  double colour_voxels; /* ... voxels that went through the colour update (8 B each)                   */
} dsr_kernel_time;
/* enable == 1 brackets every kernel launch with HIP events, enable == 2 only the two
 * dominant kernels (integrate, raycast); 0 = off.  Events cost ~3 us per bracketed launch
 * (9 % of a 5 mm frame when every kernel is bracketed): for measurement only. */
int dsr_profile_enable(dsr_engine *e, int enable);
int dsr_profile_reset(dsr_engine *e);
/* Returns the number of records written (<= cap). */
int dsr_profile_get(dsr_engine *e, dsr_kernel_time *out, int cap);

/* ---- meshing ---------------------------------------------------
 * Replaces ITMMeshingEngine<TVoxel,TIndex>::MeshScene(mesh, scene) + ITMMesh::WriteOBJ as used by
 * ITMMainEngine::SaveSceneToMesh (DynSlam.cpp:188-196 SaveStaticMap) and
 * InstanceReconstructor::SaveObjectToMesh (InstanceReconstructor.cpp:736-763).
 * Marching cubes over every allocated block in ascending entry order, voxels z/y/x, triangles in
 * table order (the serial order of the _CPU engine; upstream's CUDA engine appends with atomics,
 * i.e. in arbitrary order).  A cell is meshed when all 8 corners exist and none is at the
 * initial sdf (+1.0).  Vertices are in metres.  The triangle table is generated
 * (tools/gen_mc_tables.py): same edge table and surface as the classic tables, consistent on
 * ambiguous faces; the order of the triangles inside a cell may differ from upstream's. */
typedef struct dsr_triangle { float p0[3], p1[3], p2[3]; } dsr_triangle; /* ITMMesh::Triangle */
/* ITMMesh(memoryType, maxBlocks): noMaxTriangles = sdf_local_block_num * 32; MeshScene keeps the
 * first noMaxTriangles - 1 triangles.  The mesh stays on the device until dsr_mesh_free or the
 * next dsr_mesh_scene.  n_triangles may be NULL. */
int dsr_mesh_scene(dsr_engine *e, uint64_t *n_triangles);
/* copies triangles [first, first+count) of the current mesh to host memory */
int dsr_mesh_get(dsr_engine *e, dsr_triangle *out, uint64_t first, uint64_t count);
/* ITMMesh::WriteOBJ: "v x y z" x3 per triangle ("%f"), then "f 3i+3 3i+2 3i+1" (1-based) */
int dsr_mesh_write_obj(dsr_engine *e, const char *path);
int dsr_mesh_free(dsr_engine *e);
/* ITMMainEngine::SaveSceneToMesh(path) = MeshScene + WriteOBJ + release */
int dsr_save_scene_to_mesh(dsr_engine *e, const char *path);

#ifdef __cplusplus
}
#endif
#endif /* DSR_H_ */
