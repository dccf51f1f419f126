/*
 * ptgpu.h -- C ABI of the MI355X render path for PathTracerRust.
 *
 * The reference crate has no FFI layer; this header is the seam a Rust
 * `extern "C"` block (linked by build.rs against libptgpu.so) binds.  Every
 * entry point names the reference interface it stands in for (file:line under
 * the reference crate's root).
 *
 * Conventions
 *   - plain-old-data only: int32/uint32/uint64/double/float, no C++ types;
 *   - every input pointer is borrowed for the duration of the call;
 *   - every function returns PT_OK (0) or a negative pt_status, never throws;
 *   - one caller thread per pt_ctx (the reference has a single dispatcher
 *     thread, src/lib.rs:407-478);
 *   - geometry travels as f64, the reference's arithmetic type
 *     (src/math/vec3.rs:4-8).
 *
 * The scene travels as a serialised copy of the reference's object graph
 * (`Arc<dyn Hitable>` tree, src/hitable.rs:27-36): one pt_hitable per node,
 * children by index.  The library flattens that graph into its device layout
 * on upload; a CPU oracle can interpret the same graph directly.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 2

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID_ARG = -1,
    PT_ERR_UNSUPPORTED_SCENE = -2, /* graph shape the device flattener does not take */
    PT_ERR_NO_DEVICE = -3,
    PT_ERR_HIP = -4,          /* a HIP runtime call failed; see pt_last_error */
    PT_ERR_NO_SCENE = -5,
    PT_ERR_NO_CAMERA = -6,
    PT_ERR_OUT_OF_MEMORY = -7,
    PT_ERR_TOO_LARGE = -8,
    PT_ERR_MISSING_ASSET = -9, /* a named recipe needs an image that was not registered */
    PT_ERR_COMM = -10          /* an RCCL call failed; see pt_last_error */
} pt_status;

/* ------------------------------------------------------------------------ */
/* Scene graph wire format                                                   */
/* ------------------------------------------------------------------------ */

/* One node of the Hitable graph (src/hitable.rs:27-36 and its implementors). */
typedef enum pt_hitable_kind {
    PT_H_SPHERE = 1,        /* src/sphere.rs:27-41     f: center xyz, radius */
    PT_H_MOVING_SPHERE = 2, /* src/sphere.rs:104-133   f: center0 xyz, radius, center_range xyz, time0, time_range */
    PT_H_RECT = 3,          /* src/rect.rs:13-61       f: amin amax bmin bmax c ; axis */
    PT_H_BOX = 4,           /* src/axis_aligned_box.rs:14-29  f: pmin xyz, pmax xyz ; list = its 6 face nodes */
    PT_H_LIST = 5,          /* src/hitable.rs:40-78    list = children in order */
    PT_H_FLIP = 6,          /* src/hitable.rs:80-105   child[0] */
    PT_H_TRANSLATE = 7,     /* src/hitable.rs:107-138  child[0] ; f: translation xyz */
    PT_H_ROTATE_Y = 8,      /* src/hitable.rs:140-216  child[0] ; f: sin, cos, bbox min xyz, bbox max xyz */
    PT_H_MEDIUM = 9,        /* src/volume.rs:7-22      child[0] = boundary ; f: density ; material = Isotropic phase fn */
    PT_H_BVH = 10           /* src/bvh.rs:6-10         child[0]=left child[1]=right ; f: bbox min xyz, max xyz */
} pt_hitable_kind;

typedef struct pt_hitable {
    int32_t kind;        /* pt_hitable_kind */
    int32_t material;    /* index into materials, -1 if the node carries none */
    int32_t child[2];    /* node indices, -1 if unused */
    int32_t list_start;  /* LIST/BOX: first entry in child_index[] */
    int32_t list_count;  /* LIST/BOX: number of entries */
    int32_t axis;        /* RECT: 0 = X, 1 = Y, 2 = Z (src/rect.rs:7-11) */
    int32_t reserved;
    double f[10];        /* per-kind parameters, see pt_hitable_kind */
} pt_hitable;

typedef enum pt_material_kind {
    PT_M_DUMMY = 0,         /* src/material.rs:152-162 */
    PT_M_LAMBERTIAN = 1,    /* src/material.rs:253-306  texture, emissive */
    PT_M_DIFFUSE_LIGHT = 2, /* src/material.rs:308-332  texture */
    PT_M_METAL = 3,         /* src/material.rs:215-251  albedo, fuzz (already clamped to <= 1) */
    PT_M_DIELECTRIC = 4,    /* src/material.rs:164-213  ref_idx */
    PT_M_ISOTROPIC = 5      /* src/material.rs:334-352  texture */
} pt_material_kind;

typedef struct pt_material {
    int32_t kind;      /* pt_material_kind */
    int32_t texture;   /* index into textures, -1 if unused */
    double albedo[3];  /* METAL */
    double fuzz;       /* METAL */
    double ref_idx;    /* DIELECTRIC */
    double emissive;   /* LAMBERTIAN */
} pt_material;

typedef enum pt_texture_kind {
    PT_T_CONSTANT = 0, /* src/texture.rs:12-28   colour */
    PT_T_CHECKER = 1,  /* src/texture.rs:30-56   even, odd texture indices */
    PT_T_NOISE = 2,    /* src/texture.rs:58-75   scale */
    PT_T_IMAGE = 3     /* src/texture.rs:77-114  width, height, texel_offset (RGB8 rows top-down as decoded) */
} pt_texture_kind;

typedef struct pt_texture {
    int32_t kind;          /* pt_texture_kind */
    int32_t even;          /* CHECKER */
    int32_t odd;           /* CHECKER */
    int32_t width;         /* IMAGE */
    int32_t height;        /* IMAGE */
    int32_t reserved;
    uint64_t texel_offset; /* IMAGE: byte offset into texels[] */
    double colour[3];      /* CONSTANT */
    double scale;          /* NOISE */
} pt_texture;

/* The four lazily built Perlin tables (src/math/noise.rs:78-83). */
typedef struct pt_perlin {
    double ranvec[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} pt_perlin;

typedef struct pt_scene_desc {
    uint32_t struct_size;        /* sizeof(pt_scene_desc) */
    int32_t root;                /* node the renderer calls hit() on: SceneState.world (src/trace.rs:46) */
    const pt_hitable *hitables;
    int32_t n_hitables;
    int32_t n_child_index;
    const int32_t *child_index;  /* LIST/BOX children */
    const pt_material *materials;
    const pt_texture *textures;
    int32_t n_materials;
    int32_t n_textures;
    const uint8_t *texels;       /* RGB8 blobs of IMAGE textures */
    uint64_t n_texel_bytes;
    const pt_perlin *perlin;     /* NULL when no NOISE texture is present */
    const int32_t *lights;       /* nodes of the light-shape HitableList (src/trace.rs:123-128) */
    int32_t n_lights;
    int32_t reserved;
} pt_scene_desc;

/* The nine Camera::new parameters (src/camera.rs:25-26). */
typedef struct pt_camera {
    double origin[3];
    double look_at[3];
    double vup[3];
    double vfov;       /* degrees */
    double aspect;
    double aperture;
    double focus_dist;
    double time0;
    double time1;
} pt_camera;

typedef struct pt_ray {
    double origin[3];
    double direction[3]; /* not normalised (src/math/ray.rs:5) */
    double time;
} pt_ray;

/* HitRecord (src/hitable.rs:5-12) plus the node that produced it. */
typedef struct pt_hit {
    double t;         /* undefined when prim < 0 */
    double u, v;
    double p[3];
    double normal[3];
    int32_t prim;     /* graph index of the geometry node that was hit (sphere, rect, or medium); -1 = miss */
    int32_t material; /* index into materials */
} pt_hit;

/* ------------------------------------------------------------------------ */
/* Device context                                                            */
/* ------------------------------------------------------------------------ */

typedef struct pt_ctx pt_ctx;

/* How the library runs its passes. Every field: 0 = the library's default. None of them changes a pixel; they exist
 * for A/B measurements and for tests that compare the forms. (The environment variable PTGPU_TUNING,
 * "field=value,field=value", overrides fields by name in an unmodified host.) */
typedef struct pt_tuning {
    uint32_t pass_form;            /* pt_pass_form to use where the scene allows it; 0 = the library chooses */
    uint32_t pool_refill;          /* path pool: free path slots (1..64) at which a wave takes new camera samples */
    uint32_t pool_chunk;           /* pool kernels: path ids a wave takes from the pass's counter at a time (64..2^20) */
    uint32_t pool_grid;            /* pool kernels: workgroups per resident slot (1..256) */
    uint32_t tree_refill, tree_inner, tree_shade; /* tree pool thresholds in lanes (1..64): refill at (default 16,
                                                     plain spheres alone: small trees 24, big trees 4) / stay on
                                                     inner nodes while (default 16, trees of 32 768 nodes and more
                                                     24; plain spheres alone: small trees 8, big trees 28; realtime
                                                     frames 4) / shade when this many lanes (default 32, small
                                                     trees of plain spheres 44, realtime frames 48) ... */
    uint32_t tree_trav_plus1;      /* ... or at most this many minus one (1..65) still traverse */
    uint32_t accel_form;           /* extend stage over the SAH tree: 1 = a ray per thread, 2 = refilled lane pool */
    uint32_t accel_refill, accel_inner, accel_grid; /* lane pool thresholds (1..64, 1..65) and workgroups per slot (1..8) */
    uint32_t shards;               /* shards of the render queues: 1, 2, 4 .. 64 */
    uint32_t scan_max_ops_plus1;   /* op streams up to this length minus one are walked wave-uniformly */
    uint32_t list_max_leaves_plus1;/* scenes up to this many primitives minus one are walked as a box-free list */
    uint32_t profile_period;       /* with PT_CFG_PROFILE_STAGES: every n-th pass is bracketed with events */
} pt_tuning;

typedef struct pt_config {
    uint32_t struct_size;        /* sizeof(pt_config); the ABI-1 size (without `tuning`) is accepted too */
    int32_t device;              /* HIP device ordinal */
    uint64_t seed;               /* master seed of the counter-based RNG */
    uint32_t max_paths_in_flight;/* paths per wavefront pass; 0 = library default. 24 bytes of device memory each at
                                    pt_create, 204 more each once a pass form that keeps paths in memory, or pt_intersect, runs */
    uint32_t flags;              /* PT_CFG_* */
    pt_tuning tuning;
} pt_config;

#define PT_CFG_PROFILE_STAGES 1u /* bracket every kernel launch of sampled passes with HIP events */
#define PT_CFG_COUNT_TRAVERSAL 2u /* count node visits / primitive tests in the extend kernel */
#define PT_CFG_EXACT 4u /* IEEE division and square roots everywhere and no fused multiply-add: bit-faithful to an f64
                           CPU evaluation of the reference's expressions. The default is f64 as well, with quotients
                           and roots by refined hardware reciprocals, a*b+c fused, and sin / cos of 2 pi r from a
                           quadrant reduction and polynomials (a few ulp per operation) */

#define PT_CFG_FAT_RECORDS 8u /* diagnostic: a scene of plain spheres alone is traced through the general primitive and
                                 leaf records instead of the compact 64-byte sphere records (same image, for A/B runs) */

#define PT_CFG_WIDE_NODES 32u /* diagnostic: the one-launch pass over a tree of plain spheres reads the 64-byte nodes (f64
                                 slab tests) instead of their 32-byte form (f32 slab tests); same image, for A/B runs */

#define PT_CFG_COUNT_ACCEL 16u /* diagnostic, fast mode: pt_intersect* count the node records read and the primitives
                                  tested in the fast-mode tree (pt_stats.node_visits / prim_tests); the batch then runs
                                  through the ray-per-thread form of the traversal, which visits what the lane pool visits.
                                  PT_CFG_COUNT_TRAVERSAL counts the walk of the reference's own tree instead */

typedef enum pt_render_mode {
    PT_MODE_OFFLINE = 0,    /* overwrite rgb with the mean of spp samples (src/trace.rs:178-181) */
    PT_MODE_REALTIME = 1    /* 80 % pixel skip + 1/n running blend capped at 1001 (src/trace.rs:144-177) */
} pt_render_mode;

typedef struct pt_render_args {
    uint32_t struct_size;
    uint32_t width, height;   /* image_size (src/trace.rs:74) */
    uint32_t x0, y0, x1, y1;  /* start_xy / end_xy of the tile, y counted from the bottom (src/trace.rs:72-73,135-139) */
    uint32_t spp;             /* num_samples (src/trace.rs:71) */
    uint32_t max_depth;       /* Config.max_depth (src/lib.rs:109-123) */
    uint32_t mode;            /* pt_render_mode */
    uint32_t frame;           /* dispatch number; salts the RNG so successive realtime frames differ */
    uint32_t flags;           /* pt_render_flags */
    uint32_t row_step;        /* 0 or 1: every row of [y0, y1). k > 1: only rows y0 + row_phase + m k < y1, the split */
    uint32_t row_phase;       /* of a tile between k GPUs by interleaved rows (no reference counterpart: the        */
                              /* reference's tiles are rectangles, src/lib.rs:380-394); row_phase < row_step          */
} pt_render_args;

/* PT_RENDER_BOOK_SEMANTICS: Isotropic scatters as in the book instead of as in the reference. The reference's
 * Isotropic has no scattering_pdf (0 by the trait default, src/material.rs:142-144) and a DummyPDF that generates the
 * zero vector (439-445), so every medium scatter yields 0 or NaN. With this flag its pdf is the uniform sphere:
 * value 1/(4 pi) (the unused unit_sphere_pdf, src/material.rs:46-48), direction = the unit vector scatter() already
 * draws (349), scattering_pdf 1/(4 pi). Same number of RNG draws either way. */
enum pt_render_flags { PT_RENDER_BOOK_SEMANTICS = 1, PT_RENDER_NO_WAIT = 2 };
/* PT_RENDER_NO_WAIT: pt_render / pt_render_jobs return once the work is on the context's stream instead of waiting for
 * it. The reference's main loop hands frame k to the display while its workers already trace frame k + 1
 * (src/lib.rs:476-509); a host that waits for every realtime frame pays a quarter of the frame's time for the round trip
 * (1080p frames of the Cornell box: 5 560 per second with the wait, 5 970 without). Everything that reads the
 * image through this library (pt_readback, pt_tonemap_*, pt_display_transform, pt_present_submit, pt_group_gather*) is
 * ordered behind the frames already issued; a caller that reads a bound image by other means calls pt_wait first. An
 * error of the device surfaces at the next call that waits. */

typedef struct pt_stats {
    uint64_t paths;            /* paths traced since the last pt_stats_reset */
    uint64_t segments;         /* extend invocations (ray segments) */
    uint64_t node_visits;      /* BVH boxes tested   (needs PT_CFG_COUNT_TRAVERSAL) */
    uint64_t prim_tests;       /* primitives tested  (needs PT_CFG_COUNT_TRAVERSAL) */
    uint64_t nan_pixels;       /* pixels whose last written colour has a NaN channel */
    uint64_t launches_raygen, launches_extend, launches_shade, launches_resolve; /* (a realtime frame of one sample per
                                * pixel through a pool kernel has no resolve launch: its paths blend their own pixels) */
    double ms_raygen, ms_extend, ms_shade, ms_resolve; /* summed HIP-event time of profiled launches (a one-launch
                                                        * pass form is timed under ms_shade) */
    uint64_t profiled_segments;/* segments processed by the profiled extend/shade launches */
    uint64_t profiled_paths;   /* paths processed by the profiled raygen/resolve launches */
    uint64_t profiled_passes;  /* profiled raygen (= resolve) launches */
    uint64_t profiled_bounce_launches; /* profiled extend (= shade) launches */
    uint64_t last_pass_form;   /* pt_pass_form of the most recent pass */
} pt_stats;

/* How the library ran a pass (it chooses; none of them changes a pixel). */
enum pt_pass_form {
    PT_PASS_NONE = 0,
    PT_PASS_PER_BOUNCE = 1,       /* ray generation, then an extend and a shade launch per bounce */
    PT_PASS_PERSISTENT = 3,       /* primitive-list scenes: ray generation + one trace-and-shade launch over queue tiles */
    PT_PASS_PATH_POOL = 4,        /* primitive-list scenes, offline passes and realtime frames: one launch, paths stay in
                                     registers */
    PT_PASS_TREE_POOL = 5         /* tree scenes, offline: one launch, traversal and shading inside the wave */
};

/* No reference counterpart (the crate owns no device); allocates the per-GPU state. */
int pt_create(const pt_config *cfg, pt_ctx **out);
void pt_destroy(pt_ctx *ctx);
const char *pt_last_error(pt_ctx *ctx);

/* Replaces handing `world: Box<dyn Hitable>` to SceneState::new (src/trace.rs:55-67, src/lib.rs:359). */
int pt_scene_upload(pt_ctx *ctx, const pt_scene_desc *scene);

/* Replaces Camera::new (src/camera.rs:25-51); call again after moving the camera (Camera::update, 89-96). */
int pt_camera_set(pt_ctx *ctx, const pt_camera *cam);

/* Replaces Jobs::dispatch_jobs + wait on one TraceSceneBatchJob::run (src/trace.rs:117-196,
 * src/lib.rs:396-399,476-478): traces the tile into the context's f32 RGBA image. Synchronous. */
int pt_render(pt_ctx *ctx, const pt_render_args *args);

/* Replaces Jobs::dispatch_jobs(&batch_jobs) and the wait on the batch's counter (src/lib.rs:396-399,476-478) for the
 * batch of TraceSceneBatchJobs the reference builds, one per tile of the image (src/lib.rs:380-394). Same pixels as
 * pt_render on every job in turn. When the jobs differ only in their rectangles and the rectangles tile their bounding
 * box exactly (as the reference's tile grid does), the batch is traced as one job over that box: paths are keyed by
 * global pixel and sample, so the result is bit-identical, and it costs one pass instead of one per tile (a 1080p
 * realtime frame as 100 tiles: 8 ms; as one job: 0.22 ms). Synchronous. */
int pt_render_jobs(pt_ctx *ctx, const pt_render_args *jobs, size_t n_jobs);

/* Replaces TraceSceneBatchJob::clear_buffer + zeroing the image (src/trace.rs:112-115, src/lib.rs:462-468). */
int pt_accum_reset(pt_ctx *ctx);
/* Blocks until everything issued on the context (renders with PT_RENDER_NO_WAIT included) has completed. */
int pt_wait(pt_ctx *ctx);

/* Replaces reading SceneOutput.buffer (src/lib.rs:496-509): copies width*height*4 floats, row 0 = bottom scanline. */
int pt_readback(pt_ctx *ctx, float *rgba, size_t n_floats);

/* Device address of the f32 RGBA image (for RCCL gathers issued by the host layer); NULL before the first render. */
void *pt_image_device_ptr(pt_ctx *ctx);

/* Render into caller-owned device memory of width*height*4 floats instead of the context's image. NULL restores it. */
int pt_image_bind(pt_ctx *ctx, void *device_rgba, uint32_t width, uint32_t height);

/* Replaces Hitable::hit on SceneState.world for a batch (src/hitable.rs:28): closest hit in (t_min, t_max).
 * `medium_seed` keys the free-flight draw of ConstantMedium::hit (src/volume.rs:42) for ray i. */
int pt_intersect(pt_ctx *ctx, const pt_ray *rays, size_t n, double t_min, double t_max,
                 uint64_t medium_seed, pt_hit *out);

int pt_stats_get(pt_ctx *ctx, pt_stats *out);
int pt_stats_reset(pt_ctx *ctx);
/* Which records the most recent pass read (a diagnostic, for tests and A/B runs; no reference counterpart): a set of
 * PT_RECORDS_* bits, or a negative pt_error. */
enum pt_pass_records {
    PT_RECORDS_COMPACT_SPHERES = 1, /* the 64-byte sphere records that serve leaf test and vertex (scenes of plain spheres) */
    PT_RECORDS_NODES32 = 2          /* the 32-byte tree nodes with f32 slab tests */
};
int pt_last_pass_records(pt_ctx *ctx);

/* The HIP stream every kernel of this context is launched on (hipStream_t). */
void *pt_stream(pt_ctx *ctx);

/* Benchmark form of pt_intersect: rays already resident in device memory (SoA of 7 f64 arrays of length n:
 * ox oy oz dx dy dz time), hits left on the device; runs `repeats` launches and returns their mean duration. */
int pt_intersect_device(pt_ctx *ctx, const double *d_rays_soa, size_t n, double t_min, double t_max,
                        uint64_t medium_seed, int repeats, double *mean_ms_out);

/* No counterpart in the reference (the crate never touches a device). The variant of pt_intersect for callers that
 * cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc, processes in which another
 * thread is capturing a HIP graph (an allocation would break that capture). d_rays (n pt_ray) and d_hits (n pt_hit)
 * are DEVICE memory of the context's GPU, owned by the caller; the ray queue, the hit records and the fill count the
 * extend kernels work on live in a fixed array inside the library's device image (about 6.8 MiB per GPU, held from
 * the library's load to the end of the process), so the call performs no hipMalloc / hipFree
 * (pt_device_alloc_count() does not move). It is NOT itself capturable into a caller's graph: it launches on the
 * context's own stream, queries the two pointers and synchronises that stream before it returns. At most
 * PT_INTERSECT_STATIC_MAX_RAYS rays per call (PT_ERR_TOO_LARGE beyond that). Same hits as pt_intersect, bit for bit.
 * All contexts of one GPU share that GPU's array: calls that overlap in time on one GPU (from several contexts or
 * threads) race; calls on different GPUs do not. */
#define PT_INTERSECT_STATIC_MAX_RAYS 65536
int pt_intersect_static(pt_ctx *ctx, const pt_ray *d_rays, size_t n, double t_min, double t_max,
                        uint64_t medium_seed, pt_hit *d_hits);

/* Diagnostic, no counterpart in the reference: the number of device and page-locked host allocations (hipMalloc,
 * hipHostMalloc) this library has made in the process so far. A caller that must not allocate brackets its calls
 * with it (tests/test_zz_no_alloc.py does, around pt_intersect_static). */
uint64_t pt_device_alloc_count(void);

/* ------------------------------------------------------------------------ */
/* Several GPUs: one context per GPU, one image                              */
/* ------------------------------------------------------------------------ */
/* Replaces dealing TraceSceneBatchJobs over the image to the worker threads (src/lib.rs:380-399) and the main thread
 * reading the shared buffer (src/lib.rs:496-509) when the workers are GPUs: a GROUP is one context per GPU, each
 * with the same scene and camera. Member r of N traces rows r, r + N, r + 2N, ... of the image into its own copy
 * (pt_render with the row_step / row_phase that pt_group_rows returns; the RNG is keyed by global pixel and sample, so
 * the assembled image does not depend on N), and pt_group_gather moves every member's rows into the root's image with
 * one group of RCCL point-to-point transfers over xGMI (the rows are disjoint: a gather, not a reduction). The root
 * then presents (pt_readback, pt_tonemap_device, ...).
 *
 * Two ways to form a group:
 *   one process per GPU   rank 0 calls pt_group_unique_id and hands the bytes to the other processes by its own
 *                         means (torch.distributed, MPI, a file); every process calls pt_group_join (collective);
 *   one process, N GPUs   pt_group_join_local with the N contexts (distinct devices); gathers with
 *                         pt_group_gather_local, which issues every member's transfer from the calling thread.
 * A context alone is a group of one (row_step 1; gathering is a no-op). */
#define PT_GROUP_ID_BYTES 128
int pt_group_unique_id(uint8_t *id_out, size_t n_bytes);                   /* n_bytes >= PT_GROUP_ID_BYTES */
int pt_group_join(pt_ctx *ctx, const uint8_t *id, size_t n_bytes, int rank, int world);
int pt_group_join_local(pt_ctx *const *ctxs, int n);                       /* ctxs[k] becomes member k of n */
int pt_group_rows(pt_ctx *ctx, uint32_t *row_step, uint32_t *row_phase);   /* this member's share: world, rank */
/* The split as numbers, host only (no context, no GPU): for an image of `height` rows gathered to `root`, member
 * `rank` of `world` traces *rows rows (rank, rank + world, ...), and in the root's staging buffer they start
 * *staged_before rows in (the other members' rows lie one member behind the other in rank order; the root's own rows
 * are never staged). pt_group_staged_row: the staging-buffer row that holds image row y, -1 for a row of the root's
 * own or bad arguments. The gather's kernels run the same arithmetic (csrc/group_rows.h). */
int pt_group_layout(uint32_t height, int world, int root, int rank, uint32_t *rows, uint64_t *staged_before);
int64_t pt_group_staged_row(uint32_t height, int world, int root, uint32_t y);
/* Collective over the group: every member calls it (or the one process calls pt_group_gather_local) after rendering
 * its rows of the same width x height image. Synchronous; on return the root's image holds every row. */
int pt_group_gather(pt_ctx *ctx, int root);
int pt_group_gather_local(pt_ctx *const *ctxs, int n, int root);
/* The gather of a frame loop, without a host synchronisation (replaces the main loop reading the shared buffer while
 * the workers already write the next frame, src/lib.rs:476-509). Every member calls it after pt_render; it returns as
 * soon as the work is queued. On the render stream the member's rows are packed into one of two staging buffers (a
 * few microseconds); from then on the member may trace its next frame into its image. The transfers run on a second
 * stream behind an event, and the root assembles the whole frame (its own rows and everyone else's) in one of two
 * frame buffers of its own, so that frame k's rows land while frame k + 1 is traced and no frame is ever seen half
 * new. After the call, on the root, pt_readback, pt_tonemap_device, pt_display_transform and pt_present_submit read
 * the LATEST ASSEMBLED FRAME instead of the context's own image (their stream waits for that frame's transfers, the
 * host does not); pt_group_gather, pt_image_bind, a change of image size and pt_group_leave put them back on the
 * context's own image. pt_group_wait blocks the host until this member's part of the last asynchronous gather is
 * complete (end of a loop, before timing stops). A lone context may call it too (a group of one). Every member must
 * issue its gathers in the same order; pt_group_gather first drains the asynchronous ones. */
int pt_group_gather_async(pt_ctx *ctx, int root);
int pt_group_wait(pt_ctx *ctx);
/* The asynchronous gather's data path on ONE GPU (group of one): as pt_group_gather_async for a virtual group of two in
 * which rows 0, 2, 4, ... are the root's own and rows 1, 3, 5, ... travel from the context to itself through RCCL on
 * the second stream. The assembled frame must equal the image that was traced (the caller compares, frame by frame,
 * while it already traces the next one). */
int pt_group_loopback_async(pt_ctx *ctx);
int pt_group_leave(pt_ctx *ctx);                                           /* also done by pt_destroy */
/* Checks the gather's data path on ONE GPU (group of one): the image's odd rows are packed, wiped, sent from the
 * context to itself through RCCL and scattered back; the image must come out unchanged. */
int pt_group_loopback(pt_ctx *ctx);

/* ------------------------------------------------------------------------ */
/* Host-side scene construction (mirrors SceneBuilder / MaterialBuilder)     */
/* ------------------------------------------------------------------------ */

typedef struct pt_scene pt_scene; /* owns the arrays a pt_scene_desc points into */

/* Named scene recipes of the reference (src/lib.rs:674-1077) plus the synthetic sphere cloud.
 * `param` is recipe-specific: the sphere count for "sphere_cloud"; 1 for "cornell_smoke" and "final_book_two" makes
 * the light-shape list the scene's own light instead of the pair hard-coded in src/trace.rs:123-128 (and, for the
 * smoke box, flips the light towards the room): with PT_RENDER_BOOK_SEMANTICS that gives a finite image. */
int pt_scene_build_named(const char *name, double aspect, uint64_t seed, int64_t param,
                         pt_scene **out, pt_camera *camera_out);
/* Replaces the decode step of ImageTexture::new(EARTH_TEXTURE_BYTES) (src/texture.rs:84-98, src/lib.rs:767): the crate
 * embeds assets/textures/earthmap.jpg and decodes it with the `image` crate; this library decodes nothing. The host
 * decodes the file once and registers its RGB8 texels (rows top-down, as `to_rgb().into_vec()` yields them) under the
 * file's name; "textured_sphere" and "final_book_two" use "earthmap.jpg" (1024 x 512) and return
 * PT_ERR_MISSING_ASSET from pt_scene_build_named when it is not registered. The texels are copied. */
int pt_asset_register_image(const char *name, const uint8_t *rgb8, int width, int height);
void pt_scene_free(pt_scene *scene);
int pt_scene_get_desc(const pt_scene *scene, pt_scene_desc *out);

/* Generic builder: the operations of SceneBuilder (src/scene.rs:13-63) and MaterialBuilder
 * (src/material.rs:80-131). Functions that create something return its index (>= 0) or a pt_status. */
int pt_scene_new(uint64_t seed, pt_scene **out);
int pt_scene_texture_constant(pt_scene *s, const double rgb[3]);
int pt_scene_texture_checker(pt_scene *s, int even, int odd);
int pt_scene_texture_noise(pt_scene *s, double scale);
int pt_scene_texture_image(pt_scene *s, const uint8_t *rgb8, int width, int height);
int pt_scene_material_lambertian(pt_scene *s, int texture, double emissive);
int pt_scene_material_diffuse_light(pt_scene *s, int texture);
int pt_scene_material_metal(pt_scene *s, const double albedo[3], double fuzz);
int pt_scene_material_dielectric(pt_scene *s, double ref_idx);
int pt_scene_sphere(pt_scene *s, const double center[3], double radius, int material);
int pt_scene_moving_sphere(pt_scene *s, const double c0[3], const double c1[3], double t0, double t1,
                           double radius, int material);
int pt_scene_rect(pt_scene *s, double amin, double amax, double bmin, double bmax, double c, int axis,
                  int material);
int pt_scene_box(pt_scene *s, const double pmin[3], const double pmax[3], int material);
int pt_scene_flip_normals(pt_scene *s, int node);
int pt_scene_translate(pt_scene *s, int node, const double t[3]);
int pt_scene_rotate_y(pt_scene *s, int node, double degrees);
int pt_scene_constant_medium(pt_scene *s, int boundary, double density, int texture);
int pt_scene_list(pt_scene *s, const int32_t *nodes, int n);
/* BvhNode::from_list (src/bvh.rs:13-69) over the given nodes; axis draws come from the scene's seeded stream. */
int pt_scene_bvh(pt_scene *s, const int32_t *nodes, int n, double time0, double time1);
int pt_scene_set_root(pt_scene *s, int node);
int pt_scene_set_lights(pt_scene *s, const int32_t *nodes, int n);
/* bounding_box(t0, t1) of a node (src/hitable.rs:29); returns PT_ERR_INVALID_ARG for a LIST (unreachable!, 63-65). */
int pt_scene_bounding_box(const pt_scene *s, int node, double t0, double t1, double min_out[3],
                          double max_out[3]);

/* Flattened device layout, for inspection by tests (no GPU needed). Returns counts through the out params. */
int pt_scene_flatten_info(const pt_scene_desc *scene, int32_t *n_ops, int32_t *n_leaves, int32_t *n_xforms,
                          int32_t *max_frame_depth);

/* Replaces rebuilding the tree for a new shutter window, BvhNode::from_list(list, time0, time1) (src/bvh.rs:13,60-61;
 * the main loop advances the window every frame, src/lib.rs:412-417): moving spheres' bounds are recomputed for
 * [time0, time1] and the fast-mode tree is refitted on the device, bottom up, one launch per level. A scene walked as
 * a primitive list has no boxes and needs nothing (PT_OK). Exact mode walks the reference's host-built tree, whose
 * boxes are part of the uploaded scene: PT_ERR_UNSUPPORTED_SCENE (rebuild with pt_scene_bvh and upload). */
int pt_scene_refit(pt_ctx *ctx, double time0, double time1);

/* Fast mode traces scenes that are not a short primitive list through a SAH tree over the same primitives, built on
 * the host at upload (no reference counterpart: it stands in for walking BvhNode::from_list's tree, src/bvh.rs:13-99,
 * and returns the same closest hit up to exact ties). This builds that tree for a scene and checks it on the host:
 * every primitive in exactly one leaf, child boxes enclose their subtrees, depth within the traversal stack. */
int pt_scene_accel_info(const pt_scene_desc *scene, int32_t *n_nodes, int32_t *n_prims, int32_t *depth);

/* Replaces BvhNode::from_list (src/bvh.rs:13-69) for a scene that is already on the device: BUILDS the fast-mode tree
 * there instead of refitting the host-built one. The primitives' bounds for the shutter window [time0, time1] and
 * 63-bit Morton codes of their centres are computed on the device and the primitives are sorted along the curve
 * (radix sort); the tree is then grown level by level, one launch per level, every node splitting its run of the
 * curve where the highest differing code bit flips (in the middle once the remaining depth budget only fits a
 * balanced subtree, which bounds the traversal stack), leaves of at most four primitives; the boxes are fitted
 * bottom up, one launch per level. The host only reads back one node counter per level to size the next launch.
 * Hits are those of any other tree over the same primitives (exact ties aside). For scenes whose geometry moves a
 * lot between windows, where a refit would keep a stale topology. Same preconditions and errors as pt_scene_refit. */
int pt_scene_build_device(pt_ctx *ctx, double time0, double time1);


/* Reinhard(1.5) + sqrt gamma + u8 conversion with the row flip of the PPM writers
 * (src/trace.rs:236-242, src/lib.rs:330-340,613-640). rgb8_out is width*height*3, top row first. */
int pt_tonemap_rgb8(const float *rgba, uint32_t width, uint32_t height, uint8_t *rgb8_out);

/* The same conversion on the device, from the context's image (src/trace.rs:236-242, src/lib.rs:330-340,613-640):
 * rgb8_out is a HOST buffer of width*height*3 bytes, top row first. Bit-identical to pt_tonemap_rgb8. */
int pt_tonemap_device(pt_ctx *ctx, uint8_t *rgb8_out, size_t n_bytes);

/* Replaces the rendy tonemap pass, assets/shaders/tonemap.frag:37-70 (exposure from TonemapperArgs,
 * src/lib.rs:284-286: 1.3 by default; keys change it, src/lib.rs:425-429): colour * exposure, ACES-fitted curve
 * (input matrix, RRT+ODT fit, output matrix, clamp to [0,1]), times the vignette 1 - |uv - (0.5,0.5)| at the
 * fragment centre; f32 arithmetic as in the shader. rgba_out is a HOST buffer of width*height*4 floats, row 0 = top
 * of the screen (the shader's uv.y flip), alpha 1. */
int pt_display_transform(pt_ctx *ctx, float exposure, float *rgba_out, size_t n_floats);

/* ---- presentation bridge ----
 * Replaces the per-frame hand-over of the traced image to the presentation graph: the main loop owns the f32 RGBA
 * buffer the tile jobs write (src/lib.rs:496-509) and the CopyToImage node records a copy of the host-visible source
 * buffer into the sampled image that the tonemap pass then draws (src/node/copy_image.rs:104-165,
 * src/lib.rs:241-292).
 * Here the image lives on the GPU that traced it, and frames reach host memory through a RING of n_slots frames of
 * slot_bytes each (at most 4), attached once. host_staging NULL: the library allocates the ring as page-locked host
 * memory, which the GPU's copy engines write at the full rate of the link; pt_present_slot returns a slot's address,
 * and the host uploads from there (or imports the allocation as external host memory). host_staging non-NULL: the
 * host's own memory (the mapped source buffer of that node); its pages are pinned where the driver allows it, and
 * memory it refuses to pin is accepted too and copied through the runtime's bounce buffers at a fraction of the
 * link's rate (pt_present_pinned tells which). pt_present_submit snapshots the context's current image for a slot on
 * the render stream (PT_PRESENT_RAW: the f32 RGBA image as traced, row 0 = bottom scanline, what the node uploads;
 * PT_PRESENT_DISPLAY: pt_display_transform's output at `exposure`, row 0 = top of the screen, for a host that only
 * blits), queues the snapshot's copy into the slot on a second stream and returns: the next frame is traced while
 * the previous one crosses PCIe. pt_present_wait blocks until a slot's bytes are in host memory (a slot never
 * submitted is ready at once); submitting to a slot waits for that slot's previous copy first. The first submit of a
 * slot (and the first after the image grew) allocates its device snapshot. pt_destroy detaches. */
/* PT_PRESENT_DISPLAY8 / PT_PRESENT_DISPLAY8_SRGB: the display transform as the 8-bit swapchain image keeps it, the
 * frame the reference's tonemap pass leaves in its surface (assets/shaders/tonemap.frag:55-70 drawn into the format of
 * get_surface_format, src/lib.rs:257): one 32-bit word per pixel, R in the low byte, alpha 255, each channel the nearest
 * of 256 levels of the value in [0, 1] (floor(255 v + 0.5)), for the _SRGB kind after the sRGB transfer function that an
 * *_SRGB surface applies on write. A quarter of the bytes of the f32 kinds across PCIe: width * height * 4 per slot. */
enum pt_present_kind { PT_PRESENT_RAW = 0, PT_PRESENT_DISPLAY = 1, PT_PRESENT_DISPLAY8 = 2, PT_PRESENT_DISPLAY8_SRGB = 3 };
int pt_present_attach(pt_ctx *ctx, void *host_staging, size_t slot_bytes, uint32_t n_slots);
void *pt_present_slot(pt_ctx *ctx, uint32_t slot);   /* host address of a slot; NULL if there is none */
int pt_present_pinned(pt_ctx *ctx);                  /* 1 if the attached ring is page-locked, else 0 */
int pt_present_submit(pt_ctx *ctx, uint32_t slot, int kind, float exposure);
int pt_present_wait(pt_ctx *ctx, uint32_t slot);
int pt_present_detach(pt_ctx *ctx);

/* ---- realtime interaction (host side) ----
 * Key bits of pt_user_input: the keys the reference's main loop and camera react to
 * (src/camera.rs:121-189, src/lib.rs:425-451). */
enum pt_key {
    PT_KEY_W = 1 << 0, PT_KEY_S = 1 << 1, PT_KEY_A = 1 << 2, PT_KEY_D = 1 << 3, PT_KEY_E = 1 << 4, PT_KEY_Q = 1 << 5,
    PT_KEY_RIGHT = 1 << 6, PT_KEY_LEFT = 1 << 7, PT_KEY_UP = 1 << 8, PT_KEY_DOWN = 1 << 9,
    PT_KEY_T = 1 << 10, PT_KEY_R = 1 << 11, PT_KEY_O = 1 << 12, PT_KEY_P = 1 << 13, PT_KEY_B = 1 << 14
};
/* One frame of input::UserInput (src/input.rs:13-20): WASDQE and the app keys are read from keys_pressed, the arrow
 * keys from keys_held, as the reference does. mouse_delta is (dx, -dy) as the reference stores it. */
typedef struct pt_user_input {
    uint32_t keys_pressed;
    uint32_t keys_held;
    float mouse_dx, mouse_dy;
} pt_user_input;
typedef struct pt_frame_info {
    int32_t camera_moved;      /* Camera::update_from_input's result */
    int32_t cleared;           /* accumulation was reset this frame (src/lib.rs:462-468) */
    double exposure, numframes;/* TonemapperArgs.exposure_numframes_xx[0..1] after this frame */
    double time0, time1;       /* SceneState.time0/1 after the advance of src/lib.rs:412-417 */
    double sky_brightness;
    int32_t disable_emissive;
    uint32_t frame;            /* frames traced so far */
    pt_camera camera;          /* the Camera::new parameters that reproduce the camera after this frame */
} pt_frame_info;

/* Replaces Camera::update_from_input followed by Camera::update (src/camera.rs:89-96,109-215) on a camera given by
 * its Camera::new parameters; writes the parameters of the moved camera back. Returns 1 if it moved, 0 if not,
 * negative pt_status on bad arguments. */
int pt_camera_apply_input(pt_camera *cam, const pt_user_input *input, double frame_time);

/* Replaces the realtime main loop body (src/lib.rs:406-478) minus window and present: see pt_session_frame. */
typedef struct pt_session pt_session;
int pt_session_new(pt_ctx *ctx, const pt_camera *cam, uint32_t width, uint32_t height, uint32_t max_depth,
                   pt_session **out);
void pt_session_free(pt_session *s);
/* One iteration: time advance, T/R exposure keys, O/P/B scene keys, camera input, reset on move, then one
 * progressive frame (mode realtime, spp 1). info may be NULL. */
int pt_session_frame(pt_session *s, const pt_user_input *input, double frame_time, pt_frame_info *info);

uint32_t pt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
