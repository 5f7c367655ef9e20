/*
 * rt_hip.h - C ABI of the MI355X (gfx950) ray-tracing library.
 *
 * This library replaces ONE path of pxk5958/ray-tracer: everything that sits behind the
 * JavaScript class `RayTracer` (reference js/8-kd_tree.js:2298-2589) - the k-d tree build,
 * the per-pixel Whitted ray loop that the reference runs as a generated GLSL fragment
 * shader (js/8-kd_tree.js:196-1071), the log-average-luminance reduction
 * (js/8-kd_tree.js:2558-2569) and the tone-reproduction present pass (js/8-kd_tree.js:85-143).
 *
 * Conventions
 *   - every entry point returns 0 on success and a negative value on failure;
 *     rt_last_error() returns the message of the last failure on that context
 *     (or of the last failed rt_create when ctx is NULL).
 *   - the library copies every input during the call; output buffers are caller-owned.
 *   - one caller thread per context.
 *   - matrices are 16 floats, column-major, exactly the order of THREE.Matrix4.toArray()
 *     as handed to gl.uniformMatrix4fv (js/8-kd_tree.js:1118-1120).
 *   - image rows are bottom-up, like gl.readPixels (js/8-kd_tree.js:2558).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

/* primitive kinds: the five analytic classes of js/8-kd_tree.js:1267-1537 */
enum { RT_PRIM_PLANE = 0, RT_PRIM_RECT = 1, RT_PRIM_SPHERE = 2, RT_PRIM_CUBE = 3, RT_PRIM_CYLINDER = 4 };
/* material types: Material.PHONG_MATERIAL.. (js/8-kd_tree.js:2124-2134) */
enum { RT_MAT_PHONG = 0, RT_MAT_PHONG_BLINN = 1, RT_MAT_PHONG_CHECKERED = 2 };
/* k-d node types in the top two metadata bits (js/8-kd_tree.js:1540-1567) */
enum { RT_KD_INVALID = 0, RT_KD_INTERNAL = 1, RT_KD_EMPTY_LEAF = 2, RT_KD_LEAF = 3 };

/* The reference is configured by the constants at the top of its script
 * (js/8-kd_tree.js:25-46) plus the canvas size (8-kd_tree.html:11). */
typedef struct rt_config {
    int32_t width;               /* canvas WIDTH  */
    int32_t height;              /* canvas HEIGHT */
    int32_t supersampling;       /* SUPERSAMPLING: the tracer renders width*S x height*S */
    int32_t max_depth;           /* MAX_RECURSION_DEPTH: ray tree has 2^D - 1 slots */
    int32_t tr_method;           /* TONE_REPRODUCTION_METHOD: 0 Ward, 1 Reinhard */
    int32_t render_triangles;    /* RENDER_TRIANGLES */
    int32_t use_kd_tree;         /* USE_KD_TREE (0 = brute force over all faces) */
    int32_t max_kd_depth;        /* MAX_KD_TREE_DEPTH (<= 24) */
    float   L_max;               /* L_MAX  = 1000 */
    float   L_dmax;              /* L_DMAX = 500  */
    float   A;                   /* A = 0.18 (Reinhard key) */
    float   infinity;            /* INFINITY = 100000.0 (100000.1 in 6-transmission.js:22) */
    float   epsilon;             /* EPSILON = 1e-4 */
    float   sphere_epsilon;      /* SPHERE_EPSILON = 0.095 */
    float   plane_rect_epsilon;  /* PLANE_RECT_EPSILON = 0.095 */
    float   triangle_epsilon;    /* TRIANGLE_EPSILON = 1e-4 */
    /* multi-GPU sharding of the image: this context renders the 8x8-pixel tiles whose
     * index (blocks of 16x16 tiles row by row, Z-order inside a block) is congruent to shard_index modulo shard_count.
     * shard_count <= 1 renders the whole frame. No counterpart in the reference. */
    int32_t shard_index;
    int32_t shard_count;
    int32_t device;              /* HIP device ordinal; -1 = host-only context (mesh, k-d build and
                                    export work, every entry point that needs a GPU fails) */
    /* the shard is traced as `parts` independent slices of its tiles on separate HIP streams, so that the
     * end of one slice's kernels (a few long rays) overlaps the other slices' work. 0 = library default
     * (environment RT_PARTS, else 1); 1 = one stream. The image does not depend on it. No counterpart in the reference. */
    int32_t parts;
} rt_config;

/* One analytic primitive = the two mat4 uniforms `model<i>` / `invModel<i>` plus the
 * material constants baked into the generated GLSL (js/8-kd_tree.js:1195-1231).
 * Order in the array is scene order: ties are resolved by the strict `hitInfo.t < tMin`
 * of js/8-kd_tree.js:1204, i.e. the earlier primitive wins. */
typedef struct rt_prim {
    int32_t kind;
    int32_t mat_type;
    int32_t mat_id;
    int32_t reserved;
    float   model[16];
    float   inv_model[16];
} rt_prim;

/* One entry of phongMaterials[] / phongBlinnMaterials[] / phongCheckeredMaterials[]
 * (js/8-kd_tree.js:255-281). (type, id) select the uniform array and the slot in it.
 * Co is Co1 for the checkered material; Co2 is used only by it. */
typedef struct rt_material {
    int32_t type;
    int32_t id;
    float   ka, kd, ks, ke, kReflect, kRefract, ior;
    float   Co[3];
    float   Co2[3];
    float   Cs[3];
} rt_material;

/* pointLights[i] (js/8-kd_tree.js:284-288) */
typedef struct rt_light {
    float position[3];
    float color[3];
} rt_light;

/* counters of the last rt_trace (this context's shard only) */
typedef struct rt_stats {
    uint64_t rays_total;        /* valid ray-tree nodes: primary + reflected + refracted */
    uint64_t rays_per_level[8]; /* the same, split by ray-tree level */
    uint64_t shadow_rays;       /* lights x shaded nodes */
    uint64_t pixels;            /* pixels of this shard (supersampled resolution) */
    double   trace_ms;          /* device time of the whole trace (all steps + composite) */
    /* A frame of depth D runs as D+1 pipeline steps. Step k: analytic closest hit of level k, k-d walk with its
     * triangle tests (closest-hit rays of level k together with the shadow rays of level k-1), shading of
     * level k-1, hit resolution + children + shadow-ray set-up of level k. Step 8 (depth 8 only) is not itemised. */
    double   level_kernel_ms[8];/* device time of one step (all four stages) */
    double   level_stage_ms[8][7]; /* per step, columns 0..3: k_gen_prims, k_kd_walk, k_shade_b, k_shade_a; 4..6 unused */
    double   stage_ms[7];       /* the same, summed over all steps */
    uint64_t tri_records[2];    /* (ray, non-empty leaf) records whose triangles the closest / shadow walks tested */
    uint64_t kd_closest_rays;   /* rays that walked the k-d tree for a closest hit */
    uint64_t kd_shadow_rays;    /* shadow rays that walked the k-d tree */
    double   composite_ms;      /* device time of the bottom-up colour accumulation */
    double   tone_ms;           /* device time of the last rt_tone reduction */
    double   kd_build_ms;       /* host time of the last rt_build_kdtree */
    double   kd_upload_ms;      /* host->device copy of the compact tree */
    uint64_t kd_device_fallbacks; /* rt_build_kdtree_device calls of this context that stalled on the device and built on the host instead */
} rt_stats;

/* ---- life cycle ------------------------------------------------------------------ */
/* replaces `new RayTracer()` (js/8-kd_tree.js:2299-2345): allocates the render target */
int rt_create(const rt_config *cfg, rt_ctx **out_ctx);
void rt_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx);
/* rt_config filled with the constants of js/8-kd_tree.js:25-46 and a 640x480 canvas */
void rt_default_config(rt_config *cfg);

/* ---- scene: replaces the uniform marshalling of RayTracer.update (js/8-kd_tree.js:2435-2453,
 *      :1104-1126) and the shader regeneration of RayTracer.setScene (:2347-2431) --------
 * These calls and rt_set_mesh / rt_build_kdtree change host-side copies only; the device tables are refreshed by the
 * next rt_trace_begin, so they may be called while a frame is in flight. rt_build_kdtree_device replaces device arrays
 * the queued frame still reads and therefore fails between rt_trace_begin and rt_trace_end.
 * rt_prim.inv_model must be the inverse of rt_prim.model, as Primitive.translate / rotate / scale keep them
 * (js/8-kd_tree.js:1245-1259): the intersection routines use inv_model like the shader, a bounding sphere that lets
 * rays far from a sphere, cube or cylinder skip them is derived from model. */
int rt_set_primitives(rt_ctx *ctx, const rt_prim *prims, int32_t n);
int rt_set_materials(rt_ctx *ctx, const rt_material *mats, int32_t n);
/* n_loop is the trip count of the light loop baked into the shader (`nextPointLightId`,
 * js/8-kd_tree.js:787); lights beyond n keep zero position and colour, as uniforms that
 * were never set do. Pass n_loop = n when every light is set. */
int rt_set_lights(rt_ctx *ctx, const rt_light *lights, int32_t n, int32_t n_loop);
int rt_set_globals(rt_ctx *ctx, const float ambient[3], const float bg_color[3]);
/* cameraPos and the four corner rays of getPrimaryRay (js/8-kd_tree.js:1094-1098,:2448-2452) */
int rt_set_camera(rt_ctx *ctx, const float pos[3], const float ray00[3], const float ray01[3],
                  const float ray10[3], const float ray11[3]);

/* ---- mesh + k-d tree: replaces the per-frame vertsData/facesData fill (js/8-kd_tree.js:2461-2484)
 *      and `new KdTree().buildTree(...)` + fillDataArray + leaf copy (:2497-2520) ---------- */
/* verts: n_verts*3 doubles = g_verts as the builder sees them (f64); the device copy is
 * narrowed to f32 exactly like `vertsData[i*3+0] = g_verts[i].x` (:2462-2464).
 * faces: n_faces*3 vertex indices; face_mat: n_faces*2 = (material type, material id). */
int rt_set_mesh(rt_ctx *ctx, const double *verts, int32_t n_verts, const uint32_t *faces,
                const uint32_t *face_mat, int32_t n_faces);
/* The vertices moved, the tree stays: replaces the per-frame vertex upload of RayTracer.update
 * (`vertsData[i*3+0] = g_verts[i].x ...` + texSubImage2D, js/8-kd_tree.js:2461-2479) on the frames between two rebuilds.
 * The reference rebuilds its tree at most every 0.30 s (js/8-kd_tree.js:2494-2495); until then its shader fetches the
 * NEW vertices through the OLD leaves and the old root box (js/8-kd_tree.js:2523-2533). This call does the same: the
 * triangle records of the device are refilled from `verts` (n_verts*3 doubles, the same count and face list as the
 * last rt_set_mesh) through the node and leaf tables as they stand; root box and tree do not change until the next
 * rt_build_kdtree / rt_build_kdtree_device, which then builds from these vertices. Fails while a frame is in flight. */
int rt_update_vertices(rt_ctx *ctx, const double *verts, int32_t n_verts);
/* builds the median-split tree over all faces and uploads the compact device layout */
int rt_build_kdtree(rt_ctx *ctx);
/* The same tree, bit for bit, built level by level on the GPU and left in device memory: for geometry that changes
 * every frame (the reference rebuilds while objects are dragged, js/8-kd_tree.js:2494-2522). Only vertices, faces and
 * material slots cross PCIe. rt_kdtree_counts / rt_export_kdtree_reference work on it like on the host-built tree.
 * Should the device build stall (its in-kernel sums wait for workgroups with a lower index, see kd_build_gpu.hip; the
 * wait is bounded, environment RT_KD_CHAIN_POLL_LIMIT), the call prints a warning, builds the same tree with
 * rt_build_kdtree and counts the event in rt_stats.kd_device_fallbacks. */
int rt_build_kdtree_device(rt_ctx *ctx);
/* sizes of the canonical (reference-layout) export */
int rt_kdtree_counts(rt_ctx *ctx, int64_t *n_nodes, int64_t *n_leaves);
/* canonical export for the bit-exact check: `nodes` receives n_nodes (heapIndex, metadata)
 * pairs in ascending heap index = the non-zero entries of kdTreeData (js/8-kd_tree.js:1632-1645);
 * `leaves` receives n_leaves*4 face indices = kdLeavesData (:2504-2508); root_box = root.min, root.max */
int rt_export_kdtree_reference(rt_ctx *ctx, uint32_t *nodes, int32_t *leaves, double root_box[6]);

/* The tree's overhang, as the closest-hit walk uses it to skip subtrees that cannot hold the nearest hit: out[0..2] =
 * the largest amount by which a face listed by a leaf reaches below that leaf's cell on x, y, z, out[3..5] = above it,
 * measured on the float32 vertices and float32 cells the shader works with (js/8-kd_tree.js:873-885, :2462-2464), plus
 * a safety margin of 1/256 of the root box's largest side. A subtree entered across a split plane of axis a at ray
 * distance t holds no hit nearer than t - out[a] / |d[a]|; the reference visits such subtrees (js/8-kd_tree.js:862-962
 * has no early exit) and finds nothing nearer there, so skipping them leaves every pixel as it is. No counterpart
 * in the reference. */
int rt_kdtree_overhang(rt_ctx *ctx, float out[6]);

/* ---- frame: replaces the tracer draw call (js/8-kd_tree.js:2550-2555) -------------------- */
int rt_trace(rt_ctx *ctx);
/* The same frame in two halves, for callers that overlap work with the trace. No counterpart in the reference
 * (its update() blocks in gl.readPixels, js/8-kd_tree.js:2558).
 * rt_trace_begin queues the whole frame on the context's stream and returns without waiting for the device; the
 * entry points that queue device work (rt_tone_partial, rt_present, rt_assemble_shards) may be called right after it
 * and run behind the trace. rt_trace_end waits for the trace and checks that every buffer was large enough (they are
 * sized from the previous frames). It returns 0 when the frame is complete, 1 when buffers had to grow and the frame
 * was traced again - it is complete now, but whatever was derived from the first attempt (tone sums, presented or
 * gathered pixels) must be redone - and a negative value on failure. The rt_read_* calls refuse to run between the
 * two. rt_trace = rt_trace_begin + rt_trace_end. */
int rt_trace_begin(rt_ctx *ctx);
int rt_trace_end(rt_ctx *ctx);
/* float colour before clamping, 3 floats per pixel, width*S x height*S, bottom-up rows.
 * Pixels outside this context's shard are left untouched. */
int rt_read_float(rt_ctx *ctx, float *rgb);
/* = gl.readPixels(RGBA, UNSIGNED_BYTE) of the render target (js/8-kd_tree.js:2558) */
int rt_read_rgba8(rt_ctx *ctx, uint8_t *rgba);
/* primary hit ids, one int32 per pixel: -1 miss, 0..n_prims-1 analytic primitive,
 * n_prims + face index for a mesh triangle. Hit ids are not gathered between shards: on a sharded context the pixels
 * of the other shards read -1, and rt_pick on such a pixel fails. */
int rt_read_hit_ids(rt_ctx *ctx, int32_t *ids);
/* the primary hit id of one pixel of the last traced frame (x, y in render-target pixels, y = 0 the bottom row): what
 * the reference's mouse picking finds by intersecting the pixel's primary ray with every object on the CPU
 * (UI.onMouseDown -> getPrimaryRay + Primitive.intersect, js/8-kd_tree.js:2718-2760) */
int rt_pick(rt_ctx *ctx, int32_t x, int32_t y, int32_t *id);

/* ---- tone reproduction: replaces the CPU loop of js/8-kd_tree.js:2559-2569 --------------- */
/* partial sums over this shard: sum of log(EPSILON + L) and pixel count; the multi-GPU
 * caller adds them across ranks (all-reduce) before rt_tone_finish */
int rt_tone_partial(rt_ctx *ctx, double *sum_log, uint64_t *count);
/* Lbar = exp(sum_log / count), sf = Ward scale factor; stored in the context for rt_present */
int rt_tone_finish(rt_ctx *ctx, double sum_log, uint64_t count, double *Lbar, double *sf);
/* single-GPU convenience: rt_tone_partial + rt_tone_finish */
int rt_tone(rt_ctx *ctx, double *Lbar, double *sf);
/* present pass (js/8-kd_tree.js:85-143): SxS box filter, tone operator, canvas RGBA8.
 * canvas_rgba may be NULL: the pass is then only queued on the context's stream and the canvas stays in device
 * memory (rt_device_canvas) */
int rt_present(rt_ctx *ctx, uint8_t *canvas_rgba);
/* The same reduction with nothing returning to the host, for multi-GPU callers that all-reduce on the device:
 * rt_tone_partial_device queues the reduction and hands out the device address of THREE doubles,
 * {sum of log(EPSILON + L), pixel count, 1.0 if the last trace ran out of buffer space and rt_trace_end will repeat it,
 * else 0.0}; the caller sums them over the ranks in place (on the context's stream, see rt_stream) and passes the
 * address to rt_tone_finish_device, which queues the Lbar / sf computation from the first two; rt_present then uses
 * them. The third sum tells every rank alike - from its own device memory, without a host-side collective - whether
 * any rank's contribution to this frame was void, i.e. whether the exchange of the frame must be repeated once
 * rt_trace_end has returned. rt_tone_get waits and returns the Lbar / sf last computed by either path. Device exp/pow may differ from
 * the host's in the last bit of the double; the present pass receives them rounded to float, as in the reference. */
int rt_tone_partial_device(rt_ctx *ctx, void **d_sum_count);
int rt_tone_finish_device(rt_ctx *ctx, const void *d_sum_count);
int rt_tone_get(rt_ctx *ctx, double *Lbar, double *sf);

/* ---- device-resident access for multi-GPU gathers (no host round trip) ------------------- */
/* device pointer of the RGBA8 render target (width*S*height*S*4 bytes) */
int rt_device_rgba8(rt_ctx *ctx, void **dptr, size_t *bytes);
int rt_device_float(rt_ctx *ctx, void **dptr, size_t *bytes);
/* device pointer of the presented canvas (width*height*4 bytes) */
int rt_device_canvas(rt_ctx *ctx, void **dptr, size_t *bytes);
/* this context's shard of the RGBA8 target in compact tile order: [tile of the shard][64 pixels of the
 * 8x8 tile, row-major][4 bytes]; alpha 0 marks lanes of edge tiles that fall outside the image.
 * bytes = 256 * (number of tiles of the shard). */
int rt_device_shard_rgba8(rt_ctx *ctx, void **dptr, size_t *bytes);
/* root side of the framebuffer gather: `gathered` (device memory of this context's device) holds
 * n_shards compact shard buffers back to back, each padded to stride_bytes (a multiple of 256);
 * scatters them into this context's full-frame RGBA8 target so that rt_read_rgba8 / rt_present see
 * the whole image. The scatter is queued on the context's stream (the rt_read_* calls and rt_present follow it in
 * stream order). Replaces nothing in the reference (single GPU there). */
int rt_assemble_shards(rt_ctx *ctx, const void *gathered, size_t stride_bytes, int32_t n_shards);
/* The present pass (js/8-kd_tree.js:85-143) spread over the ranks, SUPERSAMPLING 1 only (the shader then reads exactly
 * its own pixel of the render target, so a shard's tiles can be finished where they were traced; with more it samples
 * across tiles and these calls fail: present the assembled frame on the root instead). After the tone values are set
 * (rt_tone_finish / rt_tone_finish_device), rt_present_shard runs the tone operator over this context's compact tiles
 * (queued on its stream); rt_device_shard_canvas hands out the result in the layout of rt_device_shard_rgba8; on the
 * root, rt_assemble_shards_canvas scatters the gathered canvas tiles into the canvas (rt_device_canvas) the way
 * rt_assemble_shards fills the render target. The root then holds the finished canvas but only its own tiles of the
 * render target. Replaces nothing in the reference (single GPU there). */
int rt_present_shard(rt_ctx *ctx);
int rt_device_shard_canvas(rt_ctx *ctx, void **dptr, size_t *bytes);
int rt_assemble_shards_canvas(rt_ctx *ctx, const void *gathered, size_t stride_bytes, int32_t n_shards);
/* The HIP stream (hipStream_t) on which this context queues its work. Work a caller queues on it - a collective, a
 * copy - is ordered with the library's kernels without any host synchronisation. No counterpart in the reference. */
int rt_stream(rt_ctx *ctx, void **stream);
/* Tuning knobs by name. "prune" (0 / 1, default 1): the closest-hit walk skips subtrees that cannot hold the nearest
 * hit (see rt_kdtree_overhang); 0 visits every cell of the root interval like the reference - the frames are identical.
 * "walk_waves_per_cu" (1..32, default 24): persistent waves per compute unit of the k-d walk;
 * two contexts that trace concurrently on one GPU fit side by side at 12. "stage_timing" (0 / 1, default 1): HIP events
 * around every stage kernel, which fill rt_stats.level_stage_ms / stage_ms / level_kernel_ms; without them those
 * fields are zero and a frame is about thirty queue entries shorter. "first_level_rays" (0 = default): capacity of the
 * deeper ray-tree levels on a context's first frame; later frames use what the earlier ones needed (a frame that does
 * not fit is traced again, see rt_trace_end). No counterpart in the reference. */
int rt_set_option(rt_ctx *ctx, const char *name, double value);

int rt_get_stats(rt_ctx *ctx, rt_stats *out);

/* ---- for callers that cannot allocate ----------------------------------------------------- */
/* No counterpart in the reference. The log-luminance reduction of rt_tone_partial_device (js/8-kd_tree.js:2559-2567:
 * sum of log(EPSILON + L) over the pixels whose alpha byte is not zero, and their count) for callers that must not
 * allocate device memory: code under HIP graph capture, serving loops, harnesses that forbid hipMalloc. It takes
 * no context and calls no hipMalloc, hipFree or any other allocating routine: the per-block partial sums live in a
 * 16 KB array in the library's device image, one copy per GPU, held until the process ends.
 *   d_rgba8      device memory, n_pixels RGBA8 pixels (any order; what rt_device_shard_rgba8 or rt_device_rgba8 hand out)
 *   n_pixels     1 .. RT_NOALLOC_MAX_PIXELS, larger inputs are refused
 *   L_max        rt_config.L_max
 *   d_sum_count  device memory, two doubles: receives {sum of log luminance, pixel count}, as rt_tone_partial_device yields
 *   device       the HIP device all three live on; stream: the hipStream_t to queue the two kernels on (NULL = default)
 * The result is bit-identical to rt_tone_partial's for the same pixels. All calls on one GPU share the array, so two
 * calls that overlap in time on one GPU (different streams, different threads) race: order them. Returns 0, or -1 with the
 * reason in rt_noalloc_last_error(). */
#define RT_NOALLOC_MAX_PIXELS (1ull << 30)
int rt_luminance_sums_noalloc(const void *d_rgba8, uint64_t n_pixels, float L_max, void *d_sum_count, int32_t device, void *stream);
const char *rt_noalloc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
